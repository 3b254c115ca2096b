#!/bin/bash
# Diagnostic: issue / memory-pipeline counters of the count kernel (separate rocprofv3 --pmc passes).  usage: tools/pmc_count2.sh <outdir-under-gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/ablate_count.py 50000000 sorted 0,0 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; continue; }
done
python3 - $out <<'P'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'count_kernel' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print("%-45s %16.0f  (n=%d)"%(k, v[-1], len(v)))
P
