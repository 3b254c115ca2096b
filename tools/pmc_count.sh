#!/bin/bash
# Diagnostic: hardware counters of the count kernel (separate rocprofv3 --pmc passes).  usage: [IRF_PMC_PASSES=n] tools/pmc_count.sh <outdir-under-gpurun_out>
# (IRF_PMC_PASSES=2: the two SQ passes only)
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_WAIT_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCP_UTCL1_TRANSLATION_MISS_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum"; do
  i=$((i+1))
  [ $i -gt ${IRF_PMC_PASSES:-5} ] && break
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/count_bench.py 50000000 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 - $out <<'P'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'count_kernel' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print("%-45s %16.0f  (n=%d)"%(k, sum(v)/len(v), len(v)))
P
