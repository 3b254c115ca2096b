#!/bin/bash
# Diagnostic: hardware counters of the device reader's inflate kernel (separate rocprofv3 --pmc passes).
# usage: tools/pmc_inflate.sh <outdir-under-gpurun_out> [pairs]
out=$GRAFT_REPO_ROOT/gpurun_out/$1
pairs=${2:-1000000}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_dreader.py --pairs $pairs --reps 1 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 - $out <<'P'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bgzf_inflate_kernel' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print("%-45s %16.0f  (n=%d)"%(k, sum(v)/len(v), len(v)))
P
