#!/bin/bash
# Diagnostic (GPU box): kernel + copy timeline of the bench step.  usage: tools/trace_step.sh <outdir-under-gpurun_out>; then tools/trace_step.py gpurun_out/<outdir>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-decode-bench --no-random-leg > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
