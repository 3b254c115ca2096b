#!/bin/bash
# GPU box: depth checksums of every repetition, for the variants named (diagnostic: a race shows as a repetition that differs)
pairs=$1; shift
export IRF_CHECK_EVERY_REP=1
for t in "$@"; do
  echo "== $t"
  IRF_LIB_PATH=tools/variants/libirfhip_$t.so timeout -k 10 300 python tools/count_bench.py $pairs ${REPS:-8} 2>&1 | grep -v amdgpu.ids || exit 1
done
