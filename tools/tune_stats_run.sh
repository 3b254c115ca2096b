#!/bin/bash
# Diagnostic: tools/tune_stats.py under several library variants.  usage: tools/tune_stats_run.sh "<configs>" tag...
cfgs=$1; shift
for tag in "$@"; do
  echo "== $tag"
  if [ "$tag" = base ]; then unset IRF_LIB_PATH; else export IRF_LIB_PATH=$PWD/tools/variants/libirfhip_$tag.so; fi
  timeout -k 10 300 python tools/tune_stats.py 50000000 $cfgs 2>&1 | grep -E "tune_stats|stats_prof|rror" || exit 1
done
