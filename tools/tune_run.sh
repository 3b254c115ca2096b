#!/bin/bash
# Diagnostic: time count-kernel variants built by tools/build_variant.sh.  usage: tools/tune_run.sh <masks> tag...
masks=$1; shift
for tag in "$@"; do
  echo "== $tag"
  if [ "$tag" = base ]; then unset IRF_LIB_PATH; else export IRF_LIB_PATH=$PWD/tools/variants/libirfhip_$tag.so; fi
  timeout -k 10 200 python tools/ablate_count.py 50000000 sorted $masks 2>&1 | grep ablate || exit 1
done
