#!/bin/bash
# usage (on the GPU box): tools/inflate_exp_all.sh <pairs> tag...   (tags of tools/variants/libirfhip_<tag>.so)
pairs=$1; shift
for tag in "$@"; do
  echo "== $tag"
  IRF_LIB_PATH=$PWD/tools/variants/libirfhip_$tag.so timeout -k 10 200 python tools/inflate_exp_run.py $pairs 2>&1 | grep -E "inflate_exp|Error|error" || exit 1
done
