#!/bin/bash
# GPU box: times the count kernel of every tools/variants/libirfhip_<tag>.so named on the command line (and of the shipped
# library first).  usage: tools/count_variants.sh <pairs> tag1 tag2 ...   (build the variants here first: tools/build_variant.sh)
pairs=$1; shift
timeout -k 10 400 python tools/count_bench.py $pairs 4 || exit 1
for t in "$@"; do
  IRF_LIB_PATH=tools/variants/libirfhip_$t.so timeout -k 10 200 python tools/count_bench.py $pairs 4 2>&1 | grep -v "^$" || exit 1
done
