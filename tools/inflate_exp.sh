#!/bin/bash
# Diagnostic: build libirfhip with an inflate-kernel experiment mask (bgzf.hip: IRF_INFLATE_EXP) into
# tools/variants/libirfhip_inf<mask>.so.  usage: tools/inflate_exp.sh <mask> [more hipcc flags]
set -e
cd "$(dirname "$0")/../irfinder_amd/csrc"
mask=$1; shift
mkdir -p ../../tools/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DIRF_INFLATE_EXP=$mask"
/opt/rocm/bin/hipcc $F "$@" -c bgzf.hip -o /tmp/bgzf_$mask.o
/opt/rocm/bin/hipcc $F "$@" -c dreader.hip -o /tmp/dreader_$mask.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/libirfhip_inf$mask.so /tmp/bgzf_$mask.o /tmp/dreader_$mask.o \
  count.o scan.o radix.o stats.o gather.o reorder.o engine.o bamparse.o ref_tables.o bam_reader.o reader_abi.o writers.o -lz -pthread
