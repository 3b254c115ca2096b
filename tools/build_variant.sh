#!/bin/bash
# Diagnostic: build libirfhip with count-kernel tuning macros into tools/variants/libirfhip_<tag>.so
# usage: tools/build_variant.sh <tag> -DK_UNROLL=4 ...
set -e
cd "$(dirname "$0")/../irfinder_amd/csrc"
tag=$1; shift
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c count.hip -o /tmp/count_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c stats.hip -o /tmp/stats_$tag.o
# the engine of a variant reads the diagnostic environment switches (IRF_ABLATE, IRF_ABLATE_STATS, IRF_RESCAN_PLANES, IRF_STATS_*)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DIRF_DIAG "$@" -c engine.hip -o /tmp/engine_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/libirfhip_$tag.so /tmp/count_$tag.o /tmp/stats_$tag.o scan.o radix.o gather.o reorder.o /tmp/engine_$tag.o bgzf.o bamparse.o dreader.o ref_tables.o bam_reader.o reader_abi.o writers.o -lz -pthread
