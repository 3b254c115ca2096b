#!/bin/bash
# Diagnostic: registers / LDS / spills of the kernels in one object file of irfinder_amd/csrc.  usage: tools/kernel_resources.sh count.o
o=$(cd "$(dirname "$0")/../irfinder_amd/csrc" && pwd)/$1
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=/tmp/kres.fat $o /tmp/kres.o.copy
$B/clang-offload-bundler --type=o --input=/tmp/kres.fat --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kres.co
$B/llvm-readelf --notes /tmp/kres.co | grep -E "\.name:|\.vgpr_count|group_segment_fixed_size|\.sgpr_count|spill_count|private_segment_fixed"
