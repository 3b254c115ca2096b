// gather.hip — the two element-wise kernels behind the multi-GPU collection of result tables (SURVEY §8 e;
// include/irf.h: irf_export_device / irf_import_device / irf_gather).  Upstream is one process and has no counterpart.
// A shard's table holds real values only in the rows it owns, so collecting is "zero what is not mine, then add":
// x + 0 = x exactly, also in fp64, which keeps the gathered tables bit-identical to a single-GPU run.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

template <class T>
__global__ __launch_bounds__(256) void mask_rows_kernel(T* __restrict__ data, uint32_t n_cols, uint64_t n, const uint8_t* __restrict__ code,
                                                        uint32_t keep_bits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if ((keep_bits >> code[i]) & 1u) return;
    for (uint32_t c = 0; c < n_cols; c++) data[(uint64_t)c * n + i] = T(0);
}

template <class T>
__global__ __launch_bounds__(256) void add_into_kernel(T* __restrict__ dst, const T* __restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + 255u) / 256u); }

}   // namespace

void launch_mask_rows_f64(double* data, uint32_t n_cols, uint64_t n, const uint8_t* code, uint32_t keep_bits, hipStream_t s) {
    if (n) hipLaunchKernelGGL(mask_rows_kernel<double>, dim3(blocks_for(n)), dim3(256), 0, s, data, n_cols, n, code, keep_bits);
}
void launch_mask_rows_u32(uint32_t* data, uint32_t n_cols, uint64_t n, const uint8_t* code, uint32_t keep_bits, hipStream_t s) {
    if (n) hipLaunchKernelGGL(mask_rows_kernel<uint32_t>, dim3(blocks_for(n)), dim3(256), 0, s, data, n_cols, n, code, keep_bits);
}
void launch_add_into_f64(double* dst, const double* src, uint64_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(add_into_kernel<double>, dim3(blocks_for(n)), dim3(256), 0, s, dst, src, n);
}
void launch_add_into_u32(uint32_t* dst, const uint32_t* src, uint64_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(add_into_kernel<uint32_t>, dim3(blocks_for(n)), dim3(256), 0, s, dst, src, n);
}

}   // namespace dev
}   // namespace irf
