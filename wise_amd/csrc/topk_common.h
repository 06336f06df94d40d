// Sortable 64-bit (score,row) keys shared by the scan kernels and the merge kernels.
#pragma once
#include "common.h"

namespace wise {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned f32_order(float f) {
    unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_unorder(unsigned o) {
    unsigned u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
// larger key = better: higher score first, then lower row
__device__ __forceinline__ u64 make_key(float score, unsigned row) {
    return ((u64)f32_order(score) << 32) | (u64)(0xFFFFFFFFu - row);
}

// batched (MFMA) scan, ip_topk_mfma.hip
constexpr int MFMA_QB = 32;   // queries per pass (the N of v_mfma_f32_32x32x2_f32)
constexpr int MFMA_KL = 16;   // per-lane list length: the path serves k <= 16
bool mfma_scan_supported(int d, int nq, int k);
int mfma_scan_lists(long long N);                    // P: partial lists per query the scan leaves
size_t mfma_scan_part_bytes(long long N, int k);     // P * MFMA_QB * k keys
int mfma_scan_launch(const float* X, long long N, int d, const float* qpad /*[32][d], zero rows past nq*/, int nq,
                     int k, u64* part, hipStream_t st);

}  // namespace wise
