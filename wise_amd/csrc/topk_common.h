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
constexpr int MFMA_KC = 12;   // largest k the split-bf16 candidate scan serves (16 candidates, exact re-scoring)
bool mfma_split_supported(int d, int nq, int k);
int mfma_scan_launch(const float* X, long long N, int d, const float* qpad /*[32][d], zero rows past nq*/, int nq,
                     int k, u64* part, bool split, hipStream_t st);
// the register-queue split scan over a row range, and the sample-pass threshold (ip_topk_mfma.hip)
int split_scan_launch(const float* X, long long N, long long row_offset, int d, const float* qpad, int nq, u64* part,
                      const u64* tau0, hipStream_t st);
bool split_direct_enabled();
// 64 queries per pass, one list per (block, query): part [split64_lists(N)][64][MFMA_KL]
constexpr int MFMA_QB2 = 64;
int split64_lists(long long N);
bool split64_supported(int d);
int split64_scan_launch(const float* X, long long N, long long row_offset, int d, const float* qpad, int nq, u64* part,
                        const u64* tau0, hipStream_t st, const int* gate = nullptr);
// the pass over the bf16 shadow rows, 64 or 32 queries at a time (d up to 512 / 1024): dump != null -> the scores of the
// (sampled) rows go to dump [qb][N]; otherwise every (query, row) reaching thr[query] is appended to cand [qb][cap],
// counts in ctl [qb][4].  chunk_shift >= 0: N counts SAMPLED rows, evenly spaced chunks of 2^chunk_shift groups of 32 rows,
// chunk_stride groups apart
bool shadow64_supported(int d);
bool shadow32_supported(int d);
int shadow_pass_queries(int d);   // queries one pass of the shadow scan carries at this d (0: not served)
bool shadow_one_piece();   // the batched scan takes the query as one bf16 piece (its rounding enters the error bound)
int shadow64_scan_launch(const bf16_t* Xb, long long N, int d, const float* qpad, int nq, const float* thr, int* ctl,
                         u64* cand, int cap, hipStream_t st, float* dump = nullptr, int qb = 64, int chunk_shift = -1,
                         long long chunk_stride = 0, long long row_base = 0 /*Xb points at this row of the index*/);
int sample_threshold_launch(const float* cand_scores, const long long* cand_rows, u64* tau0, hipStream_t st,
                            int kl = MFMA_KL, const int* gate = nullptr);
// exact f32 scores of cand_rows [nq][MFMA_KL], ordered, first k -> outD/outI [nq][k]
int rescore_launch(const float* X, int d, const float* Q, const long long* cand_rows, int nq, int k, const long long* ids,
                   long long id_base, float* outD, long long* outI, hipStream_t st, const int* gate = nullptr);

}  // namespace wise
