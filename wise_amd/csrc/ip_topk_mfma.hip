// HP-2, batched queries: brute-force inner-product scan for up to 32 queries per pass on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate — a serial fmaf chain per output, no reduced precision),
// so that a batch of queries shares ONE pass over the database.  Serves the batched form of the search
// the reference issues one query at a time (src/index/feature_search_index.py:113; the 3842 sequential
// queries of docs/Retrieval-Evaluation.md:36-45) — SURVEY.md §8 f3.
//
// Roofline: still HBM-bound.  Per 32 rows x 512 dims (64 KB of X) a SIMD issues 256 MFMAs of 64 cycles =
// 16.4k cycles, against ~27k cycles of HBM time for those bytes at 5.9 TB/s chip-wide.
//
// Structure (block = 8 independent waves, one block per CU, two waves per SIMD so that one wave's MFMA chain
// covers the other's wait for its LDS-DMA):
//   Q [32][d] lives in LDS for the block's lifetime (16-byte chunks XOR-swizzled by query so that the 32
//   lanes of an MFMA B-operand read hit distinct banks);
//   each wave streams 32-row x 32-column chunks of X through a private 2-deep LDS ring by 16-byte LDS-DMA
//   (lane-linear destination, swizzle on the SOURCE address), waits with a counted vmcnt, reads its A
//   fragments into registers (a third pipeline stage), re-issues the ring slot, and feeds 16 MFMAs per chunk;
//   no block barrier in the loop;  at d = 512 the LDS is exactly full: 64 KiB Q + 64 KiB rings + 32 KiB lists;
//   k-permutation: lane (i, h) holds columns h*16..h*16+15 of row i — the same permutation on the Q side;
//   selection: after a 32-row group a lane holds, for ITS query, the scores of 16 rows; candidates that beat
//   the threshold are insertion-sorted into the wave's k-entry list of that query in LDS (the two lanes of a
//   query take turns).  The threshold is the best k-th key ANY of the block's 8 lists of the query has
//   reached (a key below some list's k-th entry is dominated by k keys of the same query, so it cannot be
//   in the global top-k): inserts, which serialise the wave, fall ~6x against lane-private thresholds.
//   Measured alternatives that were slower: lists in registers with a branch-free bubble; lists in global
//   memory with a device-wide atomic threshold (every insert is a chain of dependent global loads that
//   also drains the LDS-DMA queue: 18-25 ms); 4 waves/block with a 5-deep ring (5.4 ms: nothing runs under a
//   wave's DMA wait); fragment reads software-pipelined one chunk ahead (the earlier vmcnt wait costs more
//   than the overlap gains).
//   10M x 512, 32 queries: 4.9 ms = 4.2 TB/s of X (MFMA+selection alone 3.1 ms, DMA alone 3.3 ms).
// The per-wave lists (8 per block and query) are folded by merge_keys_kernel.
#include "topk_common.h"

namespace wise {

constexpr int CW = 32;          // columns per chunk
constexpr int RING = 2;         // chunks in flight per wave
constexpr int WAVES = 8;        // waves per block (two per SIMD: one's MFMA chain covers the other's DMA wait)
constexpr int CHUNK_BYTES = 32 * CW * 4;  // 4 KiB

__device__ __forceinline__ void glds16_x(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(WAVES * 64, 1) void ip_scan_mfma_kernel(const float* __restrict__ X, long long N, int d,
                                                              const float* __restrict__ qpad, int nq, int k,
                                                              u64* __restrict__ part /*[P][32][k]*/, int abl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;  // MFMA row/query index and k-half
    const int d4 = d >> 2;                   // 16-byte chunks per row
    float* Qs = reinterpret_cast<float*>(smem);                                  // 32*d floats
    unsigned char* ring = smem + (size_t)32 * d * 4 + (size_t)wave * RING * CHUNK_BYTES;
    // one k-entry descending list per (wave, query): entry e of query j at lists[e*32 + j]; the two lanes
    // (h = 0, 1) that serve a query take turns.  The k-th entries double as the block's shared thresholds.
    u64* lists_all = reinterpret_cast<u64*>(smem + (size_t)32 * d * 4 + (size_t)WAVES * RING * CHUNK_BYTES);
    u64* lists = lists_all + (size_t)wave * MFMA_KL * 32;

    // ---- Q -> LDS, chunk c of query j stored at chunk (c & ~15) | ((c & 15) ^ (j & 15))
    for (int idx = threadIdx.x; idx < 32 * d4; idx += WAVES * 64) {
        const int j = idx / d4, c = idx - j * d4;
        const float4 v = reinterpret_cast<const float4*>(qpad)[idx];
        const int pc = (c & ~15) | ((c & 15) ^ (j & 15));
        reinterpret_cast<float4*>(Qs)[j * d4 + pc] = v;
    }
    for (int e = h; e < MFMA_KL; e += 2) lists[e * 32 + i] = 0;
    __syncthreads();

    const int nch = d / CW;
    const long long ngroups = (N + 31) / 32;
    const long long gw = (long long)blockIdx.x * WAVES + wave, nw = (long long)gridDim.x * WAVES;
    const long long my_groups = gw < ngroups ? (ngroups - gw + nw - 1) / nw : 0;
    const long long steps = my_groups * nch;

    // LDS-DMA of chunk s: 4 instructions of 8 rows x 128 B; swizzle on the source address
    long long ig = gw;   // group / chunk / ring slot of the next DMA to issue
    int ic = 0, islot = 0;
    auto issue = [&]() {
        const long long g = ig;
        const int c0 = ic * CW;
        unsigned char* dst = ring + islot * CHUNK_BYTES;
        if (++ic == nch) { ic = 0; ig += nw; }
        if (++islot == RING) islot = 0;
        if (abl == 1) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = u * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);  // logical chunk kept at physical chunk (lane & 7)
            long long grow = g * 32 + row;
            if (grow >= N) grow = N - 1;  // stay in bounds; masked at selection
            glds16_x(X + grow * d + c0 + lc * 4, dst + u * 1024);
        }
    };

    for (long long s = 0; s < RING && s < steps; ++s) issue();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u64 tau = 0;
    const bool active = i < nq;  // lanes of padded queries never select

    long long cg = gw;   // group / chunk / ring slot being consumed
    int cc = 0, cslot = 0;
    for (long long s = 0; s < steps; ++s) {
        // chunk s has landed when at most the RING-1 younger chunks (4 DMA each) are outstanding
        if (s + RING - 1 < steps) wait_vm<(RING - 1) * 4>(); else wait_vm<0>();
        const unsigned char* buf = ring + cslot * CHUNK_BYTES;
        const int c0 = cc * CW;
        if (++cslot == RING) cslot = 0;
        float4 xf[4], qf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int lc = h * 4 + t;  // logical 16-byte chunk of the 128-byte chunk row
            xf[t] = *reinterpret_cast<const float4*>(buf + i * 128 + ((lc ^ ((i >> 1) & 7)) << 4));
            const int qc = (c0 >> 2) + lc;  // chunk index within the query row
            qf[t] = reinterpret_cast<const float4*>(Qs)[i * d4 + ((qc & ~15) | ((qc & 15) ^ (i & 15)))];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (s + RING < steps) issue();  // the slot's fragments are in registers: refill it
        if (abl != 2)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].x, qf[t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].y, qf[t].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].z, qf[t].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].w, qf[t].w, acc, 0, 0, 0);
        }
        if (++cc == nch) {
            cc = 0;
            // ---- a 32-row group is complete: this lane holds query i, rows (r&3) + 8*(r>>2) + 4*h
            const long long row0 = cg * 32;
            cg += nw;
            // adopt the best k-th key any of the block's 4 lists of this query has reached
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const u64 t = lists_all[(size_t)w * MFMA_KL * 32 + (k - 1) * 32 + i];
                tau = t > tau ? t : tau;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const u64 key = make_key(acc[r], (unsigned)row);
                const bool pass = active && row < N && key > tau;
                if (__ballot(pass) != 0) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        // the list is full exactly when its k-th entry is a real key (keys are never 0)
                        if (pass && h == hh) {
                            const u64 kth = lists[(k - 1) * 32 + i];
                            if (key > kth) {
                                int pos = k - 1;
                                while (pos > 0) {
                                    const u64 prev = lists[(pos - 1) * 32 + i];
                                    if (prev >= key) break;
                                    lists[pos * 32 + i] = prev;
                                    --pos;
                                }
                                lists[pos * 32 + i] = key;
                            }
                            const u64 nk = lists[(k - 1) * 32 + i];
                            tau = nk > tau ? nk : tau;
                        }
                        // make lane (i, 0)'s writes visible to lane (i, 1) of the same wave
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
                acc[r] = 0.f;
            }
        }
    }
    // ---- publish: list (block, wave, h) of query i -> part[P_idx][i][0..k)
    const size_t pidx = (size_t)blockIdx.x * WAVES + wave;
    u64* dst = part + (pidx * MFMA_QB + i) * k;
    for (int e = h; e < k; e += 2) dst[e] = active ? lists[e * 32 + i] : 0;
}

int g_mfma_abl = 0;  // ablation knob (wise_debug_set_scan): 1 = no DMA, 2 = no MFMA

static int mfma_grid(long long N) {
    long long need = ((N + 31) / 32 + WAVES - 1) / WAVES;
    if (need < 1) need = 1;
    return need < 256 ? (int)need : 256;
}

bool mfma_scan_supported(int d, int nq, int k) {
    // Q must fit LDS beside the rings and lists (32*d*4 <= 64 KiB), chunks are 32 columns, lists hold 16
    return nq >= 8 && k <= MFMA_KL && d % CW == 0 && d >= CW && d <= 512;
}
int mfma_scan_lists(long long N) { return mfma_grid(N) * WAVES; }
size_t mfma_scan_part_bytes(long long N, int k) { return (size_t)mfma_scan_lists(N) * MFMA_QB * k * sizeof(u64); }

int mfma_scan_launch(const float* X, long long N, int d, const float* qpad, int nq, int k, u64* part, hipStream_t st) {
    const size_t lds = (size_t)32 * d * 4 + (size_t)WAVES * RING * CHUNK_BYTES + (size_t)WAVES * MFMA_KL * 32 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ip_scan_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(ip_scan_mfma_kernel, dim3(mfma_grid(N)), dim3(WAVES * 64), lds, st, X, N, d, qpad, nq, k, part,
                       g_mfma_abl);
    WISE_LAUNCH_CHECK("ip_scan_mfma_kernel");
    return WISE_OK;
}

}  // namespace wise
