// HP-2, batched queries: brute-force inner-product scan for 32 or 64 queries per pass over the database, on the
// matrix cores, so that a batch of queries shares ONE pass over X.  Serves the batched form of the search the
// reference issues one query at a time (src/index/feature_search_index.py:113; the 3842 sequential queries of
// docs/Retrieval-Evaluation.md:36-45) — SURVEY.md §8 f3.  Still an HBM-bound path: the kernels are judged against
// N*d*4 bytes per pass.
//
// Four kernels, newest last (wise_ip_topk_f32 / wise_ip_topk_shadow_f32 in ip_topk.hip pick; wise_debug_set_scan can
// force each of the first three):
//   ip_scan_mfma_kernel<false>   f32 operands on v_mfma_f32_32x32x2_f32, X through a per-wave LDS-DMA ring, lists of
//                                k <= 16 entries per (wave, query); scores final.  4.9 ms per 32 queries at 10M x 512:
//                                256 MFMAs of 64 cycles per 32 rows x 512 columns on a SIMD are two thirds of the HBM
//                                time, and loads and MFMAs do not overlap well with one 4 KiB chunk in flight per wave.
//                                Serves 12 < k <= 16.
//   ip_scan_split_direct_kernel  (k <= 12) candidate generation + exact re-scoring.  Operands are split into bf16
//                                halves, x*q ~ hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (error of a dot
//                                product <= 2^-16 sum|x_c q_c| <= 1.6e-5 for unit rows): 6 MFMAs of 32 cycles per 4 KiB
//                                chunk instead of 16 of 64.  X goes straight into a 4-deep register queue (12 KiB in
//                                flight per wave, no ring).  A sample pass over the first 32K rows gives each query a
//                                threshold that 16 rows already beat, so the main pass over the other rows almost
//                                never touches its lists (selection fell from ~1.1 ms to noise).  Lists always keep
//                                MFMA_KL = 16 candidates; rescore_topk_kernel recomputes their scores in f32, orders
//                                them and returns the first k: ids and scores are f32-exact as long as fewer than
//                                16 - k rows sit within the candidate error of the k-th score.  3.64 ms per 32 queries
//                                (5.6 TB/s of X).
//   ip_scan_split64_kernel       the same for 64 queries per pass: with inserts rare, the block's eight waves share one
//                                list per query behind a spin lock, which frees the LDS for the hi/lo images of 64
//                                queries (128 KiB at d = 512).  4.2 ms per 64 queries: 15.2k queries/s at 10M x 512.
//   ip_scan_shadow64_kernel      stage 1 of the batched two-stage search (wise_ip_topk_shadow_f32): the same 64-query
//                                structure over the bf16 SHADOW rows — MFMA operands as loaded, x*(q_hi + q_lo), 48
//                                candidates per query, a dump mode for the threshold pass.  2.5 ms per 64 queries
//                                end to end (25k queries/s); the kernels above are its gated fallback.
//
// Structure shared by all three (block = 8 independent waves, one block per CU, two waves per SIMD):
//   Q lives in LDS for the block's lifetime (16-byte chunks XOR-swizzled by query so that the 32 lanes of an MFMA
//   B-operand read hit distinct banks); no block barrier in the loop;
//   k-permutation: lane (i, h) holds columns h*16..h*16+15 of row i of a 32-column chunk — the same permutation on
//   the Q side;
//   selection: after a 32-row group a lane holds, for ITS query, the scores of 16 rows; candidates that beat the
//   threshold are insertion-sorted into the query's list in LDS (the two lanes of a query take turns).  In the ring
//   kernel the threshold is the best k-th key ANY of the block's 8 lists of the query has reached (a key below some
//   list's k-th entry is dominated by k keys of the same query, so it cannot be in the global top-k).
//   Measured alternatives that were slower: lists in registers with a branch-free bubble; lists in global memory with
//   a device-wide atomic threshold (every insert is a chain of dependent global loads: 18-25 ms); 4 waves/block with a
//   5-deep ring (5.4 ms); fragment reads software-pipelined one chunk ahead; the split products on the DMA ring
//   (4.7 ms: the ring, not the matrix cores, was the limit).
// The per-list results (part) are folded by merge_keys_kernel.
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

// x = hi + lo with both halves rounded to bf16 (|x - hi - lo| <= 2^-18 |x|): eight values -> two MFMA operands
__device__ __forceinline__ void split_bf16x8(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned hp[4], lp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned h2 = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        const float r0 = v[2 * e] - __uint_as_float(h2 << 16);
        const float r1 = v[2 * e + 1] - __uint_as_float(h2 & 0xFFFF0000u);
        hp[e] = h2;
        lp[e] = pack_bf16x2(r0, r1);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 hv = {hp[0], hp[1], hp[2], hp[3]}, lv = {lp[0], lp[1], lp[2], lp[3]};
    hi = __builtin_bit_cast(bf16x8, hv);
    lo = __builtin_bit_cast(bf16x8, lv);
}

// a 32-row group is complete: lane (i, h) holds, for query i, the scores of rows (r&3) + 8*(r>>2) + 4*h of the
// group.  Keys that beat the threshold are insertion-sorted into the wave's list of the query; acc is cleared.
__device__ __forceinline__ void select_group(f32x16& acc, u64& tau, u64* lists, const u64* lists_all, int kl, int i,
                                             int h, bool active, long long row0, long long N,
                                             long long row_offset = 0) {
    // adopt the best k-th key any of the block's 4 lists of this query has reached
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const u64 t = lists_all[(size_t)w * MFMA_KL * 32 + (kl - 1) * 32 + i];
        tau = t > tau ? t : tau;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const u64 key = make_key(acc[r], (unsigned)(row + row_offset));
        const bool pass = active && row < N && key > tau;
        if (__ballot(pass) != 0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                // the list is full exactly when its k-th entry is a real key (keys are never 0)
                if (pass && h == hh) {
                    const u64 kth = lists[(kl - 1) * 32 + i];
                    if (key > kth) {
                        int pos = kl - 1;
                        while (pos > 0) {
                            const u64 prev = lists[(pos - 1) * 32 + i];
                            if (prev >= key) break;
                            lists[pos * 32 + i] = prev;
                            --pos;
                        }
                        lists[pos * 32 + i] = key;
                    }
                    const u64 nk = lists[(kl - 1) * 32 + i];
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

// SPLIT = false: f32 operands on v_mfma_f32_32x32x2_f32, lists of k entries, scores final.
// SPLIT = true:  candidate generation.  Every f32 operand is split into two bf16 halves and a product is
//   hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation (the dropped lo*lo term and the split
//   residues bound the error of a dot product by 2^-16 * sum |x_c q_c| <= 1.6e-5 for unit rows, typically 1e-6);
//   6 MFMAs of 32 cycles replace 16 of 64 per chunk, which takes the matrix cores off the critical path, and
//   the lists always keep MFMA_KL = 16 candidates so that rescore_topk_kernel can put the exact f32 scores of
//   the best 16 in order and return the first k <= MFMA_KC of them.
template <bool SPLIT>
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
    // SPLIT: the same 32*d*4 bytes hold Qhi [32][d] bf16 then Qlo [32][d] bf16; 16-byte chunks (8 columns) of a
    // row swizzled the same way
    const int d8 = d >> 3;
    unsigned char* Qh = smem;
    unsigned char* Ql = smem + (size_t)32 * d * 2;
    for (int idx = threadIdx.x; idx < 32 * d4; idx += WAVES * 64) {
        const int j = idx / d4, c = idx - j * d4;
        const float4 v = reinterpret_cast<const float4*>(qpad)[idx];
        if (SPLIT) {
            const unsigned h01 = pack_bf16x2(v.x, v.y), h23 = pack_bf16x2(v.z, v.w);
            const unsigned l01 = pack_bf16x2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xFFFF0000u));
            const unsigned l23 = pack_bf16x2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xFFFF0000u));
            const int c8 = c >> 1;
            const size_t off = ((size_t)j * d8 + ((c8 & ~15) | ((c8 & 15) ^ (j & 15)))) * 16 + (c & 1) * 8;
            *reinterpret_cast<uint2*>(Qh + off) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(Ql + off) = make_uint2(l01, l23);
        } else {
            const int pc = (c & ~15) | ((c & 15) ^ (j & 15));
            reinterpret_cast<float4*>(Qs)[j * d4 + pc] = v;
        }
    }
    const int kl = SPLIT ? MFMA_KL : k;   // list length
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
        bf16x8 qh[2], ql[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int lc = h * 4 + t;  // logical 16-byte chunk of the 128-byte chunk row
            xf[t] = *reinterpret_cast<const float4*>(buf + i * 128 + ((lc ^ ((i >> 1) & 7)) << 4));
            if (!SPLIT) {
                const int qc = (c0 >> 2) + lc;  // chunk index within the query row
                qf[t] = reinterpret_cast<const float4*>(Qs)[i * d4 + ((qc & ~15) | ((qc & 15) ^ (i & 15)))];
            }
        }
        if (SPLIT) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {   // columns c0 + h*16 + g*8 .. +7 of query i
                const int c8 = (c0 >> 3) + h * 2 + g;
                const size_t off = ((size_t)i * d8 + ((c8 & ~15) | ((c8 & 15) ^ (i & 15)))) * 16;
                qh[g] = *reinterpret_cast<const bf16x8*>(Qh + off);
                ql[g] = *reinterpret_cast<const bf16x8*>(Ql + off);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (s + RING < steps) issue();  // the slot's fragments are in registers: refill it
        if (abl != 2) {
            if (SPLIT) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    bf16x8 xh, xl;
                    split_bf16x8(xf[2 * g], xf[2 * g + 1], xh, xl);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, qh[g], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ql[g], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, qh[g], acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].x, qf[t].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].y, qf[t].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].z, qf[t].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[t].w, qf[t].w, acc, 0, 0, 0);
                }
            }
        }
        if (++cc == nch) {
            cc = 0;
            // ---- a 32-row group is complete: this lane holds query i, rows (r&3) + 8*(r>>2) + 4*h
            const long long row0 = cg * 32;
            cg += nw;
            select_group(acc, tau, lists, lists_all, kl, i, h, active, row0, N);
        }
    }
    // ---- publish: list (block, wave, h) of query i -> part[P_idx][i][0..k)
    const size_t pidx = (size_t)blockIdx.x * WAVES + wave;
    u64* dst = part + (pidx * MFMA_QB + i) * kl;
    for (int e = h; e < kl; e += 2) dst[e] = active ? lists[e * 32 + i] : 0;
}

// ------------------------------------------------------------------------------------------------
// split-bf16 candidate scan with X straight into registers.  The LDS-DMA ring above leaves a wave one 4 KiB chunk
// in flight while it computes (the LDS is full), i.e. 32 KiB per CU against the ~60 KiB a CU's share of HBM
// bandwidth needs over the memory latency: scan and load times add up instead of overlapping (4.75 ms with
// MFMA+selection alone 2.2 ms and loads alone 3.4 ms).  Here a lane loads its 64 bytes of a chunk row with four
// global_load_dwordx4 (lane (i, h): columns h*16..h*16+15 of row i; the two lanes of a row and the four loads
// consume every 128-byte line whole) into a PF-deep register queue: PF-1 chunks = 12 KiB in flight per wave, and
// the freed 64 KiB of LDS is not needed.  Q (hi, lo) and the lists stay in LDS as above.
// ------------------------------------------------------------------------------------------------
template <int PF>
__global__ __launch_bounds__(WAVES * 64, 1) void ip_scan_split_direct_kernel(const float* __restrict__ X, long long N,
                                                                      int d, const float* __restrict__ qpad, int nq,
                                                                      u64* __restrict__ part /*[P][32][MFMA_KL]*/,
                                                                      long long row_offset,
                                                                      const u64* __restrict__ tau0 /*[32] or null*/,
                                                                      int abl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int d4 = d >> 2, d8 = d >> 3;
    unsigned char* Qh = smem;
    unsigned char* Ql = smem + (size_t)32 * d * 2;
    u64* lists_all = reinterpret_cast<u64*>(smem + (size_t)32 * d * 4);
    u64* lists = lists_all + (size_t)wave * MFMA_KL * 32;
    constexpr int kl = MFMA_KL;

    for (int idx = threadIdx.x; idx < 32 * d4; idx += WAVES * 64) {
        const int j = idx / d4, c = idx - j * d4;
        const float4 v = reinterpret_cast<const float4*>(qpad)[idx];
        const unsigned h01 = pack_bf16x2(v.x, v.y), h23 = pack_bf16x2(v.z, v.w);
        const unsigned l01 = pack_bf16x2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xFFFF0000u));
        const unsigned l23 = pack_bf16x2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xFFFF0000u));
        const int c8 = c >> 1;
        const size_t off = ((size_t)j * d8 + ((c8 & ~15) | ((c8 & 15) ^ (j & 15)))) * 16 + (c & 1) * 8;
        *reinterpret_cast<uint2*>(Qh + off) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(Ql + off) = make_uint2(l01, l23);
    }
    for (int e = h; e < MFMA_KL; e += 2) lists[e * 32 + i] = 0;
    __syncthreads();

    const int nch = d / CW;
    const long long ngroups = (N + 31) / 32;
    const long long gw = (long long)blockIdx.x * WAVES + wave, nw = (long long)gridDim.x * WAVES;
    const long long my_groups = gw < ngroups ? (ngroups - gw + nw - 1) / nw : 0;
    const long long steps = my_groups * nch;

    // register queue: slot p holds chunk s with s % PF == p
    float4 xq[PF][4];
    long long pg = gw, issued = 0;   // group / chunk of the next load; past the last step it re-reads the last chunk
    int pc = 0;
    auto prefetch = [&](float4 (&dst)[4]) {
        long long grow = pg * 32 + i;
        if (grow >= N) grow = N - 1;   // stay in bounds; masked at selection
        const float4* src = reinterpret_cast<const float4*>(X + grow * d + pc * CW + h * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = src[t];
        if (issued + 1 < steps) {
            ++issued;
            if (++pc == nch) { pc = 0; pg += nw; }
        }
    };
    if (steps > 0) {
#pragma unroll
        for (int p = 0; p < PF; ++p) prefetch(xq[p]);
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // X, N describe the rows this launch scans; row_offset is the index of its first row in the whole database.
    // tau0: a key that MFMA_KL rows scanned by an earlier launch are known to beat (the sample pass): nothing at
    // or below it can be among the best MFMA_KL, so it never enters a list
    u64 tau = tau0 ? tau0[i] : 0;
    const bool active = i < nq;
    long long cg = gw;
    int cc = 0;
    for (long long s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (s0 + p < steps) {
                const int c0 = cc * CW;
                bf16x8 qh[2], ql[2], xh[2], xl[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c8 = (c0 >> 3) + h * 2 + g;
                    const size_t off = ((size_t)i * d8 + ((c8 & ~15) | ((c8 & 15) ^ (i & 15)))) * 16;
                    qh[g] = *reinterpret_cast<const bf16x8*>(Qh + off);
                    ql[g] = *reinterpret_cast<const bf16x8*>(Ql + off);
                    split_bf16x8(xq[p][2 * g], xq[p][2 * g + 1], xh[g], xl[g]);
                }
                __builtin_amdgcn_sched_barrier(0);
                prefetch(xq[p]);   // the slot's values are in xh/xl: refill it
                __builtin_amdgcn_sched_barrier(0);
                if (!(abl & 2)) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[g], qh[g], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[g], ql[g], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[g], qh[g], acc, 0, 0, 0);
                    }
                } else {
                    acc[0] += (float)xh[0][0] + (float)xl[1][7] + (float)qh[0][0] + (float)ql[1][0];
                }
                if (++cc == nch) {
                    cc = 0;
                    const long long row0 = cg * 32;
                    cg += nw;
                    if (!(abl & 1)) select_group(acc, tau, lists, lists_all, kl, i, h, active, row0, N, row_offset);
                    else {
                        float t = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { t += acc[r]; acc[r] = 0.f; }
                        if (t == 1.2345e-30f) lists[i] = 1;
                    }
                }
            }
        }
    }
    const size_t pidx = (size_t)blockIdx.x * WAVES + wave;
    u64* dst = part + (pidx * MFMA_QB + i) * kl;
    for (int e = h; e < kl; e += 2) dst[e] = active ? lists[e * 32 + i] : 0;
}

int g_split_direct = 4;  // register-queue depth of the split scan (0 = the LDS-DMA ring variant)
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
// the split-bf16 candidate scan keeps MFMA_KL candidates per query and serves k <= MFMA_KC of them
bool mfma_split_supported(int d, int nq, int k) { return mfma_scan_supported(d, nq, k) && k <= MFMA_KC; }
int mfma_scan_lists(long long N) { return mfma_grid(N) * WAVES; }
size_t mfma_scan_part_bytes(long long N, int k) { return (size_t)mfma_scan_lists(N) * MFMA_QB * k * sizeof(u64); }

int mfma_scan_launch(const float* X, long long N, int d, const float* qpad, int nq, int k, u64* part, bool split,
                     hipStream_t st) {
    const size_t lds = (size_t)32 * d * 4 + (size_t)WAVES * RING * CHUNK_BYTES + (size_t)WAVES * MFMA_KL * 32 * 8;
    static PerDeviceOnce attr_set;
    attr_set([&] {
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_mfma_kernel<false>), 160 * 1024);
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_mfma_kernel<true>), 160 * 1024);
    });
    if (split && g_split_direct) {
        set_error("mfma_scan_launch: the register-queue scan is launched through split_scan_launch");
        return WISE_E_UNSUPPORTED;
    } else if (split)
        hipLaunchKernelGGL(ip_scan_mfma_kernel<true>, dim3(mfma_grid(N)), dim3(WAVES * 64), lds, st, X, N, d, qpad, nq, k,
                           part, g_mfma_abl);
    else
        hipLaunchKernelGGL(ip_scan_mfma_kernel<false>, dim3(mfma_grid(N)), dim3(WAVES * 64), lds, st, X, N, d, qpad, nq,
                           k, part, g_mfma_abl);
    WISE_LAUNCH_CHECK("ip_scan_mfma_kernel");
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// 64 queries per pass.  With the sample-pass threshold a list insert is a rare event, so the eight waves of a
// block can share ONE list per query behind a spin lock (LDS: 16 x 64 keys = 8 KiB instead of 8 x that), which
// leaves room for the hi/lo images of 64 queries (128 KiB at d = 512): a pass over X serves twice the queries
// and the loads are still the bound (12 MFMAs of 32 cycles per 4 KiB chunk and wave).
//   lock discipline: a lane that wins the compare-and-swap runs its critical section inside that same loop
//   trip and releases before the wave's next trip, so no lane holds a lock while a wave-mate spins (no
//   hold-and-wait); the two lanes (i, 0), (i, 1) that serve a query in a wave go in two phases as above.
// part [gridDim][64][MFMA_KL].
// ------------------------------------------------------------------------------------------------
constexpr int QB2 = 64;

template <int KLS = MFMA_KL, int QBS = 64>
__device__ __forceinline__ void select_group_shared(f32x16& acc, u64& tau, u64* lists /*[kl][QBS]*/, int* locks, int q,
                                                    int h, bool active, long long row0, long long N,
                                                    long long row_offset) {
    constexpr int kl = KLS;
    {
        const u64 t = lists[(kl - 1) * QBS + q];   // the block's current MFMA_KL-th key of this query
        tau = t > tau ? t : tau;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const u64 key = make_key(acc[r], (unsigned)(row + row_offset));
        const bool pass = active && row < N && key > tau;
        if (__ballot(pass) != 0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                bool todo = pass && h == hh;
                while (__ballot(todo) != 0) {
                    if (todo && __hip_atomic_exchange(&locks[q], 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
                        const u64 kth = lists[(kl - 1) * QBS + q];
                        if (key > kth) {
                            int pos = kl - 1;
                            while (pos > 0) {
                                const u64 prev = lists[(pos - 1) * QBS + q];
                                if (prev >= key) break;
                                lists[pos * QBS + q] = prev;
                                --pos;
                            }
                            lists[pos * QBS + q] = key;
                        }
                        const u64 nk = lists[(kl - 1) * QBS + q];
                        tau = nk > tau ? nk : tau;
                        __hip_atomic_store(&locks[q], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        todo = false;
                    }
                }
            }
        }
        acc[r] = 0.f;
    }
}

template <int PF>
__global__ __launch_bounds__(WAVES * 64, 1) void ip_scan_split64_kernel(const float* __restrict__ X, long long N, int d,
                                                                 const float* __restrict__ qpad /*[64][d]*/, int nq,
                                                                 u64* __restrict__ part /*[grid][64][MFMA_KL]*/,
                                                                 long long row_offset,
                                                                 const u64* __restrict__ tau0 /*[64] or null*/,
                                                                 const int* __restrict__ gate /*null, or run only if != 0*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate && *gate == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int d4 = d >> 2, d8 = d >> 3;
    unsigned char* Qh = smem;
    unsigned char* Ql = smem + (size_t)QB2 * d * 2;
    u64* lists = reinterpret_cast<u64*>(smem + (size_t)QB2 * d * 4);
    int* locks = reinterpret_cast<int*>(lists + MFMA_KL * QB2);
    constexpr int kl = MFMA_KL;

    for (int idx = threadIdx.x; idx < QB2 * d4; idx += WAVES * 64) {
        const int j = idx / d4, c = idx - j * d4;
        const float4 v = reinterpret_cast<const float4*>(qpad)[idx];
        const unsigned h01 = pack_bf16x2(v.x, v.y), h23 = pack_bf16x2(v.z, v.w);
        const unsigned l01 = pack_bf16x2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xFFFF0000u));
        const unsigned l23 = pack_bf16x2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xFFFF0000u));
        const int c8 = c >> 1;
        const size_t off = ((size_t)j * d8 + ((c8 & ~15) | ((c8 & 15) ^ (j & 15)))) * 16 + (c & 1) * 8;
        *reinterpret_cast<uint2*>(Qh + off) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(Ql + off) = make_uint2(l01, l23);
    }
    for (int e = threadIdx.x; e < MFMA_KL * QB2; e += WAVES * 64) lists[e] = 0;
    if (threadIdx.x < QB2) locks[threadIdx.x] = 0;
    __syncthreads();

    const int nch = d / CW;
    const long long ngroups = (N + 31) / 32;
    const long long gw = (long long)blockIdx.x * WAVES + wave, nw = (long long)gridDim.x * WAVES;
    const long long my_groups = gw < ngroups ? (ngroups - gw + nw - 1) / nw : 0;
    const long long steps = my_groups * nch;

    float4 xq[PF][4];
    long long pg = gw, issued = 0;
    int pc = 0;
    auto prefetch = [&](float4 (&dst)[4]) {
        long long grow = pg * 32 + i;
        if (grow >= N) grow = N - 1;
        const float4* src = reinterpret_cast<const float4*>(X + grow * d + pc * CW + h * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = src[t];
        if (issued + 1 < steps) {
            ++issued;
            if (++pc == nch) { pc = 0; pg += nw; }
        }
    };
    if (steps > 0) {
#pragma unroll
        for (int p = 0; p < PF; ++p) prefetch(xq[p]);
    }

    f32x16 acc0, acc1;   // queries i and 32 + i (named, not an array: the accumulators must stay in registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    u64 tau_a = tau0 ? tau0[i] : 0, tau_b = tau0 ? tau0[32 + i] : 0;
    const bool active_a = i < nq, active_b = 32 + i < nq;
    long long cg = gw;
    int cc = 0;
    for (long long s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            {   // nch % PF == 0 (host check): steps is a multiple of PF and a group ends on the last slot
                const int c0 = (cc + p) * CW;
                bf16x8 qha0, qha1, qla0, qla1, qhb0, qhb1, qlb0, qlb1, xh0, xh1, xl0, xl1;
                {
                    const int c8 = (c0 >> 3) + h * 2;
                    const size_t ra = (size_t)i * d8, rb = (size_t)(32 + i) * d8;
                    const size_t o0 = (size_t)((c8 & ~15) | ((c8 & 15) ^ (i & 15))) * 16;
                    const size_t o1 = (size_t)(((c8 + 1) & ~15) | (((c8 + 1) & 15) ^ (i & 15))) * 16;
                    qha0 = *reinterpret_cast<const bf16x8*>(Qh + ra * 16 + o0);
                    qla0 = *reinterpret_cast<const bf16x8*>(Ql + ra * 16 + o0);
                    qha1 = *reinterpret_cast<const bf16x8*>(Qh + ra * 16 + o1);
                    qla1 = *reinterpret_cast<const bf16x8*>(Ql + ra * 16 + o1);
                    qhb0 = *reinterpret_cast<const bf16x8*>(Qh + rb * 16 + o0);
                    qlb0 = *reinterpret_cast<const bf16x8*>(Ql + rb * 16 + o0);
                    qhb1 = *reinterpret_cast<const bf16x8*>(Qh + rb * 16 + o1);
                    qlb1 = *reinterpret_cast<const bf16x8*>(Ql + rb * 16 + o1);
                }
                split_bf16x8(xq[p][0], xq[p][1], xh0, xl0);
                split_bf16x8(xq[p][2], xq[p][3], xh1, xl1);
                __builtin_amdgcn_sched_barrier(0);
                prefetch(xq[p]);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl0, qha0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl0, qhb0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, qla0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, qlb0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, qha0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, qhb0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl1, qha1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl1, qhb1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, qla1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, qlb1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, qha1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, qhb1, acc1, 0, 0, 0);
            }
        }
        cc += PF;
        if (cc == nch) {
            cc = 0;
            const long long row0 = cg * 32;
            cg += nw;
            select_group_shared(acc0, tau_a, lists, locks, i, h, active_a, row0, N, row_offset);
            select_group_shared(acc1, tau_b, lists, locks, 32 + i, h, active_b, row0, N, row_offset);
        }
    }
    __syncthreads();
    u64* dst = part + (size_t)blockIdx.x * QB2 * kl;
    for (int e = threadIdx.x; e < QB2 * kl; e += WAVES * 64) {
        const int q = e / kl, r = e - q * kl;
        dst[e] = q < nq ? lists[r * QB2 + q] : 0;
    }
}

int split64_lists(long long N) { return mfma_grid(N); }
bool split64_supported(int d) { return d % (4 * CW) == 0 && d <= 512; }   // whole groups per trip of the 4-deep queue
int split64_scan_launch(const float* X, long long N, long long row_offset, int d, const float* qpad, int nq, u64* part,
                        const u64* tau0, hipStream_t st, const int* gate) {
    const size_t dl = (size_t)QB2 * d * 4 + (size_t)MFMA_KL * QB2 * 8 + QB2 * 4;
    static PerDeviceOnce dattr;
    dattr([&] {
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_split64_kernel<4>), 160 * 1024);
    });
    hipLaunchKernelGGL(ip_scan_split64_kernel<4>, dim3(mfma_grid(N)), dim3(WAVES * 64), dl, st, X, N, d, qpad, nq, part,
                       row_offset, tau0, gate);
    WISE_LAUNCH_CHECK("ip_scan_split64_kernel");
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// 64 (or 32) queries per pass over the bf16 SHADOW of the index (wise_ip_topk_shadow_f32, two or more queries): the
// pass over the bf16 rows of the batched two-stage exact search.  The rows arrive as bf16, so they are MFMA A-operands
// as loaded (no split, no VALU work): a lane's four 16-byte loads cover 32 columns of a 64-column chunk, and a product is
// x*(q_hi + q_lo): 16 MFMAs per 4 KiB chunk and wave for 64 queries.  Half the bytes of the f32 rows per pass; the
// approximate scores carry the bf16 rounding of x and the 2^-17 left over by the two-piece query (shadow_eps).
// Two modes (the threshold form of the single-query search, ip_topk.hip, for a whole batch):
//   dump     the SAMPLE pass: evenly spaced chunks of 16 groups of 32 rows (chunk_shift = 4, chunk_stride in groups); the
//            scores go to dump [QB][n sampled rows] and a per-query threshold comes out of them (batch_threshold_kernel);
//   collect  the pass over all rows: every (query, row) whose score reaches thr[query] is appended to the query's list
//            cand [QB][cap] (one atomic per hit; a hit is one score in several thousand), counts in ctl[4 q].
// LDS: the Q images only (128 KiB at d = 512, 64 queries).
// ------------------------------------------------------------------------------------------------
constexpr int CW2 = 64;   // columns per chunk of the bf16 scan

// LO: the query enters as two bf16 pieces (hi + lo: its rounding is negligible, 16 MFMAs per 4-KiB chunk at 64 queries);
// !LO: as ONE bf16 piece — half the MFMAs, LDS reads and LDS footprint per query, so a pass carries 128 queries at
// d <= 512 and 64 up to d = 1024; the rounding of the query, ||q - bf16(q)|| (max||x|| + max residual), then enters the
// error bound of the threshold form (ip_topk.hip, query_eps).  NG = QB / 32 accumulator tiles of 32 rows x 32 queries.
template <int PF, int QB, bool LO>
__global__ __launch_bounds__(WAVES * 64, 1) void ip_scan_shadow64_kernel(const bf16_t* __restrict__ Xb, long long N, int d,
                                                                  const float* __restrict__ qpad /*[QB][d]*/, int nq,
                                                                  const float* __restrict__ thr /*[QB] (collect)*/,
                                                                  int* __restrict__ ctl /*[QB][4] (collect)*/,
                                                                  u64* __restrict__ cand /*[QB][cap] (collect)*/, int cap,
                                                                  float* __restrict__ dump /*[QB][N] or null*/,
                                                                  int chunk_shift, long long chunk_stride, int abl,
                                                                  long long row_base /*added to the rows recorded in cand*/) {
#ifdef WISE_DEBUG_KNOBS
    const int ab = abl;          // timing ablations exist in the debug library only: in the product build the hot loop has no
#else                            // branches on them
    constexpr int ab = 0;
    (void)abl;
#endif
    constexpr int NG = QB / 32;
    static_assert(NG == 1 || NG == 2 || (NG == 4 && !LO), "query groups per pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int d4 = d >> 2, d8 = d >> 3;
    unsigned char* Qh = smem;
    unsigned char* Ql = smem + (size_t)QB * d * 2;

    for (int idx = threadIdx.x; idx < QB * d4; idx += WAVES * 64) {
        const int j = idx / d4, c = idx - j * d4;
        const float4 v = reinterpret_cast<const float4*>(qpad)[idx];
        const unsigned h01 = pack_bf16x2(v.x, v.y), h23 = pack_bf16x2(v.z, v.w);
        const int c8 = c >> 1;
        const size_t off = ((size_t)j * d8 + ((c8 & ~15) | ((c8 & 15) ^ (j & 15)))) * 16 + (c & 1) * 8;
        *reinterpret_cast<uint2*>(Qh + off) = make_uint2(h01, h23);
        if constexpr (LO) {
            const unsigned l01 = pack_bf16x2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xFFFF0000u));
            const unsigned l23 = pack_bf16x2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xFFFF0000u));
            *reinterpret_cast<uint2*>(Ql + off) = make_uint2(l01, l23);
        }
    }
    __syncthreads();

    const int nch = d / CW2;
    const long long ngroups = (N + 31) / 32;
    const long long gw = (long long)blockIdx.x * WAVES + wave, nw = (long long)gridDim.x * WAVES;
    const long long my_groups = gw < ngroups ? (ngroups - gw + nw - 1) / nw : 0;
    const long long steps = my_groups * nch;
    // logical group -> first row it stands for (a sample visits evenly spaced chunks of 2^chunk_shift groups)
    auto group_row = [&](long long g) {
        return (chunk_shift >= 0 ? (g >> chunk_shift) * chunk_stride + (g & ((1ll << chunk_shift) - 1)) : g) * 32;
    };
    const long long row_limit = chunk_shift >= 0 ? (long long)1 << 62 : N;   // a sample holds whole groups only

    bf16x8 xq[PF][4];
    long long pg = gw, issued = 0;
    int pc = 0;
    auto prefetch = [&](bf16x8 (&dst)[4]) {
        long long grow = group_row(pg) + i;
        if (grow >= row_limit) grow = row_limit - 1;
        const bf16x8* src = reinterpret_cast<const bf16x8*>(Xb + grow * d + pc * CW2 + h * 32);
        if (ab & 4) {
            // (timing experiment, results meaningless) the same bytes read as a TILED shadow would be: the group's chunk
            // as 4 KiB contiguous, each wave instruction 1 KiB contiguous
            const bf16x8* tb = reinterpret_cast<const bf16x8*>(Xb + (group_row(pg) * d + (long long)pc * 2048)) + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = tb[t * 64];
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = src[t];      // (non-temporal loads measured no better here)
        }
        if (issued + 1 < steps) {
            ++issued;
            if (++pc == nch) { pc = 0; pg += nw; }
        }
    };
    if (steps > 0) {
#pragma unroll
        for (int p = 0; p < PF; ++p) prefetch(xq[p]);
    }

    f32x16 acc[NG];
#pragma unroll
    for (int gq = 0; gq < NG; ++gq)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[gq][r] = 0.f;
    float thr_g[NG];
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) thr_g[gq] = (thr && gq * 32 + i < nq) ? thr[gq * 32 + i] : 3.4028234663852886e38f;
    long long cg = gw;
    int cc = 0;
    for (long long s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            // nch % PF == 0 (host check): a group ends on the last slot of a trip
            const int c8 = ((cc + p) * CW2 >> 3) + h * 4;
            bf16x8 x[4] = {xq[p][0], xq[p][1], xq[p][2], xq[p][3]};
            // Q fragments of k-step j+1 are fetched from LDS while the MFMAs of k-step j run (two register sets)
            bf16x8 qf[2][NG], lf[2][LO ? NG : 1];
            auto fetch_q = [&](int j, bf16x8 (&qd)[NG], bf16x8 (&ld)[LO ? NG : 1]) {
                const size_t o = (size_t)(((c8 + j) & ~15) | (((c8 + j) & 15) ^ (i & 15))) * 16;
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    const size_t rq = (size_t)(gq * 32 + i) * d8 * 16;        // (gq*32 + i) & 15 == i & 15: one swizzle for all
                    qd[gq] = *reinterpret_cast<const bf16x8*>(Qh + rq + o);
                    if constexpr (LO) ld[gq] = *reinterpret_cast<const bf16x8*>(Ql + rq + o);
                }
            };
            fetch_q(0, qf[0], lf[0]);
            if (ab & 2) {   // timing ablation: loads and LDS reads without the matrix products
                float t = (float)x[0][0] + (float)x[1][1] + (float)x[2][2] + (float)x[3][3];
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    fetch_q(j, qf[j & 1], lf[j & 1]);
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq) { t += (float)qf[j & 1][gq][0]; if constexpr (LO) t += (float)lf[j & 1][gq][0]; }
                }
                prefetch(xq[p]);
                acc[0][0] += t;
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < 3) fetch_q(j + 1, qf[(j + 1) & 1], lf[(j + 1) & 1]);
                if (j == 0) prefetch(xq[p]);        // x[] holds this slot's rows: the slot can take the next chunk
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    if constexpr (LO) acc[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[j], lf[j & 1][gq], acc[gq], 0, 0, 0);
                    acc[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[j], qf[j & 1][gq], acc[gq], 0, 0, 0);
                }
            }
        }
        cc += PF;
        if (cc == nch) {
            cc = 0;
            const long long lrow0 = cg * 32;            // logical (sample) position, for the dump
            const long long row0 = group_row(cg);       // physical row
            cg += nw;
            if (dump) {
                // sample pass: the scores go to dump[q][logical row] for a per-query threshold
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long lrow = lrow0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (lrow < N) {
#pragma unroll
                        for (int gq = 0; gq < NG; ++gq) dump[(size_t)(gq * 32 + i) * N + lrow] = acc[gq][r];
                    }
                }
            } else if (ab & 1) {
                float t = 0.f;
#pragma unroll
                for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[gq][r];
                if (t == 1.2345e-30f) ctl[0] = 1;
            } else {
                // collect: a lane holds queries gq*32 + i, 16 rows each
                bool hit = false;
#pragma unroll
                for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hit |= acc[gq][r] >= thr_g[gq];
                if (__ballot(hit) != 0) {
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (row < N && acc[gq][r] >= thr_g[gq]) {
                                const int q = gq * 32 + i;
                                const int pos = atomicAdd(ctl + 4 * q, 1);
                                if (pos < cap) cand[(size_t)q * cap + pos] = make_key(acc[gq][r], (unsigned)(row_base + row));
                            }
                        }
                }
            }
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[gq][r] = 0.f;
        }
    }
}

int g_shadow_one_piece = 1;   // (debug knob) 0: two-piece bf16 query in the batched shadow scan
bool shadow_one_piece() { return g_shadow_one_piece != 0; }
// queries per pass: their Q images must fit LDS (160 KiB) — two bf16 pieces per query: 64 up to d = 512, 32 up to 1024;
// one piece: 128 up to d = 512, 64 up to d = 1024
int shadow_pass_queries(int d) {
    if (d % (4 * CW2) != 0 || d > 1024) return 0;
    if (g_shadow_one_piece) return d <= 512 ? 128 : 64;
    return d <= 512 ? 64 : 32;
}
bool shadow64_supported(int d) { return shadow_pass_queries(d) >= 64; }
bool shadow32_supported(int d) { return shadow_pass_queries(d) >= 32; }
int shadow64_scan_launch(const bf16_t* Xb, long long N, int d, const float* qpad, int nq, const float* thr, int* ctl,
                         u64* cand, int cap, hipStream_t st, float* dump, int qb, int chunk_shift, long long chunk_stride,
                         long long row_base) {
    const bool lo = !g_shadow_one_piece;
    const size_t dl = (size_t)qb * d * (lo ? 4 : 2);
    static PerDeviceOnce dattr;
    dattr([&] {
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_shadow64_kernel<4, 64, true>), 160 * 1024);
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_shadow64_kernel<4, 32, true>), 160 * 1024);
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_shadow64_kernel<4, 128, false>), 160 * 1024);
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_shadow64_kernel<4, 64, false>), 160 * 1024);
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_shadow64_kernel<4, 32, false>), 160 * 1024);
    });
    WISE_CHECK_ARG(dl <= 160 * 1024 && (qb == 32 || qb == 64 || (qb == 128 && !lo)),
                   "shadow scan: %d queries per pass at d=%d not served", qb, d);
#define SH_LAUNCH(QBV, LOV)                                                                                                  \
    hipLaunchKernelGGL((ip_scan_shadow64_kernel<4, QBV, LOV>), dim3(mfma_grid(N)), dim3(WAVES * 64), dl, st, Xb, N, d, qpad, \
                       nq, thr, ctl, cand, cap, dump, chunk_shift, chunk_stride, g_mfma_abl, row_base)
    if (qb == 128) SH_LAUNCH(128, false);
    else if (qb == 64) { if (lo) SH_LAUNCH(64, true); else SH_LAUNCH(64, false); }
    else { if (lo) SH_LAUNCH(32, true); else SH_LAUNCH(32, false); }
#undef SH_LAUNCH
    WISE_LAUNCH_CHECK("ip_scan_shadow64_kernel");
    return WISE_OK;
}

// the register-queue scan over rows [row_offset, row_offset + N) of the database (X points at the first of them);
// lists go to part[0 .. mfma_scan_lists(N))
int split_scan_launch(const float* X, long long N, long long row_offset, int d, const float* qpad, int nq, u64* part,
                      const u64* tau0, hipStream_t st) {
    const size_t dl = (size_t)32 * d * 4 + (size_t)WAVES * MFMA_KL * 32 * 8;
    static PerDeviceOnce dattr;
    dattr([&] {
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_split_direct_kernel<4>), 160 * 1024);
        raise_lds_limit(reinterpret_cast<const void*>(ip_scan_split_direct_kernel<3>), 160 * 1024);
    });
    if (g_split_direct == 3)
        hipLaunchKernelGGL(ip_scan_split_direct_kernel<3>, dim3(mfma_grid(N)), dim3(WAVES * 64), dl, st, X, N, d, qpad, nq,
                           part, row_offset, tau0, g_mfma_abl);
    else
        hipLaunchKernelGGL(ip_scan_split_direct_kernel<4>, dim3(mfma_grid(N)), dim3(WAVES * 64), dl, st, X, N, d, qpad, nq,
                           part, row_offset, tau0, g_mfma_abl);
    WISE_LAUNCH_CHECK("ip_scan_split_direct_kernel");
    return WISE_OK;
}
bool split_direct_enabled() { return g_split_direct != 0; }

// tau0[q] = the key of the last of the MFMA_KL sample candidates of query q (0 while the sample holds fewer)
__global__ void sample_threshold_kernel(const float* __restrict__ cand_scores, const long long* __restrict__ cand_rows,
                                        u64* __restrict__ tau0, int kl, const int* __restrict__ gate) {
    if (gate && *gate == 0) return;
    const int q = threadIdx.x;   // launched with 64 threads: the candidate block is sized for 64 queries
    const long long row = cand_rows[(size_t)q * kl + kl - 1];
    tau0[q] = row >= 0 ? make_key(cand_scores[(size_t)q * kl + kl - 1], (unsigned)row) : 0;
}
int sample_threshold_launch(const float* cand_scores, const long long* cand_rows, u64* tau0, hipStream_t st, int kl,
                            const int* gate) {
    hipLaunchKernelGGL(sample_threshold_kernel, dim3(1), dim3(64), 0, st, cand_scores, cand_rows, tau0, kl, gate);
    WISE_LAUNCH_CHECK("sample_threshold_kernel");
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// exact f32 scores of a query's MFMA_KL candidates, put in order, first k returned: one wave per query.
// cand_rows [nq][MFMA_KL] (row index, -1 = none).  A score is the lane-strided sum of fmaf chains over the
// row's 16-byte chunks folded by a fixed butterfly: the same value whatever produced the candidates.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rescore_topk_kernel(const float* __restrict__ X, int d, const float* __restrict__ Q,
                                                          const long long* __restrict__ cand_rows, int k,
                                                          const long long* __restrict__ ids, long long id_base,
                                                          float* __restrict__ outD, long long* __restrict__ outI,
                                                          const int* __restrict__ gate) {
    if (gate && *gate == 0) return;
    const int q = blockIdx.x, lane = threadIdx.x;
    const int d4 = d >> 2;
    const float4* qv = reinterpret_cast<const float4*>(Q + (size_t)q * d);
    const long long my_row = lane < MFMA_KL ? cand_rows[(size_t)q * MFMA_KL + lane] : -1;
    float my_score = 0.f;
    for (int c = 0; c < MFMA_KL; ++c) {
        const long long row = __shfl(my_row, c, 64);
        if (row < 0) continue;   // wave-uniform
        const float4* xv = reinterpret_cast<const float4*>(X + (size_t)row * d);
        float p = 0.f;
        for (int j = lane; j < d4; j += 64) {
            const float4 a = xv[j], b = qv[j];
            p = fmaf(a.x, b.x, p); p = fmaf(a.y, b.y, p); p = fmaf(a.z, b.z, p); p = fmaf(a.w, b.w, p);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
        if (lane == c) my_score = p;
    }
    const u64 my_key = my_row >= 0 ? make_key(my_score, (unsigned)my_row) : 0;
    int rank = 0, valid = 0;
    for (int c = 0; c < MFMA_KL; ++c) {
        const u64 other = __shfl(my_key, c, 64);
        rank += other > my_key;
        valid += other != 0;
    }
    if (my_key != 0 && rank < k) {
        outD[(size_t)q * k + rank] = my_score;
        outI[(size_t)q * k + rank] = ids ? ids[my_row] : id_base + my_row;
    }
    if (lane < k && lane >= valid) {   // fewer than k rows in the index
        outD[(size_t)q * k + lane] = -3.4028234663852886e38f;
        outI[(size_t)q * k + lane] = -1;
    }
}

int rescore_launch(const float* X, int d, const float* Q, const long long* cand_rows, int nq, int k, const long long* ids,
                   long long id_base, float* outD, long long* outI, hipStream_t st, const int* gate) {
    hipLaunchKernelGGL(rescore_topk_kernel, dim3(nq), dim3(64), 0, st, X, d, Q, cand_rows, k, ids, id_base, outD, outI,
                       gate);
    WISE_LAUNCH_CHECK("rescore_topk_kernel");
    return WISE_OK;
}

}  // namespace wise
