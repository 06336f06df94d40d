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
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ip_scan_mfma_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ip_scan_mfma_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
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

// the register-queue scan over rows [row_offset, row_offset + N) of the database (X points at the first of them);
// lists go to part[0 .. mfma_scan_lists(N))
int split_scan_launch(const float* X, long long N, long long row_offset, int d, const float* qpad, int nq, u64* part,
                      const u64* tau0, hipStream_t st) {
    const size_t dl = (size_t)32 * d * 4 + (size_t)WAVES * MFMA_KL * 32 * 8;
    static bool dattr = false;
    if (!dattr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ip_scan_split_direct_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ip_scan_split_direct_kernel<3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        dattr = true;
    }
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
                                        u64* __restrict__ tau0) {
    const int q = threadIdx.x;
    if (q >= MFMA_QB) return;
    const long long row = cand_rows[(size_t)q * MFMA_KL + MFMA_KL - 1];
    tau0[q] = row >= 0 ? make_key(cand_scores[(size_t)q * MFMA_KL + MFMA_KL - 1], (unsigned)row) : 0;
}
int sample_threshold_launch(const float* cand_scores, const long long* cand_rows, u64* tau0, hipStream_t st) {
    hipLaunchKernelGGL(sample_threshold_kernel, dim3(1), dim3(64), 0, st, cand_scores, cand_rows, tau0);
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
                                                          float* __restrict__ outD, long long* __restrict__ outI) {
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
                   long long id_base, float* outD, long long* outI, hipStream_t st) {
    hipLaunchKernelGGL(rescore_topk_kernel, dim3(nq), dim3(64), 0, st, X, d, Q, cand_rows, k, ids, id_base, outD, outI);
    WISE_LAUNCH_CHECK("rescore_topk_kernel");
    return WISE_OK;
}

}  // namespace wise
