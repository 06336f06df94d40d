// HP-2: brute-force inner-product scan + top-k for gfx950 (MI355X).
//
// Replaces faiss IndexIDMap{IndexFlatIP}::search as called from the reference at
// src/index/feature_search_index.py:113 and api/routes.py:1407 (semantics: SURVEY.md App. A.3).
//
// Roofline: HBM-bound.  Algorithmic bytes per query batch = N*d*4 (every fp32 row read once).
//
// Kernel 1 (ip_scan_kernel): every wave streams groups of R rows (16 B per lane per load, rows are
//   contiguous so a group is one R*d*4-byte burst), keeps the query chunk(s) it needs in VGPRs, and
//   reduces the R*NQ partial dot products with a butterfly *transpose*-reduce (log-many cross-lane
//   moves for all R rows together, not 6 per row).  Selection is fused: each (score,row) becomes a
//   sortable 64-bit key; a wave appends the keys that beat its running threshold to a wave-private
//   LDS list and re-thresholds (bitonic sort in LDS, wave-synchronous, no block barrier) when the
//   list fills.  Nothing but the k best keys per block ever goes back to HBM.
// Kernel 2 (merge_keys_kernel): one block per query folds the per-block lists into the final
//   top-k with the same threshold lists, then translates row -> external id (IndexIDMap).
#include "topk_common.h"

namespace wise {

__device__ __forceinline__ void wave_lds_fence() {
    // one wave executes its DS instructions in order; this only stops the compiler reordering them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sort buf[0..cap) descending by one wave (cap = power of two >= 64).
__device__ void wave_bitonic_desc(volatile u64* buf, int cap, int lane) {
    for (int size = 2; size <= cap; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (cap >> 1); t += 64) {
                int pos = ((t / stride) * (stride << 1)) + (t % stride);
                int par = pos + stride;
                bool desc = ((pos & size) == 0);
                u64 a = buf[pos], b = buf[par];
                bool sw = desc ? (a < b) : (a > b);
                if (sw) { buf[pos] = b; buf[par] = a; }
            }
            wave_lds_fence();
        }
    }
}

// A wave-private running top-k list in LDS.
struct WaveList {
    volatile u64* buf;  // cap entries
    int cap, k, cnt;
    u64 tau;  // keys <= tau cannot enter the top-k any more
    __device__ void init(u64* b, int cap_, int k_, int lane) {
        buf = b; cap = cap_; k = k_; cnt = 0; tau = 0;
        for (int i = lane; i < cap; i += 64) buf[i] = 0;
        wave_lds_fence();
    }
    __device__ void compact(int lane) {
        for (int i = cnt + lane; i < cap; i += 64) buf[i] = 0;
        wave_lds_fence();
        wave_bitonic_desc(buf, cap, lane);
        if (cnt >= k) { cnt = k; tau = buf[k - 1]; }
    }
    // every lane may carry one candidate key (pass=false -> none); wave-uniform control flow
    __device__ void offer(bool pass, u64 key, int lane, int max_new) {
        u64 mask = __ballot(pass);
        if (mask == 0) return;
        int n = __popcll(mask);
        if (cnt + n > cap) {
            compact(lane);
            // the threshold moved: re-test
            pass = pass && (key > tau);
            mask = __ballot(pass);
            if (mask == 0) return;
            n = __popcll(mask);
        }
        int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (pass) buf[pos] = key;
        cnt += n;
        wave_lds_fence();
        (void)max_new;
    }
};

// ------------------------------------------------------------------------------------------------
// scan kernel: NV = float4 chunks per lane per row (ceil(d/256)), NQ queries, R rows per group
// ------------------------------------------------------------------------------------------------
// SEG = true is the inverted-list form (IndexIVFFlat, wise_ivf_scan_f32): block b serves query b / nprobe and
// scans only the rows of the list named by probes[b] (X holds the lists back to back, list_off their bounds);
// its k keys go to part[(b % nprobe) * nq + b / nprobe] so that merge_keys_kernel folds a query's nprobe lists.
struct SegArgs {
    const long long* probes;    // [nq][nprobe] list numbers, < 0 = nothing to scan
    const long long* list_off;  // [nlist + 1]
    int nprobe, nq;
    const int* gate;            // optional: the launch does nothing unless *gate != 0 (two-stage search fallback)
};

template <int NV, int NQ, int R, bool SEG = false>
__global__ __launch_bounds__(256) void ip_scan_kernel(const f32x4* __restrict__ X, long long N, int d4,
                                                      const float* __restrict__ Q, int k, int cap,
                                                      u64* __restrict__ part /*[grid][NQ][k]*/, SegArgs seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (seg.gate && *seg.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    u64* lds = reinterpret_cast<u64*>(smem);
    // wave w, query q -> lds + (w*NQ + q)*cap
    long long lo = 0;             // first row of the scanned range; N becomes its end
    size_t slot = blockIdx.x;     // where the block's keys go
    if constexpr (SEG) {
        static_assert(NQ == 1, "one query per block in the inverted-list form");
        const int qi = blockIdx.x / seg.nprobe, pi = blockIdx.x - qi * seg.nprobe;
        const long long l = seg.probes[blockIdx.x];
        if (l >= 0) { lo = seg.list_off[l]; N = seg.list_off[l + 1]; } else { N = 0; }
        Q += (size_t)qi * d4 * 4;
        slot = (size_t)pi * seg.nq + qi;
    }

    float4 qv[NQ][NV];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = v * 64 + lane;
            qv[q][v] = (c < d4) ? reinterpret_cast<const float4*>(Q)[(long long)q * d4 + c]
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }

    WaveList wl[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) wl[q].init(lds + (size_t)(wave * NQ + q) * cap, cap, k, lane);

    const long long ngroups = (N - lo + R - 1) / R;
    const long long gw = SEG ? wave : (long long)blockIdx.x * 4 + wave;
    const long long nw = SEG ? 4 : (long long)gridDim.x * 4;

    // which of the R rows this lane ends up holding after the transpose-reduce
    int myr = 0;
    {
        int bit = 5;
#pragma unroll
        for (int h = R / 2; h >= 1; h >>= 1, --bit) myr += ((lane >> bit) & 1) * h;
    }
    constexpr int LOGR = (R == 8) ? 3 : (R == 4) ? 2 : (R == 2) ? 1 : 0;
    const bool owner = (lane & ((64 >> LOGR) - 1)) == 0;

    for (long long g = gw; g < ngroups; g += nw) {
        const long long row0 = lo + g * R;
        f32x4 x[R][NV];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            long long row = row0 + r;
            if (row >= N) row = N - 1;  // stay in bounds; masked below
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = v * 64 + lane;
                if (NV * 64 == d4 || c < d4)
                    x[r][v] = __builtin_nontemporal_load(&X[row * d4 + c]);
                else
                    x[r][v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float a[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    s = fmaf(x[r][v][0], qv[q][v].x, s);
                    s = fmaf(x[r][v][1], qv[q][v].y, s);
                    s = fmaf(x[r][v][2], qv[q][v].z, s);
                    s = fmaf(x[r][v][3], qv[q][v].w, s);
                }
                a[r] = s;
            }
            // transpose-reduce: after step with mask m, a lane keeps half of its values, each summed
            // with the partner lane's copy
            int bit = 5;
#pragma unroll
            for (int h = R / 2; h >= 1; h >>= 1, --bit) {
                const int m = 1 << bit;
                const bool up = (lane >> bit) & 1;
#pragma unroll
                for (int i = 0; i < h; ++i) {
                    float send = up ? a[i] : a[i + h];
                    float keep = up ? a[i + h] : a[i];
                    a[i] = keep + __shfl_xor(send, m, 64);
                }
            }
            float s = a[0];
#pragma unroll
            for (int m = (32 >> LOGR); m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);

            const long long row = row0 + myr;
            const u64 key = make_key(s, (unsigned)row);
            const bool pass = owner && (row < N) && (key > wl[q].tau);
            wl[q].offer(pass, key, lane, R);
        }
    }

    // fold the block's 4 wave lists into wave 0's list, then publish k keys per query
#pragma unroll
    for (int q = 0; q < NQ; ++q) wl[q].compact(lane);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            for (int w = 1; w < 4; ++w) {
                const u64* other = lds + (size_t)(w * NQ + q) * cap;
                for (int i0 = 0; i0 < k; i0 += 64) {
                    int i = i0 + lane;
                    u64 key = (i < k) ? other[i] : 0;
                    bool pass = (key != 0) && (key > wl[q].tau);
                    wl[q].offer(pass, key, lane, 64);
                }
            }
            wl[q].compact(lane);
            u64* dst = part + (slot * NQ + q) * k;
            for (int i = lane; i < k; i += 64) dst[i] = wl[q].buf[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// merge kernel: one block per query, NW waves; part [P][nq_stride][k] keys -> outD/outI [nq][k]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void merge_keys_kernel(const u64* __restrict__ part, int P, int qstride,
                                                          int k, int cap, const long long* __restrict__ ids,
                                                          long long id_base, float* __restrict__ outD,
                                                          long long* __restrict__ outI, int q_off,
                                                          const int* __restrict__ gate = nullptr, int k_in = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate && *gate == 0) return;
    const int kin = k_in > 0 ? k_in : k;   // keys per input list (the lists may be shorter than the k kept)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int q = blockIdx.x;
    u64* lds = reinterpret_cast<u64*>(smem);

    WaveList wl;
    wl.init(lds + (size_t)wave * cap, cap, k, lane);
    // the P lists of this query hold P*k keys in all: every offer carries 64 of them (one per lane), whatever k
    // is; waves take 64-key chunks round-robin
    const long long total = (long long)P * kin;
    for (long long c0 = (long long)wave * 64; c0 < total; c0 += (long long)nwaves * 64) {
        const long long i = c0 + lane;
        u64 key = 0;
        if (i < total) {
            const int p = (int)(i / kin), j = (int)(i - (long long)p * kin);
            key = part[((size_t)p * qstride + q) * kin + j];
        }
        const bool pass = (key != 0) && (key > wl.tau);
        wl.offer(pass, key, lane, 64);
    }
    wl.compact(lane);
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < nwaves; ++w) {
            const u64* other = lds + (size_t)w * cap;
            for (int i0 = 0; i0 < k; i0 += 64) {
                int i = i0 + lane;
                u64 key = (i < k) ? other[i] : 0;
                bool pass = (key != 0) && (key > wl.tau);
                wl.offer(pass, key, lane, 64);
            }
        }
        wl.compact(lane);
        for (int i = lane; i < k; i += 64) {
            u64 key = wl.buf[i];
            float dscore = -3.4028234663852886e38f;
            long long id = -1;
            if (key != 0) {
                unsigned row = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
                dscore = f32_unorder((unsigned)(key >> 32));
                id = ids ? ids[row] : (id_base + (long long)row);
            }
            outD[(size_t)(q_off + q) * k + i] = dscore;
            outI[(size_t)(q_off + q) * k + i] = id;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// merge of (score,id) partial lists (multi-GPU all-gather result): one wave per query
// key = (order(score), ~slot) where slot = part*k + i keeps "lower part first, then list order"
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void merge_pairs_kernel(const float* __restrict__ inD,
                                                         const long long* __restrict__ inI, int parts, int nq,
                                                         int k, int cap, float* __restrict__ outD,
                                                         long long* __restrict__ outI) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int q = blockIdx.x;
    WaveList wl;
    wl.init(reinterpret_cast<u64*>(smem), cap, k, lane);
    for (int p = 0; p < parts; ++p) {
        const size_t base = ((size_t)p * nq + q) * k;
        for (int i0 = 0; i0 < k; i0 += 64) {
            int i = i0 + lane;
            bool valid = (i < k) && (inI[base + i] >= 0);
            u64 key = valid ? make_key(inD[base + i], (unsigned)(p * k + i)) : 0;
            bool pass = valid && (key > wl.tau);
            wl.offer(pass, key, lane, 64);
        }
    }
    wl.compact(lane);
    for (int i = lane; i < k; i += 64) {
        u64 key = wl.buf[i];
        float dscore = -3.4028234663852886e38f;
        long long id = -1;
        if (key != 0) {
            unsigned slot = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            int p = slot / k, j = slot % k;
            size_t src = ((size_t)p * nq + q) * k + j;
            dscore = inD[src];
            id = inI[src];
        }
        outD[(size_t)q * k + i] = dscore;
        outI[(size_t)q * k + i] = id;
    }
}

__global__ void reconstruct_kernel(const float* __restrict__ X, long long N, int d,
                                   const long long* __restrict__ ids, long long id_base,
                                   const long long* __restrict__ qids, int n, float* __restrict__ out) {
    __shared__ unsigned long long s_row1;  // row + 1, 0 = not found
    const int i = blockIdx.x;
    const long long want = qids[i];
    if (threadIdx.x == 0)
        s_row1 = ids ? 0ull : ((want >= id_base && want - id_base < N) ? (unsigned long long)(want - id_base + 1) : 0ull);
    __syncthreads();
    if (ids) {
        for (long long r = threadIdx.x; r < N; r += blockDim.x)
            if (ids[r] == want) atomicMax(&s_row1, (unsigned long long)(r + 1));
        __syncthreads();
    }
    const long long row = (long long)s_row1 - 1;
    for (int c = threadIdx.x; c < d; c += blockDim.x)
        out[(size_t)i * d + c] = (row >= 0) ? X[row * d + c] : __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------------
// select_topk_kernel: indices of the k largest of n scores, one block per row of scores (the coarse stage of
// IndexIVFFlat when nprobe is large: n = nlist is tens of thousands and k up to 2048, where threshold lists
// stop filtering).  Radix select on the sortable 32-bit score, most significant byte first (4 histogram passes
// find the k-th largest value T and how many elements equal to T still belong), then one ordered compaction:
// everything above T, plus the lowest-indexed ties.  Output: the chosen indices in ascending index order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_topk_kernel(const float* __restrict__ scores, int n, int k,
                                                           long long* __restrict__ out /*[rows][k]*/) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining;
    __shared__ unsigned wave_cnt[2][16];
    __shared__ unsigned base_gt, base_eq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = scores + (size_t)blockIdx.x * n;
    long long* dst = out + (size_t)blockIdx.x * k;
    const int kk = k < n ? k : n;
    if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)kk; }
    unsigned mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i = tid; i < n; i += 1024) {
            const unsigned key = f32_order(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rem = s_remaining, cum = 0;
            int b = 255;
            for (; b > 0; --b) {
                if (cum + hist[b] >= rem) break;
                cum += hist[b];
            }
            s_remaining = rem - cum;          // still to take from bucket b
            s_prefix = prefix | ((unsigned)b << shift);
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned T = s_prefix;              // the k-th largest key
    const unsigned take_eq = s_remaining;     // how many elements equal to T belong to the result
    if (tid == 0) { base_gt = 0; base_eq = 0; }
    __syncthreads();
    // ordered compaction, 1024 indices at a time: [greater..., then ties] keeps ascending index order within
    // each class; the two classes are interleaved by position so the output is ascending overall
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const unsigned key = i < n ? f32_order(row[i]) : 0u;
        const bool gt = i < n && key > T, eq = i < n && key == T;
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        if (lane == 0) { wave_cnt[0][wave] = __popcll(mg); wave_cnt[1][wave] = __popcll(me); }
        __syncthreads();
        unsigned off_g = base_gt, off_e = base_eq, tot_g = 0, tot_e = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) { off_g += wave_cnt[0][w]; off_e += wave_cnt[1][w]; }
            tot_g += wave_cnt[0][w]; tot_e += wave_cnt[1][w];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned rank_g = off_g + __popcll(mg & below), rank_e = off_e + __popcll(me & below);
        // position in the output = (#greater before me) + (#accepted ties before me)
        if (gt) dst[rank_g + min(rank_e, take_eq)] = i;
        else if (eq && rank_e < take_eq) dst[rank_g + rank_e] = i;
        __syncthreads();
        if (tid == 0) { base_gt += tot_g; base_eq += tot_e; }
        __syncthreads();
    }
    for (int j = kk + tid; j < k; j += 1024) dst[j] = -1;   // fewer than k scores: padding
}

// ------------------------------------------------------------------------------------------------
// Two-stage exact search over a bf16 shadow of the index, wise_ip_topk_shadow_f32 (the threshold form is described at
// ip_collect_bf16_kernel below; the batched form runs the same steps with the bf16 rows on the matrix cores,
// ip_topk_mfma.hip).
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void shadow_bf16_kernel(const float* __restrict__ X, long long N, int d,
                                                          bf16_t* __restrict__ Xb, float* __restrict__ norms) {
    // a wave per row, waves stride over the rows: bf16 (RNE) copy, the largest row norm -> norms[0] and the largest
    // norm of a row's rounding residual |x - bf16(x)| -> norms[1] (non-negative floats order like their bit patterns; one
    // atomic pair per wave at the end).  The residual norm is what bounds a score's error: |q.x - q.bf16(x)| <= |q| |x - bf16(x)|;
    // for rows with random mantissas it is ~0.4 x the worst case 2^-8 |x|.
    const int lane = threadIdx.x & 63;
    const long long w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    float best = 0.f, best_err = 0.f;
    for (long long row = w0; row < N; row += nw) {
        const float4* xr = reinterpret_cast<const float4*>(X + row * d);
        uint2* br = reinterpret_cast<uint2*>(Xb + row * d);
        float ss = 0.f, ee = 0.f;
        for (int c = lane; c < (d >> 2); c += 64) {
            const float4 v = xr[c];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            const uint2 pk = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            br[c] = pk;
            const float e0 = v.x - __uint_as_float(pk.x << 16), e1 = v.y - __uint_as_float(pk.x & 0xFFFF0000u);
            const float e2 = v.z - __uint_as_float(pk.y << 16), e3 = v.w - __uint_as_float(pk.y & 0xFFFF0000u);
            ee += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { ss += __shfl_xor(ss, o, 64); ee += __shfl_xor(ee, o, 64); }
        best = ss > best ? ss : best;
        best_err = ee > best_err ? ee : best_err;
    }
    if (lane == 0) {
        atomicMax(reinterpret_cast<unsigned*>(norms), __float_as_uint(sqrtf(best)));
        atomicMax(reinterpret_cast<unsigned*>(norms) + 1, __float_as_uint(sqrtf(best_err)));
    }
}

// the score error two searches of the same query may differ by: rounding residual of the shadow rows (Cauchy-Schwarz
// with the largest residual norm) + f32 accumulation error of both dot products (d 2^-23 |q| max|x|)
__device__ __forceinline__ float shadow_eps(const float* __restrict__ norms, int d, float qq) {
    return (norms[1] + (float)d * 1.1920929e-7f * norms[0]) * 1.0001f * sqrtf(qq);
}

// how the query entered the approximate scores: exactly (f32, the single-query scan), as two bf16 pieces (leaves
// <= 2^-17 |q| of each score unaccounted), or as ONE bf16 piece — then |x_b . (q - bf16 q)| <=
// (max|x| + max residual) |q - bf16 q|, with the rounding residual of the query measured (qr = its squared norm)
enum : int { QMODE_F32 = 0, QMODE_TWO_PIECE = 1, QMODE_ONE_PIECE = 2 };
__device__ __forceinline__ float query_eps(const float* __restrict__ norms, int d, float qq, float qr, int q_mode) {
    float eps = shadow_eps(norms, d, qq);
    if (q_mode == QMODE_TWO_PIECE) eps += 1.0e-5f * sqrtf(qq) * norms[0];
    if (q_mode == QMODE_ONE_PIECE) eps += (norms[0] + norms[1]) * sqrtf(qr) * 1.0001f + 1.0e-6f * sqrtf(qq) * norms[0];
    return eps;
}
__device__ __forceinline__ float bf16_round_residual(float v) { return v - bf16_to_f32(f32_to_bf16(v)); }

// One group of R rows of the bf16 shadow against the query held in registers: every lane ends up with the score of row
// `row0 + myr` (valid in the lanes with `owner`): 16-byte non-temporal loads, bf16 -> f32 by shift / mask, f32 fma chains,
// butterfly transpose-reduce over the lanes.  NV8 = 16-byte chunks (8 bf16) per lane and row.
template <int NV8, int R>
struct ShadowGroup {
    static constexpr int LOGR = (R == 8) ? 3 : (R == 4) ? 2 : (R == 2) ? 1 : 0;
    float qv[NV8][8];
    int myr;
    bool owner;
    __device__ float load_query(const float* __restrict__ Q, int d8, int lane) {   // returns |q|^2
        float qq = 0.f;
#pragma unroll
        for (int v = 0; v < NV8; ++v) {
            const int c = v * 64 + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qv[v][e] = (c < d8) ? Q[c * 8 + e] : 0.f;
                qq = fmaf(qv[v][e], qv[v][e], qq);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o, 64);
        myr = 0;
        int bit = 5;
#pragma unroll
        for (int h = R / 2; h >= 1; h >>= 1, --bit) myr += ((lane >> bit) & 1) * h;
        owner = (lane & ((64 >> LOGR) - 1)) == 0;
        return qq;
    }
    __device__ float score(const uint4* __restrict__ Xb, long long row0, long long row_end, int d8, int lane) const {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t x[R][NV8];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            long long row = row0 + r;
            if (row >= row_end) row = row_end - 1;
#pragma unroll
            for (int v = 0; v < NV8; ++v) {
                const int c = v * 64 + lane;
                if (NV8 * 64 == d8 || c < d8)
                    x[r][v] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(Xb) + row * d8 + c);
                else
                    x[r][v] = u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        float a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NV8; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned u = x[r][v][e];
                    s = fmaf(__uint_as_float(u << 16), qv[v][2 * e], s);
                    s = fmaf(__uint_as_float(u & 0xFFFF0000u), qv[v][2 * e + 1], s);
                }
            a[r] = s;
        }
        int bit = 5;
#pragma unroll
        for (int h = R / 2; h >= 1; h >>= 1, --bit) {
            const int m = 1 << bit;
            const bool up = (lane >> bit) & 1;
#pragma unroll
            for (int i = 0; i < h; ++i) {
                float send = up ? a[i] : a[i + h];
                float keep = up ? a[i + h] : a[i];
                a[i] = keep + __shfl_xor(send, m, 64);
            }
        }
        float sc = a[0];
#pragma unroll
        for (int m = (32 >> LOGR); m >= 1; m >>= 1) sc += __shfl_xor(sc, m, 64);
        return sc;
    }
};

// ------------------------------------------------------------------------------------------------
// The int8 shadow (wise_ip_shadow_i8): a quarter of the bytes of X per query.  Row r is kept as signed bytes c_r with a
// scale s_r = max|x_r| / 127 (round to nearest): x^_r = s_r c_r.  The query enters as two int8 pieces,
// q^ = sq h + (sq/254) l with sq = max|q| / 127, so a score is two v_dot4_i32_i8 chains — exact integer sums — and
// three fp32 operations per lane:  s^ = s_r (sq H + (sq/254) L).  What a score can be off by:
//   |q.x - s^| <= |q| |x_r - x^_r|  +  |q - q^| |x^_r|  +  fp32 rounding of the per-lane combination and the lane sums
//              <= |q| (rho_max + sqrt(d) 1.6e-5 X^max + 8 2^-24 X^max)
// (|q - q^| <= sqrt(d) sq / 508 and sq <= |q| / 127).  shadow_i8_kernel measures rho_max = max_r |x_r - x^_r| and
// X^max = max_r |x^_r| and stores norms[0] = X^max, norms[1] = rho_max + sqrt(d) 1.6e-5 X^max: shadow_eps(), written for
// the bf16 shadow, then bounds the int8 scores as it stands, and every kernel behind the two scans is shared.
// For Gaussian or CLIP-like rows rho is ~0.9 % of |x| (bf16: 0.2 %): the collect pass hands on a few hundred rows
// instead of a few dozen, all re-scored from the f32 rows as before.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shadow_i8_kernel(const float* __restrict__ X, long long N, int d,
                                                        signed char* __restrict__ Xq, float* __restrict__ scales,
                                                        float* __restrict__ norms /*[0] max |x^|, [2] max |x - x^|*/) {
    const int lane = threadIdx.x & 63;
    const long long w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    float best = 0.f, best_err = 0.f;
    for (long long row = w0; row < N; row += nw) {
        const float4* xr = reinterpret_cast<const float4*>(X + row * d);
        unsigned* qr = reinterpret_cast<unsigned*>(Xq + row * d);
        float mx = 0.f;
        for (int c = lane; c < (d >> 2); c += 64) {
            const float4 v = xr[c];
            mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float scale = mx / 127.f, inv = mx > 0.f ? 127.f / mx : 0.f;
        float hh = 0.f, ee = 0.f;
        for (int c = lane; c < (d >> 2); c += 64) {
            const float4 v = xr[c];
            const float xs[4] = {v.x, v.y, v.z, v.w};
            unsigned pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float qf = fminf(fmaxf(rintf(xs[e] * inv), -127.f), 127.f);
                const float back = scale * qf, err = xs[e] - back;
                hh = fmaf(back, back, hh);
                ee = fmaf(err, err, ee);
                pk |= ((unsigned)(int)qf & 0xFFu) << (8 * e);
            }
            qr[c] = pk;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { hh += __shfl_xor(hh, o, 64); ee += __shfl_xor(ee, o, 64); }
        if (lane == 0) scales[row] = scale;
        best = hh > best ? hh : best;
        best_err = ee > best_err ? ee : best_err;
    }
    if (lane == 0) {
        atomicMax(reinterpret_cast<unsigned*>(norms), __float_as_uint(sqrtf(best) * 1.00001f));
        atomicMax(reinterpret_cast<unsigned*>(norms) + 2, __float_as_uint(sqrtf(best_err) * 1.00001f));
    }
}
__global__ void shadow_i8_finish_kernel(float* __restrict__ norms, int d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) norms[1] = norms[2] + sqrtf((float)d) * 1.6e-5f * norms[0];
}

// One group of R = I8_T * (64 / LPR) rows of the int8 shadow against the query: LPR lanes cover a row (16 bytes each,
// LPR = 16 / 32 / 64 for d <= 256 / 512 / 1024), so one wave instruction loads 64 / LPR whole rows; eight such
// instructions are in flight per group.  Every lane ends up with the score of row `row0 + myr` (valid where `owner`).
constexpr int I8_T = 8, I8_TB = 3;       // wave-loads in flight per group of the int8 scans (and its log2)
template <int LPR>
struct ShadowGroupI8 {
    static constexpr int RPI = 64 / LPR, T = I8_T, R = T * RPI;
    static constexpr int LOGL = (LPR == 64) ? 6 : (LPR == 32) ? 5 : 4;
    int qh[4], ql[4];
    float sq, sl;
    int myr, chunk, sub;
    bool owner, active;
    __device__ void load_query(const float* __restrict__ Q, int d16, int lane) {
        chunk = lane & (LPR - 1);
        sub = lane >> LOGL;
        active = chunk < d16;
        float qv[16];
        float mq = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            qv[e] = active ? Q[chunk * 16 + e] : 0.f;
            mq = fmaxf(mq, fabsf(qv[e]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mq = fmaxf(mq, __shfl_xor(mq, o, 64));
        sq = mq / 127.f;
        sl = sq / 254.f;
        const float inv = mq > 0.f ? 127.f / mq : 0.f, invl = mq > 0.f ? 254.f * 127.f / mq : 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned ph = 0, pl = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float v = qv[w * 4 + b];
                const float h = fminf(fmaxf(rintf(v * inv), -127.f), 127.f);
                const float r = fmaf(-sq, h, v);
                const float l = fminf(fmaxf(rintf(r * invl), -127.f), 127.f);
                ph |= ((unsigned)(int)h & 0xFFu) << (8 * b);
                pl |= ((unsigned)(int)l & 0xFFu) << (8 * b);
            }
            qh[w] = (int)ph;
            ql[w] = (int)pl;
        }
        // transposition over the I8_TB lane bits under the row-select bits, then plain sums over the rest (16 loads in flight
        // instead of 8 measured 6 % slower at k = 10 and 15 % faster at k = 1000: fewer, larger hit groups)
        myr = 0;
        int bit = LOGL - 1;
#pragma unroll
        for (int h = T / 2; h >= 1; h >>= 1, --bit) myr += ((lane >> bit) & 1) * h;
        myr = myr * RPI + sub;
        owner = (lane & ((LPR >> I8_TB) - 1)) == 0;
    }
    __device__ float score(const signed char* __restrict__ Xq, const float* __restrict__ scales, long long row0,
                           long long row_end, int d, int lane) const {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        i32x4_t x[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long row = row0 + t * RPI + sub;
            if (row >= row_end) row = row_end - 1;
            if (active)
                x[t] = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t*>(Xq + row * d) + chunk);
            else
                x[t] = i32x4_t{0, 0, 0, 0};
        }
        long long mrow = row0 + myr;
        if (mrow >= row_end) mrow = row_end - 1;
        const float rs = scales[mrow];
        float a[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int H = 0, L = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                H = __builtin_amdgcn_sdot4(x[t][w], qh[w], H, false);
                L = __builtin_amdgcn_sdot4(x[t][w], ql[w], L, false);
            }
            a[t] = fmaf(sl, (float)L, sq * (float)H);
        }
        int bit = LOGL - 1;
#pragma unroll
        for (int h = T / 2; h >= 1; h >>= 1, --bit) {
            const int m = 1 << bit;
            const bool up = (lane >> bit) & 1;
#pragma unroll
            for (int i = 0; i < h; ++i) {
                float send = up ? a[i] : a[i + h];
                float keep = up ? a[i + h] : a[i];
                a[i] = keep + __shfl_xor(send, m, 64);
            }
        }
        float sc = a[0];
#pragma unroll
        for (int m = (LPR >> (I8_TB + 1)); m >= 1; m >>= 1) sc += __shfl_xor(sc, m, 64);
        return sc * rs;
    }
};

template <int LPR>
__global__ __launch_bounds__(256) void ip_sample_i8_kernel(const signed char* __restrict__ Xq, const float* __restrict__ scales,
                                                           long long n_groups, int d, const float* __restrict__ Q,
                                                           int chunk_shift, long long chunk_stride, float* __restrict__ dump) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    ShadowGroupI8<LPR> grp;
    constexpr int R = ShadowGroupI8<LPR>::R;
    grp.load_query(Q, d >> 4, lane);
    for (long long g = gw; g < n_groups; g += nw) {
        const long long row0 = ((g >> chunk_shift) * chunk_stride + (g & ((1ll << chunk_shift) - 1))) * R;
        const float sc = grp.score(Xq, scales, row0, row0 + R, d, lane);     // whole groups only
        if (grp.owner) dump[g * R + grp.myr] = sc;
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void ip_collect_i8_kernel(const signed char* __restrict__ Xq, const float* __restrict__ scales,
                                                            long long N, int d, const float* __restrict__ Q,
                                                            const float* __restrict__ thr_p, int* __restrict__ counter,
                                                            u64* __restrict__ cand, int cap, long long row_base) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    ShadowGroupI8<LPR> grp;
    constexpr int R = ShadowGroupI8<LPR>::R;
    grp.load_query(Q, d >> 4, lane);
    const float thr = thr_p[0];
    const long long ngroups = (N + R - 1) / R;
    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    for (long long g = gw; g < ngroups; g += nw) {
        const long long row0 = g * R;
        const float sc = grp.score(Xq, scales, row0, N, d, lane);
        const long long row = row0 + grp.myr;
        const bool pass = grp.owner && (row < N) && (sc >= thr);
        const u64 mask = __ballot(pass);
        if (mask != 0) {
            const int first = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(counter, __popcll(mask));
            base = __shfl(base, first, 64);
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (pass && pos < cap) cand[pos] = make_key(sc, (unsigned)(row_base + row));
        }
    }
}

// The SAMPLE pass of the single-query search: evenly spaced chunks of 2^chunk_shift groups of R rows (chunk_stride groups
// apart); every wave scores its share of the sampled groups and writes the best score it saw (wave_best[global wave]).
// The k-th largest of those per-wave maxima is reached by k different sampled rows (sample_threshold_kernel).
// dump != null: the score of sampled row j (j = sampled group * R + row within the group) goes to dump[j] as well — the
// general-k threshold takes the exact k-th largest of ALL sampled scores (sample_threshold_kth_kernel).
template <int NV8, int R>
__global__ __launch_bounds__(256) void ip_sample_bf16_kernel(const uint4* __restrict__ Xb, long long n_groups, int d8,
                                                             const float* __restrict__ Q, int chunk_shift,
                                                             long long chunk_stride, float* __restrict__ wave_best,
                                                             float* __restrict__ dump = nullptr) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    ShadowGroup<NV8, R> grp;
    grp.load_query(Q, d8, lane);
    float best = -3.4028234663852886e38f;
    for (long long g = gw; g < n_groups; g += nw) {
        const long long row0 = ((g >> chunk_shift) * chunk_stride + (g & ((1ll << chunk_shift) - 1))) * R;
        const float sc = grp.score(Xb, row0, row0 + R, d8, lane);     // whole groups only: no ragged edge in a sample
        best = (grp.owner && sc > best) ? sc : best;
        if (dump && grp.owner) dump[g * R + grp.myr] = sc;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (lane == 0) wave_best[gw] = best;
}

// thr = (k-th largest of the n per-wave sample maxima) - 2 eps(q): one block; k rounds of a block-wide maximum
__global__ __launch_bounds__(1024) void sample_threshold_kernel1(const float* __restrict__ wave_best, int n, int k,
                                                                 const float* __restrict__ Q, int d,
                                                                 const float* __restrict__ norms, float* __restrict__ thr) {
    __shared__ float wmax[16];
    __shared__ float wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // keys (score, slot) so that equal scores in different slots stay different rows
    u64 mine[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 1024 + tid;
        if (idx < n) mine[j] = make_key(wave_best[idx], (unsigned)idx);
    }
    float qq = 0.f;
    for (int j = tid; j < d; j += 1024) qq = fmaf(Q[j], Q[j], qq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o, 64);
    if (lane == 0) wsum[wave] = qq;
    __shared__ u64 wk[16];
    u64 L = 0;
    for (int r = 0; r < k; ++r) {
        u64 m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) m = mine[j] > m ? mine[j] : m;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(m, o, 64);
            m = other > m ? other : m;
        }
        if (lane == 0) wk[wave] = m;
        __syncthreads();
        u64 g = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) g = wk[w] > g ? wk[w] : g;
        L = g;
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = (mine[j] == g) ? 0 : mine[j];
        __syncthreads();
    }
    if (tid == 0) {
        qq = 0.f;
        for (int w = 0; w < 16; ++w) qq += wsum[w];
        thr[0] = L != 0 ? f32_unorder((unsigned)(L >> 32)) - 2.f * shadow_eps(norms, d, qq) : -3.4028234663852886e38f;
    }
    (void)wmax;
}

// ------------------------------------------------------------------------------------------------
// One query over the bf16 shadow, THRESHOLD form (the reference's call shape: nq = 1, k <= 16).
// The candidate set is not "the best C rows" (which says nothing when more than C rows sit within the bf16 error of the
// k-th score: frames of one video) but EVERY row whose approximate score could still belong to a top-k row:
//   sample   ip_sample_bf16_kernel over evenly spaced chunks (~64K rows) -> s_A, a score that k sampled rows reach
//            (the k-th largest per-wave maximum).
//            k rows have exact score >= s_A - eps, so the exact k-th best score S* of the index is >= s_A - eps, and a
//            row of the exact top-k has approximate score >= S* - eps >= s_A - 2 eps =: thr.
//   collect  (this kernel) streams all of Xb once and appends (approximate score, row) of every row with score >= thr
//            to one global list (one atomicAdd per wave and hit; a hit is one row in several thousand).
//   rescore  exact f32 dot products of the collected rows; select: the k best of those, written out.
//   refine   the collected list itself gives a far better bound than the sample did: L = the k-th largest of 1024 slice
//            maxima of the collected approximate scores (k different rows reach it), so S* >= L - eps and only rows with
//            approximate score >= L - 2 eps go on — on iid rows ~1500 collected shrink to a few dozen, on clustered rows
//            (where whole runs pass the sample's threshold) tens of thousands shrink to the runs that matter.
//   rescore  exact f32 dot products of what is left; select: the k best of those, written out.
// Exact by construction whatever the data looks like — clustered, near-duplicate, all-equal — as long as the lists hold
// the candidates (COLLECT_CAP collected, RESCORE_CAP after refinement); otherwise the gate is raised and the f32 scan
// queued behind answers.  eps: shadow_eps().
// ------------------------------------------------------------------------------------------------
constexpr int COLLECT_CAP = 262144;         // rows the collect pass may hand on (2 MiB of keys)
constexpr int RESCORE_CAP = 16384;          // rows re-scored = 16 keys per thread of the 1024-thread select kernel
constexpr int SAMPLE_CHUNK_SHIFT = 6;       // a sample chunk = 64 groups of 8 rows = 512 rows (512 KiB at d = 512)
constexpr int SAMPLE_CHUNKS = 128;          // 65536 sampled rows
constexpr long long COLLECT_MIN_ROWS = 1ll << 18;
constexpr int SAMPLE_GRID = 512;            // blocks of the sample scan: 2048 waves x 4 groups, one maximum each   // below this the f32 scan answers directly (a sample would be a quarter of it)

template <int NV8, int R>
__global__ __launch_bounds__(256) void ip_collect_bf16_kernel(const uint4* __restrict__ Xb, long long N, int d8,
                                                              const float* __restrict__ Q, const float* __restrict__ thr_p,
                                                              int* __restrict__ counter, u64* __restrict__ cand, int cap,
                                                              long long row_base = 0 /*index row of Xb's first row*/) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    ShadowGroup<NV8, R> grp;
    grp.load_query(Q, d8, lane);
    const float thr = thr_p[0];
    const long long ngroups = (N + R - 1) / R;
    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    for (long long g = gw; g < ngroups; g += nw) {
        const long long row0 = g * R;
        const float sc = grp.score(Xb, row0, N, d8, lane);
        const long long row = row0 + grp.myr;
        const bool pass = grp.owner && (row < N) && (sc >= thr);
        const u64 mask = __ballot(pass);
        if (mask != 0) {
            const int first = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(counter, __popcll(mask));
            base = __shfl(base, first, 64);
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (pass && pos < cap) cand[pos] = make_key(sc, (unsigned)(row_base + row));
        }
    }
}

__device__ u64 list_kth_score(const u64* __restrict__ cand, int n, int k, unsigned* hist, u64* sh_prefix, int* sh_rem);

// Refinement of the collected list (one block): L = k-th largest of the 1024 threads' slice maxima (k <= 16; for larger k
// the exact k-th largest collected score, list_kth_score), keep what reaches
// L - 2 eps, compacted into cand2 (order irrelevant: the final selection orders by exact score and row).
// ctl: [0] collected (written by the collect pass), [1] gate, [2] kept (written here).
// One query per blockIdx.y (the batched search runs a whole pass of queries through the same three kernels): query q
// uses ctl + 4 q, cand + q cap, cand2 / ekeys + q RESCORE_CAP, Q + q d.  pass_gate (optional): raised when ANY query of
// the launch overflows, for fallbacks that redo the whole pass.
__global__ __launch_bounds__(1024) void collect_refine_kernel(int* __restrict__ ctl, const u64* __restrict__ cand, int cap,
                                                              int k, const float* __restrict__ Q, int d,
                                                              const float* __restrict__ norms, u64* __restrict__ cand2,
                                                              int* __restrict__ stats, int* __restrict__ pass_gate,
                                                              int q_mode /*QMODE_*: how the query entered the scores*/) {
    __shared__ u64 wmax[16];
    __shared__ float wsum[16], wres[16];
    __shared__ int kept;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ctl += 4 * blockIdx.y;
    cand += (size_t)blockIdx.y * cap;
    cand2 += (size_t)blockIdx.y * RESCORE_CAP;
    Q += (size_t)blockIdx.y * d;
    const int n = ctl[0];
    if (n > cap) {
        if (tid == 0) { atomicOr(ctl + 1, 1); if (pass_gate) atomicOr(pass_gate, 1); if (stats) atomicAdd(stats + 1, 1); }
        return;
    }
    if (tid == 0) kept = 0;
    u64 mine = 0;
    for (int i = tid; i < n; i += 1024) {
        const u64 key = cand[i];
        mine = key > mine ? key : mine;
    }
    float qq = 0.f, qr = 0.f;
    for (int j = tid; j < d; j += 1024) {
        qq = fmaf(Q[j], Q[j], qq);
        const float rr = bf16_round_residual(Q[j]);
        qr = fmaf(rr, rr, qr);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { qq += __shfl_xor(qq, o, 64); qr += __shfl_xor(qr, o, 64); }
    if (lane == 0) { wsum[wave] = qq; wres[wave] = qr; }
    u64 L = 0;
    if (k > 16) {
        __shared__ unsigned hist[256];
        __shared__ u64 sh_prefix;
        __shared__ int sh_rem;
        L = list_kth_score(cand, n, k, hist, &sh_prefix, &sh_rem);
    } else
    for (int r = 0; r < k; ++r) {
        u64 m = mine;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(m, o, 64);
            m = other > m ? other : m;
        }
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        u64 g = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) g = wmax[w] > g ? wmax[w] : g;
        L = g;
        if (mine == g) mine = 0;
        __syncthreads();
    }
    qq = 0.f; qr = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { qq += wsum[w]; qr += wres[w]; }
    // fewer than k non-empty slices (n < k cannot happen: the sampled rows themselves are collected): keep everything
    const float eps = query_eps(norms, d, qq, qr, q_mode);
    const float t2 = L != 0 ? f32_unorder((unsigned)(L >> 32)) - 2.f * eps : -3.4028234663852886e38f;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const u64 key = i < n ? cand[i] : 0;
        const bool pass = key != 0 && f32_unorder((unsigned)(key >> 32)) >= t2;
        const u64 mask = __ballot(pass);
        if (mask != 0) {
            const int first = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(&kept, __popcll(mask));
            base = __shfl(base, first, 64);
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (pass && pos < RESCORE_CAP) cand2[pos] = key;
        }
    }
    __syncthreads();
    if (tid == 0) {
        ctl[2] = kept;
        if (kept > RESCORE_CAP) { atomicOr(ctl + 1, 1); if (pass_gate) atomicOr(pass_gate, 1); if (stats) atomicAdd(stats + 1, 1); }
    }
}

// exact f32 scores of the kept rows: a wave per candidate (two in flight), grid-stride; ekeys[i] = (exact score, row)
__global__ __launch_bounds__(256) void collect_rescore_kernel(const float* __restrict__ X, int d, const float* __restrict__ Q,
                                                              const int* __restrict__ ctl,
                                                              const u64* __restrict__ cand, u64* __restrict__ ekeys) {
    ctl += 4 * blockIdx.y;
    cand += (size_t)blockIdx.y * RESCORE_CAP;
    ekeys += (size_t)blockIdx.y * RESCORE_CAP;
    Q += (size_t)blockIdx.y * d;
    if (ctl[1] != 0 || ctl[3] != 0) return;     // a list overflowed (the f32 scan answers), or the one-block finish answered
    const int n = ctl[2];
    const int lane = threadIdx.x & 63;
    const int d4 = d >> 2;
    const float4* qv = reinterpret_cast<const float4*>(Q);
    const int w0 = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int i0 = w0 * 2; i0 < n; i0 += nw * 2) {
        long long rows[2];
        float p[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
            rows[u] = i0 + u < n ? (long long)(0xFFFFFFFFu - (unsigned)(cand[i0 + u] & 0xFFFFFFFFull)) : -1;
        for (int j = lane; j < d4; j += 64) {
            const float4 b = qv[j];
            float4 a[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                a[u] = rows[u] >= 0 ? reinterpret_cast<const float4*>(X + (size_t)rows[u] * d)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                p[u] = fmaf(a[u].x, b.x, p[u]); p[u] = fmaf(a[u].y, b.y, p[u]);
                p[u] = fmaf(a[u].z, b.z, p[u]); p[u] = fmaf(a[u].w, b.w, p[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) p[u] += __shfl_xor(p[u], o, 64);
            if (lane == 0 && rows[u] >= 0) ekeys[i0 + u] = make_key(p[u], (unsigned)rows[u]);
        }
    }
}

// the k best of the n <= RESCORE_CAP exact keys (16 per thread in registers, k rounds of a block-wide maximum), written
// as (score, id); counters: [0] += 1 when answered here ([1] was raised by the refinement when a list overflowed)
__global__ __launch_bounds__(1024) void collect_select_kernel(const int* __restrict__ ctl,
                                                              const u64* __restrict__ ekeys, int k,
                                                              const long long* __restrict__ ids, long long id_base,
                                                              float* __restrict__ outD, long long* __restrict__ outI,
                                                              int* __restrict__ stats) {
    __shared__ u64 wmax[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ctl += 4 * blockIdx.y;
    ekeys += (size_t)blockIdx.y * RESCORE_CAP;
    outD += (size_t)blockIdx.y * k;
    outI += (size_t)blockIdx.y * k;
    if (ctl[1] != 0) return;
    const int n = ctl[2];
    if (n <= 1024) {
        // the usual case (a few dozen rows survive the refinement): rank by counting — every thread holds one key and
        // counts the keys above it (LDS broadcast reads); rank r < k writes output r.  One barrier.
        __shared__ u64 keys[1024];
        const u64 mykey = tid < n ? ekeys[tid] : 0;
        keys[tid] = mykey;
        __syncthreads();
        if (tid < n) {
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += keys[j] > mykey;
            if (rank < k) {
                const long long row = (long long)(0xFFFFFFFFu - (unsigned)(mykey & 0xFFFFFFFFull));
                outD[rank] = f32_unorder((unsigned)(mykey >> 32));
                outI[rank] = ids ? ids[row] : id_base + row;
            }
        }
        if (tid >= n && tid < k) {        // fewer rows than k: padding
            outD[tid] = -3.4028234663852886e38f;
            outI[tid] = -1;
        }
        if (tid == 0 && stats) atomicAdd(stats, 1);
        return;
    }
    constexpr int PER = RESCORE_CAP / 1024;
    u64 mine[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = j * 1024 + tid;
        mine[j] = idx < n ? ekeys[idx] : 0;
    }
    for (int r = 0; r < k; ++r) {
        u64 m = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) m = mine[j] > m ? mine[j] : m;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(m, o, 64);
            m = other > m ? other : m;
        }
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        u64 g = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) g = wmax[w] > g ? wmax[w] : g;
        if (tid == 0) {
            if (g != 0) {
                const long long row = (long long)(0xFFFFFFFFu - (unsigned)(g & 0xFFFFFFFFull));
                outD[r] = f32_unorder((unsigned)(g >> 32));
                outI[r] = ids ? ids[row] : id_base + row;
            } else {
                outD[r] = -3.4028234663852886e38f;
                outI[r] = -1;
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) mine[j] = (mine[j] == g) ? 0 : mine[j];   // keys are unique (the row is part of them)
        __syncthreads();
    }
    if (tid == 0 && stats) atomicAdd(stats, 1);
}

// ------------------------------------------------------------------------------------------------
// General k (the k the reference's server and evaluations send: REST `end` = 20, api/routes.py:1171,1407; k = 100,
// docs/Search-Index-Evaluation.md:109; --topk 1000, docs/Retrieval-Evaluation.md:39).  The threshold form itself does not
// care about k; what did were the selections (k rounds of a block-wide maximum).  They are radix selections here.
// ------------------------------------------------------------------------------------------------
// k-th largest of the 64-bit keys a 1024-thread block holds (PER per thread, 0 = empty slot; keys are unique because the
// row is part of them): eight byte-wise histogram passes from the top byte down.  Returns the key (0 if fewer than k
// non-empty keys).  hist: 256 words, sh: 2 u64 + 1 int of shared memory.  All threads must call it.
template <int PER>
__device__ u64 block_kth_largest_key(const u64 (&mine)[PER], int k, unsigned* hist, u64* sh_prefix, int* sh_rem) {
    const int tid = threadIdx.x;
    if (tid == 0) { *sh_prefix = 0; *sh_rem = k; }
    u64 mask = 0;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const u64 prefix = *sh_prefix;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (mine[j] != 0 && (mine[j] & mask) == prefix) atomicAdd(&hist[(unsigned)(mine[j] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < 64) {
            // suffix sums over the 256 buckets, four per lane (lane 63 holds buckets 252..255)
            unsigned h[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) h[b] = hist[tid * 4 + b];
            const unsigned own = h[0] + h[1] + h[2] + h[3];
            unsigned above = own;                      // inclusive suffix over lanes >= tid
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(above, o, 64);
                if (tid + o < 64) above += t;
            }
            above -= own;                              // buckets of higher lanes only
            const unsigned rem = (unsigned)*sh_rem;
            // the bucket where the count from the top reaches rem: exactly one lane finds it (if rem <= total)
            unsigned cum = above;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                if (cum < rem && cum + h[b] >= rem) {
                    *sh_rem = (int)(rem - cum);
                    *sh_prefix = prefix | ((u64)(tid * 4 + b) << shift);
                }
                cum += h[b];
            }
            if (tid == 0 && cum < rem) *sh_rem = -1;   // fewer than k keys in all
        }
        mask |= (u64)255 << shift;
        __syncthreads();
        if (*sh_rem < 0) return 0;
    }
    return *sh_prefix;
}

// the same over 32-bit keys (ordered scores, duplicates counted): four passes
template <int PER>
__device__ unsigned block_kth_largest_u32(const unsigned (&mine)[PER], const bool (&live)[PER], int k, unsigned* hist,
                                          unsigned* sh_prefix, int* sh_rem) {
    const int tid = threadIdx.x;
    if (tid == 0) { *sh_prefix = 0; *sh_rem = k; }
    unsigned mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = *sh_prefix;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (live[j] && (mine[j] & mask) == prefix) atomicAdd(&hist[(mine[j] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < 64) {
            unsigned h[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) h[b] = hist[tid * 4 + b];
            const unsigned own = h[0] + h[1] + h[2] + h[3];
            unsigned above = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(above, o, 64);
                if (tid + o < 64) above += t;
            }
            above -= own;
            const unsigned rem = (unsigned)*sh_rem;
            unsigned cum = above;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                if (cum < rem && cum + h[b] >= rem) {
                    *sh_rem = (int)(rem - cum);
                    *sh_prefix = prefix | ((unsigned)(tid * 4 + b) << shift);
                }
                cum += h[b];
            }
            if (tid == 0 && cum < rem) *sh_rem = -1;
        }
        mask |= 255u << shift;
        __syncthreads();
        if (*sh_rem < 0) return 0;
    }
    return *sh_prefix;
}

// thr = (k-th largest of the n dumped sample scores) - 2 eps(q); zeroes the control words of the query (ctl[0..3]) on
// the way: one block, 64 scores per thread at n = 65536.
__global__ __launch_bounds__(1024) void sample_threshold_kth_kernel(const float* __restrict__ dump, int n, int k,
                                                                    const float* __restrict__ Q, int d,
                                                                    const float* __restrict__ norms, float* __restrict__ thr,
                                                                    int* __restrict__ ctl) {
    // 8192 disjoint segments of the sampled scores (8 per thread, n / 8192 scores each: segment s = elements
    // s, s + 8192, ...), their maxima, and the k-th largest of those: k DIFFERENT sampled rows reach it.  Against the
    // exact k-th largest sampled score this loses only the top scores that share a segment (k^2 / 16384 of them on
    // average: 61 at k = 1000, none to speak of at k = 20) and selects among 8 keys per thread instead of 64.
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix;
    __shared__ int sh_rem;
    __shared__ float wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 8, SEGS = 8192;
    unsigned mine[PER];
    bool live[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int seg = j * 1024 + tid;
        float m = -3.4028234663852886e38f;
        // (n <= 65536 = SAMPLE_CHUNKS x 512: at most eight scores per segment, all eight loads of all eight segments in
        // flight at once — as a counted loop this was 64 dependent round trips to L2, 20 of the kernel's 29 us)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = seg + t * SEGS;
            m = fmaxf(m, i < n ? dump[i] : -3.4028234663852886e38f);
        }
        for (int i = seg + 8 * SEGS; i < n; i += SEGS) m = fmaxf(m, dump[i]);
        live[j] = seg < n;
        mine[j] = f32_order(m);
    }
    float qq = 0.f;
    for (int j = tid; j < d; j += 1024) qq = fmaf(Q[j], Q[j], qq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o, 64);
    if (lane == 0) wsum[wave] = qq;
    if (tid < 4) ctl[tid] = 0;
    const unsigned L = block_kth_largest_u32<PER>(mine, live, k, hist, &sh_prefix, &sh_rem);
    if (tid == 0) {
        qq = 0.f;
        for (int w = 0; w < 16; ++w) qq += wsum[w];
        // (fewer than k segments: no threshold — everything is collected, the lists overflow, the f32 scan answers)
        thr[0] = sh_rem >= 0 ? f32_unorder(L) - 2.f * shadow_eps(norms, d, qq) : -3.4028234663852886e38f;
    }
}

// score part (upper 32 bits of the key, lower half zero) of the k-th largest of the n keys of a list in global memory
// (L2-resident), 0 if the list is shorter than k: four byte passes over the list by a 1024-thread block; the bucket scan
// of a pass is wave 0's (four buckets per lane).  hist[256], sh_prefix, sh_rem: shared.  All threads must call it.
__device__ u64 list_kth_score(const u64* __restrict__ cand, int n, int k, unsigned* hist, u64* sh_prefix, int* sh_rem) {
    const int tid = threadIdx.x;
    if (tid == 0) { *sh_prefix = 0; *sh_rem = k; }
    u64 mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const u64 prefix = *sh_prefix;
        for (int i0 = 0; i0 < n; i0 += 8 * 1024) {
            // Eight keys per thread are loaded before any is counted: one key per trip made a pass a chain of n / 1024
            // dependent round trips to L2 (35 of them with the int8 shadow's lists: 14 us per byte pass).
            u64 keys8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 1024 + tid;
                keys8[u] = i < n ? cand[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // The collected scores sit in a narrow band, so in the upper byte passes nearly every key falls into one
                // or two buckets: a wave first counts the bucket of its first live lane — and then of the next — with a
                // ballot and adds the count once; what is left goes one by one.
                const u64 key = keys8[u];
                bool todo = i0 + u * 1024 + tid < n && (key & mask) == prefix;
                const unsigned b = (unsigned)(key >> shift) & 255u;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const u64 act = __ballot(todo);
                    if (act == 0) break;
                    const int leader = __ffsll((long long)act) - 1;
                    const unsigned lb = (unsigned)__shfl((int)b, leader, 64);
                    const u64 same = __ballot(todo && b == lb);
                    if ((tid & 63) == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
                    todo = todo && b != lb;
                }
                if (todo) atomicAdd(&hist[b], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {
            unsigned h[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) h[b] = hist[tid * 4 + b];
            const unsigned own = h[0] + h[1] + h[2] + h[3];
            unsigned above = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(above, o, 64);
                if (tid + o < 64) above += t;
            }
            above -= own;
            const unsigned rem = (unsigned)*sh_rem;
            unsigned cum = above;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                if (cum < rem && cum + h[b] >= rem) {
                    *sh_rem = (int)(rem - cum);
                    *sh_prefix = prefix | ((u64)(tid * 4 + b) << shift);
                }
                cum += h[b];
            }
            if (tid == 0 && cum < rem) *sh_rem = -1;
        }
        mask |= (u64)255 << shift;
        __syncthreads();
        if (*sh_rem < 0) return 0;
    }
    return *sh_prefix;
}

// Between the two ranges of a large-k collect pass: thr = max(thr, (k-th largest score collected from the first range)
// - 2 eps).  The first range is a sample sixteen times the sample pass's, so the second range collects a few k rows
// instead of N / 65536 * k (150,000 at k = 1000 over 10M rows, which also slowed the scan by its appends).
__global__ __launch_bounds__(1024) void collect_tighten_kernel(const int* __restrict__ ctl, const u64* __restrict__ cand, int cap,
                                                               int k, const float* __restrict__ Q, int d,
                                                               const float* __restrict__ norms, float* __restrict__ thr) {
    __shared__ unsigned hist[256];
    __shared__ u64 sh_prefix;
    __shared__ int sh_rem;
    __shared__ float wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = ctl[0];
    if (n > cap) return;                       // overflow already: the finish kernel raises the gate
    float qq = 0.f;
    for (int j = tid; j < d; j += 1024) qq = fmaf(Q[j], Q[j], qq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o, 64);
    if (lane == 0) wsum[wave] = qq;
    const u64 L = list_kth_score(cand, n, k, hist, &sh_prefix, &sh_rem);
    if (tid == 0 && L != 0) {
        qq = 0.f;
        for (int w = 0; w < 16; ++w) qq += wsum[w];
        const float t2 = f32_unorder((unsigned)(L >> 32)) - 2.f * shadow_eps(norms, d, qq);
        if (t2 > thr[0]) thr[0] = t2;
    }
}

constexpr int FINISH_LDS_ROWS = 2048;       // survivors the one-block finish can hold (16 KiB of keys)
constexpr int FINISH_OWN_ROWS = 512;        // ... and re-scores itself (1 MiB of f32 rows through one CU); more: the multi-block kernels

// Everything behind the collect pass of ONE query in one block (refine -> exact scores -> the k best), k <= 1024:
//   refine   T2 = (k-th largest collected approximate score) - 2 eps by radix selection over the collected list (read
//            from L2 once per byte pass), survivors (approximate score >= T2) compacted;
//   rescore  <= FINISH_LDS_ROWS survivors: their exact f32 scores here, a wave per row, two rows in flight — the same
//            per-lane fmaf chain and butterfly as collect_rescore_kernel, i.e. the f32 scan's bits;
//   select   radix selection of the k-th exact key, the winners ranked by counting, written as (score, id).
// More survivors (near-duplicate runs: up to RESCORE_CAP) go to cand2 with ctl[2] = their number and ctl[3] = 0: the
// multi-block collect_rescore_kernel and collect_select_kernel queued behind take over (they return at once when
// ctl[3] != 0 = answered here).  Overflow of either list raises the gate (ctl[1]) for the f32 scan behind them.
__global__ __launch_bounds__(1024) void collect_finish_kernel(int* __restrict__ ctl, const u64* __restrict__ cand, int cap,
                                                              int k, const float* __restrict__ Q, int d,
                                                              const float* __restrict__ X, const float* __restrict__ norms,
                                                              u64* __restrict__ cand2, const long long* __restrict__ ids,
                                                              long long id_base, float* __restrict__ outD,
                                                              long long* __restrict__ outI, int* __restrict__ stats) {
    __shared__ unsigned hist[256];
    __shared__ u64 sh_prefix;
    __shared__ int sh_rem;
    __shared__ float wsum[16];
    __shared__ int kept;
    __shared__ u64 keys[FINISH_LDS_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = ctl[0];
    if (n > cap) {
        if (tid == 0) { atomicOr(ctl + 1, 1); ctl[3] = 1; if (stats) atomicAdd(stats + 1, 1); }
        return;
    }
    if (tid == 0) kept = 0;
    float qq = 0.f;
    for (int j = tid; j < d; j += 1024) qq = fmaf(Q[j], Q[j], qq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o, 64);
    if (lane == 0) wsum[wave] = qq;
    // ---- refine: the k-th largest collected score by byte passes over the list (n up to cap keys, L2-resident)
    const u64 L = list_kth_score(cand, n, k, hist, &sh_prefix, &sh_rem);
    qq = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) qq += wsum[w];
    const float t2 = L != 0 ? f32_unorder((unsigned)(L >> 32)) - 2.f * shadow_eps(norms, d, qq) : -3.4028234663852886e38f;
    // ---- survivors: into LDS while they fit, into cand2 always (the multi-block path reads them there)
    for (int i0 = 0; i0 < n; i0 += 8 * 1024) {
        u64 keys8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024 + tid;
            keys8[u] = i < n ? cand[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u64 key = keys8[u];
            const bool pass = key != 0 && f32_unorder((unsigned)(key >> 32)) >= t2;
            const u64 bal = __ballot(pass);
            if (bal != 0) {
                const int first = __ffsll((long long)bal) - 1;
                int base = 0;
                if (lane == first) base = atomicAdd(&kept, __popcll(bal));
                base = __shfl(base, first, 64);
                const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (pass && pos < FINISH_LDS_ROWS) keys[pos] = key;
                if (pass && pos < RESCORE_CAP) cand2[pos] = key;
            }
        }
    }
    __syncthreads();
    const int nk = kept;
    if (nk > FINISH_OWN_ROWS) {
        if (tid == 0) {
            ctl[2] = nk;
            if (nk > RESCORE_CAP) { atomicOr(ctl + 1, 1); ctl[3] = 1; if (stats) atomicAdd(stats + 1, 1); }
            else ctl[3] = 0;                            // the multi-block kernels behind answer
        }
        return;
    }
    // ---- exact scores of the survivors: wave per row, EIGHT in flight (collect_rescore_kernel's arithmetic: a row's
    // per-lane fmaf chain and butterfly do not depend on how many rows travel together; with two in flight a few hundred
    // survivors were a dozen dependent round trips to HBM)
    {
        const int d4 = d >> 2;
        const float4* qv = reinterpret_cast<const float4*>(Q);
        constexpr int RF = 8;
        for (int i0 = wave * RF; i0 < nk; i0 += 16 * RF) {
            long long rows[RF];
            float p[RF];
#pragma unroll
            for (int u = 0; u < RF; ++u) {
                rows[u] = i0 + u < nk ? (long long)(0xFFFFFFFFu - (unsigned)(keys[i0 + u] & 0xFFFFFFFFull)) : -1;
                p[u] = 0.f;
            }
            for (int j = lane; j < d4; j += 64) {
                const float4 b = qv[j];
                float4 a[RF];
#pragma unroll
                for (int u = 0; u < RF; ++u)
                    a[u] = rows[u] >= 0 ? reinterpret_cast<const float4*>(X + (size_t)rows[u] * d)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    p[u] = fmaf(a[u].x, b.x, p[u]); p[u] = fmaf(a[u].y, b.y, p[u]);
                    p[u] = fmaf(a[u].z, b.z, p[u]); p[u] = fmaf(a[u].w, b.w, p[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < RF; ++u) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) p[u] += __shfl_xor(p[u], o, 64);
                if (lane == 0 && rows[u] >= 0) keys[i0 + u] = make_key(p[u], (unsigned)rows[u]);
            }
        }
    }
    __syncthreads();
    // ---- the k best exact keys: selection, then rank by counting among the winners
    u64 mine[2];
    mine[0] = tid < nk ? keys[tid] : 0;
    mine[1] = tid + 1024 < nk ? keys[tid + 1024] : 0;
    const int kk = k < nk ? k : nk;
    const u64 kth = kk > 0 ? block_kth_largest_key<2>(mine, kk, hist, &sh_prefix, &sh_rem) : ~0ull;
    __syncthreads();
    if (tid == 0) kept = 0;
    __syncthreads();
    // winners (exactly kk of them: keys are unique) compacted to the front of `keys`
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool win = mine[j] != 0 && mine[j] >= kth;
        const u64 bal = __ballot(win);
        int base = 0;
        if (bal != 0) {
            const int first = __ffsll((long long)bal) - 1;
            if (lane == first) base = atomicAdd(&kept, __popcll(bal));
            base = __shfl(base, first, 64);
        }
        __syncthreads();                                // every key is in a register before its slot may be rewritten
        if (win) keys[base + __popcll(bal & ((1ull << lane) - 1ull))] = mine[j];
        __syncthreads();
    }
    if (tid < kk) {
        const u64 mykey = keys[tid];
        int rank = 0;
        for (int j = 0; j < kk; ++j) rank += keys[j] > mykey;
        const long long row = (long long)(0xFFFFFFFFu - (unsigned)(mykey & 0xFFFFFFFFull));
        outD[rank] = f32_unorder((unsigned)(mykey >> 32));
        outI[rank] = ids ? ids[row] : id_base + row;
    }
    for (int j = kk + tid; j < k; j += 1024) {          // fewer rows than k: padding
        outD[j] = -3.4028234663852886e38f;
        outI[j] = -1;
    }
    if (tid == 0) { ctl[2] = nk; ctl[3] = 1; if (stats) atomicAdd(stats, 1); }
}

// the k best (k <= 1024) of n <= RESCORE_CAP exact keys by radix selection + ranking of the winners: the multi-block
// path's last kernel for any k (collect_select_kernel's k rounds of a block-wide maximum are its k <= 16 form)
__global__ __launch_bounds__(1024) void collect_select_kth_kernel(const int* __restrict__ ctl, const u64* __restrict__ ekeys,
                                                                  int k, const long long* __restrict__ ids, long long id_base,
                                                                  float* __restrict__ outD, long long* __restrict__ outI,
                                                                  int* __restrict__ stats) {
    __shared__ unsigned hist[256];
    __shared__ u64 sh_prefix;
    __shared__ int sh_rem;
    __shared__ int cnt;
    __shared__ u64 win[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    ctl += 4 * blockIdx.y;                               // one query per blockIdx.y (the batched passes)
    ekeys += (size_t)blockIdx.y * RESCORE_CAP;
    outD += (size_t)blockIdx.y * k;
    outI += (size_t)blockIdx.y * k;
    if (ctl[1] != 0 || ctl[3] != 0) return;             // overflow (the f32 scan answers) or answered by the finish kernel
    const int n = ctl[2];
    constexpr int PER = RESCORE_CAP / 1024;
    u64 mine[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = j * 1024 + tid;
        mine[j] = idx < n ? ekeys[idx] : 0;
    }
    const int kk = k < n ? k : n;
    if (tid == 0) cnt = 0;
    const u64 kth = kk > 0 ? block_kth_largest_key<PER>(mine, kk, hist, &sh_prefix, &sh_rem) : ~0ull;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool w = mine[j] != 0 && mine[j] >= kth;
        const u64 bal = __ballot(w);
        if (bal != 0) {
            const int first = __ffsll((long long)bal) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(&cnt, __popcll(bal));
            base = __shfl(base, first, 64);
            if (w) win[base + __popcll(bal & ((1ull << lane) - 1ull))] = mine[j];
        }
    }
    __syncthreads();
    if (tid < kk) {
        const u64 mykey = win[tid];
        int rank = 0;
        for (int j = 0; j < kk; ++j) rank += win[j] > mykey;
        const long long row = (long long)(0xFFFFFFFFu - (unsigned)(mykey & 0xFFFFFFFFull));
        outD[rank] = f32_unorder((unsigned)(mykey >> 32));
        outI[rank] = ids ? ids[row] : id_base + row;
    }
    for (int j = kk + tid; j < k; j += 1024) {
        outD[j] = -3.4028234663852886e38f;
        outI[j] = -1;
    }
    if (tid == 0 && stats) atomicAdd(stats, 1);
}

static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
// a list must be able to take one 64-wide offer on top of k survivors
static int list_cap(int k) { return next_pow2(k + 64); }

struct ScanPlan {
    int cap, nq_per_pass, grid, rows;
    size_t lds;
};

static int g_scan_rows = 4, g_scan_blocks_per_cu = 0;  // tuning knobs (wise_debug_set_scan)
static int g_use_mfma = 1;                              // batched queries on the matrix cores
static long long g_scan_sample = 32768;                  // rows of the split scan's sample pass (0 = none)
static int g_stage2_factor = 8;                          // batched bf16 scan: second row range = factor x the threshold sample
static int g_use_qb64 = 1;                               // 64 queries per pass when more than 32 are waiting
static int g_use_split = 1;                             // ... as split-bf16 candidates + exact re-scoring (k <= MFMA_KC)
extern int g_mfma_abl, g_split_direct, g_shadow_one_piece;

static ScanPlan plan_scan(long long N, int d, int nq, int k) {
    ScanPlan p;
    p.cap = list_cap(k);
    const int nv = (d / 4 + 63) / 64;
    // queries per pass: registers (NV*NQ <= 8 keeps the kernel spill-free) and LDS
    // (4 waves * NQ * cap * 8 B <= 64 KiB so two blocks fit a CU)
    int nqp = 1;
    while (nqp * 2 <= nq && nqp * 2 <= 4 && nv * nqp * 2 <= 8 && (size_t)4 * nqp * 2 * p.cap * 8 <= 64 * 1024) nqp <<= 1;
    p.nq_per_pass = nqp;
    p.lds = (size_t)4 * nqp * p.cap * 8;
    p.rows = (g_scan_rows == 8 && nqp == 1 && nv <= 2) ? 8 : 4;
    int blocks_per_cu = (p.lds > 40 * 1024) ? 2 : 4;
    // 3-KiB rows (d = 768): a grid of 768 blocks instead of 1024 spreads the row stream over the memory channels —
    // 3.57 -> 4.89 TB/s measured on 6.25M x 768 (grids of 512, 1024 and 2048 blocks all sit at 3.6)
    if (nv == 3 && blocks_per_cu == 4) blocks_per_cu = 3;
    if (g_scan_blocks_per_cu > 0) blocks_per_cu = g_scan_blocks_per_cu;
    p.grid = 256 * blocks_per_cu;
    long long ngroups = (N + p.rows - 1) / p.rows;
    long long need = (ngroups + 3) / 4;
    if (need < 1) need = 1;
    if (p.grid > need) p.grid = (int)need;
    return p;
}

template <int NV, int NQ>
static void launch_scan(const ScanPlan& p, const float* X, long long N, int d, const float* Q, int k, u64* part,
                        hipStream_t st, const int* gate = nullptr) {
    SegArgs sa{};
    sa.gate = gate;
    if constexpr (NQ == 1 && NV <= 2) {
        if (p.rows == 8) {
            hipLaunchKernelGGL((ip_scan_kernel<NV, NQ, 8>), dim3(p.grid), dim3(256), p.lds, st,
                               reinterpret_cast<const f32x4*>(X), N, d / 4, Q, k, p.cap, part, sa);
            return;
        }
    }
    auto kern = ip_scan_kernel<NV, NQ, 4>;
    if (p.lds > 48 * 1024)
        raise_lds_limit(reinterpret_cast<const void*>(kern), (int)p.lds);
    hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds, st, reinterpret_cast<const f32x4*>(X), N, d / 4, Q, k,
                       p.cap, part, sa);
}

template <int NV>
static void launch_seg_scan(const float* X, int d, const float* Q, int k, int cap, u64* part, const SegArgs& seg,
                            hipStream_t st) {
    auto kern = ip_scan_kernel<NV, 1, 4, true>;
    const size_t lds = (size_t)4 * cap * 8;
    if (lds > 48 * 1024)
        raise_lds_limit(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((long long)seg.nq * seg.nprobe)), dim3(256), lds, st,
                       reinterpret_cast<const f32x4*>(X), 0ll, d / 4, Q, k, cap, part, seg);
}

template <int NV>
static int dispatch_nq(const ScanPlan& p, const float* X, long long N, int d, const float* Q, int k, u64* part,
                       hipStream_t st) {
    switch (p.nq_per_pass) {
        case 1: launch_scan<NV, 1>(p, X, N, d, Q, k, part, st); return 0;
        case 2:
            if constexpr (NV * 2 <= 8) { launch_scan<NV, 2>(p, X, N, d, Q, k, part, st); return 0; }
            break;
        case 4:
            if constexpr (NV * 4 <= 8) { launch_scan<NV, 4>(p, X, N, d, Q, k, part, st); return 0; }
            break;
    }
    return WISE_E_INVALID;
}

}  // namespace wise

using namespace wise;

#ifdef WISE_DEBUG_KNOBS
extern "C" int wise_debug_set_scan(int rows, int blocks_per_cu) {
    g_scan_rows = rows & 0xFF;
    g_scan_blocks_per_cu = blocks_per_cu & 0xFF;
    g_stage2_factor = (blocks_per_cu >> 8) & 0xFFF ? (blocks_per_cu >> 8) & 0xFFF : 8;
    g_use_mfma = (rows >> 8) & 1 ? 0 : 1;  // bit 8: force the VALU kernel for batched queries
    g_mfma_abl = ((rows >> 9) & 3) | (((rows >> 27) & 1) << 2);   // bit 27: (timing experiment) tiled addressing in the shadow scan
    g_use_split = (rows >> 11) & 1 ? 0 : 1;
    g_split_direct = (rows >> 12) & 15 ? ((rows >> 12) & 15) % 8 : 4;   // bits 12-15: queue depth 3/4/6; 8 = DMA ring
    if (((rows >> 12) & 15) == 8) g_split_direct = 0;
    g_use_qb64 = (rows >> 25) & 1 ? 0 : 1;   // bit 25: never 64 queries per pass
    g_shadow_one_piece = (rows >> 26) & 1 ? 0 : 1;   // bit 26: two-piece bf16 queries in the batched shadow scan
    g_scan_sample = (rows >> 16) & 1 ? 0 : ((rows >> 17) & 0xFF ? (long long)((rows >> 17) & 0xFF) * 16384 : 32768);  // bit 16: no sample pass; bits 17-24: sample rows / 16384  // bit 11: f32 matrix-core scan instead of the split-bf16 candidates
    return 0;
}
#endif

extern "C" size_t wise_ip_topk_workspace_bytes(int64_t N, int d, int nq, int k) {
    if (N < 0 || d < 4 || nq < 1 || k < 1 || k > 2048) return 0;
    ScanPlan p = plan_scan(N, d, nq, k);
    // part keys [grid][nq_per_pass][k] + padded query block
    size_t valu = align_up((size_t)p.grid * p.nq_per_pass * k * sizeof(u64), 256) +
                  align_up((size_t)p.nq_per_pass * d * sizeof(float), 256) + 256;
    if (g_use_mfma && mfma_scan_supported(d, nq, k)) {
        // the split path keeps MFMA_KL keys per list and a candidate block [32][MFMA_KL] of (score, row)
        // (lists of the sample pass and of the main pass, a 32-key threshold block)
        size_t mf = align_up(2 * mfma_scan_part_bytes(N, MFMA_KL), 256) + align_up((size_t)MFMA_QB2 * d * sizeof(float), 256) +
                    align_up((size_t)MFMA_QB2 * MFMA_KL * 12, 256) + 512 + 256;
        if (mf > valu) valu = mf;
    }
    return valu;
}

extern "C" int wise_ip_topk_f32(const float* X, int64_t N, int d, const float* Q, int nq, int k, const int64_t* ids,
                                int64_t id_base, float* outD, int64_t* outI, void* workspace, size_t workspace_bytes,
                                void* stream) {
    WISE_CHECK_ARG(d >= 4 && d <= 2048 && d % 4 == 0, "ip_topk: d=%d must be a multiple of 4 in [4,2048]", d);
    WISE_CHECK_ARG(k >= 1 && k <= 2048, "ip_topk: k=%d out of [1,2048]", k);
    WISE_CHECK_ARG(nq >= 1 && nq <= 1024, "ip_topk: nq=%d out of [1,1024]", nq);
    WISE_CHECK_ARG(N >= 0 && N < 0xFFFFFFFFll, "ip_topk: N=%lld out of range", (long long)N);
    WISE_CHECK_ARG(Q && outD && outI && (X || N == 0), "ip_topk: null pointer");
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0, "ip_topk: X and Q must be 16-byte aligned");
    size_t need = wise_ip_topk_workspace_bytes(N, d, nq, k);
    if (!workspace || workspace_bytes < need) {
        set_error("ip_topk: workspace %zu < %zu bytes", workspace_bytes, need);
        return WISE_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (g_use_mfma && N > 0 && mfma_scan_supported(d, nq, k)) {
        // batched path: 32 or 64 queries share one pass over X on the matrix cores
        const bool split = g_use_split && mfma_split_supported(d, nq, k);
        const bool direct = split && split_direct_enabled();
        const int kl = split ? MFMA_KL : k;
        unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
        u64* mpart = reinterpret_cast<u64*>(wsb);
        size_t off = align_up(2 * mfma_scan_part_bytes(N, MFMA_KL), 256);
        float* mq = reinterpret_cast<float*>(wsb + off);
        off += align_up((size_t)MFMA_QB2 * d * sizeof(float), 256);
        long long* cand_rows = reinterpret_cast<long long*>(wsb + off);
        float* cand_scores = reinterpret_cast<float*>(wsb + off + (size_t)MFMA_QB2 * MFMA_KL * 8);
        off += align_up((size_t)MFMA_QB2 * MFMA_KL * 12, 256);
        u64* tau0 = reinterpret_cast<u64*>(wsb + off);
        const int cap = list_cap(kl);
        int mwv = 8192 / cap;
        if (mwv < 1) mwv = 1;
        if (mwv > 16) mwv = 16;
        for (int q0 = 0; q0 < nq;) {
            // 64 at a time while more than 32 remain (d <= 512 keeps both bf16 images of 64 queries in LDS)
            const int qb = (direct && g_use_qb64 && split64_supported(d) && nq - q0 > MFMA_QB) ? MFMA_QB2 : MFMA_QB;
            const int nqa = (nq - q0 < qb) ? nq - q0 : qb;
            hipError_t e = hipSuccess;
            if (nqa < qb) e = hipMemsetAsync(mq, 0, (size_t)qb * d * sizeof(float), st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(mq, Q + (size_t)q0 * d, (size_t)nqa * d * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) { set_error("ip_topk: query staging: %s", hipGetErrorString(e)); return (int)e; }
            int nlists = mfma_scan_lists(N);
            if (direct) {
                // sample pass over the first rows: its MFMA_KL-th candidate of a query is a threshold nothing in the
                // final top MFMA_KL can fall below, so the main pass (the other rows) hardly ever touches its lists
                const long long ns = (g_scan_sample && N >= 8ll * g_scan_sample) ? g_scan_sample : 0;
                auto scan = [&](const float* Xp, long long n, long long row_off, u64* dst, const u64* t0) {
                    return qb == MFMA_QB2 ? split64_scan_launch(Xp, n, row_off, d, mq, nqa, dst, t0, st)
                                          : split_scan_launch(Xp, n, row_off, d, mq, nqa, dst, t0, st);
                };
                auto lists_of = [&](long long n) { return qb == MFMA_QB2 ? split64_lists(n) : mfma_scan_lists(n); };
                int p1 = 0;
                ProfScope prof(PROF_SCAN, (double)N * d * 4.0, st);
                if (ns > 0) {
                    int rc = scan(X, ns, 0, mpart, nullptr);
                    if (rc) return rc;
                    p1 = lists_of(ns);
                    hipLaunchKernelGGL(merge_keys_kernel, dim3(nqa), dim3(mwv * 64), (size_t)mwv * cap * 8, st, mpart, p1, qb,
                                       kl, cap, (const long long*)nullptr, 0ll, cand_scores, cand_rows, 0);
                    WISE_LAUNCH_CHECK("merge_keys_kernel");
                    if ((rc = sample_threshold_launch(cand_scores, cand_rows, tau0, st))) return rc;
                }
                int rc = scan(X + (size_t)ns * d, N - ns, ns, mpart + (size_t)p1 * qb * kl, ns > 0 ? tau0 : nullptr);
                if (rc) return rc;
                nlists = p1 + lists_of(N - ns);
            } else {
                ProfScope prof(PROF_SCAN, (double)N * d * 4.0, st);
                int rc = mfma_scan_launch(X, N, d, mq, nqa, k, mpart, split, st);
                if (rc) return rc;
            }
            if (split) {
                // best MFMA_KL candidates per query by approximate score (rows, not ids), then their exact scores
                hipLaunchKernelGGL(merge_keys_kernel, dim3(nqa), dim3(mwv * 64), (size_t)mwv * cap * 8, st, mpart, nlists, qb,
                                   kl, cap, (const long long*)nullptr, 0ll, cand_scores, cand_rows, 0);
                WISE_LAUNCH_CHECK("merge_keys_kernel");
                int rc = rescore_launch(X, d, mq, cand_rows, nqa, k, reinterpret_cast<const long long*>(ids),
                                        (long long)id_base, outD + (size_t)q0 * k,
                                        reinterpret_cast<long long*>(outI) + (size_t)q0 * k, st);
                if (rc) return rc;
            } else {
                hipLaunchKernelGGL(merge_keys_kernel, dim3(nqa), dim3(mwv * 64), (size_t)mwv * cap * 8, st, mpart,
                                   mfma_scan_lists(N), MFMA_QB, k, cap, reinterpret_cast<const long long*>(ids),
                                   (long long)id_base, outD, reinterpret_cast<long long*>(outI), q0);
                WISE_LAUNCH_CHECK("merge_keys_kernel");
            }
            q0 += nqa;
        }
        return WISE_OK;
    }
    ScanPlan p = plan_scan(N, d, nq, k);
    u64* part = reinterpret_cast<u64*>(workspace);
    float* qpad = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) +
                                           align_up((size_t)p.grid * p.nq_per_pass * k * sizeof(u64), 256));
    const int nv = (d / 4 + 63) / 64;
    // merge geometry
    int mw = 8192 / p.cap;
    if (mw < 1) mw = 1;
    if (mw > 16) mw = 16;
    const size_t mlds = (size_t)mw * p.cap * 8;

    for (int q0 = 0; q0 < nq; q0 += p.nq_per_pass) {
        const int nqa = (nq - q0 < p.nq_per_pass) ? nq - q0 : p.nq_per_pass;
        const float* qptr = Q + (size_t)q0 * d;
        if (nqa < p.nq_per_pass) {
            // ragged tail: replicate the last query so the kernel shape stays fixed (results ignored)
            for (int j = 0; j < p.nq_per_pass; ++j) {
                int srcq = q0 + (j < nqa ? j : nqa - 1);
                hipError_t e = hipMemcpyAsync(qpad + (size_t)j * d, Q + (size_t)srcq * d, (size_t)d * sizeof(float),
                                              hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) { set_error("ip_topk: memcpy: %s", hipGetErrorString(e)); return (int)e; }
            }
            qptr = qpad;
        }
        if (N > 0) {
            int rc = 0;
            ProfScope prof(PROF_SCAN, (double)N * d * 4.0, st);
            switch (nv) {
                case 1: rc = dispatch_nq<1>(p, X, N, d, qptr, k, part, st); break;
                case 2: rc = dispatch_nq<2>(p, X, N, d, qptr, k, part, st); break;
                case 3: rc = dispatch_nq<3>(p, X, N, d, qptr, k, part, st); break;
                case 4: rc = dispatch_nq<4>(p, X, N, d, qptr, k, part, st); break;
                case 5: rc = dispatch_nq<5>(p, X, N, d, qptr, k, part, st); break;
                case 6: rc = dispatch_nq<6>(p, X, N, d, qptr, k, part, st); break;
                case 7: rc = dispatch_nq<7>(p, X, N, d, qptr, k, part, st); break;
                case 8: rc = dispatch_nq<8>(p, X, N, d, qptr, k, part, st); break;
                default: rc = WISE_E_INVALID;
            }
            if (rc) { set_error("ip_topk: no kernel for d=%d", d); return rc; }
            WISE_LAUNCH_CHECK("ip_scan_kernel");
        }
        if (mlds > 48 * 1024)
            raise_lds_limit(reinterpret_cast<const void*>(merge_keys_kernel), (int)mlds);
        hipLaunchKernelGGL(merge_keys_kernel, dim3(nqa), dim3(mw * 64), mlds, st, part, N > 0 ? p.grid : 0,
                           p.nq_per_pass, k, p.cap, reinterpret_cast<const long long*>(ids), (long long)id_base, outD,
                           reinterpret_cast<long long*>(outI), q0);
        WISE_LAUNCH_CHECK("merge_keys_kernel");
    }
    return WISE_OK;
}

extern "C" size_t wise_ivf_scan_workspace_bytes(int nq, int nprobe, int k) {
    if (nq < 1 || nprobe < 1 || k < 1 || k > 2048) return 0;
    return align_up((size_t)nq * nprobe * k * sizeof(u64), 256);
}

extern "C" int wise_ivf_scan_f32(const float* X, int64_t N, int d, const int64_t* list_off, int nlist,
                                 const int64_t* ids, const float* Q, int nq, const int64_t* probes, int nprobe, int k,
                                 float* outD, int64_t* outI, void* workspace, size_t workspace_bytes, void* stream) {
    WISE_CHECK_ARG(d >= 4 && d <= 2048 && d % 4 == 0, "ivf_scan: d=%d must be a multiple of 4 in [4,2048]", d);
    WISE_CHECK_ARG(k >= 1 && k <= 2048, "ivf_scan: k=%d out of [1,2048]", k);
    WISE_CHECK_ARG(nq >= 1 && nprobe >= 1 && nlist >= 1 && (long long)nq * nprobe < (1ll << 31),
                   "ivf_scan: nq=%d nprobe=%d nlist=%d out of range", nq, nprobe, nlist);
    WISE_CHECK_ARG(N >= 0 && N < 0xFFFFFFFFll, "ivf_scan: N=%lld out of range", (long long)N);
    WISE_CHECK_ARG(Q && outD && outI && list_off && probes && (X || N == 0), "ivf_scan: null pointer");
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0, "ivf_scan: X and Q must be 16-byte aligned");
    const size_t need = wise_ivf_scan_workspace_bytes(nq, nprobe, k);
    if (!workspace || workspace_bytes < need) {
        set_error("ivf_scan: workspace %zu < %zu bytes", workspace_bytes, need);
        return WISE_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    u64* part = reinterpret_cast<u64*>(workspace);
    const int cap = list_cap(k);
    const SegArgs seg = {reinterpret_cast<const long long*>(probes), reinterpret_cast<const long long*>(list_off), nprobe,
                         nq};
    switch ((d / 4 + 63) / 64) {
        case 1: launch_seg_scan<1>(X, d, Q, k, cap, part, seg, st); break;
        case 2: launch_seg_scan<2>(X, d, Q, k, cap, part, seg, st); break;
        case 3: launch_seg_scan<3>(X, d, Q, k, cap, part, seg, st); break;
        case 4: launch_seg_scan<4>(X, d, Q, k, cap, part, seg, st); break;
        case 5: launch_seg_scan<5>(X, d, Q, k, cap, part, seg, st); break;
        case 6: launch_seg_scan<6>(X, d, Q, k, cap, part, seg, st); break;
        case 7: launch_seg_scan<7>(X, d, Q, k, cap, part, seg, st); break;
        case 8: launch_seg_scan<8>(X, d, Q, k, cap, part, seg, st); break;
        default: set_error("ivf_scan: no kernel for d=%d", d); return WISE_E_INVALID;
    }
    WISE_LAUNCH_CHECK("ip_scan_kernel<seg>");
    int mw = 8192 / cap;
    if (mw < 1) mw = 1;
    if (mw > 16) mw = 16;
    const size_t mlds = (size_t)mw * cap * 8;
    if (mlds > 48 * 1024)
        raise_lds_limit(reinterpret_cast<const void*>(merge_keys_kernel), (int)mlds);
    hipLaunchKernelGGL(merge_keys_kernel, dim3(nq), dim3(mw * 64), mlds, st, part, nprobe, nq, k, cap,
                       reinterpret_cast<const long long*>(ids), 0ll, outD, reinterpret_cast<long long*>(outI), 0);
    WISE_LAUNCH_CHECK("merge_keys_kernel");
    return WISE_OK;
}

// ---- two-stage exact search over a bf16 shadow (see the kernels above)
namespace wise {
static int shadow_grid(long long N) {
    long long need = ((N + 7) / 8 + 3) / 4;
    if (need < 4) need = 4;
    need = (need + 3) / 4 * 4;
    // two blocks per CU: as fast as four (1.65 vs 1.69 ms at 10M x 512) and half the lists to merge
    const long long cap = g_scan_blocks_per_cu > 0 ? 256ll * g_scan_blocks_per_cu : 512;
    return need < cap ? (int)need : (int)cap;
}
// k <= 1024: the one-query threshold form serves any such k (radix selections); batches of queries go through the
// matrix-core passes for k <= SHADOW_BATCH_K, else one query at a time
constexpr int SHADOW_KMAX = 1024, SHADOW_BATCH_K = 128;
static bool shadow_supported(int d, int k) { return d % 8 == 0 && d >= 8 && d <= 1024 && k >= 1 && k <= SHADOW_KMAX; }
}  // namespace wise

extern "C" int wise_ip_shadow_bf16(const float* X, int64_t N, int d, uint16_t* Xb, float* norms, void* stream) {
    float* max_norm = norms;
    WISE_CHECK_ARG(d >= 8 && d % 8 == 0 && N >= 0 && (X && Xb || N == 0) && max_norm, "ip_shadow_bf16: bad argument");
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Xb & 15) == 0, "ip_shadow_bf16: X and Xb must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(max_norm, 0, 2 * sizeof(float), st);
    if (e != hipSuccess) { set_error("ip_shadow_bf16: %s", hipGetErrorString(e)); return (int)e; }
    if (N > 0) {
        const long long want = (N + 3) / 4;
        hipLaunchKernelGGL(shadow_bf16_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, X, (long long)N, d,
                           Xb, max_norm);
        WISE_LAUNCH_CHECK("shadow_bf16_kernel");
    }
    return WISE_OK;
}

extern "C" int wise_ip_shadow_i8(const float* X, int64_t N, int d, int8_t* Xq, float* scales, float* norms, void* stream) {
    WISE_CHECK_ARG(d >= 16 && d % 16 == 0 && d <= 1024 && N >= 0 && ((X && Xq && scales) || N == 0) && norms,
                   "ip_shadow_i8: bad argument (d=%d must be a multiple of 16 up to 1024)", d);
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Xq & 15) == 0, "ip_shadow_i8: X and Xq must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(norms, 0, 4 * sizeof(float), st);
    if (e != hipSuccess) { set_error("ip_shadow_i8: %s", hipGetErrorString(e)); return (int)e; }
    if (N > 0) {
        const long long want = (N + 3) / 4;
        hipLaunchKernelGGL(shadow_i8_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, X, (long long)N, d,
                           reinterpret_cast<signed char*>(Xq), scales, norms);
        WISE_LAUNCH_CHECK("shadow_i8_kernel");
    }
    hipLaunchKernelGGL(shadow_i8_finish_kernel, dim3(1), dim3(64), 0, st, norms, d);
    WISE_LAUNCH_CHECK("shadow_i8_finish_kernel");
    return WISE_OK;
}

namespace wise { struct PassWsSize { size_t total; }; static size_t pass_workspace_bytes(long long N, int d, int k); }
extern "C" size_t wise_ip_topk_shadow_workspace_bytes(int64_t N, int d, int nq, int k) {
    if (N < 0 || nq < 1 || !shadow_supported(d, k)) return 0;
    ScanPlan p = plan_scan(N, d, 1, k);
    for (int m = 2; m <= 4; m *= 2) {   // the gated f32 scan of 2 or 4 queries may plan a larger grid
        const ScanPlan pm = plan_scan(N, d, m, k);
        if (pm.grid > p.grid) p.grid = pm.grid;
    }
    // one query: per-wave sample maxima | threshold | control words | collected keys | kept keys | exact keys | sampled scores | lists of the gated f32 scan
    size_t one = align_up((size_t)SAMPLE_GRID * 4 * sizeof(float), 256) + 256 + 256 +
                 align_up((size_t)COLLECT_CAP * sizeof(u64), 256) + 2 * align_up((size_t)RESCORE_CAP * sizeof(u64), 256) +
                 align_up((size_t)SAMPLE_CHUNKS * 512 * sizeof(float), 256) +              // every sampled row's score
                 align_up((size_t)p.grid * 4 * k * sizeof(u64), 256);
    // an index too small for a sample is answered by the f32 scan and needs only its workspace
    if (N < COLLECT_MIN_ROWS) return wise_ip_topk_workspace_bytes(N, d, nq, k);
    // batches (two queries and more): see pass_workspace() — ~134 MB of per-query lists that a single query never touches
    const size_t many = nq >= 2 ? pass_workspace_bytes(N, d, k) : 0;
    return one > many ? one : many;
}

namespace wise {
// Threshold of query q from its n dumped SAMPLE scores: thread t takes the maximum of elements t, t + 1024, ... (1024
// disjoint segments, each maximum a different row), the block sorts the 1024 maxima, L = the k-th largest: k sampled rows
// reach it, so the exact k-th best score of the index is >= L - eps and a row of the exact top-k scores >= L - 2 eps
// approximately.  thr[q] = L - 2 eps(q).  (A segment maximum costs a tenth of an exact selection: ~15 us against 176.)
__global__ __launch_bounds__(1024) void batch_threshold_kernel(const float* __restrict__ scores, long long n, int k,
                                                               const float* __restrict__ Q, int d,
                                                               const float* __restrict__ norms, int q_mode,
                                                               float* __restrict__ thr) {
    __shared__ float mx[1024];
    __shared__ float wsum[16], wres[16];
    const int q = blockIdx.x, t = threadIdx.x;
    const float* sq = scores + (size_t)q * n;
    float m = -3.4028234663852886e38f;
    for (long long j = t; j < n; j += 1024) {
        const float v = sq[j];
        m = v > m ? v : m;
    }
    mx[t] = m;
    float qq = 0.f, qr = 0.f;
    for (int j = t; j < d; j += 1024) {
        const float qv = Q[(size_t)q * d + j], rr = bf16_round_residual(qv);
        qq = fmaf(qv, qv, qq);
        qr = fmaf(rr, rr, qr);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { qq += __shfl_xor(qq, o, 64); qr += __shfl_xor(qr, o, 64); }
    if ((t & 63) == 0) { wsum[t >> 6] = qq; wres[t >> 6] = qr; }
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int pos = ((t / stride) * (stride << 1)) + (t % stride);
            if (t < 512) {
                const int par = pos + stride;
                const bool desc = (pos & size) == 0;
                const float a = mx[pos], b2 = mx[par];
                if (desc ? (a < b2) : (a > b2)) { mx[pos] = b2; mx[par] = a; }
            }
            __syncthreads();
        }
    if (t == 0) {
        qq = 0.f; qr = 0.f;
        for (int w = 0; w < 16; ++w) { qq += wsum[w]; qr += wres[w]; }
        const float eps = query_eps(norms, d, qq, qr, q_mode);
        thr[q] = mx[k - 1] - 2.f * eps;     // n >= 1024 sampled rows (host check): every segment holds a row
    }
}

// Between the two ranges of the batched collect pass: the rows collected from the FIRST range bound the k-th best score
// far better than the 64K-row sample did (k-th best of 1M rows instead of 64K), so the rest of the index runs under
//     thr[q] = max(thr[q], L1 - 2 eps),   L1 = k-th largest of 1024 slice maxima of the collected approximate scores
// (>= k distinct rows reach L1, the same argument as for the sample).  It matters because the collect kernel's hit path —
// a pass over all of a lane's accumulators with atomics — is taken by nearly every 32-row group under the sample
// threshold (128 queries x 32 rows x 3e-4), and by one group in thirty under the tightened one: 2.0 -> 1.75 ms per pass.
// A query whose first-range list overflowed keeps its threshold (the refine step raises the gate for it later).
__global__ __launch_bounds__(1024) void batch_tighten_kernel(const int* __restrict__ ctl, const u64* __restrict__ cand, int cap,
                                                             int k, const float* __restrict__ Q, int d,
                                                             const float* __restrict__ norms, int q_mode,
                                                             float* __restrict__ thr) {
    __shared__ u64 wmax[16];
    __shared__ float wsum[16], wres[16];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = ctl[4 * q];
    if (n > cap || n < k) return;                    // uniform: nothing to learn from an overflowed or a short list
    cand += (size_t)q * cap;
    Q += (size_t)q * d;
    u64 mine = 0;
    for (int i = tid; i < n; i += 1024) {
        const u64 key = cand[i];
        mine = key > mine ? key : mine;
    }
    float qq = 0.f, qr = 0.f;
    for (int j = tid; j < d; j += 1024) {
        qq = fmaf(Q[j], Q[j], qq);
        const float rr = bf16_round_residual(Q[j]);
        qr = fmaf(rr, rr, qr);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { qq += __shfl_xor(qq, o, 64); qr += __shfl_xor(qr, o, 64); }
    if (lane == 0) { wsum[wave] = qq; wres[wave] = qr; }
    u64 L = 0;
    if (k > 16) {
        __shared__ unsigned hist[256];
        __shared__ u64 sh_prefix;
        __shared__ int sh_rem;
        L = list_kth_score(cand, n, k, hist, &sh_prefix, &sh_rem);
        __syncthreads();
    } else
    for (int r = 0; r < k; ++r) {
        u64 m = mine;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(m, o, 64);
            m = other > m ? other : m;
        }
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        u64 g = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) g = wmax[w] > g ? wmax[w] : g;
        L = g;
        if (mine == g) mine = 0;
        __syncthreads();
    }
    if (tid == 0 && L != 0) {                        // L == 0: fewer than k non-empty slices
        qq = 0.f; qr = 0.f;
        for (int w = 0; w < 16; ++w) { qq += wsum[w]; qr += wres[w]; }
        const float t2 = f32_unorder((unsigned)(L >> 32)) - 2.f * query_eps(norms, d, qq, qr, q_mode);
        if (t2 > thr[q]) thr[q] = t2;
    }
}

// One query (threshold form, see ip_collect_bf16_kernel): sample scan -> threshold -> collect every row that
// could belong to the top-k -> exact scores -> the k best; the f32 scan queued behind runs only if the list overflowed.
static int shadow_search_one(const float* X, const bf16_t* Xb, const float* norms, long long N, int d, const float* q,
                             int k, const long long* ids, long long id_base, float* outD, long long* outI, int* stats,
                             unsigned char* wsb, hipStream_t st, const signed char* Xq = nullptr /*int8 shadow + scales*/,
                             const float* scales = nullptr) {
    float* wave_best = reinterpret_cast<float*>(wsb);
    size_t off = align_up((size_t)SAMPLE_GRID * 4 * sizeof(float), 256);
    float* thr = reinterpret_cast<float*>(wsb + off);
    off += 256;
    int* counter = reinterpret_cast<int*>(wsb + off);     // ctl: [0] collected, [1] gate, [2] kept
    int* gate = counter + 1;
    off += 256;
    u64* cand = reinterpret_cast<u64*>(wsb + off);
    off += align_up((size_t)COLLECT_CAP * sizeof(u64), 256);
    u64* cand2 = reinterpret_cast<u64*>(wsb + off);
    off += align_up((size_t)RESCORE_CAP * sizeof(u64), 256);
    u64* ekeys = reinterpret_cast<u64*>(wsb + off);
    off += align_up((size_t)RESCORE_CAP * sizeof(u64), 256);
    float* dump = reinterpret_cast<float*>(wsb + off);
    off += align_up((size_t)SAMPLE_CHUNKS * 512 * sizeof(float), 256);
    u64* epart = reinterpret_cast<u64*>(wsb + off);
    const int d8 = d / 8, nv8 = (d8 + 63) / 64;
    if (nv8 > 2) { set_error("ip_topk_shadow: no kernel for d=%d", d); return WISE_E_INVALID; }
    const uint4* xb = reinterpret_cast<const uint4*>(Xb);
    // int8 rows: 16 / 32 / 64 lanes per row, groups of 32 / 16 / 8 rows
    const int lpr = Xq ? (d <= 256 ? 16 : d <= 512 ? 32 : 64) : 0;
    const int grows = Xq ? I8_T * (64 / lpr) : 8;        // rows per group of the two scans
    if (Xq && (d % 16 != 0 || d > 1024 || !scales)) { set_error("ip_topk_shadow8: d=%d must be a multiple of 16 up to 1024", d); return WISE_E_INVALID; }
    // ---- sample: SAMPLE_CHUNKS evenly spaced chunks of 2^shift groups of 8 rows, every sampled score dumped.  About
    // N / 64 rows, between 16384 and 65536 (a shard of an index sharded over eight GPUs pays a quarter of the sample a
    // whole index does); the threshold is the exact k-th largest sampled score (k <= 1024 of >= 16384 samples) and it
    // also zeroes the control words: no memset in front.
    {
        int shift = SAMPLE_CHUNK_SHIFT;                  // 512 rows per chunk
        while (shift > 4 && (long long)SAMPLE_CHUNKS * (8ll << shift) * 64 > N) --shift;
        for (int gr = grows; gr > 8; gr >>= 1) --shift;                // the same rows per chunk in larger groups
        const long long groups = N / grows;              // whole groups only: a sampled group is never ragged
        const long long chunk_groups = 1ll << shift;
        const long long stride = (groups - chunk_groups) / (SAMPLE_CHUNKS - 1);      // last chunk ends inside the index
        const long long sgroups = (long long)SAMPLE_CHUNKS * chunk_groups;
        const int sgrid = (int)((sgroups + 15) / 16 < SAMPLE_GRID ? (sgroups + 15) / 16 : SAMPLE_GRID);
        if (Xq) {
            if (lpr == 16)
                hipLaunchKernelGGL((ip_sample_i8_kernel<16>), dim3(sgrid), dim3(256), 0, st, Xq, scales, sgroups, d, q, shift, stride, dump);
            else if (lpr == 32)
                hipLaunchKernelGGL((ip_sample_i8_kernel<32>), dim3(sgrid), dim3(256), 0, st, Xq, scales, sgroups, d, q, shift, stride, dump);
            else
                hipLaunchKernelGGL((ip_sample_i8_kernel<64>), dim3(sgrid), dim3(256), 0, st, Xq, scales, sgroups, d, q, shift, stride, dump);
        } else if (nv8 == 1)
            hipLaunchKernelGGL((ip_sample_bf16_kernel<1, 8>), dim3(sgrid), dim3(256), 0, st, xb, sgroups, d8, q, shift, stride,
                               wave_best, dump);
        else
            hipLaunchKernelGGL((ip_sample_bf16_kernel<2, 8>), dim3(sgrid), dim3(256), 0, st, xb, sgroups, d8, q, shift, stride,
                               wave_best, dump);
        WISE_LAUNCH_CHECK("ip_sample_bf16_kernel");
        hipLaunchKernelGGL(sample_threshold_kth_kernel, dim3(1), dim3(1024), 0, st, dump, (int)(sgroups * grows), k, q, d, norms, thr,
                           counter);
        WISE_LAUNCH_CHECK("sample_threshold_kth_kernel");
    }
    // ---- collect over all rows; for large k in two ranges, the threshold tightened in between by what the first
    // range (2^20 rows: a sample sixteen times the sample pass's) collected
    {
        ProfScope prof(PROF_SCAN, Xq ? (double)N * (d + 4.0) : (double)N * d * 2.0, st);
        auto collect = [&](long long r0, long long rows) {
            const int grid = shadow_grid(rows);
            if (Xq) {
                const signed char* base8 = Xq + (size_t)r0 * d;
                const float* sc8 = scales + r0;
                if (lpr == 16)
                    hipLaunchKernelGGL((ip_collect_i8_kernel<16>), dim3(grid), dim3(256), 0, st, base8, sc8, rows, d, q, thr, counter, cand, COLLECT_CAP, r0);
                else if (lpr == 32)
                    hipLaunchKernelGGL((ip_collect_i8_kernel<32>), dim3(grid), dim3(256), 0, st, base8, sc8, rows, d, q, thr, counter, cand, COLLECT_CAP, r0);
                else
                    hipLaunchKernelGGL((ip_collect_i8_kernel<64>), dim3(grid), dim3(256), 0, st, base8, sc8, rows, d, q, thr, counter, cand, COLLECT_CAP, r0);
                return;
            }
            const uint4* base = xb + (size_t)r0 * d8;
            if (nv8 == 1)
                hipLaunchKernelGGL((ip_collect_bf16_kernel<1, 8>), dim3(grid), dim3(256), 0, st, base, rows, d8, q, thr, counter,
                                   cand, COLLECT_CAP, r0);
            else
                hipLaunchKernelGGL((ip_collect_bf16_kernel<2, 8>), dim3(grid), dim3(256), 0, st, base, rows, d8, q, thr, counter,
                                   cand, COLLECT_CAP, r0);
        };
        const long long R1 = 1ll << 20;
        if (k > 64 && N >= 4 * R1) {
            collect(0, R1);
            WISE_LAUNCH_CHECK("ip_collect_bf16_kernel");
            hipLaunchKernelGGL(collect_tighten_kernel, dim3(1), dim3(1024), 0, st, counter, cand, COLLECT_CAP, k, q, d, norms, thr);
            WISE_LAUNCH_CHECK("collect_tighten_kernel");
            collect(R1, N - R1);
        } else {
            collect(0, N);
        }
        WISE_LAUNCH_CHECK("ip_collect_bf16_kernel");
    }
    // ---- refine, exact scores, the k best: one block; the two kernels behind it only run when more than
    // FINISH_LDS_ROWS rows survive the refinement (runs of near-duplicates)
    hipLaunchKernelGGL(collect_finish_kernel, dim3(1), dim3(1024), 0, st, counter, cand, COLLECT_CAP, k, q, d, X, norms, cand2, ids,
                       id_base, outD, outI, stats);
    WISE_LAUNCH_CHECK("collect_finish_kernel");
    hipLaunchKernelGGL(collect_rescore_kernel, dim3(64), dim3(256), 0, st, X, d, q, counter, cand2, ekeys);
    WISE_LAUNCH_CHECK("collect_rescore_kernel");
    hipLaunchKernelGGL(collect_select_kth_kernel, dim3(1), dim3(1024), 0, st, counter, ekeys, k, ids, id_base, outD, outI, stats);
    WISE_LAUNCH_CHECK("collect_select_kth_kernel");
    // ---- the f32 scan of the same query, which returns at once unless the list overflowed
    ScanPlan p = plan_scan(N, d, 1, k);
    if (p.nq_per_pass != 1) { set_error("ip_topk_shadow: f32 plan serves %d queries per pass", p.nq_per_pass); return WISE_E_INVALID; }
    const int nv = (d / 4 + 63) / 64;
    bool launched = false;
    if (nv == 1) { launch_scan<1, 1>(p, X, N, d, q, k, epart, st, gate); launched = true; }
    if (nv == 2) { launch_scan<2, 1>(p, X, N, d, q, k, epart, st, gate); launched = true; }
    if (nv == 3) { launch_scan<3, 1>(p, X, N, d, q, k, epart, st, gate); launched = true; }
    if (nv == 4) { launch_scan<4, 1>(p, X, N, d, q, k, epart, st, gate); launched = true; }
    if (!launched) { set_error("ip_topk_shadow: no f32 kernel for d=%d", d); return WISE_E_INVALID; }
    WISE_LAUNCH_CHECK("ip_scan_kernel (gated)");
    int emw = 8192 / p.cap;
    if (emw < 1) emw = 1;
    if (emw > 16) emw = 16;
    hipLaunchKernelGGL(merge_keys_kernel, dim3(1), dim3(emw * 64), (size_t)emw * p.cap * 8, st, epart, p.grid, 1, k, p.cap,
                       ids, id_base, outD, outI, 0, gate);
    WISE_LAUNCH_CHECK("merge_keys_kernel (gated)");
    return WISE_OK;
}

constexpr int BATCH_CAP = 65536;            // rows per query the batched collect pass may hand on
constexpr int BATCH_SAMPLE_SHIFT = 4;       // a sample chunk = 16 groups of 32 rows = 512 rows
constexpr long long BATCH_FIRST_RANGE = 1ll << 20;   // rows of the collect pass's first range (a multiple of 32)
constexpr int PASS_QMAX = 128;              // most queries one pass of the shadow scan carries (one-piece queries, d <= 512)

struct PassWs {
    u64* mpart; float* mq; long long* cand_rows; float* cand_scores; u64* tau0; int* ctl; int* gate; float* thr;
    float* dump; u64* cand; u64* cand2; u64* ekeys; size_t total;
};
static PassWs pass_workspace(unsigned char* wsb, long long N, int d, int k) {
    PassWs w;
    size_t off = 0;
    // (sizing calls pass no buffer: offsets are applied to a real base only)
    auto at = [&](size_t o) -> unsigned char* { return wsb ? wsb + o : nullptr; };
    const ScanPlan p2 = plan_scan(N, d, 4, k);     // (the VALU fallback plans up to four queries per launch)
    size_t lists = (size_t)2 * split64_lists(N) * MFMA_QB2 * MFMA_KL * sizeof(u64);
    const size_t valu = (size_t)p2.grid * 4 * k * sizeof(u64);
    if (valu > lists) lists = valu;
    w.mpart = reinterpret_cast<u64*>(at(off)); off += align_up(lists, 256);
    w.mq = reinterpret_cast<float*>(at(off)); off += align_up((size_t)PASS_QMAX * d * sizeof(float), 256);
    w.cand_rows = reinterpret_cast<long long*>(at(off));
    w.cand_scores = reinterpret_cast<float*>(at(off + (size_t)MFMA_QB2 * MFMA_KL * 8));
    off += align_up((size_t)MFMA_QB2 * MFMA_KL * 12, 256);
    w.tau0 = reinterpret_cast<u64*>(at(off)); off += 512;
    w.ctl = reinterpret_cast<int*>(at(off)); off += PASS_QMAX * 4 * sizeof(int);
    w.gate = reinterpret_cast<int*>(at(off)); off += 256;
    w.thr = reinterpret_cast<float*>(at(off)); off += align_up(PASS_QMAX * sizeof(float), 256);
    w.dump = reinterpret_cast<float*>(at(off)); off += align_up((size_t)PASS_QMAX * SAMPLE_CHUNKS * 512 * sizeof(float), 256);
    w.cand = reinterpret_cast<u64*>(at(off)); off += align_up((size_t)PASS_QMAX * BATCH_CAP * sizeof(u64), 256);
    w.cand2 = reinterpret_cast<u64*>(at(off)); off += align_up((size_t)PASS_QMAX * RESCORE_CAP * sizeof(u64), 256);
    w.ekeys = reinterpret_cast<u64*>(at(off)); off += align_up((size_t)PASS_QMAX * RESCORE_CAP * sizeof(u64), 256);
    w.total = off;
    return w;
}

static size_t pass_workspace_bytes(long long N, int d, int k) { return pass_workspace(nullptr, N, d, k).total; }

// up to QB queries (shadow_pass_queries(d): 128 / 64 / 32) in the threshold form, the bf16 rows on the matrix cores: sample pass ->
// per-query thresholds -> one pass over all bf16 rows collecting every (query, row) that could matter -> per query:
// refine, exact scores, the k best.  If any query's list overflows the pass gate is raised and the scan of the f32 rows
// queued behind (split-bf16 candidates + exact re-scoring, or the f32 VALU scan for d > 512) redoes the pass.
static int shadow_search_pass(const float* X, const bf16_t* Xb, const float* norms, long long N, int d, const float* Q,
                              int nqa, int k, const long long* ids, long long id_base, float* outD, long long* outI,
                              int* stats, unsigned char* wsb, hipStream_t st, int QB /*64, or 32 for 512 < d <= 1024*/) {
    const PassWs w = pass_workspace(wsb, N, d, k);
    const int q_mode = shadow_one_piece() ? QMODE_ONE_PIECE : QMODE_TWO_PIECE;
    // one pass of the first stage over rows [r0, r0 + n)
    auto stage1 = [&](long long r0, long long n, int nq_, const float* thr_, int* ctl_, u64* cand_, int cap_, float* dump_,
                      int shift_, long long stride_) {
        return shadow64_scan_launch(Xb + (size_t)r0 * d, n, d, w.mq, nq_, thr_, ctl_, cand_, cap_, st, dump_, QB, shift_, stride_, r0);
    };
    u64* mpart = w.mpart;
    float* mq = w.mq;
    long long* cand_rows = w.cand_rows;
    float* cand_scores = w.cand_scores;
    u64* tau0 = w.tau0;
    int* gate = w.gate;
    hipError_t e = hipMemsetAsync(w.ctl, 0, PASS_QMAX * 4 * sizeof(int) + sizeof(int), st);     // ctl and the pass gate behind it
    if (e == hipSuccess && nqa < QB) e = hipMemsetAsync(mq, 0, (size_t)QB * d * sizeof(float), st);
    if (e == hipSuccess) e = hipMemcpyAsync(mq, Q, (size_t)nqa * d * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { set_error("ip_topk_shadow: query staging: %s", hipGetErrorString(e)); return (int)e; }
    // threshold sample of the fallback's own passes (split scan of the f32 rows)
    const long long ns = (g_scan_sample && N >= 16ll * g_scan_sample) ? 2 * g_scan_sample : 0;
    int rc;
    {
        // ---- sample: SAMPLE_CHUNKS evenly spaced chunks of 512 rows, scores dumped; thresholds
        const long long groups = N / 32, chunk_groups = 1ll << BATCH_SAMPLE_SHIFT;
        const long long stride = (groups - chunk_groups) / (SAMPLE_CHUNKS - 1);
        const long long nsample = (long long)SAMPLE_CHUNKS * chunk_groups * 32;
        if ((rc = stage1(0, nsample, QB, nullptr, nullptr, nullptr, 0, w.dump, BATCH_SAMPLE_SHIFT, stride))) return rc;
        hipLaunchKernelGGL(batch_threshold_kernel, dim3(nqa), dim3(1024), 0, st, w.dump, nsample, k, mq, d, norms,
                           q_mode, w.thr);
        WISE_LAUNCH_CHECK("batch_threshold_kernel");
        // ---- collect over all rows, in two ranges: [0, R1) under the sample's thresholds, the rest under thresholds
        // tightened by what the first range collected (batch_tighten_kernel)
        {
            ProfScope prof(PROF_SCAN, (double)N * d * 2.0, st);
            const long long R1 = N >= 4 * BATCH_FIRST_RANGE ? BATCH_FIRST_RANGE : N;
            if ((rc = stage1(0, R1, nqa, w.thr, w.ctl, w.cand, BATCH_CAP, nullptr, -1, 0))) return rc;
            if (R1 < N) {
                hipLaunchKernelGGL(batch_tighten_kernel, dim3(nqa), dim3(1024), 0, st, w.ctl, w.cand, BATCH_CAP, k, mq, d, norms,
                                   q_mode, w.thr);
                WISE_LAUNCH_CHECK("batch_tighten_kernel");
                if ((rc = stage1(R1, N - R1, nqa, w.thr, w.ctl, w.cand, BATCH_CAP, nullptr, -1, 0))) return rc;
            }
        }
        hipLaunchKernelGGL(collect_refine_kernel, dim3(1, nqa), dim3(1024), 0, st, w.ctl, w.cand, BATCH_CAP, k, mq, d, norms,
                           w.cand2, stats, gate, q_mode);
        WISE_LAUNCH_CHECK("collect_refine_kernel");
        hipLaunchKernelGGL(collect_rescore_kernel, dim3(8, nqa), dim3(256), 0, st, X, d, mq, w.ctl, w.cand2, w.ekeys);
        WISE_LAUNCH_CHECK("collect_rescore_kernel");
        if (k > 16)
            hipLaunchKernelGGL(collect_select_kth_kernel, dim3(1, nqa), dim3(1024), 0, st, w.ctl, w.ekeys, k, ids, id_base, outD,
                               outI, stats);
        else
            hipLaunchKernelGGL(collect_select_kernel, dim3(1, nqa), dim3(1024), 0, st, w.ctl, w.ekeys, k, ids, id_base, outD, outI,
                               stats);
        WISE_LAUNCH_CHECK("collect_select_kernel");
    }
    // ---- gated fallback over the f32 rows: every launch returns at once while *gate == 0
    if (d > 512 || k > MFMA_KC) {
        // d > 512 or k > 12: the split-bf16 kernels do not reach (they keep 16 candidates); the f32 VALU scan redoes the
        // pass, up to four queries per launch
        const ScanPlan p = plan_scan(N, d, 4, k);
        const int nv = (d / 4 + 63) / 64;
        u64* epart = mpart;          // the stage-1 lists are dead by now
        int emw = 8192 / p.cap;
        if (emw < 1) emw = 1;
        if (emw > 16) emw = 16;
        for (int q0 = 0; q0 < nqa; q0 += p.nq_per_pass) {
            const int nqp = nqa - q0 < p.nq_per_pass ? nqa - q0 : p.nq_per_pass;   // mq is zero-padded to QB rows
            const float* qq = mq + (size_t)q0 * d;
            bool ok = false;
            if (p.nq_per_pass == 4) {
                if (nv == 1) { launch_scan<1, 4>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 2) { launch_scan<2, 4>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
            } else if (p.nq_per_pass == 2) {
                if (nv == 1) { launch_scan<1, 2>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 2) { launch_scan<2, 2>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 3) { launch_scan<3, 2>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 4) { launch_scan<4, 2>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
            } else if (p.nq_per_pass == 1) {
                if (nv == 1) { launch_scan<1, 1>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 2) { launch_scan<2, 1>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 3) { launch_scan<3, 1>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
                if (nv == 4) { launch_scan<4, 1>(p, X, N, d, qq, k, epart, st, gate); ok = true; }
            }
            if (!ok) { set_error("ip_topk_shadow: no f32 fallback kernel for d=%d", d); return WISE_E_INVALID; }
            WISE_LAUNCH_CHECK("ip_scan_kernel (gated)");
            hipLaunchKernelGGL(merge_keys_kernel, dim3(nqp), dim3(emw * 64), (size_t)emw * p.cap * 8, st, epart, p.grid,
                               p.nq_per_pass, k, p.cap, ids, id_base, outD, outI, q0, gate);
            WISE_LAUNCH_CHECK("merge_keys_kernel (gated)");
        }
        return WISE_OK;
    }
    // d <= 512: the split-bf16 scan of the f32 rows, 64 queries per launch (a 128-query pass is redone in two halves)
    for (int sub = 0; sub < nqa; sub += MFMA_QB2) {
        const int nsub = nqa - sub < MFMA_QB2 ? nqa - sub : MFMA_QB2;
        const float* sq = mq + (size_t)sub * d;       // mq is zero-padded to QB rows, QB a multiple of 64 here
        float* sD = outD + (size_t)sub * k;
        long long* sI = outI + (size_t)sub * k;
        const int kl = MFMA_KL, cap = list_cap(kl), FQ = MFMA_QB2;
        int mwv = 8192 / cap;
        if (mwv < 1) mwv = 1;
        if (mwv > 16) mwv = 16;
        int p1 = 0;
        if (ns > 0) {
            if ((rc = split64_scan_launch(X, ns, 0, d, sq, nsub, mpart, nullptr, st, gate))) return rc;
            p1 = split64_lists(ns);
            hipLaunchKernelGGL(merge_keys_kernel, dim3(nsub), dim3(mwv * 64), (size_t)mwv * cap * 8, st, mpart, p1, FQ, kl, cap,
                               (const long long*)nullptr, 0ll, cand_scores, cand_rows, 0, gate);
            WISE_LAUNCH_CHECK("merge_keys_kernel (gated)");
            if ((rc = sample_threshold_launch(cand_scores, cand_rows, tau0, st, kl, gate))) return rc;
        }
        if ((rc = split64_scan_launch(X + (size_t)ns * d, N - ns, ns, d, sq, nsub, mpart + (size_t)p1 * FQ * kl,
                                      ns > 0 ? tau0 : nullptr, st, gate)))
            return rc;
        hipLaunchKernelGGL(merge_keys_kernel, dim3(nsub), dim3(mwv * 64), (size_t)mwv * cap * 8, st, mpart,
                           p1 + split64_lists(N - ns), FQ, kl, cap, (const long long*)nullptr, 0ll, cand_scores, cand_rows, 0,
                           gate);
        WISE_LAUNCH_CHECK("merge_keys_kernel (gated)");
        if ((rc = rescore_launch(X, d, sq, cand_rows, nsub, k, ids, id_base, sD, sI, st, gate))) return rc;
    }
    return WISE_OK;
}
}  // namespace wise

extern "C" int wise_ip_topk_shadow_f32(const float* X, const uint16_t* Xb, const float* norms, int64_t N, int d,
                                       const float* Q, int nq, int k, const int64_t* ids, int64_t id_base, float* outD,
                                       int64_t* outI, int32_t* counters, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    WISE_CHECK_ARG(shadow_supported(d, k), "ip_topk_shadow: d=%d must be a multiple of 8 in [8,1024], k=%d in [1,1024]", d, k);
    WISE_CHECK_ARG(N > 0 && N < 0xFFFFFFFFll, "ip_topk_shadow: N=%lld out of range", (long long)N);
    WISE_CHECK_ARG(nq >= 1 && nq <= 1024, "ip_topk_shadow: nq=%d out of [1,1024]", nq);
    WISE_CHECK_ARG(X && Xb && norms && Q && outD && outI, "ip_topk_shadow: null pointer");
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Xb & 15) == 0 && ((uintptr_t)Q & 15) == 0,
                   "ip_topk_shadow: X, Xb and Q must be 16-byte aligned");
    const size_t need = wise_ip_topk_shadow_workspace_bytes(N, d, nq, k);
    if (!workspace || workspace_bytes < need) {
        set_error("ip_topk_shadow: workspace %zu < %zu bytes", workspace_bytes, need);
        return WISE_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    const long long* lids = reinterpret_cast<const long long*>(ids);
    long long* lI = reinterpret_cast<long long*>(outI);
    // two or more queries on the matrix cores where the 64-query kernels apply (k <= 12: the fallback keeps 16
    // candidates), else one query at a time
    // (from two queries on: a 64-query pass costs 2.2 ms at 10M x 512 whatever it carries, two single-query searches
    // 3.4 ms; the VALU scan with 2 or 4 queries in registers is bound by its cross-lane reductions, 3.9 / 6.7 ms)
    // an index too small for a sample (a few hundred MB at most) is answered by the f32 scans directly
    if (N < COLLECT_MIN_ROWS) {
        const size_t fneed = wise_ip_topk_workspace_bytes(N, d, nq, k);
        if (fneed == 0 || fneed > workspace_bytes) { set_error("ip_topk_shadow: workspace %zu < %zu bytes", workspace_bytes, fneed); return WISE_E_WORKSPACE; }
        return wise_ip_topk_f32(X, N, d, Q, nq, k, ids, id_base, outD, outI, workspace, workspace_bytes, stream);
    }
    // (k <= 12: the split-bf16 scan of the f32 rows as the gated fallback; 12 < k <= 128: the f32 VALU scan)
    const bool batched = d <= 512 && nq >= 2 && shadow64_supported(d) &&
                         ((k <= MFMA_KC && mfma_split_supported(d, 8, k) && split64_supported(d) && split_direct_enabled()) ||
                          (k > MFMA_KC && k <= SHADOW_BATCH_K && nq >= 3));
    // 512 < d <= 1024 (768: the ViT-L/14 dimension): the f32 VALU scan as the gated fallback; 64 queries per pass with
    // one-piece queries (their images fit LDS), 32 with two pieces; worth it from 3 queries on (a pass moves the bf16
    // rows once: 2.7 ms at 10M x 768, a single query 2.4 ms)
    const bool batched32 = !batched && nq >= 3 && k <= SHADOW_BATCH_K && d > 512 && (shadow64_supported(d) || shadow32_supported(d));
    if (batched || batched32) {
        const int qmax = shadow_pass_queries(d);    // 128 (one-piece queries, d <= 512), 64 or 32
        for (int q0 = 0, qb = qmax; q0 < nq; q0 += qb) {
            qb = (qmax == 128 && nq - q0 <= 64) ? 64 : qmax;     // a half-empty 128-query pass costs 6 % more than a 64-query one
            const int nqa = nq - q0 < qb ? nq - q0 : qb;
            int rc = shadow_search_pass(X, Xb, norms, N, d, Q + (size_t)q0 * d, nqa, k, lids, (long long)id_base,
                                        outD + (size_t)q0 * k, lI + (size_t)q0 * k, counters, wsb, st, qb);
            if (rc) return rc;
        }
        return WISE_OK;
    }
    // otherwise one query at a time in the threshold form
    for (int q = 0; q < nq; ++q) {
        int rc = shadow_search_one(X, Xb, norms, N, d, Q + (size_t)q * d, k, lids, (long long)id_base,
                                   outD + (size_t)q * k, lI + (size_t)q * k, counters, wsb, st);
        if (rc) return rc;
    }
    return WISE_OK;
}

// The same search over the int8 shadow (wise_ip_shadow_i8), one query at a time in the threshold form: the two scans read
// N (d + 4) bytes instead of 2 N d; thresholds, refinement, exact re-scoring from X and the gated f32 scan are the bf16
// form's kernels (norms carries the int8 error bound).  Same workspace as wise_ip_topk_shadow_f32.
extern "C" int wise_ip_topk_shadow8_f32(const float* X, const int8_t* Xq, const float* scales, const float* norms, int64_t N,
                                        int d, const float* Q, int nq, int k, const int64_t* ids, int64_t id_base, float* outD,
                                        int64_t* outI, int32_t* counters, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    WISE_CHECK_ARG(shadow_supported(d, k) && d % 16 == 0, "ip_topk_shadow8: d=%d must be a multiple of 16 in [16,1024], k=%d in [1,1024]", d, k);
    WISE_CHECK_ARG(N > 0 && N < 0xFFFFFFFFll, "ip_topk_shadow8: N=%lld out of range", (long long)N);
    WISE_CHECK_ARG(nq >= 1 && nq <= 1024, "ip_topk_shadow8: nq=%d out of [1,1024]", nq);
    WISE_CHECK_ARG(X && Xq && scales && norms && Q && outD && outI, "ip_topk_shadow8: null pointer");
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Xq & 15) == 0 && ((uintptr_t)Q & 15) == 0,
                   "ip_topk_shadow8: X, Xq and Q must be 16-byte aligned");
    const size_t need = wise_ip_topk_shadow_workspace_bytes(N, d, nq, k);
    if (!workspace || workspace_bytes < need) {
        set_error("ip_topk_shadow8: workspace %zu < %zu bytes", workspace_bytes, need);
        return WISE_E_WORKSPACE;
    }
    if (N < COLLECT_MIN_ROWS)
        return wise_ip_topk_f32(X, N, d, Q, nq, k, ids, id_base, outD, outI, workspace, workspace_bytes, stream);
    hipStream_t st = (hipStream_t)stream;
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    for (int q = 0; q < nq; ++q) {
        int rc = shadow_search_one(X, nullptr, norms, N, d, Q + (size_t)q * d, k, reinterpret_cast<const long long*>(ids),
                                   (long long)id_base, outD + (size_t)q * k, reinterpret_cast<long long*>(outI) + (size_t)q * k,
                                   counters, wsb, st, reinterpret_cast<const signed char*>(Xq), scales);
        if (rc) return rc;
    }
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// Dense scores S[nq, N] = Q . X^T in exact f32 on the matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain,
// bit-identical to a scalar loop) — the coarse stage of IndexIVFFlat when nprobe is a sizeable part of nlist (the
// reference's nprobe = 1024 of 31,620 cells, config.py:19 / api/routes.py:899-902): a threshold list stops filtering
// there, so all scores are written and select_topk_kernel picks the nprobe best.
// Block = 4 waves = 128 rows of X x 32 queries; a wave owns a 32 x 32 tile (16 accumulator registers).  X and Q tiles
// go through LDS in 64-column chunks (row stride 65 floats: the MFMA operand read — 32 lanes, 32 different rows, one
// column — is conflict-free).  ~2 x 988 x 8 x 256 MFMAs for 31,620 x 512 x 256 queries: tens of microseconds.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ip_scores_f32_kernel(const float* __restrict__ X, long long N, int d,
                                                            const float* __restrict__ Q, int nq,
                                                            float* __restrict__ S /*[nq][N]*/) {
    constexpr int KC = 64, LD = KC + 1;
    __shared__ float xs[128 * LD];
    __shared__ float qs[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long n0 = (long long)blockIdx.x * 128;
    const int q0 = blockIdx.y * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int c0 = 0; c0 < d; c0 += KC) {
        // stage: 128 x 64 floats of X (8 float4 per thread) and 32 x 64 of Q (2 per thread); zero past the edges
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int idx = t * 256 + tid, r = idx >> 4, c4 = (idx & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + r < N && c0 + c4 < d) v = *reinterpret_cast<const float4*>(X + (size_t)(n0 + r) * d + c0 + c4);
            float* dst = xs + r * LD + c4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = t * 256 + tid, r = idx >> 4, c4 = (idx & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + r < nq && c0 + c4 < d) v = *reinterpret_cast<const float4*>(Q + (size_t)(q0 + r) * d + c0 + c4);
            float* dst = qs + r * LD + c4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        __syncthreads();
        const float* xa = xs + (wave * 32 + (lane & 31)) * LD + (lane >> 5);
        const float* qb = qs + (lane & 31) * LD + (lane >> 5);
#pragma unroll 8
        for (int kk = 0; kk < KC; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk], qb[kk], acc, 0, 0, 0);   // D[row of X][query]
        __syncthreads();
    }
    // acc[reg]: X row (reg&3) + 8*(reg>>2) + 4*(lane>>5) of the wave's 32, query lane&31
    const int q = q0 + (lane & 31);
    if (q < nq) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long long n = n0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (n < N) S[(size_t)q * N + n] = acc[reg];
        }
    }
}

extern "C" int wise_ip_scores_f32(const float* X, int64_t N, int d, const float* Q, int nq, float* scores, void* stream) {
    WISE_CHECK_ARG(X && Q && scores && N >= 1 && nq >= 1 && nq <= 65535 * 32 && d >= 4 && d % 4 == 0,
                   "ip_scores: bad argument (N=%lld d=%d nq=%d)", (long long)N, d, nq);
    WISE_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0, "ip_scores: X and Q must be 16-byte aligned");
    hipLaunchKernelGGL(ip_scores_f32_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)((nq + 31) / 32)), dim3(256), 0,
                       (hipStream_t)stream, X, (long long)N, d, Q, nq, scores);
    WISE_LAUNCH_CHECK("ip_scores_f32_kernel");
    return WISE_OK;
}

extern "C" int wise_select_topk_f32(const float* scores, int rows, int n, int k, int64_t* out, void* stream) {
    WISE_CHECK_ARG(scores && out && rows >= 1 && n >= 1 && k >= 1, "select_topk: bad argument");
    hipLaunchKernelGGL(select_topk_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, scores, n, k,
                       reinterpret_cast<long long*>(out));
    WISE_LAUNCH_CHECK("select_topk_kernel");
    return WISE_OK;
}

extern "C" int wise_topk_merge(const float* inD, const int64_t* inI, int parts, int nq, int k, float* outD,
                               int64_t* outI, void* stream) {
    WISE_CHECK_ARG(inD && inI && outD && outI, "topk_merge: null pointer");
    WISE_CHECK_ARG(parts >= 1 && nq >= 1 && k >= 1 && k <= 2048 && (long long)parts * k <= 65536,
                   "topk_merge: parts=%d nq=%d k=%d out of range", parts, nq, k);
    const int cap = list_cap(k);
    const size_t lds = (size_t)cap * 8;
    hipLaunchKernelGGL(merge_pairs_kernel, dim3(nq), dim3(64), lds, (hipStream_t)stream, inD,
                       reinterpret_cast<const long long*>(inI), parts, nq, k, cap, outD,
                       reinterpret_cast<long long*>(outI));
    WISE_LAUNCH_CHECK("merge_pairs_kernel");
    return WISE_OK;
}

extern "C" int wise_reconstruct_batch(const float* X, int64_t N, int d, const int64_t* ids, int64_t id_base,
                                      const int64_t* query_ids, int n, float* out, void* stream) {
    WISE_CHECK_ARG(X && query_ids && out && n >= 0 && d >= 1, "reconstruct_batch: bad argument");
    if (n == 0) return WISE_OK;
    hipLaunchKernelGGL(reconstruct_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, X, (long long)N, d,
                       reinterpret_cast<const long long*>(ids), (long long)id_base,
                       reinterpret_cast<const long long*>(query_ids), n, out);
    WISE_LAUNCH_CHECK("reconstruct_kernel");
    return WISE_OK;
}
