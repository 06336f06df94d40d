// HP-2: building an IndexIVFFlat — the k-means of `index.train(features)` and the list assignment of `index.add_with_ids(...)`
// (src/index/feature_search_index.py:53-76: faiss IndexIVFFlat over 10 round(sqrt(N)) cells; faiss Clustering / IndexIVF.add).
// Through round 3 the bookkeeping around this library's own score kernel ran on ATen (argmax, index_add_, bincount, argsort,
// cumsum, fancy indexing): the last vendor kernels doing numeric work inside wise_amd/ (VERDICT r03, weak item 10).  Here:
//   wise_ivf_argmax          nearest centroid of every row from its score row (ties: the lowest list, as torch.argmax)
//   wise_ivf_group           rows grouped by list, STABLE (a list keeps its rows in order of insertion): least-significant-digit
//                            radix sort of (list, row) by 8-bit digits — per-workgroup digit histograms, one scan, a scatter
//                            whose rank inside a workgroup comes from wave ballots — plus the lists' sizes and offsets
//   wise_ivf_list_sums       per list the sum of its rows IN THAT ORDER (one workgroup per list, fixed order of addition: the same
//                            bits run after run, which a float atomicAdd scatter does not give), the spherical k-means update
//   wise_ivf_normalize_rows  c / max(||c||, 1e-20); wise_ivf_reseed: an empty cell becomes a perturbed copy of a full one
//   wise_ivf_gather_rows / _i64, wise_ivf_expand_lists   the copies of add() / finalize
// All HBM-bound integer / copy work: n * d * 4 bytes per pass over the rows, n * 16 bytes per sort pass.
#include "common.h"

namespace wise {
namespace ivf_build {

constexpr int SORT_ITEMS = 8;                     // keys per thread of a sort workgroup
constexpr int SORT_BLOCK = 256 * SORT_ITEMS;      // keys per workgroup

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ s, int rows, int n, long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = s + (size_t)row * n;
    float best = -INFINITY;
    int bi = 0x7fffffff;                      // (a row of NaNs keeps it: list 0)
    for (int c = lane; c < n; c += 64) {
        const float v = r[c];
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

// ---- radix sort pass: keys k[i] (list numbers, 32 bits used), payload v[i]; digit = (k >> shift) & 255
__global__ __launch_bounds__(256) void sort_hist_kernel(const int* __restrict__ k, long long n, int shift, unsigned* __restrict__ hist /*[256][nblk]*/,
                                                        int nblk) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * SORT_BLOCK;
#pragma unroll
    for (int t = 0; t < SORT_ITEMS; ++t) {
        const long long i = base + t * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(k[i] >> shift) & 255], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];      // digit-major: one scan gives every (digit, workgroup) its base
}

// exclusive scan of m unsigned counters in place (one workgroup; m up to a few million: 256 digits x workgroups)
__global__ __launch_bounds__(1024) void scan_kernel(unsigned* __restrict__ a, long long m) {
    __shared__ unsigned long long part[1024];
    const long long per = (m + 1023) / 1024, lo = per * threadIdx.x, hi = lo + per < m ? lo + per : m;
    unsigned long long s = 0;
    for (long long i = lo; i < hi; ++i) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (long long i = lo; i < hi; ++i) { const unsigned v = a[i]; a[i] = (unsigned)run; run += v; }
}

// stable scatter: element i of the workgroup goes to base(digit, workgroup) + (number of earlier elements of the workgroup with
// that digit).  Earlier = lower i: the workgroup walks its keys in SORT_ITEMS rounds of 256 consecutive keys; within a round the
// rank is (waves before mine with the digit: counts in LDS) + (lanes before mine in my wave: ballots over the digit's 8 bits).
__global__ __launch_bounds__(256) void sort_scatter_kernel(const int* __restrict__ k, const long long* __restrict__ v, long long n,
                                                           int shift, const unsigned* __restrict__ hist, int nblk,
                                                           int* __restrict__ ko, long long* __restrict__ vo) {
    __shared__ unsigned run[256];          // elements of each digit placed by earlier rounds
    __shared__ unsigned wcnt[4][256];      // this round: per wave, elements of each digit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    run[threadIdx.x] = hist[(size_t)threadIdx.x * nblk + blockIdx.x];
    const long long base = (long long)blockIdx.x * SORT_BLOCK;
    for (int t = 0; t < SORT_ITEMS; ++t) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
        const long long i = base + t * 256 + threadIdx.x;
        const bool live = i < n;
        const int key = live ? k[i] : 0;
        const unsigned dg = live ? (unsigned)((key >> shift) & 255) : 256u;
        // lanes of my wave with my digit
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long b = __ballot(live && ((dg >> bit) & 1));
            peers &= ((dg >> bit) & 1) ? b : ~b;
        }
        const unsigned before = (unsigned)__popcll(peers & ((1ull << lane) - 1));
        if (live && before == 0) wcnt[wave][dg] = (unsigned)__popcll(peers);      // the first lane of a digit group records its size
        __syncthreads();
        if (live) {
            unsigned off = run[dg] + before;
            for (int w = 0; w < wave; ++w) off += wcnt[w][dg];
            ko[off] = key;
            vo[off] = v[i];
        }
        __syncthreads();
        run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void iota_key_kernel(const long long* __restrict__ assign, long long n, int* __restrict__ k,
                                                       long long* __restrict__ v) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { k[i] = (int)assign[i]; v[i] = i; }
}
__global__ __launch_bounds__(256) void count_kernel(const long long* __restrict__ assign, long long n, int nlist, unsigned* __restrict__ cnt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const long long a = assign[i]; if (a >= 0 && a < nlist) atomicAdd(cnt + a, 1u); }
}
__global__ __launch_bounds__(1024) void offsets_kernel(const unsigned* __restrict__ cnt, int nlist, long long* __restrict__ off /*[nlist + 1]*/,
                                                       long long* __restrict__ cnt64 /*[nlist] or null*/) {
    __shared__ unsigned long long part[1024];
    const int per = (nlist + 1023) / 1024, lo = per * threadIdx.x, hi = lo + per < nlist ? lo + per : nlist;
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; }
        off[nlist] = (long long)run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { off[i] = (long long)run; if (cnt64) cnt64[i] = cnt[i]; run += cnt[i]; }
}

// sums[c][:] = sum over the list's rows in order (block = list, thread = 4 columns; 8 rows in flight per step, added in order)
__global__ __launch_bounds__(256) void list_sums_kernel(const float* __restrict__ x, const long long* __restrict__ order,
                                                        const long long* __restrict__ off, int d, float* __restrict__ sums) {
    const int c = blockIdx.x;
    const long long lo = off[c], hi = off[c + 1];
    for (int col = threadIdx.x * 4; col < d; col += 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long i = lo; i < hi; i += 8) {
            float4 v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[t] = i + t < hi ? *reinterpret_cast<const float4*>(x + (size_t)order[i + t] * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 8; ++t) { s.x += v[t].x; s.y += v[t].y; s.z += v[t].z; s.w += v[t].w; }
        }
        *reinterpret_cast<float4*>(sums + (size_t)c * d + col) = s;
    }
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, int rows, int d, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = in + (size_t)row * d;
    float q = 0.f;
    for (int c = lane; c < d; c += 64) q = fmaf(r[c], r[c], q);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float inv = 1.0f / fmaxf(sqrtf(q), 1e-20f);
    for (int c = lane; c < d; c += 64) out[(size_t)row * d + c] = r[c] * inv;
}

// sums[empty[e]][:] = sums[donor[e]][:] * (1 + 1e-3 sign(.)): an empty cell restarts next to a full one
__global__ __launch_bounds__(256) void reseed_kernel(float* __restrict__ sums, const long long* __restrict__ empty,
                                                     const long long* __restrict__ donor, int d) {
    const long long e = empty[blockIdx.x], s = donor[blockIdx.x];
    for (int c = threadIdx.x; c < d; c += 256) {
        const float v = sums[(size_t)s * d + c];
        sums[(size_t)e * d + c] = v * (1.0f + 1e-3f * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)));
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const long long* __restrict__ idx, long long n, int d,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)idx[row] * d);
    float4* dst = reinterpret_cast<float4*>(out + (size_t)row * d);
    for (int c = lane; c < d / 4; c += 64) dst[c] = src[c];
}
__global__ __launch_bounds__(256) void gather_i64_kernel(const long long* __restrict__ a, const long long* __restrict__ idx, long long n,
                                                         long long* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[idx[i]];
}
__global__ __launch_bounds__(256) void expand_lists_kernel(const long long* __restrict__ off, int nlist, long long* __restrict__ out) {
    const int c = blockIdx.x;
    for (long long i = off[c] + threadIdx.x; i < off[c + 1]; i += 256) out[i] = c;
}

}  // namespace ivf_build
}  // namespace wise

using namespace wise;
using namespace wise::ivf_build;

extern "C" int wise_ivf_argmax(const float* scores, int rows, int n, int64_t* out, void* stream) {
    WISE_CHECK_ARG(scores && out && rows >= 0 && n > 0, "ivf_argmax: bad argument");
    if (rows == 0) return WISE_OK;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores, rows, n, (long long*)out);
    WISE_LAUNCH_CHECK("ivf argmax_rows_kernel");
    return WISE_OK;
}

extern "C" size_t wise_ivf_group_workspace_bytes(int64_t n, int nlist) {
    if (n < 0 || nlist <= 0) return 0;
    const size_t nblk = (size_t)((n + SORT_BLOCK - 1) / SORT_BLOCK) + 1;
    return (size_t)n * (4 + 8) * 2 + nblk * 256 * 4 + (size_t)nlist * 4 + 256;
}

// order [n] = the rows grouped by list, stable; list_off [nlist + 1]; counts [nlist] (may be null).  assign values in [0, nlist).
extern "C" int wise_ivf_group(const int64_t* assign, int64_t n, int nlist, int64_t* order, int64_t* list_off, int64_t* counts,
                              void* workspace, size_t workspace_bytes, void* stream) {
    WISE_CHECK_ARG(list_off && n >= 0 && (n == 0 || (assign && order)) && nlist > 0 && nlist <= (1 << 24), "ivf_group: bad argument");
    WISE_CHECK_ARG(workspace && workspace_bytes >= wise_ivf_group_workspace_bytes(n, nlist), "ivf_group: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)((n + SORT_BLOCK - 1) / SORT_BLOCK);
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    long long* v0 = reinterpret_cast<long long*>(w);                 w += (size_t)n * 8;
    long long* v1 = reinterpret_cast<long long*>(w);                 w += (size_t)n * 8;
    int* k0 = reinterpret_cast<int*>(w);                             w += (size_t)n * 4;
    int* k1 = reinterpret_cast<int*>(w);                             w += (size_t)n * 4;
    w = reinterpret_cast<unsigned char*>(((uintptr_t)w + 255) & ~(uintptr_t)255);
    unsigned* hist = reinterpret_cast<unsigned*>(w);                 w += (size_t)(nblk + 1) * 256 * 4;
    unsigned* cnt = reinterpret_cast<unsigned*>(w);
    if (hipMemsetAsync(cnt, 0, (size_t)nlist * 4, st) != hipSuccess) { set_error("ivf_group: memset failed"); return WISE_E_INVALID; }
    if (n > 0) {
        const unsigned g256 = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(count_kernel, dim3(g256), dim3(256), 0, st, (const long long*)assign, (long long)n, nlist, cnt);
        hipLaunchKernelGGL(iota_key_kernel, dim3(g256), dim3(256), 0, st, (const long long*)assign, (long long)n, k0, v0);
        int passes = 1;
        while ((1ll << (8 * passes)) < nlist) ++passes;
        int* kin = k0; int* kout = k1; long long* vin = v0; long long* vout = v1;
        for (int p = 0; p < passes; ++p) {
            const bool last = p == passes - 1;
            long long* vdst = last ? reinterpret_cast<long long*>(order) : vout;      // the last pass writes the rows where they are wanted
            hipLaunchKernelGGL(sort_hist_kernel, dim3(nblk), dim3(256), 0, st, kin, (long long)n, 8 * p, hist, nblk);
            hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, hist, (long long)nblk * 256);
            hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk), dim3(256), 0, st, kin, vin, (long long)n, 8 * p, hist, nblk, kout, vdst);
            int* tk = kin; kin = kout; kout = tk;
            if (!last) { long long* tv = vin; vin = vout; vout = tv; }
        }
    }
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(1024), 0, st, cnt, nlist, (long long*)list_off, (long long*)counts);
    WISE_LAUNCH_CHECK("ivf_group");
    return WISE_OK;
}

extern "C" int wise_ivf_list_sums(const float* x, const int64_t* order, const int64_t* list_off, int nlist, int d, float* sums, void* stream) {
    WISE_CHECK_ARG(x && order && list_off && sums && nlist > 0 && d > 0 && d % 4 == 0, "ivf_list_sums: bad argument (d %% 4 == 0)");
    hipLaunchKernelGGL(list_sums_kernel, dim3(nlist), dim3(256), 0, (hipStream_t)stream, x, (const long long*)order, (const long long*)list_off, d, sums);
    WISE_LAUNCH_CHECK("ivf list_sums_kernel");
    return WISE_OK;
}
extern "C" int wise_ivf_normalize_rows(const float* in, int rows, int d, float* out, void* stream) {
    WISE_CHECK_ARG(in && out && rows >= 0 && d > 0, "ivf_normalize_rows: bad argument");
    if (rows == 0) return WISE_OK;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, in, rows, d, out);
    WISE_LAUNCH_CHECK("ivf normalize_rows_kernel");
    return WISE_OK;
}
extern "C" int wise_ivf_reseed(float* sums, const int64_t* empty, const int64_t* donor, int n_empty, int d, void* stream) {
    WISE_CHECK_ARG(sums && empty && donor && n_empty >= 0 && d > 0, "ivf_reseed: bad argument");
    if (n_empty == 0) return WISE_OK;
    hipLaunchKernelGGL(reseed_kernel, dim3(n_empty), dim3(256), 0, (hipStream_t)stream, sums, (const long long*)empty, (const long long*)donor, d);
    WISE_LAUNCH_CHECK("ivf reseed_kernel");
    return WISE_OK;
}
extern "C" int wise_ivf_gather_rows(const float* x, const int64_t* idx, int64_t n, int d, float* out, void* stream) {
    WISE_CHECK_ARG(x && idx && out && n >= 0 && d > 0 && d % 4 == 0, "ivf_gather_rows: bad argument (d %% 4 == 0)");
    if (n == 0) return WISE_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (const long long*)idx, (long long)n, d, out);
    WISE_LAUNCH_CHECK("ivf gather_rows_kernel");
    return WISE_OK;
}
extern "C" int wise_ivf_gather_i64(const int64_t* a, const int64_t* idx, int64_t n, int64_t* out, void* stream) {
    WISE_CHECK_ARG(a && idx && out && n >= 0, "ivf_gather_i64: bad argument");
    if (n == 0) return WISE_OK;
    hipLaunchKernelGGL(gather_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)a, (const long long*)idx, (long long)n, (long long*)out);
    WISE_LAUNCH_CHECK("ivf gather_i64_kernel");
    return WISE_OK;
}
extern "C" int wise_ivf_expand_lists(const int64_t* list_off, int nlist, int64_t* out, void* stream) {
    WISE_CHECK_ARG(list_off && out && nlist > 0, "ivf_expand_lists: bad argument");
    hipLaunchKernelGGL(expand_lists_kernel, dim3(nlist), dim3(256), 0, (hipStream_t)stream, (const long long*)list_off, nlist, (long long*)out);
    WISE_LAUNCH_CHECK("ivf expand_lists_kernel");
    return WISE_OK;
}
