// (debug) LDS canary: workgroups that fill their LDS allocation with a known pattern, wait, and check it, for as long
// as asked.  Run beside another kernel on a second stream it tells whether that kernel's LDS traffic stays inside its
// own allocation when both share a compute unit.  Not part of the product path; bound by tools/ and tests only.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__device__ __forceinline__ uint32_t canary(uint32_t block, uint32_t it, uint32_t i) {
    uint32_t h = block * 0x9E3779B1u ^ it * 0x85EBCA77u ^ i * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h | 1u;
}

// report: [0] mismatching words seen, [1..7] first mismatch: block, iteration, word index, got, expected, xcc/cu id, lds words
__global__ __launch_bounds__(256) void lds_canary_kernel(int iters, int spin, int words, uint32_t* report) {
    extern __shared__ uint32_t lds[];
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < words; i += 256) lds[i] = canary(blockIdx.x, it, i);
        __syncthreads();
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += 256) {
            const uint32_t got = lds[i], exp = canary(blockIdx.x, it, i);
            if (got != exp) {
                if (atomicAdd(report, 1u) == 0) {
                    report[1] = blockIdx.x; report[2] = it; report[3] = i; report[4] = got; report[5] = exp;
                    uint32_t hwid;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    report[6] = hwid; report[7] = words;
                }
            }
        }
        __syncthreads();
    }
}

// VGPR canary: every lane parks R known words in registers, idles, and checks them.
// report: [0] mismatches, then 8 words per event (up to 60): block, iteration, register slot, lane, got, expected, hwid
template <int R>
__global__ __launch_bounds__(256, 2) void vgpr_canary_kernel(int iters, int spin, uint32_t* report) {
    extern __shared__ uint32_t lds[];
    if (threadIdx.x == 0) lds[0] = 0;   // keep the allocation
    uint32_t x[R];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[r] = canary(blockIdx.x, it, r * 256 + threadIdx.x);
            asm volatile("" : "+v"(x[r]));
        }
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            asm volatile("" : "+v"(x[r]));
            const uint32_t exp = canary(blockIdx.x, it, r * 256 + threadIdx.x);
            if (x[r] != exp) {
                const uint32_t slot = atomicAdd(report, 1u);
                if (slot < 60) {
                    uint32_t hwid;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    uint32_t* q = report + 16 + slot * 8;
                    q[0] = blockIdx.x; q[1] = it; q[2] = r; q[3] = threadIdx.x; q[4] = x[r]; q[5] = exp; q[6] = hwid;
                }
            }
        }
    }
}

// synthetic neighbours: one hardware feature each, LDS sized like the fused HTSAT kernels (so they share a CU
// with whatever runs on the other stream)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256, 2) void neighbour_kernel(int what, int iters, const uint32_t* __restrict__ src,
                                                          uint32_t* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nb_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (what == 0) {          // LDS-DMA, 16 bytes per lane
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + ((size_t)((blockIdx.x * 64 + it * 8 + t) & 4095) * 256 + lane) * 4),
                    (__attribute__((address_space(3))) void*)(nb_lds + (wave * 8 + t) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += reinterpret_cast<uint32_t*>(nb_lds)[threadIdx.x];
        }
    } else if (what == 1) {   // LDS-DMA, 4 bytes per lane
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + (size_t)((blockIdx.x * 64 + it * 8 + t) & 4095) * 1024 + lane),
                    (__attribute__((address_space(3))) void*)(nb_lds + (wave * 8 + t) * 256), 4, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += reinterpret_cast<uint32_t*>(nb_lds)[threadIdx.x];
        }
    } else if (what == 2) {   // cross-lane permutes through the LDS crossbar
        uint32_t v = src[threadIdx.x];
        for (int it = 0; it < iters * 16; ++it) v += __builtin_amdgcn_ds_bpermute(((lane ^ (it & 63)) << 2), (int)v);
        acc = v;
    } else if (what == 3) {   // matrix cores only
        f32x4_t c = {0.f, 0.f, 0.f, 0.f};
        bf16x8_t a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(wave + i); }
        for (int it = 0; it < iters * 16; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        acc = __float_as_uint(c[0] + c[1] + c[2] + c[3]);
    } else if (what == 4) {   // plain LDS traffic, 16-byte reads and 8-byte writes
        uint4* l4 = reinterpret_cast<uint4*>(nb_lds);
        uint2* l2 = reinterpret_cast<uint2*>(nb_lds);
        for (int it = 0; it < iters * 4; ++it) {
            l2[(threadIdx.x * 5 + it) & 4095] = make_uint2(it, lane);
            __syncthreads();
            const uint4 q = l4[(threadIdx.x * 3 + it) & 2047];
            acc += q.x ^ q.y ^ q.z ^ q.w;
            __syncthreads();
        }
    } else {                  // global loads to registers only
        for (int it = 0; it < iters * 8; ++it)
            acc += src[((size_t)((blockIdx.x * 64 + it) & 4095) * 1024) + threadIdx.x];
    }
    if (acc == 0x13572468u) sink[0] = acc;
}

// packed-fp32 probe: a wave runs the same dependent chain of one VALU instruction form twice from the same start
// and compares the two results bit for bit.  report[what] counts lanes whose two runs disagree.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int WHAT>
__device__ __forceinline__ f32x2_t pk_step(f32x2_t x, f32x2_t c, f32x2_t d, f32x2_t sc) {
    f32x2_t r;
    if (WHAT == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(c), "v"(d));
    else if (WHAT == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(d));
    else if (WHAT == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
    else if (WHAT == 3) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(d));
    else if (WHAT == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(c), "v"(d));
    else if (WHAT == 5) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(d));
    else if (WHAT == 6) {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "v"(x.x), "v"(c.x), "v"(d.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(x.y), "v"(c.y), "v"(d.y));
    } else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(sc), "v"(x));
    return r;
}
template <int WHAT>
__device__ __forceinline__ f32x2_t pk_chain(f32x2_t x, f32x2_t c, f32x2_t d, f32x2_t sc) {
#pragma unroll
    for (int i = 0; i < 64; ++i) x = pk_step<WHAT>(x, c, d, sc);
    return x;
}
// chains that alternate a packed op with a consumer of another kind reading its result straight away
template <int WHAT>
__device__ __forceinline__ f32x2_t mixed_chain(f32x2_t x, f32x2_t c, f32x2_t d, uint32_t lds_slot, uint32_t& h) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        f32x2_t r;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(c), "v"(d));
        if (WHAT == 8) {            // 32-bit integer add reads both halves
            asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %2" : "+v"(h) : "v"(r.x), "v"(r.y));
        } else if (WHAT == 9) {     // LDS store data, read back
            asm volatile("ds_write_b64 %0, %1" ::"v"(lds_slot), "v"(r) : "memory");
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_slot) : "memory");
        } else if (WHAT == 10) {    // single-rate fp32 op reads the result
            float t;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(r.x), "v"(r.y), "v"(r.x));
            h += __float_as_uint(t);
        } else if (WHAT == 11) {    // quarter-rate integer multiply reads the result
            uint32_t t;
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"(r.x), "v"(r.y));
            h += t;
        } else if (WHAT == 14) {    // write-after-read: LDS store data overwritten by the next packed op
            f32x2_t t;
            asm volatile("ds_write_b64 %1, %0\n\tv_pk_add_f32 %0, %0, %2" : "+v"(r) : "v"(lds_slot), "v"(d) : "memory");
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds_slot) : "memory");
            h += __float_as_uint(r.x);
            r = t;
        } else if (WHAT == 15) {    // the same with the two-address form the FFT exchange uses
            f32x2_t r2 = r + d;
            f32x4_t q;
            asm volatile("ds_write2_b64 %2, %0, %1 offset1:1\n\tv_pk_add_f32 %1, %1, %3\n\tv_pk_mul_f32 %0, %0, %3"
                         : "+v"(r), "+v"(r2) : "v"(lds_slot), "v"(d) : "memory");
            asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(lds_slot) : "memory");
            h += __float_as_uint(q[2]) + __float_as_uint(q[3]) + __float_as_uint(r.x) + __float_as_uint(r2.y);
            r = f32x2_t{q[0], q[1]};
        } else if (WHAT == 12) {    // packed move with op_sel
            asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(r) : "v"(r));
        } else {                    // transcendental reads the result
            float t;
            asm volatile("v_rcp_f32 %0, %1" : "=v"(t) : "v"(r.x));
            h += __float_as_uint(t);
        }
        x = r;
    }
    return x;
}
template <int WHAT>
__global__ __launch_bounds__(256, 2) void mixed_probe_kernel(int iters, uint32_t* report) {
    extern __shared__ uint32_t lds[];
    if (threadIdx.x == 0) lds[8191] = 0;
    const int lane = threadIdx.x & 63;
    const f32x2_t c = {0.5f + 0.001f * lane, 0.25f + 0.002f * lane}, d = {1.f + lane, 2.f - 0.5f * lane};
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        const f32x2_t x0 = {0.125f * (it & 15) + lane, 3.f - lane};
        uint32_t ha = 0, hb = 0;
        const f32x2_t a = mixed_chain<WHAT>(x0, c, d, threadIdx.x * 16u, ha);
        const f32x2_t b = mixed_chain<WHAT>(x0, c, d, threadIdx.x * 16u, hb);
        bad += (__float_as_uint(a.x) != __float_as_uint(b.x)) || (__float_as_uint(a.y) != __float_as_uint(b.y)) || ha != hb;
    }
    if (bad) atomicAdd(report + WHAT, bad);
}

template <int WHAT>
__global__ __launch_bounds__(256, 2) void pk_probe_kernel(int iters, uint32_t* report) {
    extern __shared__ uint32_t lds[];
    if (threadIdx.x == 0) lds[0] = 0;
    const int lane = threadIdx.x & 63;
    const f32x2_t c = {0.5f + 0.001f * lane, 0.25f + 0.002f * lane}, d = {1.f + lane, 2.f - 0.5f * lane};
    f32x2_t sc = {0.999f, 1.001f};
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        const f32x2_t x0 = {0.125f * (it & 15) + lane, 3.f - lane};
        const f32x2_t a = pk_chain<WHAT>(x0, c, d, sc);
        const f32x2_t b = pk_chain<WHAT>(x0, c, d, sc);
        bad += (__float_as_uint(a.x) != __float_as_uint(b.x)) || (__float_as_uint(a.y) != __float_as_uint(b.y));
    }
    if (bad) atomicAdd(report + WHAT, bad);
}

// the shape of the kernel that failed in the product (candidate re-scoring): dot products of rows of X with a query,
// four rows per wave, operands straight from global loads, accumulated with PACKED f32 fmas (vector types keep the
// packing even with the SLP vectoriser off).  out[block][wave][4].
template <int FORM>   // 0: pairs of one row's elements; 1: the same element of two rows, query element broadcast (op_sel)
__global__ __launch_bounds__(1024) void pk_dot_probe_kernel(const float* __restrict__ X, int d, const float* __restrict__ Q,
                                                            const long long* __restrict__ rows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d4 = d >> 2;
    const float4* qv = reinterpret_cast<const float4*>(Q + (size_t)blockIdx.x * d);
    long long r[4];
    f32x2_t lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        r[u] = rows[(blockIdx.x * 16 + wave) * 4 + u];
        lo[u] = f32x2_t{0.f, 0.f};
        hi[u] = f32x2_t{0.f, 0.f};
    }
    for (int j = lane; j < d4; j += 64) {
        const float4 b = qv[j];
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)   // conditional loads, as in the product kernel (a negative row is "no candidate")
            a[u] = r[u] >= 0 ? reinterpret_cast<const float4*>(X + (size_t)r[u] * d)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (FORM == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lo[u] = __builtin_elementwise_fma(f32x2_t{a[u].x, a[u].y}, f32x2_t{b.x, b.y}, lo[u]);
                hi[u] = __builtin_elementwise_fma(f32x2_t{a[u].z, a[u].w}, f32x2_t{b.z, b.w}, hi[u]);
            }
        } else {
            // lo[0] = (p0, p1), lo[1] = (p2, p3): rows paired, one query element at a time
            lo[0] = __builtin_elementwise_fma(f32x2_t{a[0].x, a[1].x}, f32x2_t{b.x, b.x}, lo[0]);
            lo[1] = __builtin_elementwise_fma(f32x2_t{a[2].x, a[3].x}, f32x2_t{b.x, b.x}, lo[1]);
            lo[0] = __builtin_elementwise_fma(f32x2_t{a[0].y, a[1].y}, f32x2_t{b.y, b.y}, lo[0]);
            lo[1] = __builtin_elementwise_fma(f32x2_t{a[2].y, a[3].y}, f32x2_t{b.y, b.y}, lo[1]);
            lo[0] = __builtin_elementwise_fma(f32x2_t{a[0].z, a[1].z}, f32x2_t{b.z, b.z}, lo[0]);
            lo[1] = __builtin_elementwise_fma(f32x2_t{a[2].z, a[3].z}, f32x2_t{b.z, b.z}, lo[1]);
            lo[0] = __builtin_elementwise_fma(f32x2_t{a[0].w, a[1].w}, f32x2_t{b.w, b.w}, lo[0]);
            lo[1] = __builtin_elementwise_fma(f32x2_t{a[2].w, a[3].w}, f32x2_t{b.w, b.w}, lo[1]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float p = FORM == 0 ? (lo[u][0] + lo[u][1]) + (hi[u][0] + hi[u][1]) : lo[u >> 1][u & 1];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
        if (lane == 0) out[(blockIdx.x * 16 + wave) * 4 + u] = p;
    }
}

// the narrowed form: v_pk_fma_f32 with op_sel:[0,1,0] (both result halves read the HIGH half of src1), once with the
// destination allocated over src1 (what the register allocator did in both failing kernels) and once with a
// destination of its own; every result is checked against scalar fmas in the same lane.  report[form] counts wrong
// halves.
template <bool OVERLAP, int NOPS = 0>
__global__ __launch_bounds__(256) void pk_overlap_probe_kernel(int iters, uint32_t* report) {
    const int lane = threadIdx.x & 63;
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2_t a = {1.f + 0.001f * lane + it, 2.f - 0.003f * lane}, b = {0.5f + 0.01f * (it & 63), 1.25f + 0.002f * lane};
        f32x2_t c = {3.f + lane, -1.f + 0.5f * it};
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float e0 = __builtin_fmaf(a[0], b[1], c[0]), e1 = __builtin_fmaf(a[1], b[1], c[1]);
            f32x2_t d;
            if (OVERLAP) {
                d = b;
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(d) : "v"(a), "v"(c));
            } else if (NOPS == 0) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            } else if (NOPS == 1) {
                asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            } else {
                asm volatile("s_nop 3\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            }
            const bool w0 = __float_as_uint(d[0]) != __float_as_uint(e0), w1 = __float_as_uint(d[1]) != __float_as_uint(e1);
            bad += w0 + w1;
            if (!OVERLAP && (w0 || w1)) {
                // what did the wrong half compute?  the same product with src1's LOW half (op_sel dropped)?
                const float alt0 = __builtin_fmaf(a[0], b[0], c[0]), alt1 = __builtin_fmaf(a[1], b[0], c[1]);
                if (w0 && __float_as_uint(d[0]) == __float_as_uint(alt0)) atomicAdd(report + 2, 1u);
                if (w1 && __float_as_uint(d[1]) == __float_as_uint(alt1)) atomicAdd(report + 3, 1u);
                if (w0) atomicAdd(report + 4, 1u);
                if (w1) atomicAdd(report + 5, 1u);
                if (atomicAdd(report + 6, 1u) == 0) {
                    report[8] = __float_as_uint(a[0]); report[9] = __float_as_uint(a[1]); report[10] = __float_as_uint(b[0]);
                    report[11] = __float_as_uint(b[1]); report[12] = __float_as_uint(c[0]); report[13] = __float_as_uint(c[1]);
                    report[14] = __float_as_uint(d[0]); report[15] = __float_as_uint(d[1]);
                }
            }
            // next operands: keep magnitudes bounded
            a = f32x2_t{d[1] * 0.5f + 1.f, d[0] * 0.25f - 1.f};
            b = f32x2_t{b[1] * 0.75f + 0.1f, b[0] * 0.5f + 0.7f};
            c = f32x2_t{c[1] * 0.5f, c[0] * 0.5f + 0.3f};
        }
    }
    if (bad) atomicAdd(report + (OVERLAP ? 1 : NOPS == 0 ? 0 : 15 + NOPS), bad);
}
}  // namespace

extern "C" int wise_debug_pk_overlap_probe(int blocks, int iters, uint32_t* report, void* stream) {
    hipLaunchKernelGGL(pk_overlap_probe_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, report);
    hipLaunchKernelGGL(pk_overlap_probe_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, report);
    hipLaunchKernelGGL((pk_overlap_probe_kernel<false, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, report);
    hipLaunchKernelGGL((pk_overlap_probe_kernel<false, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, report);
    return (int)hipGetLastError();
}

extern "C" int wise_debug_pk_dot_probe(const float* X, int d, const float* Q, const long long* rows, int blocks, float* out,
                                       void* stream, int form) {
    if (form == 0)
        hipLaunchKernelGGL(pk_dot_probe_kernel<0>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, X, d, Q, rows, out);
    else
        hipLaunchKernelGGL(pk_dot_probe_kernel<1>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, X, d, Q, rows, out);
    return (int)hipGetLastError();
}

extern "C" int wise_debug_pk_probe(int what, int blocks, int lds_bytes, int iters, uint32_t* report, void* stream) {
    void (*k)(int, uint32_t*) = nullptr;
    switch (what) {
        case 0: k = pk_probe_kernel<0>; break;
        case 1: k = pk_probe_kernel<1>; break;
        case 2: k = pk_probe_kernel<2>; break;
        case 3: k = pk_probe_kernel<3>; break;
        case 4: k = pk_probe_kernel<4>; break;
        case 5: k = pk_probe_kernel<5>; break;
        case 6: k = pk_probe_kernel<6>; break;
        case 7: k = pk_probe_kernel<7>; break;
        case 8: k = mixed_probe_kernel<8>; break;
        case 9: k = mixed_probe_kernel<9>; break;
        case 10: k = mixed_probe_kernel<10>; break;
        case 11: k = mixed_probe_kernel<11>; break;
        case 12: k = mixed_probe_kernel<12>; break;
        case 13: k = mixed_probe_kernel<13>; break;
        case 14: k = mixed_probe_kernel<14>; break;
        default: k = mixed_probe_kernel<15>; break;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, iters, report);
    return (int)hipGetLastError();
}

extern "C" int wise_debug_neighbour(int what, int blocks, int lds_bytes, int iters, const uint32_t* src, uint32_t* sink,
                                    void* stream) {
    hipLaunchKernelGGL(neighbour_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, what, iters,
                       src, sink);
    return (int)hipGetLastError();
}

extern "C" int wise_debug_vgpr_canary(int blocks, int lds_bytes, int iters, int spin, uint32_t* report, void* stream) {
    hipLaunchKernelGGL(vgpr_canary_kernel<200>, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, iters,
                       spin, report);
    return (int)hipGetLastError();
}

extern "C" int wise_debug_lds_canary(int blocks, int lds_bytes, int iters, int spin, uint32_t* report, void* stream) {
    if (lds_bytes > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, iters, spin,
                       lds_bytes / 4, report);
    return (int)hipGetLastError();
}
