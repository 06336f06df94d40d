// Pieces shared by the bf16 GEMM kernels (gemm_bf16.hip, gemm_w4.h): epilogue modes, activations, the XCD-aware tile
// order, counted waits.  Device-only helpers; gfx950.
#pragma once
#include "common.h"

namespace wise {

enum : int { EPI_BF16 = 0, EPI_QUICKGELU = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_F32 = 4, EPI_GELU_TANH = 5, EPI_RELU = 6 };
constexpr bool bf16_out(int mode) { return mode == EPI_BF16 || mode == EPI_QUICKGELU || mode == EPI_GELU || mode == EPI_GELU_TANH || mode == EPI_RELU; }

// x * sigmoid(1.702 x) on the two native transcendentals (v_exp_f32 is 2^x, v_rcp_f32 ~1 ulp): an IEEE divide
// costs ~10 more instructions per element and the epilogue applies this to 128-160 elements per thread
__device__ __forceinline__ float act_quickgelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
// erf GELU, 0.5 x (1 + erf(x / sqrt 2)) = x Phi(x), as x * sigmoid(p(x)) with an odd degree-5 p fitted (minimax on
// [-8, 8], tools/fit_gelu.py) to the erf form: |error| <= 2.6e-5 everywhere — a hundredth of the bf16 rounding the
// result gets right after (2^-9 relative), and 20x closer than the tanh form (4.7e-4).  Six plain VALU operations,
// one v_exp_f32 and one v_rcp_f32: half the issue slots of the Abramowitz-Stegun erf it replaces (|error| 5e-7, a
// precision the bf16 output could not carry).  It matters: HTSAT applies GELU to 0.96 G values per forward — at 20
// issue slots each that alone was ~0.6 ms of the 4.8 ms forward.  x^2 is clamped at 64 so that the x^5 term cannot turn
// p around for |x| > 10; beyond |x| = 8 the sigmoid is saturated either way.  -log2(e) is folded into the coefficients.
__device__ __forceinline__ float act_gelu(float x) {
    const float x2 = fminf(x * x, 64.f);
    const float p = x * fmaf(x2, fmaf(x2, 0.0010142630198970437f, -0.10677572339773178f), -2.301121234893799f);
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(p));
}
// GPT-2's gelu_new, 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x * sigmoid(2u): one exp2 and one rcp
__device__ __forceinline__ float act_gelu_tanh(float x) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * x, x * x, x);
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.f * 1.4426950408889634f * u));
}
// The three sigmoid-shaped activations are x * rcp(1 + exp2(p(x))): act_arg gives p.  act_apply_n evaluates N values stage
// by stage (all arguments, all exp2, all +1, all rcp, all products), pinned with scheduling barriers: written value by
// value the compiler emits each chain serially — five dependent instructions with a wait state behind each of the two
// transcendentals — and a wave that is alone on its SIMD (gemm_w4.h) pays that latency in full: 7.4k cycles for the
// 160 values a lane holds of a 160 x 256 tile against ~5k staged.
template <int MODE>
__device__ __forceinline__ float act_arg(float x) {
    if (MODE == EPI_QUICKGELU) return -1.702f * 1.4426950408889634f * x;
    if (MODE == EPI_GELU) {
        const float x2 = fminf(x * x, 64.f);
        return x * fmaf(x2, fmaf(x2, 0.0010142630198970437f, -0.10677572339773178f), -2.301121234893799f);
    }
    if (MODE == EPI_GELU_TANH) return -2.f * 1.4426950408889634f * (0.7978845608028654f * fmaf(0.044715f * x, x * x, x));
    return x;
}
template <int MODE, int N>
__device__ __forceinline__ void act_apply_n(float (&x)[N]) {
    if (MODE == EPI_RELU) {
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = fmaxf(x[k], 0.f);
    } else if (MODE == EPI_QUICKGELU || MODE == EPI_GELU || MODE == EPI_GELU_TANH) {
        float t[N];
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = act_arg<MODE>(x[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = __builtin_amdgcn_exp2f(t[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = 1.f + t[k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = __builtin_amdgcn_rcpf(t[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] *= t[k];
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MODE>
__device__ __forceinline__ float act_apply(float x) {
    if (MODE == EPI_QUICKGELU) return act_quickgelu(x);
    if (MODE == EPI_GELU) return act_gelu(x);
    if (MODE == EPI_GELU_TANH) return act_gelu_tanh(x);
    if (MODE == EPI_RELU) return fmaxf(x, 0.f);
    return x;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile order.  Blocks b, b+8, ... share an XCD (round-robin dispatch), so the remap first gives each
// XCD a contiguous run of virtual ids, then walks them in GROUP_M x tiles_n bands, m fastest inside a
// band ("grouped ordering"): the ~64 tiles an XCD has resident at once form a ~8x8 patch of the output
// that shares 8 A panels and 8 W panels, which fits the XCD's 4 MiB L2.  With plain n-fastest order the
// resident set spans every W panel (3.5 MB at N=2304) and thrashes: PMC showed 139 MB fetched from
// the memory side for a GEMM whose operands total 23 MB.  Placement only affects speed.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int group_m, int* tm, int* tn) {
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = group_m * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * group_m;
    const int gsize = min(tiles_m - first_m, group_m);
    const int in_group = bid - gid * per_group;
    *tm = first_m + in_group % gsize;
    *tn = in_group / gsize;
}

}  // namespace wise
