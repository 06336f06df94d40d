// HP-1: OpenCLIP VisionTransformer image tower on gfx950.
//
// Replaces `self.model.encode_image(x).float()` + L2 normalise at
// src/feature/mlfoundation_openclip.py:99-100 (arithmetic: SURVEY.md App. A.1, open_clip 2.24.0
// VisionTransformer; the same computation as transformers' CLIPVisionModelWithProjection,
// which is what the oracle is pinned against).
//
// Data layout in HBM (workspace, all row-major, rows padded to a multiple of 256 so no GEMM
// tile, 128- or 256-row, needs an M edge):
//   x    fp32 [Mp, W]    residual stream (kept fp32; LN statistics and the residual adds are fp32)
//   h    bf16 [Mp, W]    LayerNorm output / attention output (GEMM A operand)
//   qkv  bf16 [Mp, 3W]   in_proj output;       also patch-embed fp32 output [Mpp, W] before layer 0
//   a    bf16 [Mp, F]    MLP hidden;           also the im2col patches bf16 [Mpp, Kp] before layer 0
#include "transformer.h"
#include "gemm_w4.h"   // the volatile-asm MFMA statement and its retire / pin helpers (attn_oproj_fold_kernel)

namespace wise {
extern int g_ablate;  // timing-only ablation switches (wise_debug_set_gemm_flags)

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row in registers, exact two-pass statistics in fp32
// ------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x /* may be xo: no restrict */, const float* __restrict__ w,
                                                        const float* __restrict__ b, int rows, int W, float eps,
                                                        bf16_t* __restrict__ y, float* xo = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int w4 = W >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * W);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        v[i] = (c < w4) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
    uint2* yr = reinterpret_cast<uint2*>(y + (size_t)row * W);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            const float4 ww = reinterpret_cast<const float4*>(w)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
            uint2 pk;
            const float y0 = (v[i].x - mean) * rstd * ww.x + bb.x, y1 = (v[i].y - mean) * rstd * ww.y + bb.y;
            const float y2 = (v[i].z - mean) * rstd * ww.z + bb.z, y3 = (v[i].w - mean) * rstd * ww.w + bb.w;
            pk.x = pack_bf16x2(y0, y1);
            pk.y = pack_bf16x2(y2, y3);
            yr[c] = pk;
            // post-LN blocks (BERT family): the normalised row is also the fp32 residual stream (xo may alias x: a wave
            // owns its row and has read all of it)
            if (xo) reinterpret_cast<float4*>(xo + (size_t)row * W)[c] = make_float4(y0, y1, y2, y3);
        }
    }
}

// The same, R rows per wave (W <= 512 with many rows: HTSAT's stages 2-3, the CLIP text tower): a wave that normalises ONE row
// of 192 floats has a single 768-byte load in flight and lives for two butterflies — 131072 such waves per launch ran at
// 4.5 TB/s (33 -> 28.6 us with four rows per wave; at W = 768 / 12800 rows the one-row form is as fast: 11.4 vs 11.6 us).
// Here a wave takes R consecutive rows, all R x NV loads issued before the first sum and the R butterflies interleaved.
// Per row the arithmetic is layernorm_kernel's, operation for operation (same bits).
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, int rows, int W, float eps,
                                                             bf16_t* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int w4 = W >> 2;
    float4 v[R][NV];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * W);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            v[r][i] = (c < w4) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o, 64);
    float mean[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = s[r] / (float)W;
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < w4) {
                float a0 = v[r][i].x - mean[r], a1 = v[r][i].y - mean[r], a2 = v[r][i].z - mean[r], a3 = v[r][i].w - mean[r];
                q[r] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o, 64);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            const float4 ww = reinterpret_cast<const float4*>(w)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    const float rstd = rsqrtf(q[r] / (float)W + eps);
                    uint2 pk;
                    const float y0 = (v[r][i].x - mean[r]) * rstd * ww.x + bb.x, y1 = (v[r][i].y - mean[r]) * rstd * ww.y + bb.y;
                    const float y2 = (v[r][i].z - mean[r]) * rstd * ww.z + bb.z, y3 = (v[r][i].w - mean[r]) * rstd * ww.w + bb.w;
                    pk.x = pack_bf16x2(y0, y1);
                    pk.y = pack_bf16x2(y2, y3);
                    reinterpret_cast<uint2*>(y + (size_t)(row0 + r) * W)[c] = pk;
                }
            }
        }
    }
}

// Narrow rows (W <= 128, HTSAT's first stage has W = 96): half a wave per row, so a wave normalises two rows and
// 24 of every 32 lanes work instead of 24 of 64.  Same two-pass arithmetic; the reductions stay inside 32 lanes.
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, int rows, int W, float eps,
                                                               bf16_t* __restrict__ y) {
    const int lane = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = row < rows;            // both halves of a wave take part in the shuffles
    const int w4 = W >> 2;
    const bool on = live && lane < w4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) v = reinterpret_cast<const float4*>(x + (size_t)row * W)[lane];
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)W;
    float q = 0.f;
    if (on) {
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)W + eps);
    if (on) {
        const float4 ww = reinterpret_cast<const float4*>(w)[lane];
        const float4 bb = reinterpret_cast<const float4*>(b)[lane];
        uint2 pk;
        pk.x = pack_bf16x2((v.x - mean) * rstd * ww.x + bb.x, (v.y - mean) * rstd * ww.y + bb.y);
        pk.y = pack_bf16x2((v.z - mean) * rstd * ww.z + bb.z, (v.w - mean) * rstd * ww.w + bb.w);
        reinterpret_cast<uint2*>(y + (size_t)row * W)[lane] = pk;
    }
}

int layernorm_f32_bf16(const float* x, const float* w, const float* b, int rows, int W, float eps, bf16_t* y,
                       hipStream_t st) {
    WISE_CHECK_ARG(x && w && b && y, "layernorm: null pointer");
    WISE_CHECK_ARG(rows >= 0 && W >= 4 && W % 4 == 0 && W <= 4096, "layernorm: W=%d must be a multiple of 4, <= 4096", W);
    if (rows == 0 || (g_ablate & 2)) return WISE_OK;
    if (W <= 128) {
        hipLaunchKernelGGL(layernorm_narrow_kernel, dim3((rows + 7) / 8), dim3(256), 0, st, x, w, b, rows, W, eps, y);
        WISE_LAUNCH_CHECK("layernorm_narrow_kernel");
        return WISE_OK;
    }
    const int nv = (W / 4 + 63) / 64;
    // many rows of a hot width: several rows per wave (same bits as one row per wave; ablation bit 4 = one row per wave)
    if (nv <= 2 && rows >= 16384 && !(g_ablate & 16)) {
        const dim3 block(256);
        if (nv == 1) hipLaunchKernelGGL((layernorm_rows_kernel<1, 4>), dim3((rows + 15) / 16), block, 0, st, x, w, b, rows, W, eps, y);
        else hipLaunchKernelGGL((layernorm_rows_kernel<2, 4>), dim3((rows + 15) / 16), block, 0, st, x, w, b, rows, W, eps, y);
        WISE_LAUNCH_CHECK("layernorm_rows_kernel");
        return WISE_OK;
    }
    const dim3 grid((rows + 3) / 4), block(256);
#define LN_CASE(n) \
    case n: hipLaunchKernelGGL(layernorm_kernel<n>, grid, block, 0, st, x, w, b, rows, W, eps, y, (float*)nullptr); break;
    switch (nv) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        LN_CASE(9) LN_CASE(10) LN_CASE(11) LN_CASE(12) LN_CASE(13) LN_CASE(14) LN_CASE(15) LN_CASE(16)
    }
#undef LN_CASE
    WISE_LAUNCH_CHECK("layernorm_kernel");
    return WISE_OK;
}

// the same with a second output: xo fp32 [rows, W] = the normalised rows (may be x itself) — post-LN blocks
int layernorm_f32_dual(const float* x, const float* w, const float* b, int rows, int W, float eps, float* xo, bf16_t* y,
                       hipStream_t st) {
    WISE_CHECK_ARG(x && w && b && y && xo, "layernorm_dual: null pointer");
    WISE_CHECK_ARG(rows >= 0 && W > 128 && W % 4 == 0 && W <= 4096, "layernorm_dual: W=%d must be a multiple of 4 in (128, 4096]", W);
    if (rows == 0) return WISE_OK;
    const int nv = (W / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
#define LN_CASE(n) \
    case n: hipLaunchKernelGGL(layernorm_kernel<n>, grid, block, 0, st, x, w, b, rows, W, eps, y, xo); break;
    switch (nv) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        LN_CASE(9) LN_CASE(10) LN_CASE(11) LN_CASE(12) LN_CASE(13) LN_CASE(14) LN_CASE(15) LN_CASE(16)
    }
#undef LN_CASE
    WISE_LAUNCH_CHECK("layernorm_kernel (dual)");
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// Attention, head dim 64, no mask: one wave per (image, head, 64-query chunk), flash-style over
// 64-key blocks.  Computes S^T = K Q^T so that a lane owns whole query columns: the softmax
// reduction is in-lane plus two cross-lane steps, and the exponentiated accumulators are, as they
// stand, the B operand of O^T = V^T P^T (MFMA k-slot order permuted consistently on the V^T side).
// V^T comes from a wave-private LDS image [64 dh][64 keys (+4 pad)].
// ------------------------------------------------------------------------------------------------
constexpr int V_RS = 144;  // bytes per V row in LDS (64 dh bf16 = 128 B + 16 B pad)
using short4v = __attribute__((ext_vector_type(4))) short;

// One 64-key block of the online softmax for QT query tiles, shared by the attention kernels.  In: sacc = raw
// scores S^T (this lane: keys kbase + kt*16 + r, query column qt*16 + (lane & 15)).  Out: pf = the exponentiated
// block as the next MFMA's B operand; oacc / mrun / lrun updated.  The loop around it is bound by these VALU
// instructions, not by the 64 MFMAs of a block, so they are kept few: the 1/sqrt(64) * log2(e) scale rides in
// the FMA that subtracts the running maximum, 2^x is the bare v_exp_f32 (arguments are <= 0; results below
// 2^-126 may flush), and the key-validity compare exists only in the MASK instantiation (the last, partial key
// block, and causal attention).
template <int QT, bool MASK, int ND = 4>
__device__ __forceinline__ void softmax_block(f32x4 (&sacc)[4][QT], bf16x8 (&pf)[2][QT], f32x4 (&oacc)[ND][QT],
                                              float (&mrun)[QT], float (&lrun)[QT], int kbase, const int (&klim)[QT]) {
    // 1/sqrt(head dim) * log2(e); head dim = 16 ND (64, or 80 for ViT-H/14)
    const float sc = (ND == 4 ? 0.125f : 0.11180339887498948f) * 1.4426950408889634f;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MASK && kbase + kt * 16 + r >= klim[qt]) sacc[kt][qt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kt][qt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun[qt], mx * sc);
        const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        mrun[qt] = mnew;
        float ps = 0.f;
        float p[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[kt][r] = __builtin_amdgcn_exp2f(fmaf(sacc[kt][qt][r], sc, -mnew));
                ps += p[kt][r];
            }
        lrun[qt] = lrun[qt] * alpha + ps;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            oacc[dt][qt][0] *= alpha; oacc[dt][qt][1] *= alpha;
            oacc[dt][qt][2] *= alpha; oacc[dt][qt][3] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f[r] = (__bf16)p[2 * ks][r];
                f[4 + r] = (__bf16)p[2 * ks + 1][r];
            }
            pf[ks][qt] = f;
        }
    }
}

// (register budget stated explicitly: left alone the compiler spends 200-340 registers on load hoisting and the kernel,
// which lives on latency hiding, drops to 1-2 waves per SIMD)
// DH = head dim: 64, or 80 (ViT-H/14: width 1280 over 16 heads) — then the QK^T contraction runs three 32-deep steps
// with the last 16 channels zero, the PV product has five 16-channel output tiles and a V row is 160 bytes.
// PF: the K fragments and the V rows of the NEXT key block travel while the current one is computed (64 more registers:
// the kernel is bound by the latency of those loads, block after block — a wave holding ONE query took as long over a
// sequence as a wave holding 64).
template <int QT, bool CAUSAL, int DH = 64, bool PF = false>
__global__ __launch_bounds__(256, (QT == 2 && DH == 64 && !PF) ? 4 : 2) void attention_kernel(const bf16_t* __restrict__ qkv, int B, int T, int H,
                                                        bf16_t* __restrict__ o, const int* __restrict__ lens = nullptr) {
    constexpr int NS = (DH + 31) / 32, ND = DH / 16, VRS = DH * 2 + 16;   // k-steps of QK^T, dh tiles of PV, V row stride
    __shared__ __attribute__((aligned(16))) unsigned char v_all[4][64 * VRS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nqc = (T + 16 * QT - 1) / (16 * QT);
    const long long item = (long long)blockIdx.x * 4 + wave;  // (b, h, qc)
    if (item >= (long long)B * H * nqc) return;
    const int qc = (int)(item % nqc);
    const int h = (int)((item / nqc) % H);
    const int b = (int)(item / ((long long)nqc * H));
    const int W = H * DH, W3 = 3 * W;
    const bf16_t* base = qkv + (size_t)b * T * W3;
    unsigned char* vimg = v_all[wave];
    const int l15 = lane & 15, g = lane >> 4;
    // keys the sequence has (right-padded batches of the BERT-family text towers: keys past lens[b] are masked for every
    // query); everything key-side below runs on Tk, the query side on T
    const int Tk = lens ? max(1, min(T, lens[b])) : T;

    // Q fragments: B operand, lane holds Q[query = qt*16 + l15][dh = s*32 + 8g .. +7]
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[QT][NS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int t = qc * (16 * QT) + qt * 16 + l15;
        if (t >= T) t = T - 1;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            qf[qt][s] = (s * 32 + g * 8 < DH) ? *reinterpret_cast<const bf16x8*>(base + (size_t)t * W3 + h * DH + s * 32 + g * 8)
                                              : zero8;
    }

    f32x4 oacc[ND][QT];  // [dt][qt]: O^T[dh = dt*16 + g*4 + r][query = qt*16 + l15]
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { mrun[qt] = -INFINITY; lrun[qt] = 0.f; }

    // causal (text tower): query t sees keys <= t, so key blocks past the chunk's last query are skipped
    const int q_last = min(T, (qc + 1) * (16 * QT)) - 1;
    const int nkb = CAUSAL ? (q_last >> 6) + 1 : (Tk + 63) >> 6;
    constexpr int CPR = DH / 8;                         // 16-byte chunks per V row (8, or 10)
    // loads of one key block: K fragments [kt][k-step] (A operand of S^T = K Q^T) and the lane's share of the V rows
    auto load_k = [&](int kb, bf16x8 (&kf)[4][NS]) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            int t = kb * 64 + kt * 16 + l15;
            if (t >= Tk) t = Tk - 1;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
                kf[kt][s2] = (s2 * 32 + g * 8 < DH) ? *reinterpret_cast<const bf16x8*>(base + (size_t)t * W3 + W + h * DH + s2 * 32 + g * 8)
                                                    : zero8;
        }
    };
    auto load_v = [&](int kb, uint4 (&vr)[CPR]) {
#pragma unroll
        for (int p = 0; p < CPR; ++p) {
            const int c = p * 64 + lane, key = c / CPR, part = c - key * CPR;
            int t = kb * 64 + key;
            const bool valid = t < Tk;
            if (!valid) t = Tk - 1;
            uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)t * W3 + 2 * W + h * DH + part * 8);
            if (!valid) v = make_uint4(0u, 0u, 0u, 0u);
            vr[p] = v;
        }
    };
    // HOIST (without PF): all of a block's V rows and K fragments are requested before the first is used — more loads in
    // flight per wave (T = 576: 960 -> 700 us per launch).  Not where registers are the budget: 32 queries per wave at four
    // waves per SIMD (T <= 64) and head width 80 keep the load-use-load-use form.
    constexpr bool HOIST = PF || (QT != 2 && DH == 64 && !CAUSAL) || (DH == 80 && QT == 2);
    bf16x8 kf_n[PF ? 4 : 1][NS];
    uint4 vr_n[PF ? CPR : 1];
    if constexpr (PF) { load_k(0, kf_n); load_v(0, vr_n); }
    for (int kb = 0; kb < nkb; ++kb) {
        // ---- V image, row-major [64 keys][64 dh]: 8 passes, lane copies 16 B of V row key = p*8 + lane/8;
        //      the transpose the PV product needs is done by ds_read_b64_tr_b16 on the way out
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 kf[HOIST ? 4 : 1][NS];
        if constexpr (HOIST) {
            uint4 vr[CPR];
            if constexpr (PF) {
#pragma unroll
                for (int p = 0; p < CPR; ++p) vr[p] = vr_n[p];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) kf[kt][s2] = kf_n[kt][s2];
            } else {
                load_v(kb, vr);
                load_k(kb, kf);
            }
#pragma unroll
            for (int p = 0; p < CPR; ++p) {
                const int c = p * 64 + lane, key = c / CPR, part = c - key * CPR;
                *reinterpret_cast<uint4*>(vimg + key * VRS + part * 16) = vr[p];
            }
        } else {
#pragma unroll
            for (int p = 0; p < CPR; ++p) {
                const int c = p * 64 + lane, key = c / CPR, part = c - key * CPR;
                int t = kb * 64 + key;
                const bool valid = t < Tk;
                if (!valid) t = Tk - 1;
                uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)t * W3 + 2 * W + h * DH + part * 8);
                if (!valid) v = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(vimg + key * VRS + part * 16) = v;
            }
        }
        if constexpr (PF) {
            if (kb + 1 < nkb) { load_k(kb + 1, kf_n); load_v(kb + 1, vr_n); }   // in flight through this block's MFMAs and softmax
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- S^T[kt][qt] = K Q^T
        f32x4 sacc[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            constexpr int KI = HOIST ? 1 : 0;          // kf row of this tile: kt (hoisted) or the single scratch row
            if constexpr (!HOIST) {
                int t = kb * 64 + kt * 16 + l15;
                if (t >= Tk) t = Tk - 1;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2)
                    kf[0][s2] = (s2 * 32 + g * 8 < DH) ? *reinterpret_cast<const bf16x8*>(base + (size_t)t * W3 + W + h * DH + s2 * 32 + g * 8)
                                                       : zero8;
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt * KI][s2], qf[qt][s2], c, 0, 0, 0);
                sacc[kt][qt] = c;
            }
        }
        // ---- online softmax per query column (qt, l15); this lane holds keys kt*16 + g*4 + r
        bf16x8 pf[2][QT];  // [ks][qt] B operand of the PV product
        {
            int klim[QT];  // keys this query sees
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) klim[qt] = CAUSAL ? min(T, qc * (16 * QT) + qt * 16 + l15 + 1) : Tk;
            // masking is a separate small pass (last, partial key block and causal attention only) so that the
            // softmax body is instantiated once: with both variants inlined the kernel needed 236 registers
            // instead of ~130 and lost half its occupancy
            if (CAUSAL || kb * 64 + 64 > Tk) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kb * 64 + kt * 16 + g * 4 + r >= klim[qt]) sacc[kt][qt][r] = -INFINITY;
            }
            softmax_block<QT, false, ND>(sacc, pf, oacc, mrun, lrun, kb * 64 + g * 4, klim);
        }
        // ---- O^T += V^T P^T ; A operand: V^T[dh = dt*16 + l15][slot 8g + j] with
        //      slot j<4 -> key ks*32 + g*4 + j ; j>=4 -> key ks*32 + 16 + g*4 + (j-4)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                // transposed LDS read: the 16 lanes of group g fetch a 4-key x 16-dh block; lane 4q+p supplies
                // the address of key row q, dh 4p..4p+3 and receives dh column (dt*16 + l15) of the 4 keys
                const unsigned char* blk = vimg + (ks * 32 + g * 4 + (l15 >> 2)) * VRS + (dt * 16 + (l15 & 3) * 4) * 2;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4v*)(blk));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4v*)(blk + 16 * VRS));
                union { short8 s; bf16x8 f; } cv;
                cv.s = short8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cv.f, pf[ks][qt], oacc[dt][qt], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store O[query][h*64 + dt*16 + g*4 + r]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        const int t = qc * (16 * QT) + qt * 16 + l15;
        if (t < T) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                uint2 pk;
                pk.x = pack_bf16x2(oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv);
                pk.y = pack_bf16x2(oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv);
                *reinterpret_cast<uint2*>(o + ((size_t)b * T + t) * W + h * DH + dt * 16 + g * 4) = pk;
            }
        }
    }
}

static int g_attn_qt = 0;  // query tiles (of 16) per wave; 0 = by sequence length (debug knob overrides)

int attention_bf16(const bf16_t* qkv, int B, int T, int H, bf16_t* o, hipStream_t st, bool causal, int dh, const int* lens) {
    WISE_CHECK_ARG(qkv && o && B > 0 && T > 0 && H > 0, "attention: bad argument");
    WISE_CHECK_ARG(dh == 64 || (dh == 80 && !causal), "attention: head dim %d (64, or 80 without a mask)", dh);
    WISE_CHECK_ARG(!lens || (dh == 64 && !causal), "attention: per-sequence key counts go with head dim 64 and no causal mask");
    if (g_ablate & 4) return WISE_OK;
    if (dh == 80) {
        // ViT-H/14 (T = 257): 32 queries per wave with the block's loads hoisted (227 registers); 48 per wave without the
        // hoisting measured 433 us per launch against 400 (64 would spill: 80 accumulator registers for O alone)
        const long long items = (long long)B * H * ((T + 31) / 32);
        hipLaunchKernelGGL((attention_kernel<2, false, 80>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, qkv, B, T, H, o, (const int*)nullptr);
        WISE_LAUNCH_CHECK("attention_kernel");
        return WISE_OK;
    }
    // 64 queries per wave with a block's loads hoisted from T = 33 on (T = 50, ViT-B/32: 24.4 -> 21.9 us per launch: K and
    // V are read once per head instead of twice); 32 queries per wave below that.  (Measured and dropped: a block-per-head kernel that stages K/V once in
    // LDS for all query chunks of a head — at T = 257 it was 5 % slower: the loop is bound by the softmax VALU work
    // and by latency at 2-3 waves per SIMD, not by the K/V re-reads, which hit L2.)
    // measured per launch (bs 256): T = 257: 64 queries/wave 327 us as it was, 290 with the hoisted loads, 48 queries +
    // prefetch 248; T = 576: 960 / 700 / 717; T = 197: 147 / 133 / 132 (tools/attn_bench.py)
    if (!causal && g_attn_qt == 3) {   // (debug variant) 48 queries per wave, next key block prefetched
        const long long it3 = (long long)B * H * ((T + 47) / 48);
        hipLaunchKernelGGL((attention_kernel<3, false, 64, true>), dim3((unsigned)((it3 + 3) / 4)), dim3(256), 0, st, qkv, B, T, H, o, lens);
        WISE_LAUNCH_CHECK("attention_kernel");
        return WISE_OK;
    }
    if (!causal && g_attn_qt == 5) {   // 32 queries per wave, next key block prefetched
        const long long it2 = (long long)B * H * ((T + 31) / 32);
        hipLaunchKernelGGL((attention_kernel<2, false, 64, true>), dim3((unsigned)((it2 + 3) / 4)), dim3(256), 0, st, qkv, B, T, H, o, lens);
        WISE_LAUNCH_CHECK("attention_kernel");
        return WISE_OK;
    }
    const int qt = g_attn_qt ? (g_attn_qt == 4 ? 4 : 2) : (T <= 32 ? 2 : 4);
    const int nqc = (T + 16 * qt - 1) / (16 * qt);
    const long long items = (long long)B * H * nqc;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    if (causal) {
        if (qt == 4)
            hipLaunchKernelGGL((attention_kernel<4, true>), grid, block, 0, st, qkv, B, T, H, o, (const int*)nullptr);
        else
            hipLaunchKernelGGL((attention_kernel<2, true>), grid, block, 0, st, qkv, B, T, H, o, (const int*)nullptr);
    } else if (qt == 4)
        hipLaunchKernelGGL((attention_kernel<4, false>), grid, block, 0, st, qkv, B, T, H, o, lens);
    else
        hipLaunchKernelGGL((attention_kernel<2, false>), grid, block, 0, st, qkv, B, T, H, o, lens);
    WISE_LAUNCH_CHECK("attention_kernel");
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// Fold mode, T <= 64 (ViT-B/32: 50 tokens): attention, out-projection, residual add and the rows' statistics in ONE kernel —
// replaces attention_kernel + gemm_w4_kernel<EPI_RESID, .., FOLD 2> of a block.  One workgroup per
// FRAME, one wave per head (H waves, H / 4 per SIMD):
//   A. wave h: the head's V rows into LDS, then its 64 query rows in two chunks of 32 (S^T = K Q^T, softmax, O^T = V^T P^T:
//      the arithmetic of attention_kernel<2> statement for statement); O_h as bf16 into columns 64h .. 64h+63 of an LDS image
//      [64 rows][W] — the very bytes its V rows occupied (a head's V image IS that column block: nothing else in LDS);
//   B. the frame's 64 x W rows times W_o^T: wave w owns output columns 64w .. 64w+63 (16 accumulator tiles), the weights come
//      straight from L2 as MFMA operands (a W x W bf16 matrix is 1.2 MB, resident in every XCD's L2; each workgroup reads it
//      once), three K-steps of them in flight; the activation fragments are ds_read_b128 from the image;
//   C. x = (acc + bias) + (hi + lo), written back as hi + lo; the row's sums over the wave's 64 columns by the SAME tree as
//      the GEMM epilogue's (gemm_w4.h epi_f32_pass: quads of columns, then 4-chunk groups, pairs of groups), the H partials
//      added in column order, rstd = rsqrt(var + eps).  The whole row lives in this workgroup: no arrival counter, no second
//      pass — and every bit equals the two-kernel form's (tests/test_gpu_vit_fold.py holds it to that).
// Time: the QKV rows are read once (59 MB at bs 256), the attention output never reaches HBM (-39 MB), one launch instead of
// two, and a frame's rows wait for nobody else's.  Roofline: HBM for phase A and C, L2 -> CU bandwidth + MFMA for phase B.
// ------------------------------------------------------------------------------------------------
// (debug twin only: phase ablation — bit 0 / 1 / 2 skip phase A / B / C — and s_memtime stamps of workgroup 0's waves, bit 3;
//  tools/attn_oproj_bench.py.  In the product `dbg` is the constant 0 and all of it folds away.)
#ifdef WISE_DEBUG_KNOBS
__device__ unsigned long long g_ao_stamps[12][8];
static int g_ao_dbg = 0;
#define AO_STAMP(k) do { if ((dbg & 8) && blockIdx.x == 0 && lane == 0) g_ao_stamps[wave][k] = __builtin_amdgcn_s_memtime(); } while (0)
#define AO_DBG_PARAM , int dbg
#define AO_DBG_ARG , g_ao_dbg
#else
#define AO_STAMP(k) do {} while (0)
#define AO_DBG_PARAM
#define AO_DBG_ARG
constexpr int dbg = 0;
#endif
template <int H>
__global__ __launch_bounds__(64 * H, 1) void attn_oproj_fold_kernel(const bf16_t* __restrict__ qkv, int T,
                                                                    const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                                    bf16_t* __restrict__ hi, long long lo_off,
                                                                    float* __restrict__ rstd, float eps AO_DBG_PARAM) {
    constexpr int W = 64 * H, W3 = 3 * W, ARS = 2 * W + 16;    // image row stride in bytes: +16 keeps ds_read_b128 of 16 rows conflict-free
    extern __shared__ __attribute__((aligned(16))) unsigned char ao_smem[];
    unsigned char* aimg = ao_smem;
    float* part = reinterpret_cast<float*>(ao_smem + 64 * ARS);   // [64 rows][H waves][2]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: it addresses the weights' descriptor)
    const int l15 = lane & 15, g = lane >> 4;
    const size_t row_first = (size_t)blockIdx.x * T;
    const bf16_t* base = qkv + row_first * W3;

    AO_STAMP(0);
    // ---- A: attention of head `wave`
    if (!(dbg & 1)) {
        const int h = wave;
        unsigned char* vimg = aimg + h * 128;
        // every load of the head up front: V rows, K fragments, the Q fragments of both chunks (one round trip, not five)
        uint4 vr[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int c = p * 64 + lane, key = c >> 3, part8 = c & 7;
            const int t = key < T ? key : T - 1;
            vr[p] = *reinterpret_cast<const uint4*>(base + (size_t)t * W3 + 2 * W + h * 64 + part8 * 8);
            if (key >= T) vr[p] = make_uint4(0u, 0u, 0u, 0u);
        }
        bf16x8 kf[4][2], qfa[2][2][2];   // (the second chunk's Q is asked for once the first chunk's scores are out: registers)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            int t = kt * 16 + l15;
            if (t >= T) t = T - 1;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) kf[kt][s2] = *reinterpret_cast<const bf16x8*>(base + (size_t)t * W3 + W + h * 64 + s2 * 32 + g * 8);
        }
        auto load_q = [&](int qc) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                int t = qc * 32 + qt * 16 + l15;
                if (t >= T) t = T - 1;
#pragma unroll
                for (int s = 0; s < 2; ++s) qfa[qc][qt][s] = *reinterpret_cast<const bf16x8*>(base + (size_t)t * W3 + h * 64 + s * 32 + g * 8);
            }
        };
        load_q(0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int c = p * 64 + lane, key = c >> 3, part8 = c & 7;
            *reinterpret_cast<uint4*>(vimg + key * ARS + part8 * 16) = vr[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint2 okeep[2][2][4];   // [chunk][qt][dt]: both chunks' O, written once the V rows they overwrite have been read
#pragma unroll
        for (int qc = 0; qc < 2; ++qc) {
            const bf16x8 (&qf)[2][2] = qfa[qc];
            f32x4 oacc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
            f32x4 sacc[4][2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][s2], qf[qt][s2], c, 0, 0, 0);
                    sacc[kt][qt] = c;
                }
            if (qc == 0) load_q(1);
            bf16x8 pf[2][2];
            {
                const int klim[2] = {T, T};
                if (T < 64) {
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (kt * 16 + g * 4 + r >= T) sacc[kt][qt][r] = -INFINITY;
                }
                softmax_block<2, false, 4>(sacc, pf, oacc, mrun, lrun, g * 4, klim);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const unsigned char* blk = vimg + (ks * 32 + g * 4 + (l15 >> 2)) * ARS + (dt * 16 + (l15 & 3) * 4) * 2;
                    const short4v lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(blk));
                    const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(blk + 16 * ARS));
                    union { short8 s; bf16x8 f; } cv;
                    cv.s = short8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cv.f, pf[ks][qt], oacc[dt][qt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float l = lrun[qt];
                l += __shfl_xor(l, 16, 64);
                l += __shfl_xor(l, 32, 64);
                const float inv = 1.f / l;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    okeep[qc][qt][dt].x = pack_bf16x2(oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv);
                    okeep[qc][qt][dt].y = pack_bf16x2(oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv);
                }
            }
        }
        // every transposed read of the V rows is behind us (their results fed the MFMAs above): O_h takes their place
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qc = 0; qc < 2; ++qc)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<uint2*>(aimg + (qc * 32 + qt * 16 + l15) * ARS + (h * 64 + dt * 16 + g * 4) * 2) = okeep[qc][qt][dt];
    }
    AO_STAMP(1);
    __syncthreads();
    AO_STAMP(2);

    // ---- B: C[64, 64w .. 64w+63] = A_img[64, W] @ Wt[64w .. , :]^T.  Weights as the MFMA A operand: a lane ends up with
    //      C[row = i*16 + l15][col = 64w + j*16 + 4g + r] in acc[i][j][r], like the GEMM family.
    const int n0 = wave * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(dbg & 2)) {
        // the weights in TILE order (wise_hip.h, wise_attention_oproj_fold): the fragment of (wave, K-step, k-half, column tile) is
        // 1 KiB contiguous, lane l's 16 bytes at l * 16 — every load instruction reads eight whole 128-byte lines, and
        // consecutive instructions walk forward through memory.  (Row-major W_o: 16 half-lines per instruction at a 1536-byte
        // stride, every wave of every workgroup of an XCD on the same few L2 channels at the same moment — measured 10 TB/s of
        // L2 hits chip-wide, 30 us for this phase, whatever the prefetch depth.)
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (size_t)n0 * W), 0, 64 * W * 2, 0x00020000);
        const int wv = lane * 16;
        const unsigned char* ap = aimg + l15 * ARS + g * 16;
        auto wload = [&](int j, int s32) {      // s32 = 2 * K-step + k-half
            union { w4::u32x4_t u; bf16x8 f; } cv;
            cv.u = __builtin_amdgcn_raw_buffer_load_b128(rW, wv, (s32 * 4 + j) * 1024, 0);
            return cv.f;
        };
        // K-steps of 64: a column tile's two fragments are the two halves of the same 128-byte lines, requested back to back
        // (32-deep steps asked for the second half a step later, when 12 waves' lines had long pushed it out of the 32 KiB L1).
        // ONE set of fragment registers per operand, one step ahead: the MFMAs run column tile by column tile, and a tile's
        // registers take the fragments of step s + 1 as soon as its four MFMAs have issued; the activation fragments of a
        // k-half likewise once the half's sixteen MFMAs have issued.
        constexpr int KS2 = W / 64;
        bf16x8 wf[2][4], af[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { wf[0][j] = wload(j, 0); wf[1][j] = wload(j, 1); }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[h2][i] = *reinterpret_cast<const bf16x8*>(ap + i * 16 * ARS + h2 * 64);
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) w4::mfma16v(acc[i][j], wf[h2][j], af[h2][i]);
                    if (s + 1 < KS2) wf[h2][j] = wload(j, 2 * (s + 1) + h2);
                }
                if (s + 1 < KS2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[h2][i] = *reinterpret_cast<const bf16x8*>(ap + i * 16 * ARS + (s + 1) * 128 + h2 * 64);
                }
            }
        }
        w4::mfma_retire();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) w4::pin_v(acc[i][j]);
    }

    AO_STAMP(3);
    // ---- C: residual add on hi + lo, the rows' statistics
    if (!(dbg & 4)) {
        // lane coordinates recomputed from an opaque copy of the thread index: nothing derived from them up there has to stay in
        // a register through phases A and B (at 168 registers per wave one such value went to scratch)
        int tid_c = threadIdx.x;
        asm volatile("" : "+v"(tid_c));
        const int l15 = tid_c & 15, g = (tid_c >> 4) & 3;
        float4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + n0 + j * 16 + g * 4);
        // the tile's residual values, all 32 loads in flight at once (the weights' registers are free now)
        uint2 rha[4][4], rla[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + l15;
            const bf16_t* ph = hi + (row_first + (row < T ? row : T - 1)) * W + n0 + g * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rha[i][j] = *reinterpret_cast<const uint2*>(ph + j * 16);
                rla[i][j] = *reinterpret_cast<const uint2*>(ph + lo_off + j * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + l15;
            const bool valid = row < T;
            bf16_t* ph = hi + (row_first + (valid ? row : T - 1)) * W + n0 + g * 4;
            const uint2 (&rh)[4] = rha[i];
            const uint2 (&rl)[4] = rla[i];
            float q1[4], q2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = acc[i][j];
                float4 v = make_float4(a[0] + bv[j].x, a[1] + bv[j].y, a[2] + bv[j].z, a[3] + bv[j].w);
                v.x += __uint_as_float(rh[j].x << 16) + __uint_as_float(rl[j].x << 16);
                v.y += __uint_as_float(rh[j].x & 0xffff0000u) + __uint_as_float(rl[j].x & 0xffff0000u);
                v.z += __uint_as_float(rh[j].y << 16) + __uint_as_float(rl[j].y << 16);
                v.w += __uint_as_float(rh[j].y & 0xffff0000u) + __uint_as_float(rl[j].y & 0xffff0000u);
                uint2 h2, l2;
                h2.x = pack_bf16x2(v.x, v.y); h2.y = pack_bf16x2(v.z, v.w);
                l2.x = pack_bf16x2(v.x - __uint_as_float(h2.x << 16), v.y - __uint_as_float(h2.x & 0xffff0000u));
                l2.y = pack_bf16x2(v.z - __uint_as_float(h2.y << 16), v.w - __uint_as_float(h2.y & 0xffff0000u));
                if (valid) {
                    *reinterpret_cast<uint2*>(ph + j * 16) = h2;
                    *reinterpret_cast<uint2*>(ph + lo_off + j * 16) = l2;
                }
                // chunk (4 columns) -> pair of chunks (g ^ 1) -> the four chunks of this 16-column tile (g ^ 2)
                float s1 = ((v.x + v.y) + v.z) + v.w;
                float s2 = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                q1[j] = s1; q2[j] = s2;
            }
            if (g == 0) {
                part[(row * H + wave) * 2] = (q1[0] + q1[1]) + (q1[2] + q1[3]);
                part[(row * H + wave) * 2 + 1] = (q2[0] + q2[1]) + (q2[2] + q2[3]);
            }
        }
    }
    AO_STAMP(4);
    __syncthreads();
    AO_STAMP(5);
    if (threadIdx.x < 64 && (int)threadIdx.x < T) {
        const int row = threadIdx.x;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < H; ++w) { s1 += part[(row * H + w) * 2]; s2 += part[(row * H + w) * 2 + 1]; }
        const float inv_n = 1.0f / (float)W;
        const float mean = s1 * inv_n;
        const float var = fmaxf(fmaf(-mean, mean, s2 * inv_n), 0.f);
        rstd[row_first + row] = rsqrtf(var + eps);
    }
    AO_STAMP(6);
}

static PerDeviceOnce g_attn_oproj_once;
bool attention_oproj_fold_ok(int T, int H, int dh) { return T >= 1 && T <= 64 && H == 12 && dh == 64; }
int attention_oproj_fold(const bf16_t* qkv, int B, int T, int H, const bf16_t* Wt, const float* bias, bf16_t* hi, long long lo_off,
                         float* rstd, float eps, hipStream_t st) {
    WISE_CHECK_ARG(qkv && Wt && bias && hi && rstd && B > 0, "attention_oproj_fold: bad argument");
    WISE_CHECK_ARG(attention_oproj_fold_ok(T, H, 64), "attention_oproj_fold: 1 <= T <= 64 and 12 heads of 64 (T=%d, H=%d)", T, H);
    constexpr int HH = 12;
    const int lds = 64 * (2 * 64 * HH + 16) + 64 * HH * 2 * 4;
    g_attn_oproj_once([&] { raise_lds_limit(reinterpret_cast<const void*>(attn_oproj_fold_kernel<HH>), lds); });
    hipLaunchKernelGGL((attn_oproj_fold_kernel<HH>), dim3((unsigned)B), dim3(64 * HH), (size_t)lds, st, qkv, T, Wt, bias, hi, lo_off, rstd, eps AO_DBG_ARG);
    WISE_LAUNCH_CHECK("attn_oproj_fold_kernel");
    return WISE_OK;
}

// ------------------------------------------------------------------------------------------------
// patch gather (im2col): images [B,3,S,S] fp32 or u8 -> patches bf16 [B*g*g, Kp], k = c*P*P + py*P + px
// u8 input applies (x/255 - mean)/std of the OpenAI CLIP transform (mlfoundation_openclip.py:81-90)
// ------------------------------------------------------------------------------------------------
// (u8 input) Normalize constants: OpenAI CLIP's for the CLIP towers, 0.5 / 0.5 for the SigLIP towers (open_clip preprocess_cfg)
struct NormConst { float mean[3], stdv[3]; };
static NormConst norm_const(int arch) {
    if (arch == 1) return NormConst{{0.5f, 0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}};
    return NormConst{{0.48145466f, 0.4578275f, 0.40821073f}, {0.26862954f, 0.26130258f, 0.27577711f}};
}

template <typename TIN>
__global__ __launch_bounds__(256) void patchify_kernel(const TIN* __restrict__ img, int B, int S, int P, int g,
                                                       int Kp, bf16_t* __restrict__ patches, NormConst nc) {
    const float* mean = nc.mean;
    const float* stdv = nc.stdv;
    const int prow = blockIdx.x;  // b*g*g + gy*g + gx
    const int b = prow / (g * g), gy = (prow / g) % g, gx = prow % g;
    const int K = 3 * P * P;
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
        float v = 0.f;
        if (k < K) {
            const int c = k / (P * P), py = (k / P) % P, px = k % P;
            const size_t src = (((size_t)b * 3 + c) * S + gy * P + py) * S + gx * P + px;
            if (sizeof(TIN) == 1)
                v = ((float)img[src] / 255.f - (c == 0 ? mean[0] : c == 1 ? mean[1] : mean[2])) /
                    (c == 0 ? stdv[0] : c == 1 ? stdv[1] : stdv[2]);  // ToTensor's true division, then Normalize
            else
                v = (float)img[src];
        }
        patches[(size_t)prow * Kp + k] = f32_to_bf16(v);
    }
}

// the same gather, 8 consecutive pixels of a patch row per thread (P % 8 == 0, S % 4 == 0, Kp == 3*P*P, 16-byte
// aligned input): two 16-byte loads (fp32) or one 8-byte load (u8) and one 16-byte store
template <typename TIN>
__global__ __launch_bounds__(256) void patchify8_kernel(const TIN* __restrict__ img, long long total8, int S, int P, int g,
                                                        bf16_t* __restrict__ patches, NormConst nc) {
    const float* mean = nc.mean;
    const float* stdv = nc.stdv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total8) return;
    const int K8 = 3 * P * P / 8;
    const long long prow = idx / K8;
    const int k = (int)(idx - prow * K8) * 8;
    const int b = (int)(prow / (g * g)), gy = (int)((prow / g) % g), gx = (int)(prow % g);
    const int c = k / (P * P), py = (k / P) % P, px = k % P;
    const size_t src = (((size_t)b * 3 + c) * S + gy * P + py) * S + gx * P + px;
    float v[8];
    if (sizeof(TIN) == 1) {
        const uint2 raw = *reinterpret_cast<const uint2*>(img + src);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned byte = ((e < 4 ? raw.x : raw.y) >> (8 * (e & 3))) & 0xFFu;
            v[e] = ((float)byte / 255.f - (c == 0 ? mean[0] : c == 1 ? mean[1] : mean[2])) /
                   (c == 0 ? stdv[0] : c == 1 ? stdv[1] : stdv[2]);  // ToTensor's true division, then Normalize
        }
    } else {
        const float4 a = *reinterpret_cast<const float4*>(img + src), bq = *reinterpret_cast<const float4*>(img + src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
    }
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(patches + (size_t)prow * (3 * P * P) + k) = o;
}

// timm towers without a class token and without ln_pre (SigLIP): x[row, :] = patch_out[row, :] + pos[row % T, :]
// (the patch embedding's bias is folded into pos by the packer).  float4 per thread.
__global__ __launch_bounds__(256) void embed_pos_kernel(const float* __restrict__ patch_out, const float* __restrict__ pos,
                                                        long long total4, int T, int W, float* __restrict__ x) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int w4 = W >> 2;
    const long long row = idx / w4;
    const int c = (int)(idx - row * w4);
    const float4 a = reinterpret_cast<const float4*>(patch_out)[idx];
    const float4 p = reinterpret_cast<const float4*>(pos)[(size_t)(row % T) * w4 + c];
    reinterpret_cast<float4*>(x)[idx] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
}

// Attention pooling with one latent query (timm AttentionPoolLatent, the 'map' head of the SigLIP towers): per (image,
// head) o = softmax(q_h . K_h^T / 8) V_h over the image's T tokens, head dim 64.  kv bf16 [B*T, 2W] (keys | values),
// qv fp32 [W] = q(latent) (constant: computed by the packer).  One wave per (image, head); lane = token slice.
__global__ __launch_bounds__(64) void map_pool_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ qv, int T, int W,
                                                      bf16_t* __restrict__ o /*[B, W]*/) {
    const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int W2 = 2 * W;
    float q[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) q[j] = qv[h * 64 + j] * 0.125f;
    const bf16_t* base = kv + (size_t)b * T * W2 + h * 64;
    float mx = -INFINITY;
    constexpr int MAXT = 16;                       // tokens per lane: T <= 1024
    float sc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int t = i * 64 + lane;
        float sdot = -INFINITY;
        if (t < T) {
            const uint4* kr = reinterpret_cast<const uint4*>(base + (size_t)t * W2);
            sdot = 0.f;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const uint4 u = kr[c8];
                const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sdot = fmaf(__uint_as_float(w4[e] << 16), q[c8 * 8 + 2 * e], sdot);
                    sdot = fmaf(__uint_as_float(w4[e] & 0xFFFF0000u), q[c8 * 8 + 2 * e + 1], sdot);
                }
            }
        }
        sc[i] = sdot;
        mx = fmaxf(mx, sdot);
    }
    mx = wave_max(mx);
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        sc[i] = (i * 64 + lane < T) ? __expf(sc[i] - mx) : 0.f;
        ps += sc[i];
    }
    ps = wave_sum(ps);
    float acc[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) acc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int t = i * 64 + lane;
        if (t < T) {
            const uint4* vr = reinterpret_cast<const uint4*>(base + (size_t)t * W2 + W);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const uint4 u = vr[c8];
                const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[c8 * 8 + 2 * e] = fmaf(sc[i], __uint_as_float(w4[e] << 16), acc[c8 * 8 + 2 * e]);
                    acc[c8 * 8 + 2 * e + 1] = fmaf(sc[i], __uint_as_float(w4[e] & 0xFFFF0000u), acc[c8 * 8 + 2 * e + 1]);
                }
            }
        }
    }
    const float inv = 1.f / ps;
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float tot = wave_sum(acc[j]);
        if (lane == j) mine = tot * inv;
    }
    o[(size_t)b * W + h * 64 + lane] = f32_to_bf16(mine);
}

// x[b*T + t, :] = ln_pre( (t == 0 ? cls : patch_out[b*g*g + t-1, :]) + pos[t, :] ), one wave per row
template <int NV>
__global__ __launch_bounds__(256) void embed_lnpre_kernel(const float* __restrict__ patch_out,
                                                          const float* __restrict__ cls, const float* __restrict__ pos,
                                                          const float* __restrict__ w, const float* __restrict__ bb,
                                                          int rows, int T, int W, float eps, float* __restrict__ x,
                                                          bf16_t* __restrict__ hcopy, float* __restrict__ rstd_out, long long lo_off) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / T, t = row % T;
    const int w4 = W >> 2;
    const float4* src = (t == 0) ? reinterpret_cast<const float4*>(cls)
                                 : reinterpret_cast<const float4*>(patch_out + ((size_t)b * (T - 1) + (t - 1)) * W);
    const float4* pr = reinterpret_cast<const float4*>(pos + (size_t)t * W);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            float4 a = src[c], p = pr[c];
            v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        } else
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * W);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            const float4 ww = reinterpret_cast<const float4*>(w)[c];
            const float4 b4 = reinterpret_cast<const float4*>(bb)[c];
            v[i] = make_float4((v[i].x - mean) * rstd * ww.x + b4.x, (v[i].y - mean) * rstd * ww.y + b4.y,
                               (v[i].z - mean) * rstd * ww.z + b4.z, (v[i].w - mean) * rstd * ww.w + b4.w);
            if (!hcopy) xr[c] = v[i];
        }
    }
    if (hcopy) {
        // fold mode (gemm_w4.h FoldArgs): the residual stream is hi + lo (two bf16 arrays where the fp32 rows would be); the
        // first block's QKV GEMM takes hi as its operand and the row's rstd as a scale
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * 64 + lane < w4) s2 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean2 = wave_sum(s2) / (float)W;
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < w4) {
                float a0 = v[i].x - mean2, a1 = v[i].y - mean2, a2 = v[i].z - mean2, a3 = v[i].w - mean2;
                q2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                const unsigned h0 = pack_bf16x2(v[i].x, v[i].y), h1 = pack_bf16x2(v[i].z, v[i].w);
                reinterpret_cast<uint2*>(hcopy + (size_t)row * W)[c] = make_uint2(h0, h1);
                reinterpret_cast<uint2*>(hcopy + lo_off + (size_t)row * W)[c] =
                    make_uint2(pack_bf16x2(v[i].x - __uint_as_float(h0 << 16), v[i].y - __uint_as_float(h0 & 0xffff0000u)),
                               pack_bf16x2(v[i].z - __uint_as_float(h1 << 16), v[i].w - __uint_as_float(h1 & 0xffff0000u)));
            }
        }
        const float r2 = rsqrtf(wave_sum(q2) / (float)W + eps);
        if (lane == 0) rstd_out[row] = r2;
    }
}

// LayerNorm of ONE row per sequence: x[b*T + pos[b], :] -> y[b, :] bf16 (pos == nullptr: row 0, the class
// token; the text tower passes its end-of-text positions) ; wave per sequence
template <int NV>
__global__ __launch_bounds__(256) void cls_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, int B, int T, int W, float eps,
                                                     bf16_t* __restrict__ y, const int* __restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (img >= B) return;
    const int w4 = W >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + ((size_t)img * T + (pos ? pos[img] : 0)) * W);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        v[i] = (c < w4) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i * 64 + lane < w4) {
            float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
    uint2* yr = reinterpret_cast<uint2*>(y + (size_t)img * W);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < w4) {
            const float4 ww = reinterpret_cast<const float4*>(w)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
            uint2 pk;
            pk.x = pack_bf16x2((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y);
            pk.y = pack_bf16x2((v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
            yr[c] = pk;
        }
    }
}

// out[r,:] = e[r,:] / ||e[r,:]||_2 (no epsilon: mlfoundation_openclip.py:100) ; wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ e, int rows, int D,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* er = e + (size_t)r * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += er[c] * er[c];
    const float nrm = sqrtf(wave_sum(s));
    for (int c = lane; c < D; c += 64) out[(size_t)r * D + c] = er[c] / nrm;
}

int l2norm_rows(const float* e, int rows, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, e, rows, D, out);
    WISE_LAUNCH_CHECK("l2norm_rows_kernel");
    return WISE_OK;
}

size_t transformer_splitk_bytes(int W, int F, int batch, int T) {
    // (a workspace sized for `batch` sequences serves every smaller batch: the skinny case is ONE sequence whatever
    // `batch` is, and the number of K slices depends on N and K alone)
    (void)batch;
    const int M = T, Mp = (M + 255) / 256 * 256;
    size_t b = gemm_splitk_bytes(Mp, M, 3 * W, W);
    b = std::max(b, gemm_splitk_bytes(Mp, M, W, W));
    b = std::max(b, gemm_splitk_bytes(Mp, M, F, W));
    return std::max(b, gemm_splitk_bytes(Mp, M, W, F));
}

// L pre-LN residual blocks over x fp32 [B*T (padded to 256), W]; h / qkv / a are the bf16 scratch operands
int transformer_blocks(const BlockWeights& bw, int L, int W, int H, int F, int act, int batch, int T, bool causal,
                       float* x, bf16_t* h, bf16_t* qkv, bf16_t* a, hipStream_t st, float eps, bool skinny, float* sk,
                       size_t sk_bytes) {
    // skinny (the text towers): a call of <= 128 rows may take the split-K kernels.  The image towers do not: their
    // one-stream, two-half-batch and two-batches-in-flight forms must return the same bits, and a half batch could fall
    // under 128 rows where the whole batch does not.
    const int M = batch * T, Mp = (M + 255) / 256 * 256;
    const int Mv = skinny ? M : Mp;
    int rc;
    // rows the LayerNorms touch: all M of them (the image towers also feed rows M..Mp of h to the GEMMs, which they never
    // wrote and whose results nobody reads)
    if (L > 0 && (rc = layernorm_f32_bf16(x, bw.pf + bw.ln1_w, bw.pf + bw.ln1_b, M, W, eps, h, st))) return rc;
    for (int l = 0; l < L; ++l) {
        const bf16_t* lwb = bw.wb + bw.per_layer_b * l;
        const float* lpf = bw.pf + bw.per_layer_f * l;
        if ((rc = gemm_bf16_rows(h, lwb + bw.in_proj, lpf + bw.in_b, Mp, Mv, 3 * W, W, 0, qkv, st, sk, sk_bytes))) return rc;
        if ((rc = attention_bf16(qkv, batch, T, H, h, st, causal, W / H))) return rc;
        // x += out_proj(attention); h = ln_2(x)   (one launch behind the split-K kernel for a single text query)
        if ((rc = gemm_resid_ln_rows(h, lwb + bw.out_proj, lpf + bw.out_b, Mp, Mv, M, W, W, x, lpf + bw.ln2_w, lpf + bw.ln2_b, eps,
                                     false, h, st, sk, sk_bytes)))
            return rc;
        // act: 0 QuickGELU, 1 erf GELU, 2 gelu_new (tanh) -> epilogue modes 1, 2, 5
        if ((rc = gemm_bf16_rows(h, lwb + bw.c_fc, lpf + bw.fc_b, Mp, Mv, F, W, act == 0 ? 1 : (act == 1 ? 2 : 5), a, st, sk, sk_bytes))) return rc;
        if (l + 1 < L) {   // x += c_proj(...); h = ln_1 of the NEXT block
            const float* npf = bw.pf + bw.per_layer_f * (l + 1);
            if ((rc = gemm_resid_ln_rows(a, lwb + bw.c_proj, lpf + bw.proj_b, Mp, Mv, M, W, F, x, npf + bw.ln1_w, npf + bw.ln1_b, eps,
                                         false, h, st, sk, sk_bytes)))
                return rc;
        } else if ((rc = gemm_bf16_rows(a, lwb + bw.c_proj, lpf + bw.proj_b, Mp, Mv, W, F, 3, x, st, sk, sk_bytes))) {
            return rc;
        }
    }
    return WISE_OK;
}

// The same L pre-LN blocks with every LayerNorm folded into the GEMMs around it (gemm_w4.h FoldArgs).  The residual stream is
// hi + lo — two bf16 arrays in the region the fp32 rows would take: hi [Mp, W], then lo [Mp, W] — and on entry rstd holds the
// rows' 1 / sqrt(var + eps) (embed_lnpre_kernel wrote all three); the weights are the packer's folded ones (in_proj / c_fc:
// gamma-scaled, row-centred; their biases: + W beta).  Seven launches per block become five, a residual GEMM reads and
// writes 4 bytes per element as before and NOTHING else touches the rows (the LayerNorm's pass over them is gone, and hi is
// the next GEMM's operand as it stands), and no row's result depends on the batch it sits in (one epilogue implementation,
// one reduction tree).  The attention output goes to `h`, which the unfolded form uses for the LayerNorm output.
static int transformer_blocks_fold(const BlockWeights& bw, int L, int W, int H, int F, int act, int batch, int T, float* x,
                                   bf16_t* h, bf16_t* qkv, bf16_t* a, float* stats, hipStream_t st, float eps, bool fuse_attn) {
    const int M = batch * T, Mp = (M + 255) / 256 * 256;
    int rc;
    float* rstd = stats;                        // [Mp] row scales; arrival counters and partial sums behind them
    if (hipMemsetAsync(stats + Mp, 0, gemm_fold_counters_bytes(Mp), st) != hipSuccess) {
        set_error("vit_forward: memset of the fold counters failed");
        return WISE_E_INVALID;
    }
    bf16_t* hi = reinterpret_cast<bf16_t*>(x);
    const long long lo_off = (long long)Mp * W;
    bf16_t* ao = h;     // attention output [Mp, W]
    for (int l = 0; l < L; ++l) {
        const bf16_t* lwb = bw.wb + bw.per_layer_b * l;
        const float* lpf = bw.pf + bw.per_layer_f * l;
        if ((rc = gemm_fold_bf16(hi, lwb + bw.in_proj, lpf + bw.in_b, rstd, Mp, 3 * W, W, 0, qkv, st))) return rc;
        if (fuse_attn) {   // ln_fold = 2: attention, out-projection, residual add and statistics of a frame in one workgroup
            if ((rc = attention_oproj_fold(qkv, batch, T, H, lwb + bw.out_proj, lpf + bw.out_b, hi, lo_off, rstd, eps, st))) return rc;
        } else {
            if ((rc = attention_bf16(qkv, batch, T, H, ao, st, false, W / H))) return rc;
            if ((rc = gemm_fold_resid(ao, lwb + bw.out_proj, lpf + bw.out_b, Mp, W, W, hi, lo_off, stats, eps, st))) return rc;
        }
        if ((rc = gemm_fold_bf16(hi, lwb + bw.c_fc, lpf + bw.fc_b, rstd, Mp, F, W, act == 0 ? 1 : (act == 1 ? 2 : 5), a, st))) return rc;
        if ((rc = gemm_fold_resid(a, lwb + bw.c_proj, lpf + bw.proj_b, Mp, W, F, hi, lo_off, stats, eps, st))) return rc;
    }
    return WISE_OK;
}

// out[b,:] = normalize( LN(x[b*T + pos[b], :]) @ proj ), projT bf16 [D,W]; hb bf16 [Bp,W] and e fp32 [Bp,D] scratch
int pooled_ln(const float* x, const float* ln_w, const float* ln_b, int batch, int T, int W, const int* pos,
              bf16_t* hb, hipStream_t st, float eps) {
    const int nv = (W / 4 + 63) / 64;
    const dim3 grid((batch + 3) / 4), block(256);
#define CLS_CASE(n) case n: hipLaunchKernelGGL(cls_ln_kernel<n>, grid, block, 0, st, x, ln_w, ln_b, batch, T, W, eps, \
                                               hb, pos); break;
    switch (nv) {
        CLS_CASE(1) CLS_CASE(2) CLS_CASE(3) CLS_CASE(4) CLS_CASE(5) CLS_CASE(6) CLS_CASE(7) CLS_CASE(8)
        default: set_error("pooled_ln: width %d too large", W); return WISE_E_UNSUPPORTED;
    }
#undef CLS_CASE
    WISE_LAUNCH_CHECK("cls_ln_kernel");
    return WISE_OK;
}

int pooled_head(const float* x, const float* ln_w, const float* ln_b, const bf16_t* projT, int batch, int T, int W,
                int D, const int* pos, bf16_t* hb, float* e, float* out, hipStream_t st, float eps, const float* proj_bias) {
    const int Bp = (batch + 255) / 256 * 256;
    int rc;
    if ((rc = pooled_ln(x, ln_w, ln_b, batch, T, W, pos, hb, st, eps))) return rc;
    if ((rc = gemm_bf16(hb, projT, proj_bias, Bp, D, W, 4, e, st))) return rc;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((batch + 3) / 4), dim3(256), 0, st, e, batch, D, out);
    WISE_LAUNCH_CHECK("l2norm_rows_kernel");
    return WISE_OK;
}

struct VitDims {
    int S, P, W, L, H, F, D, g, T, K, Kp, arch, fold;
};
static int vit_dims(const wise_vit_config* c, VitDims* d) {
    WISE_CHECK_ARG(c, "vit: null config");
    d->S = c->image_size; d->P = c->patch; d->W = c->width; d->L = c->layers; d->H = c->heads;
    d->F = c->mlp; d->D = c->embed_dim;
    WISE_CHECK_ARG(d->P > 0 && d->S > 0 && d->S % d->P == 0, "vit: image_size %d not a multiple of patch %d", d->S, d->P);
    WISE_CHECK_ARG(d->W > 0 && d->W % 128 == 0 && (d->H * 64 == d->W || d->H * 80 == d->W),
                   "vit: width %d must be heads * 64 (or heads * 80: ViT-H/14) and a multiple of 128", d->W);
    WISE_CHECK_ARG(d->F > 0 && d->F % 128 == 0, "vit: mlp %d must be a multiple of 128", d->F);
    WISE_CHECK_ARG(d->D > 0 && d->L >= 0 && d->W <= 4096 && d->W % 8 == 0, "vit: bad dims");
    WISE_CHECK_ARG(c->act >= 0 && c->act <= 2, "vit: act must be 0 (quick_gelu), 1 (gelu) or 2 (gelu, tanh form)");
    WISE_CHECK_ARG(c->arch == 0 || c->arch == 1, "vit: arch must be 0 (CLIP) or 1 (timm SigLIP: no class token, attention-pool head)");
    d->arch = c->arch;
    WISE_CHECK_ARG(c->ln_fold == 0 || ((c->ln_fold == 1 || c->ln_fold == 2) && c->arch == 0 && c->layers >= 1 && d->W >= 256),
                   "vit: ln_fold is 0, 1 or 2, and not 0 only for arch 0 with at least one block and width >= 256");
    d->fold = c->ln_fold;
    WISE_CHECK_ARG(d->arch == 0 || (d->D == d->W && d->H * 64 == d->W), "vit: the attention-pool head has no projection (embed_dim == width) and head dim 64");
    d->g = d->S / d->P; d->T = d->g * d->g + (d->arch == 0 ? 1 : 0); d->K = 3 * d->P * d->P; d->Kp = (d->K + 63) / 64 * 64;
    WISE_CHECK_ARG(d->arch == 0 || d->T <= 1024, "vit: the attention-pool head serves up to 1024 tokens");
    WISE_CHECK_ARG(d->fold != 2 || attention_oproj_fold_ok(d->T, d->H, d->W / d->H),
                   "vit: ln_fold = 2 (attention and out-projection in one kernel) serves up to 64 tokens and 12 heads of 64 (T=%d, H=%d)", d->T, d->H);
    return WISE_OK;
}

struct VitOffsets {
    // bf16 blob (elements)
    size_t conv1, layer0_b, per_layer_b, in_proj, out_proj, c_fc, c_proj, projT, total_b;
    // fp32 blob (elements)
    size_t cls, pos, ln_pre_w, ln_pre_b, layer0_f, per_layer_f, ln1_w, ln1_b, in_b, out_b, ln2_w, ln2_b, fc_b, proj_b,
        ln_post_w, ln_post_b, total_f;
    // arch 1 (attention-pool head): bf16 W_kv [2W,W], W_proj [W,W], W_fc1 [F,W], W_fc2 [W,F] from projT on; fp32 after the
    // final norm (ln_post_w/b): q(latent) [W], b_kv [2W], b_proj [W], head norm w,b, b_fc1 [F], b_fc2 [W]
    size_t h_kv, h_proj, h_fc1, h_fc2, hq, h_kvb, h_projb, hn_w, hn_b, h_fc1b, h_fc2b;
};
static VitOffsets vit_offsets(const VitDims& d) {
    VitOffsets o;
    const size_t W = d.W, F = d.F;
    o.conv1 = 0;
    o.layer0_b = W * d.Kp;
    o.in_proj = 0; o.out_proj = 3 * W * W; o.c_fc = o.out_proj + W * W; o.c_proj = o.c_fc + F * W;
    o.per_layer_b = o.c_proj + W * F;
    o.projT = o.layer0_b + o.per_layer_b * d.L;
    o.total_b = o.projT + (size_t)d.D * W;
    o.cls = 0; o.pos = W; o.ln_pre_w = o.pos + (size_t)d.T * W; o.ln_pre_b = o.ln_pre_w + W;
    o.layer0_f = o.ln_pre_b + W;
    if (d.arch == 1) {   // no class token, no ln_pre
        o.pos = 0; o.layer0_f = (size_t)d.T * W;
        o.h_kv = o.projT; o.h_proj = o.h_kv + 2 * W * W; o.h_fc1 = o.h_proj + W * W; o.h_fc2 = o.h_fc1 + F * W;
        o.total_b = o.h_fc2 + W * F;
    }
    o.ln1_w = 0; o.ln1_b = W; o.in_b = 2 * W; o.out_b = 5 * W; o.ln2_w = 6 * W; o.ln2_b = 7 * W; o.fc_b = 8 * W;
    o.proj_b = o.fc_b + F;
    o.per_layer_f = o.proj_b + W;
    o.ln_post_w = o.layer0_f + o.per_layer_f * d.L; o.ln_post_b = o.ln_post_w + W;
    o.total_f = o.ln_post_b + W;
    if (d.arch == 1) {
        o.hq = o.ln_post_b + W; o.h_kvb = o.hq + W; o.h_projb = o.h_kvb + 2 * W; o.hn_w = o.h_projb + W; o.hn_b = o.hn_w + W;
        o.h_fc1b = o.hn_b + W; o.h_fc2b = o.h_fc1b + F;
        o.total_f = o.h_fc2b + W;
    }
    return o;
}

struct VitWs {
    size_t x, h, qkv, a, rstd, total;
    int M, Mp, Mpatch, Mpp;
};
static VitWs vit_ws(const VitDims& d, int B) {
    VitWs w;
    // rows padded to 256 so every GEMM tiling (128- and 256-row tiles) sees whole tiles
    w.M = B * d.T; w.Mp = (w.M + 255) / 256 * 256;
    w.Mpatch = B * d.g * d.g; w.Mpp = (w.Mpatch + 255) / 256 * 256;
    size_t off = 0;
    w.x = off; off += align_up((size_t)w.Mp * d.W * 4, 256);
    w.h = off; off += align_up((size_t)w.Mp * d.W * 2, 256);
    size_t qkv_b = (size_t)w.Mp * 3 * d.W * 2, po_b = (size_t)w.Mpp * d.W * 4;
    w.qkv = off; off += align_up(qkv_b > po_b ? qkv_b : po_b, 256);
    size_t a_b = (size_t)w.Mp * d.F * 2, pa_b = (size_t)w.Mpp * d.Kp * 2;
    const size_t head_b = (size_t)((B + 255) / 256 * 256) * (d.W + d.F) * 2;      // arch 1: the head's bf16 rows live here too
    if (pa_b > a_b) a_b = pa_b;
    if (d.arch == 1 && head_b > a_b) a_b = head_b;
    w.a = off; off += align_up(a_b, 256);
    // fold mode (gemm_w4.h FoldArgs): row scales, partial row sums per 64 columns, arrival counters per row stripe
    w.rstd = off; off += align_up(gemm_fold_stats_bytes(w.Mp, d.W), 256);
    w.total = off;
    return w;
}

template <int NV>
static void launch_embed(const float* po, const float* cls, const float* pos, const float* w, const float* b, int rows,
                         int T, int W, float* x, hipStream_t st, bf16_t* hcopy, float* rstd_out, long long lo_off) {
    hipLaunchKernelGGL(embed_lnpre_kernel<NV>, dim3((rows + 3) / 4), dim3(256), 0, st, po, cls, pos, w, b, rows, T, W,
                       1e-5f, x, hcopy, rstd_out, lo_off);
}

// rows r * stride of a hi + lo residual stream -> fp32 [n, W] (the class rows in front of ln_post; the parity tap)
__global__ __launch_bounds__(256) void hilo_rows_kernel(const bf16_t* __restrict__ hi, long long lo_off, int n, int stride, int W,
                                                        float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    const size_t src = (size_t)r * stride * W + c;
    out[i] = bf16_to_f32(hi[src]) + bf16_to_f32(hi[src + lo_off]);
}

// one contiguous part of the batch on one stream; `wsb` is that part's own workspace region
static int vit_forward_part(const wise_vit_config* cfg, const VitDims& d, const VitOffsets& o, const bf16_t* wb,
                            const float* pf, const void* images, int in_kind, int batch, float* out,
                            unsigned char* wsb, hipStream_t st) {
    const VitWs ws = vit_ws(d, batch);
    float* x = reinterpret_cast<float*>(wsb + ws.x);
    bf16_t* h = reinterpret_cast<bf16_t*>(wsb + ws.h);
    bf16_t* qkv = reinterpret_cast<bf16_t*>(wsb + ws.qkv);
    bf16_t* a = reinterpret_cast<bf16_t*>(wsb + ws.a);
    float* patch_out = reinterpret_cast<float*>(wsb + ws.qkv);
    bf16_t* patches = a;
    const int W = d.W;
    int rc;

    // 1. patch gather + conv1-as-GEMM (no bias)
    const NormConst nc = norm_const(d.arch);
    const bool fast_gather = d.P % 8 == 0 && d.S % 8 == 0 && d.Kp == 3 * d.P * d.P && ((uintptr_t)images & 15) == 0;
    if (fast_gather) {
        const long long total8 = (long long)ws.Mpatch * (d.Kp / 8);
        const unsigned blocks = (unsigned)((total8 + 255) / 256);
        if (in_kind == WISE_VIT_IN_U8)
            hipLaunchKernelGGL(patchify8_kernel<unsigned char>, dim3(blocks), dim3(256), 0, st,
                               reinterpret_cast<const unsigned char*>(images), total8, d.S, d.P, d.g, patches, nc);
        else
            hipLaunchKernelGGL(patchify8_kernel<float>, dim3(blocks), dim3(256), 0, st,
                               reinterpret_cast<const float*>(images), total8, d.S, d.P, d.g, patches, nc);
    } else if (in_kind == WISE_VIT_IN_U8)
        hipLaunchKernelGGL(patchify_kernel<unsigned char>, dim3(ws.Mpatch), dim3(256), 0, st,
                           reinterpret_cast<const unsigned char*>(images), batch, d.S, d.P, d.g, d.Kp, patches, nc);
    else
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(ws.Mpatch), dim3(256), 0, st,
                           reinterpret_cast<const float*>(images), batch, d.S, d.P, d.g, d.Kp, patches, nc);
    WISE_LAUNCH_CHECK("patchify_kernel");
    rc = gemm_bf16(patches, wb + o.conv1, nullptr, ws.Mpp, W, d.Kp, 4, patch_out, st);
    if (rc) return rc;
    // 2. cls + pos + ln_pre -> x   (arch 1: patch rows + pos, nothing else)
    if (d.arch == 1) {
        const long long total4 = (long long)ws.M * (W >> 2);
        hipLaunchKernelGGL(embed_pos_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, patch_out, pf + o.pos,
                           total4, d.T, W, x);
        WISE_LAUNCH_CHECK("embed_pos_kernel");
    } else {

        const int nv = (W / 4 + 63) / 64;
        // fold mode: the fp32 rows' region holds the residual stream as hi [Mp, W] bf16, then lo [Mp, W] bf16
        bf16_t* fh = d.fold ? reinterpret_cast<bf16_t*>(x) : nullptr;
        float* frs = d.fold ? reinterpret_cast<float*>(wsb + ws.rstd) : nullptr;
        const long long flo = (long long)ws.Mp * W;
        const float* cls = pf + o.cls; const float* pos = pf + o.pos;
        const float* lw = pf + o.ln_pre_w; const float* lb = pf + o.ln_pre_b;
        switch (nv) {
            case 1: launch_embed<1>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 2: launch_embed<2>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 3: launch_embed<3>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 4: launch_embed<4>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 5: launch_embed<5>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 6: launch_embed<6>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 7: launch_embed<7>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            case 8: launch_embed<8>(patch_out, cls, pos, lw, lb, ws.M, d.T, W, x, st, fh, frs, flo); break;
            default: set_error("vit_forward: width %d too large", W); return WISE_E_UNSUPPORTED;
        }
        WISE_LAUNCH_CHECK("embed_lnpre_kernel");
    }
    // 3. transformer blocks
    const BlockWeights bw = {wb + o.layer0_b, o.per_layer_b, o.in_proj, o.out_proj, o.c_fc, o.c_proj,
                             pf + o.layer0_f, o.per_layer_f, o.ln1_w, o.ln1_b, o.in_b, o.out_b, o.ln2_w, o.ln2_b,
                             o.fc_b, o.proj_b};
    if (d.fold) {
        if ((rc = transformer_blocks_fold(bw, d.L, W, d.H, d.F, cfg->act, batch, d.T, x, h, qkv, a,
                                          reinterpret_cast<float*>(wsb + ws.rstd), st, 1e-5f, d.fold == 2)))
            return rc;
    } else if ((rc = transformer_blocks(bw, d.L, W, d.H, d.F, cfg->act, batch, d.T, false, x, h, qkv, a, st,
                                        d.arch == 1 ? 1e-6f : 1e-5f)))
        return rc;
    if (d.arch == 1) {
        // 4'. timm 'map' head: final norm over every token -> keys | values GEMM -> one latent query per head attends the
        // image's tokens -> projection -> y + mlp(norm(y)) -> L2 normalise (the tower has no further projection)
        const int Bp = (batch + 255) / 256 * 256;
        if ((rc = layernorm_f32_bf16(x, pf + o.ln_post_w, pf + o.ln_post_b, ws.M, W, 1e-6f, h, st))) return rc;
        if ((rc = gemm_bf16(h, wb + o.h_kv, pf + o.h_kvb, ws.Mp, 2 * W, W, 0, qkv, st))) return rc;
        bf16_t* pooled = h;                                    // [Bp, W] bf16 (h is free once the kv GEMM has read it)
        hipLaunchKernelGGL(map_pool_kernel, dim3(batch, d.H), dim3(64), 0, st, qkv, pf + o.hq, d.T, W, pooled);
        WISE_LAUNCH_CHECK("map_pool_kernel");
        float* y = reinterpret_cast<float*>(qkv);              // [Bp, W] fp32 (keys | values are consumed; x stays for the parity tap)
        bf16_t* yn = a;                                        // [Bp, W] bf16, then the hidden rows [Bp, F] behind it
        bf16_t* hid = a + (size_t)Bp * W;
        if ((rc = gemm_bf16(pooled, wb + o.h_proj, pf + o.h_projb, Bp, W, W, 4, y, st))) return rc;
        if ((rc = layernorm_f32_bf16(y, pf + o.hn_w, pf + o.hn_b, batch, W, 1e-6f, yn, st))) return rc;
        if ((rc = gemm_bf16(yn, wb + o.h_fc1, pf + o.h_fc1b, Bp, d.F, W, 2, hid, st))) return rc;     // timm Mlp: erf GELU
        if ((rc = gemm_bf16(hid, wb + o.h_fc2, pf + o.h_fc2b, Bp, W, d.F, 3, y, st))) return rc;
        return l2norm_rows(y, batch, W, out, st);
    }
    // 4. ln_post(cls) -> bf16 [Bp,W] (aliases h) ; @ proj -> fp32 [Bp,D] (aliases qkv) ; L2 normalise rows
    if (d.fold) {   // the class rows of the hi + lo stream as fp32 [batch, W] (in `a`, free now), then the same head with T = 1
        float* xc = reinterpret_cast<float*>(a);
        const long long n = (long long)batch * W;
        hipLaunchKernelGGL(hilo_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(x),
                           (long long)ws.Mp * W, batch, d.T, W, xc);
        WISE_LAUNCH_CHECK("hilo_rows_kernel");
        return pooled_head(xc, pf + o.ln_post_w, pf + o.ln_post_b, wb + o.projT, batch, 1, W, d.D, nullptr, h,
                           reinterpret_cast<float*>(qkv), out, st);
    }
    if ((rc = pooled_head(x, pf + o.ln_post_w, pf + o.ln_post_b, wb + o.projT, batch, d.T, W, d.D, nullptr, h,
                          reinterpret_cast<float*>(qkv), out, st)))
        return rc;
    return WISE_OK;
}

// Parts of a batch are independent; running them on separate internal streams lets one part's GEMM
// tails and barrier bubbles be filled by the other parts' kernels (and their LayerNorm / attention
// traffic overlap the GEMM main loops).  Fork/join with events, so the call stays stream-ordered for
// the caller and graph-capturable once the internal streams exist (first call creates them).
constexpr int MAX_PARTS = 4;
static hipStream_t g_side[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
static hipEvent_t g_ev_fork = nullptr, g_ev_join[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
static int g_vit_streams = 2;

// Part 0 runs on the caller's own stream; parts 1.. each get a side stream, created on first need.  (HIP maps
// streams onto 4 hardware queues: with four side streams created up front, the two half-batch streams could land
// on one queue behind the caller's other streams and stop overlapping — measured 3.4 -> 4.45 ms per step.)
static int ensure_side_streams(int nside) {
    if (!g_ev_fork) {
        hipError_t e = hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming);
        if (e != hipSuccess) { set_error("vit_forward: event: %s", hipGetErrorString(e)); return (int)e; }
    }
    for (int i = 0; i < nside && i < MAX_PARTS; ++i) {
        if (g_side[i]) continue;
        hipError_t e = hipStreamCreateWithFlags(&g_side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g_ev_join[i], hipEventDisableTiming);
        if (e != hipSuccess) { set_error("vit_forward: side stream: %s", hipGetErrorString(e)); return (int)e; }
    }
    return WISE_OK;
}

// number of parts for a batch under `streams`, and part i's [begin, end) images
static int parts_for(int batch, int streams) {
    int p = streams < 1 ? 1 : (streams > MAX_PARTS ? MAX_PARTS : streams);
    while (p > 1 && batch / p < 32) --p;  // keep every part at least 32 images
    return p;
}
// (fold mode: ONE part — a half batch's row counts are no multiple of the 160-row tiles the folded GEMMs are fastest on, and the
//  whole batch on one stream measured faster than two halves on two: 2.97 against 3.06 ms at bs = 256, tools/vit_fold_ab.py)
static int vit_parts(int batch, int fold = 0) { return fold ? 1 : parts_for(batch, g_vit_streams); }
static void part_range(int batch, int parts, int i, int* lo, int* hi) {
    *lo = (int)((long long)batch * i / parts);
    *hi = (int)((long long)batch * (i + 1) / parts);
}
static size_t total_ws_for(const VitDims& d, int batch, int parts) {
    size_t t = 0;
    for (int i = 0; i < parts; ++i) {
        int lo, hi;
        part_range(batch, parts, i, &lo, &hi);
        t += vit_ws(d, hi - lo).total;
    }
    return t;
}

}  // namespace wise

using namespace wise;

#ifdef WISE_DEBUG_KNOBS
extern "C" int wise_debug_set_vit_streams(int n) {
    g_vit_streams = n & 0xFF;
    if (n >> 8) g_attn_qt = (n >> 8) == 255 ? 0 : (n >> 8);  // bits 8..: attention variant (2 / 4: 32 / 64 queries per wave; 3 / 5: 48 / 32 with prefetch; 255: heuristic)
    return 0;
}
#endif

extern "C" int wise_vit_layout(const wise_vit_config* cfg, int64_t* wb_elems, int64_t* pf_elems) {
    VitDims d;
    int rc = vit_dims(cfg, &d);
    if (rc) return rc;
    VitOffsets o = vit_offsets(d);
    if (wb_elems) *wb_elems = (int64_t)o.total_b;
    if (pf_elems) *pf_elems = (int64_t)o.total_f;
    return WISE_OK;
}

extern "C" size_t wise_vit_workspace_bytes(const wise_vit_config* cfg, int batch) {
    VitDims d;
    if (vit_dims(cfg, &d) || batch < 1) return 0;
    size_t best = 0;  // valid for every stream setting
    for (int s = 1; s <= MAX_PARTS; ++s) {
        const size_t t = total_ws_for(d, batch, parts_for(batch, s));
        if (t > best) best = t;
    }
    return best;
}

extern "C" int wise_vit_forward(const wise_vit_config* cfg, const uint16_t* wb, const float* pf, const void* images,
                                int in_kind, int batch, float* out, void* workspace, size_t workspace_bytes,
                                void* stream) {
    VitDims d;
    int rc = vit_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(wb && pf && images && out, "vit_forward: null pointer");
    WISE_CHECK_ARG(batch >= 1, "vit_forward: batch=%d", batch);
    WISE_CHECK_ARG(in_kind == WISE_VIT_IN_F32 || in_kind == WISE_VIT_IN_U8, "vit_forward: in_kind=%d", in_kind);
    const int parts = vit_parts(batch, d.fold);
    const size_t need = total_ws_for(d, batch, parts);
    if (!workspace || workspace_bytes < need) {
        set_error("vit_forward: workspace %zu < %zu bytes", workspace_bytes, need);
        return WISE_E_WORKSPACE;
    }
    WISE_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wb & 15) == 0 && ((uintptr_t)pf & 15) == 0,
                   "vit_forward: workspace must be 256-byte and weight blobs 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const VitOffsets o = vit_offsets(d);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    if (parts == 1) return vit_forward_part(cfg, d, o, wb, pf, images, in_kind, batch, out, wsb, st);

    if ((rc = ensure_side_streams(parts - 1))) return rc;
    const size_t img_elem = (in_kind == WISE_VIT_IN_U8) ? 1 : 4;
    const size_t img_stride = (size_t)3 * d.S * d.S * img_elem;
    const unsigned char* img = reinterpret_cast<const unsigned char*>(images);
    hipError_t e = hipEventRecord(g_ev_fork, st);
    for (int i = 1; i < parts && e == hipSuccess; ++i) e = hipStreamWaitEvent(g_side[i - 1], g_ev_fork, 0);
    if (e != hipSuccess) { set_error("vit_forward: fork: %s", hipGetErrorString(e)); return (int)e; }
    size_t base = 0;
    gemm_set_overlapped(true);
    // side parts first, the caller's own part last: the side streams start while this thread still enqueues
    size_t bases[MAX_PARTS];
    for (int i = 0; i < parts; ++i) {
        int lo, hi;
        part_range(batch, parts, i, &lo, &hi);
        bases[i] = base;
        base += vit_ws(d, hi - lo).total;
    }
    for (int k = 0; k < parts && !rc; ++k) {
        const int i = (k + 1) % parts;   // 1, 2, ..., 0
        int lo, hi;
        part_range(batch, parts, i, &lo, &hi);
        rc = vit_forward_part(cfg, d, o, wb, pf, img + (size_t)lo * img_stride, in_kind, hi - lo,
                              out + (size_t)lo * d.D, wsb + bases[i], i == 0 ? st : g_side[i - 1]);
    }
    gemm_set_overlapped(false);
    // always join, even after an error, so the caller's stream never runs ahead of the side streams
    for (int i = 1; i < parts; ++i) {
        hipError_t e2 = hipEventRecord(g_ev_join[i - 1], g_side[i - 1]);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(st, g_ev_join[i - 1], 0);
        if (e2 != hipSuccess && !rc) { set_error("vit_forward: join: %s", hipGetErrorString(e2)); rc = (int)e2; }
    }
    return rc;
}

extern "C" int wise_vit_forward_single(const wise_vit_config* cfg, const uint16_t* wb, const float* pf, const void* images,
                                       int in_kind, int batch, float* out, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    VitDims d;
    int rc = vit_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(wb && pf && images && out, "vit_forward: null pointer");
    WISE_CHECK_ARG(batch >= 1, "vit_forward: batch=%d", batch);
    WISE_CHECK_ARG(in_kind == WISE_VIT_IN_F32 || in_kind == WISE_VIT_IN_U8, "vit_forward: in_kind=%d", in_kind);
    const size_t need = vit_ws(d, batch).total;
    if (!workspace || workspace_bytes < need) {
        set_error("vit_forward: workspace %zu < %zu bytes", workspace_bytes, need);
        return WISE_E_WORKSPACE;
    }
    WISE_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wb & 15) == 0 && ((uintptr_t)pf & 15) == 0,
                   "vit_forward: workspace must be 256-byte and weight blobs 16-byte aligned");
    // (a caller that keeps two whole batches in flight on two streams brackets this call with wise_overlap_hint so that
    // the GEMMs tile for co-residency; a lone stream gets the lone-stream tiles)
    return vit_forward_part(cfg, d, vit_offsets(d), wb, pf, images, in_kind, batch, out,
                            reinterpret_cast<unsigned char*>(workspace), (hipStream_t)stream);
}

extern "C" int wise_vit_tap_residual(const wise_vit_config* cfg, int batch, const void* workspace, float* dst,
                                     void* stream) {
    VitDims d;
    int rc = vit_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(workspace && dst && batch >= 1, "vit_tap_residual: bad argument");
    const unsigned char* wsb = reinterpret_cast<const unsigned char*>(workspace);
    const int parts = vit_parts(batch, d.fold);
    size_t base = 0;
    float* dp = dst;
    for (int i = 0; i < parts; ++i) {
        int lo, hi;
        part_range(batch, parts, i, &lo, &hi);
        const VitWs ws = vit_ws(d, hi - lo);
        if (d.fold) {
            const long long n = (long long)ws.M * d.W;
            hipLaunchKernelGGL(hilo_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const bf16_t*>(wsb + base + ws.x), (long long)ws.Mp * d.W, ws.M, 1, d.W, dp);
            WISE_LAUNCH_CHECK("hilo_rows_kernel");
        } else {
            hipError_t e = hipMemcpyAsync(dp, wsb + base + ws.x, (size_t)ws.M * d.W * 4, hipMemcpyDeviceToDevice,
                                          (hipStream_t)stream);
            if (e != hipSuccess) { set_error("vit_tap_residual: %s", hipGetErrorString(e)); return (int)e; }
        }
        dp += (size_t)ws.M * d.W;
        base += ws.total;
    }
    return WISE_OK;
}

extern "C" int wise_layernorm_f32_bf16(const float* x, const float* w, const float* b, int rows, int W, float eps,
                                       uint16_t* y, void* stream) {
    return layernorm_f32_bf16(x, w, b, rows, W, eps, y, (hipStream_t)stream);
}

extern "C" int wise_attention_bf16(const uint16_t* qkv, int B, int T, int H, uint16_t* o, void* stream) {
    return attention_bf16(qkv, B, T, H, o, (hipStream_t)stream, false, 64);
}

#ifdef WISE_DEBUG_KNOBS
extern "C" int wise_debug_ao_set(int flags) { wise::g_ao_dbg = flags & 15; return 0; }
extern "C" int wise_debug_ao_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wise::g_ao_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int wise_attention_oproj_fold(const uint16_t* qkv, int B, int T, int H, const uint16_t* Wt, const float* bias,
                                         uint16_t* hi, int64_t lo_off, float* rstd, float eps, void* stream) {
    WISE_CHECK_ARG(lo_off >= (int64_t)B * T * H * 64, "attention_oproj_fold: lo_off %lld overlaps the hi rows", (long long)lo_off);
    return attention_oproj_fold(qkv, B, T, H, Wt, bias, hi, (long long)lo_off, rstd, eps, (hipStream_t)stream);
}

extern "C" int wise_attention_dh_bf16(const uint16_t* qkv, int B, int T, int H, int dh, uint16_t* o, void* stream) {
    return attention_bf16(qkv, B, T, H, o, (hipStream_t)stream, false, dh);
}
extern "C" int wise_attention_lens_bf16(const uint16_t* qkv, int B, int T, int H, const int32_t* lens, uint16_t* o, void* stream) {
    WISE_CHECK_ARG(lens, "attention_lens: null pointer");
    return attention_bf16(qkv, B, T, H, o, (hipStream_t)stream, false, 64, lens);
}
extern "C" int wise_attention_causal_bf16(const uint16_t* qkv, int B, int T, int H, uint16_t* o, void* stream) {
    return attention_bf16(qkv, B, T, H, o, (hipStream_t)stream, true, 64);
}
