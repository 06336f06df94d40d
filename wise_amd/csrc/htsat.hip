// HP-1 audio: MS-CLAP (2023) HTSAT audio encoder on gfx950.
//
// Replaces `self.model.clap.audio_encoder(x)[0]` + L2 normalise at
// src/feature/microsoft_clap.py:49-50 (msclap 1.3.3 HTSAT_Swin_Transformer + Projection; arithmetic:
// SURVEY.md App. A.2, restated in oracle/htsat_ref.py, whose Swin body is pinned to transformers'
// ClapAudioModel).
//
// Pipeline (all hand-written HIP):
//   frontend   one wave per STFT frame: reflect-padded, Hann-windowed 1024-point FFT in LDS, power,
//              sparse Slaney mel filterbank (lane = band), 10*log10, BatchNorm folded to scale/shift
//   embed      time-axis bicubic resample to 1024 frames + fold to the 256x256 image + 4x4/4 conv + LN,
//              one wave per token, straight from the log-mel (the image is never materialised)
//   Swin       LN -> qkv GEMM -> window attention (cyclic shift and window partition are pure
//              addressing; rel-pos bias + shift mask fused into the softmax) -> proj GEMM(+resid)
//              -> LN -> fc1 GEMM(+GELU) -> fc2 GEMM(+resid); PatchMerging = gather+LN kernel + GEMM
//   head       final LN, token mean, Projection (linear1, GELU, linear2, LN(e1+e2)), L2 normalise
#include "common.h"
#include "transformer.h"
#include <type_traits>

namespace wise {

int gemm_bf16(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, int mode, void* out,
              hipStream_t st);
int layernorm_f32_bf16(const float* x, const float* w, const float* b, int rows, int W, float eps, bf16_t* y,
                       hipStream_t st);
// LayerNorm fused into the GEMM's A tile (gemm_bf16.hip), for K = 96 / 192
bool gemm_ln_supported(int N, int K, int mode);
int gemm_ln_bf16(const float* x, const float* lnw, const float* lnb, const bf16_t* Wt, const float* bias, int M, int N,
                 int K, float eps, int mode, bf16_t* out, hipStream_t st);
// x[M,96] += fc2(gelu(fc1(LN(x)))) with the hidden layer kept on chip (gemm_bf16.hip)
int mlp96_fused(float* x, const float* lnw, const float* lnb, const bf16_t* W1, const float* b1, const bf16_t* W2,
                const float* b2, int M, float eps, hipStream_t st);

namespace htsat {
// htsat_frontend.hip: frontend() — declared in transformer.h
static int g_frontend_only = 0;  // (debug) stop after the front end: concurrency tests tap the log-mel
static int g_fuse_ln = 15;  // bit 0: LayerNorm-in-GEMM fusion, bit 1: fused MLP (stage 1), bit 2: fused attention half of a stage-1 block, bit 3: token-per-lane embedding; wise_debug_set_htsat flips them off
constexpr int N_FFT = 1024, HOP = 320, N_MELS = 64, MELW = FRONT_MELW, MAXF = 1024;
constexpr int EMBED = 96, LATENT = 768, OUT = 1024;
constexpr int DEPTHS[4] = {2, 2, 6, 2};
// 16 bytes per lane from global memory straight into LDS (base wave-uniform, lane i lands at base + 16 i)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
constexpr int HEADS[4] = {4, 8, 16, 32};

// ------------------------------------------------------------------------------------------------
// embed: wave per token (b, i, j) of the 64x64 grid
// image[R][c] with R = r*64 + f, c = t' holds mel[t = r*256 + t'][f] resampled to 1024 frames
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float v) { const float A = -0.75f; return ((A + 2.f) * v - (A + 3.f)) * v * v + 1.f; }
__device__ __forceinline__ float cubic2(float v) { const float A = -0.75f; return ((A * v - 5.f * A) * v + 8.f * A) * v - 4.f * A; }

__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ melbn, int B, int Fc,
                                                    const float* __restrict__ pw /*[96][16]*/,
                                                    const float* __restrict__ pb, const float* __restrict__ lnw,
                                                    const float* __restrict__ lnb, float* __restrict__ x) {
    // block = one row i of the 64x64 token grid of one clip: image rows 4i..4i+3 = mel bins f0..f0+3 of
    // time block r, all 256 image columns.  Thread t' resamples its 4 pixels (one float4 of 4 adjacent mel
    // bins per tap), then each wave embeds 16 tokens with the conv weights held in registers.
    __shared__ float pix[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x >> 6, i = blockIdx.x & 63;
    {
        const int r = i >> 4, f0 = (4 * i) & 63;
        const int t = r * 256 + threadIdx.x;
        const float* m = melbn + (size_t)b * Fc * 64 + f0;
        float4 v;
        if (Fc == MAXF) {
            v = *reinterpret_cast<const float4*>(m + (size_t)t * 64);
        } else {
            const float scale = (float)(Fc - 1) / (float)(MAXF - 1);
            const float src = (float)t * scale;
            const float fl = floorf(src);
            const int i0 = (int)fl;
            const float tt = src - fl;
            const float w0 = cubic2(tt + 1.f), w1 = cubic1(tt), w2 = cubic1(1.f - tt), w3 = cubic2(2.f - tt);
            const int a0 = min(max(i0 - 1, 0), Fc - 1), a1 = min(max(i0, 0), Fc - 1), a2 = min(max(i0 + 1, 0), Fc - 1),
                      a3 = min(max(i0 + 2, 0), Fc - 1);
            const float4 q0 = *reinterpret_cast<const float4*>(m + (size_t)a0 * 64);
            const float4 q1 = *reinterpret_cast<const float4*>(m + (size_t)a1 * 64);
            const float4 q2 = *reinterpret_cast<const float4*>(m + (size_t)a2 * 64);
            const float4 q3 = *reinterpret_cast<const float4*>(m + (size_t)a3 * 64);
            // same accumulation order as the oracle: taps 0..3 added in turn
            v.x = q0.x * w0; v.x += q1.x * w1; v.x += q2.x * w2; v.x += q3.x * w3;
            v.y = q0.y * w0; v.y += q1.y * w1; v.y += q2.y * w2; v.y += q3.y * w3;
            v.z = q0.z * w0; v.z += q1.z * w1; v.z += q2.z * w2; v.z += q3.z * w3;
            v.w = q0.w * w0; v.w += q1.w * w1; v.w += q2.w * w2; v.w += q3.w * w3;
        }
        pix[0][threadIdx.x] = v.x; pix[1][threadIdx.x] = v.y; pix[2][threadIdx.x] = v.z; pix[3][threadIdx.x] = v.w;
    }
    const int c0 = lane, c1 = lane + 64;
    const bool has1 = c1 < EMBED;
    float w0r[16], w1r[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        w0r[p] = pw[c0 * 16 + p];
        w1r[p] = has1 ? pw[c1 * 16 + p] : 0.f;
    }
    const float b0 = pb[c0], b1 = has1 ? pb[c1] : 0.f;
    const float g0 = lnw[c0], g1 = has1 ? lnw[c1] : 0.f, h0 = lnb[c0], h1 = has1 ? lnb[c1] : 0.f;
    __syncthreads();
    for (int jj = 0; jj < 16; ++jj) {
        const int j = wv * 16 + jj;
        float y0 = b0, y1 = b1;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float pv = pix[p >> 2][4 * j + (p & 3)];
            y0 = fmaf(w0r[p], pv, y0);
            y1 = fmaf(w1r[p], pv, y1);
        }
        const float mean = wave_sum(y0 + (has1 ? y1 : 0.f)) / (float)EMBED;
        const float d0 = y0 - mean, d1 = has1 ? y1 - mean : 0.f;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) / (float)EMBED + 1e-5f);
        float* xr = x + ((size_t)blockIdx.x * 64 + j) * EMBED;
        xr[c0] = d0 * rstd * g0 + h0;
        if (has1) xr[c1] = d1 * rstd * g1 + h1;
    }
}

// The same embedding with lane = token (the form that runs): a block = four rows i of one clip's 64 x 64 token grid, a wave
// = the 64 tokens of one row.  embed_kernel above (lane = channel) spends ~90 instructions per token — two six-step
// cross-lane sums for the LayerNorm of every token and 32 of 64 lanes idle on the second channel — and ran at 0.23 of
// the HBM rate of its 235 MB; here a lane keeps its token's 16 pixels and all 96 channels in registers, the conv weights
// are wave-uniform (scalar loads), the LayerNorm sums are per-lane and serial, and the [64 tokens x 96] tile leaves through
// a wave-private LDS image in three 32-channel rounds so that every store instruction writes whole 128-byte lines.
// The convolution's fmaf chain is the one of embed_kernel (bias, then taps 0..15); the LayerNorm sums run in a different
// order (serial per lane instead of a butterfly), i.e. the result differs in the last bits.
__global__ __launch_bounds__(256) void embed_tok_kernel(const float* __restrict__ melbn, int B, int Fc,
                                                        const float* __restrict__ pw /*[96][16]*/,
                                                        const float* __restrict__ pb, const float* __restrict__ lnw,
                                                        const float* __restrict__ lnb, float* __restrict__ x) {
    __shared__ __attribute__((aligned(16))) float pix[4][4][256];          // [row of the block][mel bin][time column]
    __shared__ __attribute__((aligned(16))) float tile[4][64 * 36];        // per wave: 64 tokens x 32 channels, 144-byte rows
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x >> 4, i0 = (blockIdx.x & 15) * 4;             // rows i0 .. i0+3: one time block r, 16 mel bins
    {
        const int r = i0 >> 4, f0 = (4 * i0) & 63;
        const int t = r * 256 + threadIdx.x;
        const float* m = melbn + (size_t)b * Fc * 64 + f0;
        float4 v[4];
        if (Fc == MAXF) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(m + (size_t)t * 64 + 4 * q);
        } else {
            const float scale = (float)(Fc - 1) / (float)(MAXF - 1);
            const float src = (float)t * scale;
            const float fl = floorf(src);
            const int j0 = (int)fl;
            const float tt = src - fl;
            const float w0 = cubic2(tt + 1.f), w1 = cubic1(tt), w2 = cubic1(1.f - tt), w3 = cubic2(2.f - tt);
            const int a0 = min(max(j0 - 1, 0), Fc - 1), a1 = min(max(j0, 0), Fc - 1), a2 = min(max(j0 + 1, 0), Fc - 1),
                      a3 = min(max(j0 + 2, 0), Fc - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 q0 = *reinterpret_cast<const float4*>(m + (size_t)a0 * 64 + 4 * q);
                const float4 q1 = *reinterpret_cast<const float4*>(m + (size_t)a1 * 64 + 4 * q);
                const float4 q2 = *reinterpret_cast<const float4*>(m + (size_t)a2 * 64 + 4 * q);
                const float4 q3 = *reinterpret_cast<const float4*>(m + (size_t)a3 * 64 + 4 * q);
                // same accumulation order as the oracle: taps 0..3 added in turn
                v[q].x = q0.x * w0; v[q].x += q1.x * w1; v[q].x += q2.x * w2; v[q].x += q3.x * w3;
                v[q].y = q0.y * w0; v[q].y += q1.y * w1; v[q].y += q2.y * w2; v[q].y += q3.y * w3;
                v[q].z = q0.z * w0; v[q].z += q1.z * w1; v[q].z += q2.z * w2; v[q].z += q3.z * w3;
                v[q].w = q0.w * w0; v[q].w += q1.w * w1; v[q].w += q2.w * w2; v[q].w += q3.w * w3;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pix[q][0][threadIdx.x] = v[q].x; pix[q][1][threadIdx.x] = v[q].y;
            pix[q][2][threadIdx.x] = v[q].z; pix[q][3][threadIdx.x] = v[q].w;
        }
    }
    __syncthreads();
    // ---- this wave's row, lane = token j: pixel p = (mel bin p >> 2, time column 4 j + (p & 3))
    float pv[16];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const float4 q = *reinterpret_cast<const float4*>(&pix[wv][rr][4 * lane]);
        pv[rr * 4 + 0] = q.x; pv[rr * 4 + 1] = q.y; pv[rr * 4 + 2] = q.z; pv[rr * 4 + 3] = q.w;
    }
    float y[EMBED];
#pragma unroll
    for (int c = 0; c < EMBED; ++c) {
        float a = pb[c];
#pragma unroll
        for (int p = 0; p < 16; ++p) a = fmaf(pw[c * 16 + p], pv[p], a);
        y[c] = a;
    }
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < EMBED; ++c) s4[c & 3] += y[c];
    const float mean = ((s4[0] + s4[1]) + (s4[2] + s4[3])) / (float)EMBED;
    float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < EMBED; ++c) { y[c] -= mean; q4[c & 3] = fmaf(y[c], y[c], q4[c & 3]); }
    const float rstd = rsqrtf(((q4[0] + q4[1]) + (q4[2] + q4[3])) / (float)EMBED + 1e-5f);
    // ---- out through the wave's image, 32 channels at a time
    float* tl = tile[wv];
    float* xrow = x + ((size_t)(b * 64 + i0 + wv) * 64) * EMBED;
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const int c = rd * 32 + k4 * 4;
            float4 o;
            o.x = y[c + 0] * rstd * lnw[c + 0] + lnb[c + 0];
            o.y = y[c + 1] * rstd * lnw[c + 1] + lnb[c + 1];
            o.z = y[c + 2] * rstd * lnw[c + 2] + lnb[c + 2];
            o.w = y[c + 3] * rstd * lnw[c + 3] + lnb[c + 3];
            *reinterpret_cast<float4*>(tl + lane * 36 + k4 * 4) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tok = q * 8 + (lane >> 3), seg = lane & 7;
            const float4 o = *reinterpret_cast<const float4*>(tl + tok * 36 + seg * 4);
            *reinterpret_cast<float4*>(xrow + (size_t)tok * EMBED + rd * 32 + seg * 4) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------
// window attention: wave per (image, window, head); 64 tokens, head dim 24 (padded to 32 for MFMA)
// qkv [B*H*W, 3C] bf16 in ORIGINAL token order; the cyclic shift is done by addressing
// ------------------------------------------------------------------------------------------------
constexpr int VT_LD = 68;

__global__ __launch_bounds__(256, 4) void swin_attention_kernel(const bf16_t* __restrict__ qkv, int B, int H, int W, int C,
                                                             int heads, int shift,
                                                             const float* __restrict__ bias /*[heads][64][64]*/,
                                                             bf16_t* __restrict__ o) {
    __shared__ __attribute__((aligned(16))) bf16_t vt_all[4][32 * VT_LD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nwx = W >> 3, nwin = (H >> 3) * nwx;
    const long long item = (long long)blockIdx.x * 4 + wv;
    if (item >= (long long)B * nwin * heads) return;
    const int h = (int)(item % heads);
    const int win = (int)((item / heads) % nwin);
    const int b = (int)(item / ((long long)heads * nwin));
    const int wy = win / nwx, wx = win % nwx;
    const int C3 = 3 * C;
    const bf16_t* base = qkv + (size_t)b * H * W * C3;
    bf16_t* vt = vt_all[wv];
    const int l15 = lane & 15, g = lane >> 4;

    // token p (0..63) of this window -> row in the original layout, and its shift-region id
    auto row_of = [&](int p) {
        const int ys = wy * 8 + (p >> 3), xs = wx * 8 + (p & 7);
        int y = ys + shift, x = xs + shift;
        if (y >= H) y -= H;
        if (x >= W) x -= W;
        return y * W + x;
    };
    auto region_of = [&](int p) {
        const int ys = wy * 8 + (p >> 3), xs = wx * 8 + (p & 7);
        const int ry = (ys >= H - 8) + (ys >= H - 4), rx = (xs >= W - 8) + (xs >= W - 4);
        return ry * 3 + rx;
    };

    // Q (B operand) and K (A operand) fragments: 8 bf16 of head dim 8g..8g+7, zero for g == 3
    bf16x8 qf[4], kf[4];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = row_of(t * 16 + l15);
        if (g < 3) {
            qf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)r * C3 + h * 24 + g * 8);
            kf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)r * C3 + C + h * 24 + g * 8);
        } else {
            qf[t] = zero8;
            kf[t] = zero8;
        }
    }
    // V^T image [32 dh][64 keys]: lane = key, 3 chunks of 8 dh; rows 24..31 zero
    {
        const int r = row_of(lane);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const short8 v = *reinterpret_cast<const short8*>(base + (size_t)r * C3 + 2 * C + h * 24 + c * 8);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) vt[(c * 8 + jj) * VT_LD + lane] = (bf16_t)v[jj];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) vt[(24 + jj) * VT_LD + lane] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // S^T[kt][qt] = K Q^T (one k-step of 32 >= 24)
    f32x4 sacc[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
            sacc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], c, 0, 0, 0);
        }
    const float scale = 0.20412414523193154f;  // 24^-0.5
    const float LOG2E = 1.4426950408889634f;
    const float* bh = bias + (size_t)h * 4096;
    int regk[4][4];
    if (shift > 0) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) regk[kt][r] = region_of(kt * 16 + g * 4 + r);
    }
    f32x4 oacc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float linv[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = qt * 16 + l15;
        const int regq = (shift > 0) ? region_of(q) : 0;
        float mx = -INFINITY;
        float s2[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float4 bb = *reinterpret_cast<const float4*>(bh + q * 64 + kt * 16 + g * 4);
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = sacc[kt][qt][r] * scale + bv[r];
                if (shift > 0 && regk[kt][r] != regq) s += -100.f;
                s *= LOG2E;
                s2[kt][r] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float ps = 0.f;
        float p[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[kt][r] = __builtin_amdgcn_exp2f(s2[kt][r] - mx);  // arguments <= 0: the bare v_exp_f32
                ps += p[kt][r];
            }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        linv[qt] = 1.f / ps;
        // O^T += V^T P^T, two k-steps of 32 key slots (slot order permuted consistently on both sides)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (__bf16)p[2 * ks][r];
                pf[4 + r] = (__bf16)p[2 * ks + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16_t* rowp = vt + (dt * 16 + l15) * VT_LD + ks * 32 + g * 4;
                const uint2 lo = *reinterpret_cast<const uint2*>(rowp);
                const uint2 hi = *reinterpret_cast<const uint2*>(rowp + 16);
                union { uint4 u; bf16x8 f; } cv;
                cv.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cv.f, pf, oacc[dt][qt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int r = row_of(qt * 16 + l15);
        bf16_t* orow = o + ((size_t)b * H * W + r) * C + h * 24;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int dh = dt * 16 + g * 4;
            if (dh < 24) {
                uint2 pk;
                pk.x = pack_bf16x2(oacc[dt][qt][0] * linv[qt], oacc[dt][qt][1] * linv[qt]);
                pk.y = pack_bf16x2(oacc[dt][qt][2] * linv[qt], oacc[dt][qt][3] * linv[qt]);
                *reinterpret_cast<uint2*>(orow + dh) = pk;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 1 (C = 96, 4 heads of 24, 64 x 64 tokens per clip): the whole attention half of a Swin block in one kernel,
//     x += proj( window_attention( LN1(x) W_qkv^T + b_qkv ) ) + b_proj          (in place; a window owns its 64 tokens)
// so that per block the activations cross HBM once each way (24 KiB in, 24 KiB out per window) instead of four times
// (LN-GEMM out 36 KiB, attention in 36 + out 12, projection in 12 + 24 + out 24).
// A workgroup = 4 waves = the 4 heads of one window, persistent over windows (blockIdx, +gridDim, ...):
//   1. all 256 threads: LayerNorm of the window's 64 rows (4 threads per row, exact two-pass statistics) -> bf16 image
//      in LDS (the MFMA operand layout: row-major, 208-byte rows);
//   2. wave h: q_h, k_h as TRANSPOSED products W x^T (lane = token, registers = head-dim) and v_h as x W^T (lane = head
//      dim, registers = token): the accumulators, packed to bf16, ARE the operands of S^T = K Q^T and O^T = V^T P^T —
//      the head-dim (resp. key) index is permuted the same way on both sides of each contraction, so nothing is
//      exchanged.  W_{q,k,v} of the head stay in registers for the life of the workgroup (18 fragments);
//   3. softmax with relative-position bias and shift mask exactly as swin_attention_kernel; O_h -> bf16 image in LDS;
//   4. wave w: output projection of token tile w (W_proj image in LDS, loaded once per workgroup) + bias + residual,
//      lane = token with 4 consecutive channels -> 16-byte read-modify-write of x.
// ------------------------------------------------------------------------------------------------
constexpr int SB_LD = 104;    // bf16 per LDS row: 96 + 8 (208 B = 13 x 16 B: rows fall on distinct 16-byte slots mod 256 B)

template <int SHIFT /*0, or 4: the odd block of a stage*/>
__global__ __launch_bounds__(256, 2) void swin96_block_attn_kernel(float* __restrict__ x, int B,
                                                                   const float* __restrict__ n1w, const float* __restrict__ n1b,
                                                                   const bf16_t* __restrict__ wqkv /*[288][96]*/,
                                                                   const float* __restrict__ qkvb /*[288]*/,
                                                                   const float* __restrict__ bias /*[4][64][64]*/,
                                                                   const bf16_t* __restrict__ wproj /*[96][96]*/,
                                                                   const float* __restrict__ pb /*[96]*/) {
    constexpr int H = 64, W = 64, C = 96, shift = SHIFT;
    __shared__ __attribute__((aligned(16))) float x_img[64 * C];     // the window's rows as fetched (LDS-DMA, dense)
    __shared__ __attribute__((aligned(16))) bf16_t a_img[64 * SB_LD];
    __shared__ __attribute__((aligned(16))) bf16_t o_img[64 * SB_LD];
    __shared__ __attribute__((aligned(16))) bf16_t wp_img[96 * SB_LD];
    __shared__ __attribute__((aligned(16))) float ln_g[96], ln_b[96];
    // relative-position bias, compact: the [64][64] table of a head is Toeplitz in (dy, dx), 15 x 15 distinct values;
    // stored reversed and pre-multiplied by log2(e):  rpb[h][224 - ((dy+7)*15 + dx+7)] = bias_h(dy, dx) * log2(e)
    __shared__ float rpb[4][228];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = wv;                                  // this wave's head
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const float LOG2E = 1.4426950408889634f;
    const int nwin_total = B * 64;

    // rows of window `it` (token p of the window -> row of the clip's 64 x 64 grid, cyclic shift by addressing)
    auto row_of = [&](int it, int p) {
        const int win = it & 63;
        int y = (win >> 3) * 8 + (p >> 3) + shift, xx = (win & 7) * 8 + (p & 7) + shift;
        if (y >= H) y -= H;
        if (xx >= W) xx -= W;
        return (it >> 6) * (H * W) + y * W + xx;
    };
    // fetch window `it` into x_img: 24 pieces of 1 KiB, 6 per wave, each lane 16 bytes of one row
    auto fetch_window = [&](int it) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int c = wv * 6 + j;
            const int off = c * 1024 + lane * 16;      // byte offset in the dense [64][384 B] image
            const int row = off / 384, colb = off - row * 384;
            glds16(reinterpret_cast<const unsigned char*>(x + (size_t)row_of(it, row) * C) + colb,
                   reinterpret_cast<unsigned char*>(x_img) + c * 1024);
        }
    };
    if ((int)blockIdx.x < nwin_total) fetch_window(blockIdx.x);

    // ---- once per workgroup: W_proj image, LayerNorm affine terms, the compact bias tables; this head's weight
    // fragments and biases
    for (int c = tid; c < 96 * 12; c += 256) {
        const int r = c / 12, part = c - r * 12;
        *reinterpret_cast<uint4*>(wp_img + r * SB_LD + part * 8) = *reinterpret_cast<const uint4*>(wproj + r * 96 + part * 8);
    }
    if (tid < 96) { ln_g[tid] = n1w[tid]; ln_b[tid] = n1b[tid]; }
    for (int e = tid; e < 4 * 225; e += 256) {
        const int hh = e / 225, j = e - hh * 225;
        const int ti = 224 - j, dy = ti / 15 - 7, dx = ti % 15 - 7;
        const int qy = dy > 0 ? dy : 0, qx = dx > 0 ? dx : 0, ky = qy - dy, kx = qx - dx;
        rpb[hh][j] = bias[(size_t)hh * 4096 + (qy * 8 + qx) * 64 + ky * 8 + kx] * LOG2E;
    }
    bf16x8 wf[3][2][3];                                // [q,k,v][head-dim tile][k-step]: row = head dim dt*16+l15, 8 channels at ks*32+g*8
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int dh = dt * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                wf[m][dt][ks] = dh < 24 ? *reinterpret_cast<const bf16x8*>(wqkv + (size_t)(m * C + h * 24 + dh) * C + ks * 32 + g * 8)
                                        : zero8;
        }
    float bq[2][4], bk[2][4], bvv[2];                  // q/k: head dim dt*16+g*4+r (registers); v: head dim dt*16+l15 (lane)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dh = dt * 16 + g * 4 + r;
            bq[dt][r] = dh < 24 ? qkvb[h * 24 + dh] : 0.f;
            bk[dt][r] = dh < 24 ? qkvb[C + h * 24 + dh] : 0.f;
        }
        bvv[dt] = (dt * 16 + l15) < 24 ? qkvb[2 * C + h * 24 + dt * 16 + l15] : 0.f;
    }
    const int lrow = tid >> 2, lq = tid & 3;           // LayerNorm role: row tid>>2, channels (tid&3)*24 .. +23
    const float sc = 0.20412414523193154f * LOG2E;     // 24^-0.5, in the exp2 domain
    // bias address of this lane: query (qt*2 + (l15>>3), l15&7), key (kt*2 + (g>>1), (g&1)*4 + r)
    //   reversed index = 224 - ((qy-ky+7)*15 + qx-kx+7) = [22 - c_lane] + [90 - (qt-kt)*30 + r],  c_lane in [-19, 22]
    const float* rp_lane = &rpb[h][22 - (((l15 >> 3) - (g >> 1)) * 15 + (l15 & 7) - (g & 1) * 4)];

    for (int item = blockIdx.x; item < nwin_total; item += gridDim.x) {
        const int win = item & 63;
        const int wy = win >> 3, wx = win & 7;
        auto region_of = [&](int p) {
            const int ys = wy * 8 + (p >> 3), xs = wx * 8 + (p & 7);
            const int ry = (ys >= H - 8) + (ys >= H - 4), rx = (xs >= W - 8) + (xs >= W - 4);
            return ry * 3 + rx;
        };
        // b_proj does not change from window to window; reloading it (cache hits) is cheaper than the registers the
        // compiler would pin for it across the loop — so its address is made opaque per trip
        const float* pbw = pb;
        asm volatile("" : "+v"(pbw));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of x_img have landed
        __syncthreads();                                   // ... and everybody else's (first trip: the setup above too)
        // ---- 1. LayerNorm of the window's rows: x_img -> a_img
        {
            const float* xr = x_img + lrow * C + lq * 24;
            float v[24];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(xr + i * 4);
                v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
            }
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) sm += v[i];
            sm += __shfl_xor(sm, 1, 64);
            sm += __shfl_xor(sm, 2, 64);
            const float mean = sm * (1.f / 96.f);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i) { v[i] -= mean; sq = fmaf(v[i], v[i], sq); }
            sq += __shfl_xor(sq, 1, 64);
            sq += __shfl_xor(sq, 2, 64);
            const float rstd = rsqrtf(sq * (1.f / 96.f) + 1e-5f);
            bf16_t* dst = a_img + lrow * SB_LD + lq * 24;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 gg = *reinterpret_cast<const float4*>(ln_g + lq * 24 + 8 * i + 4 * j);
                    const float4 bb = *reinterpret_cast<const float4*>(ln_b + lq * 24 + 8 * i + 4 * j);
                    y[4 * j] = fmaf(v[8 * i + 4 * j] * rstd, gg.x, bb.x);
                    y[4 * j + 1] = fmaf(v[8 * i + 4 * j + 1] * rstd, gg.y, bb.y);
                    y[4 * j + 2] = fmaf(v[8 * i + 4 * j + 2] * rstd, gg.z, bb.z);
                    y[4 * j + 3] = fmaf(v[8 * i + 4 * j + 3] * rstd, gg.w, bb.w);
                }
                uint4 pk;
                pk.x = pack_bf16x2(y[0], y[1]); pk.y = pack_bf16x2(y[2], y[3]);
                pk.z = pack_bf16x2(y[4], y[5]); pk.w = pack_bf16x2(y[6], y[7]);
                *reinterpret_cast<uint4*>(dst + 8 * i) = pk;
            }
        }
        __syncthreads();
        // x_img is free: the next window's rows travel while this one is computed
        if (item + (int)gridDim.x < nwin_total) fetch_window(item + gridDim.x);
        // ---- 2. q, k (transposed products) and v of this head, straight into operand form
        bf16x8 qf[4], kf[4], vf[2][2];
        {
            f32x4 vacc[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 xf[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
                    xf[ks] = *reinterpret_cast<const bf16x8*>(a_img + (t * 16 + l15) * SB_LD + ks * 32 + g * 8);
                f32x4 qa[2], ka[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    qa[dt] = f32x4{bq[dt][0], bq[dt][1], bq[dt][2], bq[dt][3]};
                    ka[dt] = f32x4{bk[dt][0], bk[dt][1], bk[dt][2], bk[dt][3]};
                    vacc[t][dt] = f32x4{bvv[dt], bvv[dt], bvv[dt], bvv[dt]};
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        qa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][dt][ks], xf[ks], qa[dt], 0, 0, 0);
                        ka[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][dt][ks], xf[ks], ka[dt], 0, 0, 0);
                        vacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks], wf[2][dt][ks], vacc[t][dt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qf[t][r] = (__bf16)qa[0][r]; qf[t][4 + r] = (__bf16)qa[1][r];
                    kf[t][r] = (__bf16)ka[0][r]; kf[t][4 + r] = (__bf16)ka[1][r];
                }
            }
            // V^T operand of k-step ks (key slots: registers 0-3 = keys (2ks)*16+g*4+r, 4-7 = keys (2ks+1)*16+g*4+r — the
            // order the probabilities are packed in below)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vf[ks][dt][r] = (__bf16)vacc[2 * ks][dt][r];
                        vf[ks][dt][4 + r] = (__bf16)vacc[2 * ks + 1][dt][r];
                    }
        }
        // ---- 3. per query tile: S^T = K Q^T, softmax (+ relative-position bias, shift mask), O^T = V^T P^T
        // Only the windows of the last row / column of a shifted block hold tokens of different regions: the mask is
        // compiled into its own copy of the loop, chosen per window (uniform branch)
        auto attend = [&](auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        unsigned long long regk = 0ull;                // shift-region id of key kt*16+g*4+r in nibble kt*4+r
        if (MASK) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) regk |= (unsigned long long)region_of(kt * 16 + g * 4 + r) << (4 * (kt * 4 + r));
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const int q = qt * 16 + l15;
            const int regq = MASK ? region_of(q) : 0;
            float mx = -INFINITY;
            float s2[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv = fmaf(sa[r], sc, rp_lane[90 - (qt - kt) * 30 + r]);
                    if (MASK && (int)((regk >> (4 * (kt * 4 + r))) & 15ull) != regq) sv += -100.f * LOG2E;
                    s2[kt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float ps = 0.f;
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kt][r] = __builtin_amdgcn_exp2f(s2[kt][r] - mx);
                    ps += p[kt][r];
                }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            const float linv = 1.f / ps;
            f32x4 oa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (__bf16)p[2 * ks][r];
                    pf[4 + r] = (__bf16)p[2 * ks + 1][r];
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) oa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks][dt], pf, oa[dt], 0, 0, 0);
            }
            // lane (query l15, g): head dims dt*16 + g*4 + r
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int dh = dt * 16 + g * 4;
                if (dh < 24) {
                    uint2 pk;
                    pk.x = pack_bf16x2(oa[dt][0] * linv, oa[dt][1] * linv);
                    pk.y = pack_bf16x2(oa[dt][2] * linv, oa[dt][3] * linv);
                    *reinterpret_cast<uint2*>(o_img + q * SB_LD + h * 24 + dh) = pk;
                }
            }
        }
        };
        if (shift > 0 && (wy == 7 || wx == 7)) attend(std::true_type{});
        else attend(std::false_type{});
        __syncthreads();
        // ---- 4. projection of token tile wv, + bias + residual, in place (the rows come from cache: they were fetched for
        // x_img moments ago)
        {
            bf16x8 of[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                of[ks] = *reinterpret_cast<const bf16x8*>(o_img + (wv * 16 + l15) * SB_LD + ks * 32 + g * 8);
            float* xr = x + (size_t)row_of(item, wv * 16 + l15) * C + g * 4;
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) {
                const float4 xv = *reinterpret_cast<const float4*>(xr + nt * 16);
                const float4 pbv = *reinterpret_cast<const float4*>(pbw + nt * 16 + g * 4);
                f32x4 acc = f32x4{xv.x + pbv.x, xv.y + pbv.y, xv.z + pbv.z, xv.w + pbv.w};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const bf16x8 wpf = *reinterpret_cast<const bf16x8*>(wp_img + (nt * 16 + l15) * SB_LD + ks * 32 + g * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wpf, of[ks], acc, 0, 0, 0);
                }
                *reinterpret_cast<float4*>(xr + nt * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
        // LDS reuse: a_img is rewritten only after the next trip's first barrier (its reads ended before the barrier above);
        // o_img only after the next trip's second barrier (this trip's reads end before its first)
    }
}

// ------------------------------------------------------------------------------------------------
// PatchMerging gather + LN: x fp32 [B,H,W,C] -> y bf16 [B,(H/2)(W/2),4C], segments x0|x1|x2|x3 =
// (2i,2j) (2i+1,2j) (2i,2j+1) (2i+1,2j+1); wave per output token; NV = ceil(C/64) float4 per lane
// ------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void merge_ln_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                       const float* __restrict__ w, const float* __restrict__ bb,
                                                       bf16_t* __restrict__ y, long long lo_off /* != 0: x is a hi + lo bf16 stream */) {
    const int lane = threadIdx.x & 63;
    const int H2 = H >> 1, W2 = W >> 1;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (long long)B * H2 * W2) return;
    const int b = (int)(tok / (H2 * W2)), ij = (int)(tok % (H2 * W2)), i = ij / W2, j = ij % W2;
    const int c4 = C >> 2;  // float4 per segment; C float4 in total
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int idx = u * 64 + lane;
        if (idx < C) {
            const int seg = idx / c4, off = idx % c4;
            const int yy = 2 * i + (seg & 1), xx = 2 * j + (seg >> 1);
            const size_t e0 = (((size_t)b * H + yy) * W + xx) * C;
            if (lo_off) {
                const bf16_t* hp = reinterpret_cast<const bf16_t*>(x) + e0 + 4 * off;
                const uint2 hv = *reinterpret_cast<const uint2*>(hp), lv = *reinterpret_cast<const uint2*>(hp + lo_off);
                v[u] = make_float4(__uint_as_float(hv.x << 16) + __uint_as_float(lv.x << 16),
                                   __uint_as_float(hv.x & 0xffff0000u) + __uint_as_float(lv.x & 0xffff0000u),
                                   __uint_as_float(hv.y << 16) + __uint_as_float(lv.y << 16),
                                   __uint_as_float(hv.y & 0xffff0000u) + __uint_as_float(lv.y & 0xffff0000u));
            } else
                v[u] = reinterpret_cast<const float4*>(x + e0)[off];
        } else
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    const float n = (float)(4 * C);
    const float mean = wave_sum(s) / n;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (u * 64 + lane < C) {
            const float a0 = v[u].x - mean, a1 = v[u].y - mean, a2 = v[u].z - mean, a3 = v[u].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / n + 1e-5f);
    uint2* yr = reinterpret_cast<uint2*>(y + (size_t)tok * 4 * C);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int idx = u * 64 + lane;
        if (idx < C) {
            const float4 ww = reinterpret_cast<const float4*>(w)[idx];
            const float4 b4 = reinterpret_cast<const float4*>(bb)[idx];
            uint2 pk;
            pk.x = pack_bf16x2((v[u].x - mean) * rstd * ww.x + b4.x, (v[u].y - mean) * rstd * ww.y + b4.y);
            pk.y = pack_bf16x2((v[u].z - mean) * rstd * ww.z + b4.z, (v[u].w - mean) * rstd * ww.w + b4.w);
            yr[idx] = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// head: block per clip. x [B*64, 768] fp32 -> LN -> token mean -> Projection -> L2 normalise
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

__global__ __launch_bounds__(256) void hilo_to_f32_kernel(const bf16_t* __restrict__ hi, long long lo_off, long long n,
                                                          float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = bf16_to_f32(hi[i]) + bf16_to_f32(hi[i + lo_off]);
}

// latent: block per clip. x [B*64, 768] fp32 -> final LN -> token mean -> bf16 [B, 768]
__global__ __launch_bounds__(256) void latent_kernel(const float* __restrict__ x, const float* __restrict__ nw,
                                                     const float* __restrict__ nb, bf16_t* __restrict__ lat, long long lo_off) {
    __shared__ float part[4][LATENT];  // per-wave partial token sums, added in a fixed order (deterministic)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* xb = x + (size_t)blockIdx.x * 64 * LATENT;
    float accl[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) accl[u] = 0.f;
    for (int t = wv; t < 64; t += 4) {
        const float* r = xb + (size_t)t * LATENT;
        const bf16_t* rh = reinterpret_cast<const bf16_t*>(x) + ((size_t)blockIdx.x * 64 + t) * LATENT;
        float v[12], s = 0.f;
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            v[u] = lo_off ? bf16_to_f32(rh[u * 64 + lane]) + bf16_to_f32(rh[u * 64 + lane + lo_off]) : r[u * 64 + lane];
            s += v[u];
        }
        const float mean = wave_sum(s) / (float)LATENT;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 12; ++u) { const float d = v[u] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / (float)LATENT + 1e-5f);
#pragma unroll
        for (int u = 0; u < 12; ++u) accl[u] += (v[u] - mean) * rstd * nw[u * 64 + lane] + nb[u * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) part[wv][u * 64 + lane] = accl[u];
    __syncthreads();
    for (int c = tid; c < LATENT; c += 256)
        lat[(size_t)blockIdx.x * LATENT + c] =
            f32_to_bf16(((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) * (1.f / 64.f));
}

// out[b,:] = normalize( LayerNorm(e[b,:]) ) with e = e1 + e2 already summed by the GEMM epilogue; wave per clip
__global__ __launch_bounds__(256) void proj_ln_norm_kernel(const float* __restrict__ e, int B,
                                                           const float* __restrict__ lw, const float* __restrict__ lb,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* er = e + (size_t)b * OUT;
    float v[16], s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { v[u] = er[u * 64 + lane]; s += v[u]; }
    const float mean = wave_sum(s) / (float)OUT;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { const float d = v[u] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)OUT + 1e-5f);
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        v[u] = (v[u] - mean) * rstd * lw[u * 64 + lane] + lb[u * 64 + lane];
        sq += v[u] * v[u];
    }
    const float nrm = sqrtf(wave_sum(sq));
#pragma unroll
    for (int u = 0; u < 16; ++u) out[(size_t)b * OUT + u * 64 + lane] = v[u] / nrm;
}

}  // namespace htsat

// msclap `Projection` (shared by the audio head and the caption encoder): e1 = lat @ W1^T, e2 = gelu(e1) @ W2^T,
// out = normalize(LayerNorm(e1 + e2)); three small GEMMs (e fp32 and g bf16 are scratch [Bp,1024], Bp = B
// rounded up to 128; lat rows past B must be readable) and one wave-per-row kernel.
int clap_projection(const bf16_t* lat, const bf16_t* W1, const bf16_t* W2, const float* lw, const float* lb, int B,
                    int d_in, float* e, bf16_t* g, float* out, hipStream_t st) {
    const int Bp = (B + 127) / 128 * 128;
    int rc;
    if ((rc = gemm_bf16(lat, W1, nullptr, Bp, htsat::OUT, d_in, 4, e, st))) return rc;
    if ((rc = gemm_bf16(lat, W1, nullptr, Bp, htsat::OUT, d_in, 2, g, st))) return rc;
    if ((rc = gemm_bf16(g, W2, nullptr, Bp, htsat::OUT, htsat::OUT, 3, e, st))) return rc;
    hipLaunchKernelGGL(htsat::proj_ln_norm_kernel, dim3((B + 3) / 4), dim3(256), 0, st, e, B, lw, lb, out);
    WISE_LAUNCH_CHECK("proj_ln_norm_kernel");
    return WISE_OK;
}

namespace htsat {
// ------------------------------------------------------------------------------------------------
// blob layout + workspace
// ------------------------------------------------------------------------------------------------
struct Offsets {
    // fp32 blob
    size_t bn_scale, bn_shift, mel_start, mel_len, mel_wt, hann, pe_w, pe_b, pe_nw, pe_nb;
    size_t blk_f[4][6];    // start of each block's fp32 params
    size_t merge_f[3];     // merge norm w,b
    size_t fin_nw, fin_nb, pj_lw, pj_lb, total_f;
    // bf16 blob
    size_t blk_b[4][6];
    size_t merge_b[3];
    size_t pj_w1, pj_w2, total_b;
};
static Offsets offsets() {
    Offsets o;
    size_t f = 0, w = 0;
    o.bn_scale = f; f += 64; o.bn_shift = f; f += 64; o.mel_start = f; f += 64; o.mel_len = f; f += 64;
    o.mel_wt = f; f += 64 * MELW; o.hann = f; f += N_FFT; o.pe_w = f; f += EMBED * 16; o.pe_b = f; f += EMBED;
    o.pe_nw = f; f += EMBED; o.pe_nb = f; f += EMBED;
    for (int i = 0; i < 4; ++i) {
        const size_t C = (size_t)EMBED << i;
        for (int j = 0; j < DEPTHS[i]; ++j) {
            o.blk_f[i][j] = f; f += 13 * C + 4096 * (size_t)HEADS[i];
            o.blk_b[i][j] = w; w += 12 * C * C;
        }
        if (i < 3) {
            o.merge_f[i] = f; f += 8 * C;
            o.merge_b[i] = w; w += 8 * C * C;
        }
    }
    o.fin_nw = f; f += LATENT; o.fin_nb = f; f += LATENT; o.pj_lw = f; f += OUT; o.pj_lb = f; f += OUT;
    o.total_f = f;
    o.pj_w1 = w; w += (size_t)OUT * LATENT; o.pj_w2 = w; w += (size_t)OUT * OUT;
    o.total_b = w;
    return o;
}

struct Ws {
    size_t mel, x, h, qkv, a, stats, total;
    int Fc;
};
static Ws workspace(int B, int N) {
    Ws w;
    const int frames = 1 + N / HOP;
    w.Fc = frames < MAXF ? frames : MAXF;
    const size_t rows1 = align_up((size_t)B * 4096, 128) + 128;  // stage-1 rows (padded); rows*C is largest at stage 1
    size_t off = 0;
    w.mel = off; off += align_up((size_t)B * w.Fc * 64 * 4, 256);
    w.x = off; off += align_up(rows1 * EMBED * 4, 256);
    w.h = off; off += align_up(rows1 * EMBED * 2, 256);
    w.qkv = off; off += align_up(rows1 * EMBED * 3 * 2, 256);
    w.a = off; off += align_up(rows1 * EMBED * 4 * 2, 256);
    // fold mode (stages 2 - 4): row scales, arrival counters, partial sums per 32 columns; rows * width halves per stage
    w.stats = off; off += align_up(gemm_fold_stats_bytes((int)((rows1 / 4 + 127) / 128 * 128), 2 * EMBED), 256);
    w.total = off;
    return w;
}

}  // namespace htsat
}  // namespace wise

using namespace wise;
using namespace wise::htsat;

extern "C" int wise_htsat_layout(int64_t* wb_elems, int64_t* pf_elems) {
    const Offsets o = offsets();
    if (wb_elems) *wb_elems = (int64_t)o.total_b;
    if (pf_elems) *pf_elems = (int64_t)o.total_f;
    return WISE_OK;
}

extern "C" size_t wise_htsat_workspace_bytes(int batch, int samples) {
    if (batch < 1 || samples < N_FFT / 2 + 1) return 0;
    return workspace(batch, samples).total;
}

static int htsat_forward_impl(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                              float* out, void* workspace_ptr, size_t workspace_bytes, void* stream, int flags) {
    WISE_CHECK_ARG(wb && pf && wave && out, "htsat_forward: null pointer");
    WISE_CHECK_ARG((flags & ~7) == 0 && (!(flags & 1) || flags == 1),
                   "htsat_forward: flags %d (bit 0: LayerNorm fold — alone; bit 1: one-kernel MLP, bit 2: one-kernel norm1 + QKV + window attention of stages 2 and 3)", flags);
    WISE_CHECK_ARG(batch >= 1 && samples >= N_FFT / 2 + 1, "htsat_forward: batch=%d samples=%d", batch, samples);
    const Ws ws = workspace(batch, samples);
    if (!workspace_ptr || workspace_bytes < ws.total) {
        set_error("htsat_forward: workspace %zu < %zu bytes", workspace_bytes, ws.total);
        return WISE_E_WORKSPACE;
    }
    WISE_CHECK_ARG(((uintptr_t)workspace_ptr & 255) == 0 && ((uintptr_t)wb & 15) == 0 && ((uintptr_t)pf & 15) == 0,
                   "htsat_forward: workspace must be 256-byte and weight blobs 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const Offsets o = offsets();
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace_ptr);
    float* mel = reinterpret_cast<float*>(wsb + ws.mel);
    float* x = reinterpret_cast<float*>(wsb + ws.x);
    bf16_t* h = reinterpret_cast<bf16_t*>(wsb + ws.h);
    bf16_t* qkv = reinterpret_cast<bf16_t*>(wsb + ws.qkv);
    bf16_t* a = reinterpret_cast<bf16_t*>(wsb + ws.a);
    const int B = batch, Fc = ws.Fc;
    int rc;

    if ((rc = frontend(wave, B, samples, Fc, pf + o.hann, pf + o.mel_start, pf + o.mel_len, pf + o.mel_wt, pf + o.bn_scale,
                       pf + o.bn_shift, mel, st, 16 /* widest band of the 50..8000 Hz filterbank (pack_htsat_weights asserts it) */)))
        return rc;
    if (g_frontend_only) return WISE_OK;
    if (g_fuse_ln & 8)
        hipLaunchKernelGGL(embed_tok_kernel, dim3((unsigned)(B * 16)), dim3(256), 0, st, mel, B, Fc,
                           pf + o.pe_w, pf + o.pe_b, pf + o.pe_nw, pf + o.pe_nb, x);
    else
        hipLaunchKernelGGL(embed_kernel, dim3((unsigned)(B * 64)), dim3(256), 0, st, mel, B, Fc,
                           pf + o.pe_w, pf + o.pe_b, pf + o.pe_nw, pf + o.pe_nb, x);
    WISE_LAUNCH_CHECK("htsat embed_kernel");

    // flags bit 0: stages 2 - 4 with their LayerNorms folded into the GEMMs (gemm_w4.h FoldArgs; the packer stored the folded
    // qkv / fc1 weights and biases): the residual stream of those stages is bf16 hi + lo where the fp32 rows would be, started
    // by the patch-merging projection in front of the stage; norm1 / norm2 launches are gone and nothing reads the rows twice
    const bool fold_on = (flags & 1) != 0;
    float* stats = reinterpret_cast<float*>(wsb + ws.stats);
    long long x_lo_off = 0;     // != 0: x currently holds a hi + lo stream, lo that many elements behind hi
    int H = 64;
    for (int i = 0; i < 4; ++i) {
        const int C = EMBED << i, heads = HEADS[i];
        const int M = B * H * H;
        const int Mp = (M + 127) / 128 * 128;
        for (int j = 0; j < DEPTHS[i]; ++j) {
            const float* p = pf + o.blk_f[i][j];
            const bf16_t* wq = wb + o.blk_b[i][j];
            const float* n1w = p; const float* n1b = p + C; const float* rb = p + 2 * C;
            const float* qb = rb + 4096 * (size_t)heads; const float* pb = qb + 3 * C;
            const float* n2w = pb + C; const float* n2b = n2w + C; const float* f1b = n2b + C; const float* f2b = f1b + 4 * C;
            const bf16_t* wproj = wq + (size_t)3 * C * C; const bf16_t* wf1 = wproj + (size_t)C * C;
            const bf16_t* wf2 = wf1 + (size_t)4 * C * C;
            const int shift = (j % 2 == 1 && H > 8) ? 4 : 0;
            if (fold_on && i >= 1) {
                bf16_t* hi = reinterpret_cast<bf16_t*>(x);
                const long long lo_off = (long long)Mp * C;
                if ((rc = gemm_fold_bf16(hi, wq, qb, stats, Mp, 3 * C, C, 0, qkv, st))) return rc;
                const long long items = (long long)B * (H / 8) * (H / 8) * heads;
                hipLaunchKernelGGL(swin_attention_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, qkv, B, H,
                                   H, C, heads, shift, rb, h);
                WISE_LAUNCH_CHECK("htsat swin_attention_kernel");
                if ((rc = gemm_fold_resid(h, wproj, pb, Mp, C, C, hi, lo_off, stats, 1e-5f, st, 1))) return rc;
                if ((rc = gemm_fold_bf16(hi, wf1, f1b, stats, Mp, 4 * C, C, 2, a, st))) return rc;
                if ((rc = gemm_fold_resid(a, wf2, f2b, Mp, C, 4 * C, hi, lo_off, stats, 1e-5f, st, 1))) return rc;
                continue;
            }
            // stage 1 (C = 96): a block holds whole rows, so LayerNorm happens inside the GEMM's A-tile build and the
            // normalised activations never travel through HBM (LN 71 + GEMM 151 us -> 142 us; LN 71 + 200 -> 205).
            // At C = 192 the fused kernel (one block per CU) loses to LayerNorm + the tuned GEMMs (113 vs 93 us,
            // 173 vs 140 us), so stage 2 keeps the two-kernel form.
            const bool fuse_ln = (g_fuse_ln & 1) && C == 96 && gemm_ln_supported(3 * C, C, 0);
            if (C == 96 && H == 64 && (g_fuse_ln & 4)) {
                // stage 1: LayerNorm, QKV, window attention, projection and residual add in one kernel (2 workgroups per CU)
                const int nwin = B * 64;
                const int grid = nwin < 512 ? nwin : 512;
                // counted with the GEMM family: the QKV and projection products (the 64 x 64 attention products are not)
                ProfScope prof(PROF_GEMM, 2.0 * (double)M * 96.0 * (288.0 + 96.0), st);
                if (shift)
                    hipLaunchKernelGGL(swin96_block_attn_kernel<4>, dim3(grid), dim3(256), 0, st, x, B, n1w, n1b, wq, qb, rb,
                                       wproj, pb);
                else
                    hipLaunchKernelGGL(swin96_block_attn_kernel<0>, dim3(grid), dim3(256), 0, st, x, B, n1w, n1b, wq, qb, rb,
                                       wproj, pb);
                WISE_LAUNCH_CHECK("htsat swin96_block_attn_kernel");
            } else if ((flags & 4) && swin_qkv_attn_ok(C, H)) {
                // flags bit 2: norm1, the QKV projection and the window attention in one kernel (swin_stream.hip: the normalised rows
                // and the qkv rows never written); the packer stored W_qkv and its bias as that kernel's stream
                if ((rc = swin_qkv_attn(x, n1w, n1b, 1e-5f, wq, qb, rb, h, B, H, C, shift, st))) return rc;
                if ((rc = gemm_bf16(h, wproj, pb, Mp, C, C, 3, x, st))) return rc;
            } else {
            if (fuse_ln) {
                if ((rc = gemm_ln_bf16(x, n1w, n1b, wq, qb, Mp, 3 * C, C, 1e-5f, 0, qkv, st))) return rc;
            } else {
                if ((rc = layernorm_f32_bf16(x, n1w, n1b, M, C, 1e-5f, h, st))) return rc;
                if ((rc = gemm_bf16(h, wq, qb, Mp, 3 * C, C, 0, qkv, st))) return rc;
            }
            {
                const long long items = (long long)B * (H / 8) * (H / 8) * heads;
                hipLaunchKernelGGL(swin_attention_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, qkv, B, H,
                                   H, C, heads, shift, rb, h);
                WISE_LAUNCH_CHECK("htsat swin_attention_kernel");
            }
            if ((rc = gemm_bf16(h, wproj, pb, Mp, C, C, 3, x, st))) return rc;
            }
            if (C == 96 && (g_fuse_ln & 2)) {
                // the whole MLP in one kernel: the 384-wide hidden activations (403 MB at batch 128) stay on chip
                if ((rc = mlp96_fused(x, n2w, n2b, wf1, f1b, wf2, f2b, Mp, 1e-5f, st))) return rc;
            } else {
                if ((flags & 2) && (C == 192 || C == 384)) {
                    // flags bit 1: fc1, GELU and fc2 in one kernel, the hidden activations never written (mlp_stream.hip); the
                    // packer stored the two matrices as that kernel's stream in the fc1 + fc2 slots
                    // (norm2 happens inside the kernel, on the way into its operand registers: no LayerNorm launch, no h)
                    if ((rc = mlp_stream(nullptr, wf1, f1b, f2b, x, Mp, C, st, n2w, n2b, 1e-5f))) return rc;
                    continue;
                }
                if (fuse_ln) {
                    if ((rc = gemm_ln_bf16(x, n2w, n2b, wf1, f1b, Mp, 4 * C, C, 1e-5f, 2, a, st))) return rc;
                } else {
                    if ((rc = layernorm_f32_bf16(x, n2w, n2b, M, C, 1e-5f, h, st))) return rc;
                    if ((rc = gemm_bf16(h, wf1, f1b, Mp, 4 * C, C, 2, a, st))) return rc;
                }
                if ((rc = gemm_bf16(a, wf2, f2b, Mp, C, 4 * C, 3, x, st))) return rc;
            }
        }
        if (i < 3) {
            const float* mp = pf + o.merge_f[i];
            const long long toks = (long long)B * (H / 2) * (H / 2);
            const dim3 grid((unsigned)((toks + 3) / 4)), block(256);
            switch ((C + 63) / 64) {
                case 2: hipLaunchKernelGGL(merge_ln_kernel<2>, grid, block, 0, st, x, B, H, H, C, mp, mp + 4 * C, h, x_lo_off); break;
                case 3: hipLaunchKernelGGL(merge_ln_kernel<3>, grid, block, 0, st, x, B, H, H, C, mp, mp + 4 * C, h, x_lo_off); break;
                case 6: hipLaunchKernelGGL(merge_ln_kernel<6>, grid, block, 0, st, x, B, H, H, C, mp, mp + 4 * C, h, x_lo_off); break;
                default: set_error("htsat: unexpected C=%d", C); return WISE_E_UNSUPPORTED;
            }
            WISE_LAUNCH_CHECK("htsat merge_ln_kernel");
            const int M2 = (int)toks, M2p = (M2 + 127) / 128 * 128;
            if (fold_on) {   // the projection STARTS the next stage's hi + lo stream and its first statistics
                if (hipMemsetAsync(stats + M2p, 0, gemm_fold_counters_bytes(M2p), st) != hipSuccess) {
                    set_error("htsat_forward: memset of the fold counters failed");
                    return WISE_E_INVALID;
                }
                x_lo_off = (long long)M2p * 2 * C;
                if ((rc = gemm_fold_resid(h, wb + o.merge_b[i], nullptr, M2p, 2 * C, 4 * C, reinterpret_cast<bf16_t*>(x), x_lo_off,
                                          stats, 1e-5f, st, 1, false)))
                    return rc;
            } else if ((rc = gemm_bf16(h, wb + o.merge_b[i], nullptr, M2p, 2 * C, 4 * C, 4, x, st))) return rc;
            H >>= 1;
        }
    }
    // head: final LN + token mean -> latent bf16 [Bp,768] (aliases h); then the msclap Projection
    {
        hipLaunchKernelGGL(latent_kernel, dim3(B), dim3(256), 0, st, x, pf + o.fin_nw, pf + o.fin_nb, h, x_lo_off);
        WISE_LAUNCH_CHECK("htsat latent_kernel");
        if ((rc = clap_projection(h, wb + o.pj_w1, wb + o.pj_w2, pf + o.pj_lw, pf + o.pj_lb, B, LATENT,
                                  reinterpret_cast<float*>(qkv), a, out, st)))
            return rc;
    }
    return WISE_OK;
}

extern "C" int wise_htsat_forward(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                                  float* out, void* workspace_ptr, size_t workspace_bytes, void* stream) {
    return htsat_forward_impl(wb, pf, wave, batch, samples, out, workspace_ptr, workspace_bytes, stream, 0);
}
extern "C" int wise_htsat_forward2(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                                   float* out, void* workspace_ptr, size_t workspace_bytes, int flags, void* stream) {
    return htsat_forward_impl(wb, pf, wave, batch, samples, out, workspace_ptr, workspace_bytes, stream, flags);
}

#ifdef WISE_DEBUG_KNOBS
extern "C" int wise_debug_set_htsat(int flags) {
    wise::htsat::g_fuse_ln = 15 & ~((flags & 7) | ((flags >> 4) & 1) << 3);   // flags bit 0: no LayerNorm fusion at all, bit 1: no fused MLP, bit 2: no fused attention half, bit 4: lane-per-channel embedding
    wise::htsat::g_frontend_only = (flags >> 3) & 1;   // bit 3: front end only
    wise::htsat::frontend_set_variant((flags >> 5) & 1);   // bit 5: the full-length FFT front end
    return 0;
}
#endif

extern "C" int wise_htsat_tap(int what, const void* workspace_ptr, int batch, int samples, float* dst, int64_t count,
                              void* stream) {
    WISE_CHECK_ARG(workspace_ptr && dst && count > 0 && batch >= 1, "htsat_tap: bad argument");
    const Ws ws = workspace(batch, samples);
    const unsigned char* wsb = reinterpret_cast<const unsigned char*>(workspace_ptr);
    if (what == 2) {   // the last stage's rows of a fold-mode forward: hi + lo -> fp32
        const long long Mp = ((long long)batch * 64 + 127) / 128 * 128;
        hipLaunchKernelGGL(hilo_to_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const bf16_t*>(wsb + ws.x), Mp * LATENT, (long long)count, dst);
        WISE_LAUNCH_CHECK("hilo_to_f32_kernel");
        return WISE_OK;
    }
    const size_t off = (what == 0) ? ws.mel : ws.x;
    hipError_t e = hipMemcpyAsync(dst, wsb + off, (size_t)count * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("htsat_tap: %s", hipGetErrorString(e)); return (int)e; }
    return WISE_OK;
}
