// MS-CLAP version '2022' audio encoder: PANNs Cnn14 + msclap Projection (gfx950).
//
// Stands behind  self.model.clap.audio_encoder(preprocessed_audio)[0]  of the reference's
// src/feature/microsoft_clap.py:49 when the feature id is microsoft/clap/2022/... (:20-31 accepts every key of
// msclap's CLAP.model_name; '2022' builds AudioEncoder('Cnn14', 2048, 1024)).  CPU restatement: oracle/cnn14_ref.py.
//
//   front end     htsat_frontend.hip's kernel (STFT power -> sparse mel -> dB -> folded bn0); the 2022 config differs
//                 from the 2023 one only in the filterbank (fmax 14000), which the packer builds       fp32 [B, T, 64]
//   conv_first    block 1's first convolution, 1 -> 64 channels: nine taps per output, VALU            bf16 NHWC
//   conv3x3       the other eleven convolutions as implicit GEMMs on the matrix cores (gemm_bf16.hip:
//                 no im2col buffer — a K-tile is 64 channels of one tap, gathered by the LDS-DMA);
//                 BatchNorm folded into the weights and a bias by the packer, ReLU in the epilogue     bf16 NHWC
//   avgpool2      2x2 average pooling (floor) after blocks 1-5: fused into the second convolution of the block (the tile's
//                 rows are the members of pooling windows; see conv3x3_kernel), the unpooled tensor is never written
//   head_pool     mean over the mel axis, max over time + mean over time                               bf16 [Bp, 2048]
//   fc1 + ReLU    gemm_bf16 (mode 6), then msclap's Projection (clap_projection, htsat.hip) and the L2 normalisation
//
// Layout: activations are position-major ("NHWC"): row p = (b*T + t)*F + f holds the C channels of one
// time-frequency cell, so a convolution's A operand rows are contiguous channel vectors.  Two activation buffers
// alternate; rows are padded to the 256-row tile (padding rows are written, never read as data).
#include <hip/hip_runtime.h>

#include "common.h"
#include "transformer.h"

namespace wise {
namespace cnn14 {
constexpr int N_FFT = 1024, HOP = 320, MELW = htsat::FRONT_MELW;
constexpr int NBLK = 6, EMB = 2048, OUT = 1024, MIN_FRAMES = 32;
constexpr int CH[NBLK] = {64, 128, 256, 512, 1024, 2048};

// ------------------------------------------------------------------------------------------------
// block 1, conv 1: 1 -> 64 channels.  A thread owns ONE group of 8 output channels (lane & 7) for the life of the
// kernel — its 72 weights (BatchNorm scale folded) and 8 shifts stay in registers — and walks cells with a grid
// stride; the eight threads of a cell write its 128 bytes together.  Nine input values per cell come from the log-mel
// (cache hits: neighbouring cells share them).  Bound by the bf16 tensor it writes (787 MB at 64 clips x 10 s).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ mel /*[B,T,64]*/,
                                                         const float* __restrict__ w /*[64][9]*/,
                                                         const float* __restrict__ shift /*[64]*/, int T, long long cells,
                                                         bf16_t* __restrict__ out /*[cells, 64]*/) {
    const int g = threadIdx.x & 7;
    float wr[8][9], sh[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        sh[c] = shift[g * 8 + c];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wr[c][tap] = w[(g * 8 + c) * 9 + tap];
    }
    const long long stride = (long long)gridDim.x * 32;
    for (long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); p < cells; p += stride) {
        const int f = (int)(p & 63);
        const int t = (int)((p >> 6) % T);
        float x[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool in = (unsigned)(t + dy) < (unsigned)T && (unsigned)(f + dx) < 64u;
            x[tap] = in ? mel[p + dy * 64 + dx] : 0.f;
        }
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a = fmaf(x[tap], wr[c][tap], a);
            v[c] = fmaxf(a + sh[c], 0.f);
        }
        uint4 pk;
        pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
        pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(out + p * 64 + g * 8) = pk;
    }
}

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// [B, T, F, C] -> lat [B, C]: mean over F, then max over T + mean over T; thread = (clip, 8 channels)
__global__ __launch_bounds__(256) void head_pool_kernel(const bf16_t* __restrict__ x, int B, int T, int F, int C,
                                                        bf16_t* __restrict__ lat) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int cg = C >> 3;
    if (idx >= B * cg) return;
    const int g = idx % cg, b = idx / cg;
    float mx[8], sm[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { mx[c] = -3.0e38f; sm[c] = 0.f; }
    const float invF = 1.f / (float)F;
    for (int t = 0; t < T; ++t) {
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a[8];
        for (int f = 0; f < F; ++f) {
            unpack8(*reinterpret_cast<const uint4*>(x + (((size_t)b * T + t) * F + f) * C + g * 8), a);
#pragma unroll
            for (int c = 0; c < 8; ++c) m[c] += a[c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float v = m[c] * invF; mx[c] = fmaxf(mx[c], v); sm[c] += v; }
    }
    const float invT = 1.f / (float)T;
    uint4 pk;
    pk.x = pack_bf16x2(mx[0] + sm[0] * invT, mx[1] + sm[1] * invT); pk.y = pack_bf16x2(mx[2] + sm[2] * invT, mx[3] + sm[3] * invT);
    pk.z = pack_bf16x2(mx[4] + sm[4] * invT, mx[5] + sm[5] * invT); pk.w = pack_bf16x2(mx[6] + sm[6] * invT, mx[7] + sm[7] * invT);
    *reinterpret_cast<uint4*>(lat + (size_t)b * C + g * 8) = pk;
}

// ------------------------------------------------------------------------------------------------
// blob layout + workspace
// ------------------------------------------------------------------------------------------------
struct Offsets {
    // fp32
    size_t bn_scale, bn_shift, mel_start, mel_len, mel_wt, hann, c0_w, c0_b;
    size_t cb[NBLK][2];          // shift of conv (block, 0|1); [0][0] unused (c0_b)
    size_t fc1_b, pj_lw, pj_lb, total_f;
    // bf16
    size_t cw[NBLK][2];          // [Cout][9][Cin]; [0][0] unused (the first conv is fp32)
    size_t fc1_w, pj_w1, pj_w2, total_b;
};
static Offsets offsets() {
    Offsets o{};
    size_t f = 0, w = 0;
    o.bn_scale = f; f += 64; o.bn_shift = f; f += 64; o.mel_start = f; f += 64; o.mel_len = f; f += 64;
    o.mel_wt = f; f += 64 * MELW; o.hann = f; f += N_FFT; o.c0_w = f; f += 64 * 9; o.c0_b = f; f += 64;
    for (int i = 0; i < NBLK; ++i) {
        const size_t cin = i ? CH[i - 1] : 1, cout = CH[i];
        for (int j = 0; j < 2; ++j) {
            if (i == 0 && j == 0) continue;
            o.cb[i][j] = f; f += cout;
            o.cw[i][j] = w; w += cout * 9 * (j ? cout : cin);
        }
    }
    o.fc1_b = f; f += EMB; o.pj_lw = f; f += OUT; o.pj_lb = f; f += OUT;
    o.total_f = f;
    o.fc1_w = w; w += (size_t)EMB * EMB; o.pj_w1 = w; w += (size_t)OUT * EMB; o.pj_w2 = w; w += (size_t)OUT * OUT;
    o.total_b = w;
    return o;
}

struct Ws {
    size_t zeros, mel, a, b, lat, h, e, g, total;
    int T;
};
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static Ws workspace(int B, int samples) {
    Ws w{};
    w.T = samples / HOP + 1;
    const size_t Bp = (size_t)(B + 127) / 128 * 128;
    const size_t rows1 = ((size_t)B * w.T * 64 + 255) / 256 * 256;     // block 1 holds the largest tensors (rows padded to the 256-row tile)
    size_t off = 0;
    w.zeros = off; off += 256;
    w.mel = off; off += up256((size_t)B * w.T * 64 * 4);
    w.a = off; off += up256(rows1 * 64 * 2);
    w.b = off; off += up256(rows1 * 64 * 2);
    w.lat = off; off += up256(Bp * EMB * 2);
    w.h = off; off += up256(Bp * EMB * 2);
    w.e = off; off += up256(Bp * OUT * 4);
    w.g = off; off += up256(Bp * OUT * 2);
    w.total = off;
    return w;
}
static bool shape_ok(int B, int samples) {
    if (B < 1 || samples < N_FFT / 2 + 1) return false;
    const long long T = samples / HOP + 1;
    return T >= MIN_FRAMES && (long long)B * T * 64 < (1ll << 31) - 256;
}

static int forward(const bf16_t* wb, const float* pf, const float* wave, int B, int samples, float* out,
                   unsigned char* wsb, hipStream_t st) {
    const Offsets o = offsets();
    const Ws ws = workspace(B, samples);
    const bf16_t* zeros = reinterpret_cast<const bf16_t*>(wsb + ws.zeros);
    float* mel = reinterpret_cast<float*>(wsb + ws.mel);
    bf16_t* cur = reinterpret_cast<bf16_t*>(wsb + ws.a);
    bf16_t* nxt = reinterpret_cast<bf16_t*>(wsb + ws.b);
    bf16_t* lat = reinterpret_cast<bf16_t*>(wsb + ws.lat);
    bf16_t* h = reinterpret_cast<bf16_t*>(wsb + ws.h);
    const int Bp = (B + 127) / 128 * 128;
    hipError_t e = hipMemsetAsync(wsb + ws.zeros, 0, 256, st);
    if (e == hipSuccess) e = hipMemsetAsync(lat, 0, (size_t)Bp * EMB * 2, st);   // padding rows of the head's GEMMs
    if (e != hipSuccess) { set_error("cnn14: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    int rc;
    int T = ws.T, F = 64;
    if ((rc = htsat::frontend(wave, B, samples, T, pf + o.hann, pf + o.mel_start, pf + o.mel_len, pf + o.mel_wt,
                              pf + o.bn_scale, pf + o.bn_shift, mel, st)))
        return rc;
    {
        const long long cells = (long long)B * T * 64;
        const long long want = (cells + 31) / 32;
        hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)(want < 256 * 16 ? want : 256 * 16)), dim3(256), 0, st, mel,
                           pf + o.c0_w, pf + o.c0_b, T, cells, cur);
        WISE_LAUNCH_CHECK("cnn14 conv_first_kernel");
    }
    for (int i = 0; i < NBLK; ++i) {
        const int cin = i ? CH[i - 1] : 1, cout = CH[i];
        for (int j = 0; j < 2; ++j) {
            if (i == 0 && j == 0) continue;
            // the block's second convolution carries the 2x2 average pooling that follows it (blocks 1-5)
            if ((rc = conv3x3_bf16(cur, wb + o.cw[i][j], pf + o.cb[i][j], zeros, B, T, F, j ? cout : cin, cout,
                                   j == 1 && i < NBLK - 1, nxt, st)))
                return rc;
            bf16_t* t = cur; cur = nxt; nxt = t;
        }
        if (i < NBLK - 1) { T /= 2; F /= 2; }
    }
    hipLaunchKernelGGL(head_pool_kernel, dim3((B * (EMB / 8) + 255) / 256), dim3(256), 0, st, cur, B, T, F, EMB, lat);
    WISE_LAUNCH_CHECK("cnn14 head_pool_kernel");
    if ((rc = gemm_bf16(lat, wb + o.fc1_w, pf + o.fc1_b, Bp, EMB, EMB, 6 /*ReLU*/, h, st))) return rc;
    return clap_projection(h, wb + o.pj_w1, wb + o.pj_w2, pf + o.pj_lw, pf + o.pj_lb, B, EMB,
                           reinterpret_cast<float*>(wsb + ws.e), reinterpret_cast<bf16_t*>(wsb + ws.g), out, st);
}

}  // namespace cnn14
}  // namespace wise

using namespace wise;

extern "C" int wise_cnn14_layout(int64_t* wb_elems, int64_t* pf_elems) {
    WISE_CHECK_ARG(wb_elems && pf_elems, "cnn14_layout: null pointer");
    const cnn14::Offsets o = cnn14::offsets();
    *wb_elems = (int64_t)o.total_b;
    *pf_elems = (int64_t)o.total_f;
    return WISE_OK;
}

extern "C" size_t wise_cnn14_workspace_bytes(int batch, int samples) {
    if (!cnn14::shape_ok(batch, samples)) return 0;
    return cnn14::workspace(batch, samples).total;
}

extern "C" int wise_cnn14_forward(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                                  float* out, void* workspace, size_t workspace_bytes, void* stream) {
    WISE_CHECK_ARG(wb && pf && wave && out && workspace, "cnn14_forward: null pointer");
    WISE_CHECK_ARG(cnn14::shape_ok(batch, samples), "cnn14_forward: batch=%d samples=%d unsupported (at least %d STFT frames)",
                   batch, samples, cnn14::MIN_FRAMES);
    WISE_CHECK_ARG(workspace_bytes >= cnn14::workspace(batch, samples).total, "cnn14_forward: workspace too small (%zu < %zu)",
                   workspace_bytes, cnn14::workspace(batch, samples).total);
    return cnn14::forward(wb, pf, wave, batch, samples, out, reinterpret_cast<unsigned char*>(workspace),
                          (hipStream_t)stream);
}

// parity taps of the last forward in this workspace: 0 = log-mel + bn0 fp32 [B*T*64], 1 = pooled latent bf16 [B*2048],
// 2 = fc1 output ('embedding') bf16 [B*2048]
extern "C" int wise_cnn14_tap(int what, const void* workspace_ptr, int batch, int samples, void* dst, int64_t bytes,
                              void* stream) {
    WISE_CHECK_ARG(workspace_ptr && dst && bytes > 0 && cnn14::shape_ok(batch, samples) && what >= 0 && what <= 2,
                   "cnn14_tap: bad argument");
    const cnn14::Ws ws = cnn14::workspace(batch, samples);
    const unsigned char* wsb = reinterpret_cast<const unsigned char*>(workspace_ptr);
    const size_t off = what == 0 ? ws.mel : what == 1 ? ws.lat : ws.h;
    hipError_t e = hipMemcpyAsync(dst, wsb + off, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("cnn14_tap: %s", hipGetErrorString(e)); return (int)e; }
    return WISE_OK;
}

// relu(conv3x3(x [B, T, F, Cin] bf16 NHWC, padding 1) + bias) -> out [ceil256(B*T*F), Cout] bf16, or (pool != 0) its 2x2
// average pooling [B*(T/2)*(F/2), Cout]; wt [Cout, 9*Cin],
// k = (kh*3 + kw)*Cin + c; zeros = 16 bytes of zeros on the device.  The building block above, exposed for parity tests
// and for callers with other convolutional encoders.
extern "C" int wise_conv3x3_relu_bf16(const uint16_t* x, const uint16_t* wt, const float* bias, const uint16_t* zeros,
                                      int batch, int T, int F, int cin, int cout, int pool, uint16_t* out, void* stream) {
    return conv3x3_bf16(x, wt, bias, zeros, batch, T, F, cin, cout, pool != 0, out, (hipStream_t)stream);
}
