// MS-CLAP HTSAT audio front end: STFT power spectrum -> log-mel -> bn0, one kernel (htsat.hip calls it first).
//
// This file — like every product file since the same fault showed up in the search re-scoring kernel — is compiled
// without packed f32 math (wise_amd/build.py).  With the SLP vectoriser on, the butterflies
// become v_pk_{fma,mul,add}_f32 and the kernel, correct on its own, returns a few wrong frames whenever its waves
// share a SIMD with waves of another stream's kernel that issues MFMA (the fused LayerNorm GEMM / fused MLP of a
// second forward in flight, or a bare MFMA loop: tools/htsat_neighbours.py, DESIGN.md "Two HTSAT batches in flight").
// Built from scalar fp32 instructions the same experiment is bit-identical to the serial run over every
// neighbour tried, at the same speed (the kernel is bound by its LDS exchanges and the log-mel tail).
#include <hip/hip_runtime.h>

#include <mutex>

#include "common.h"
#include "transformer.h"

namespace wise {
namespace htsat {
constexpr int N_FFT = 1024, HOP = 320, MELW = FRONT_MELW;

// ------------------------------------------------------------------------------------------------
// frontend: a wave computes one STFT frame at a time and loops over frames (persistent grid)
//
// 1024-point radix-2 decimation-in-time FFT, butterflies in exactly the textbook order, but each lane keeps 16
// points in registers and runs 4 + 4 + 2 stages there; only the two regroupings between them go through LDS:
//   group 1 (stages 1-4):  lane L owns the contiguous points 16L .. 16L+15 of the bit-reversed sequence.  It
//                          loads them straight from global memory: x[brev4(e)*64 + brev6(L)] — for a fixed e
//                          the wave reads one contiguous 256-byte window;
//   exchange 1:            64 x 16 transpose through a pitch-17 image;
//   group 2 (stages 5-8):  lane (blk, r) owns points blk*256 + r + 16q, q < 16;
//   exchange 2:            natural order, 16 points of padding per 256 so the four blk groups use both
//                          halves of the banks;
//   group 3 (stages 9-10): lane owns points lane + 64u + 256q', u, q' < 4; |X|^2 of bins 0..512 go to LDS
//                          for the mel filterbank.
// Twiddles tw[half + j] = exp(-i*pi*j/half) come from a table built once per process (fft_twiddle_kernel);
// every per-lane twiddle, the lane's Hann taps and its mel weights are loaded once, before the frame loop.
// ------------------------------------------------------------------------------------------------
__device__ float2 g_fft_tw[1024];

__global__ void fft_twiddle_kernel() {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 1024) return;
    const int half = idx ? (1 << (31 - __clz(idx))) : 1;
    const int j = idx - half;
    float s, c;
    sincospif(idx ? (float)j / (float)half : 0.f, &s, &c);
    g_fft_tw[idx] = make_float2(c, -s);
}

// one radix-2 butterfly, the arithmetic of the stage loop it replaces
__device__ __forceinline__ void bfly(float2& a, float2& q, const float2 t) {
    const float tr = t.x * q.x - t.y * q.y, ti = t.x * q.y + t.y * q.x;
    const float2 a0 = a;
    a = make_float2(a0.x + tr, a0.y + ti);
    q = make_float2(a0.x - tr, a0.y - ti);
}

constexpr int FFT_LDS = 1088;  // float2 per wave: max(64 * 17, 1024 + 3 * 16)

// MW = mel weights a lane keeps and applies per frame: the widest band of the filterbank in use (16 FFT bins for the 2023
// config's 50..8000 Hz, 35 for the 2022 config's 50..14000 Hz; the table's row stride is MELW either way).  The kernel is
// bound by its LDS operations (~108 per frame and lane), so the 20 reads a narrow filterbank does not need are worth leaving out.
template <int MW>
__global__ __launch_bounds__(256, 2) void frontend_kernel(const float* __restrict__ wave, int B, int N, int Fc,
                                                          const float* __restrict__ hann,
                                                          const float* __restrict__ mel_start,
                                                          const float* __restrict__ mel_len,
                                                          const float* __restrict__ mel_wt /*[MELW][64]*/,
                                                          const float* __restrict__ bn_scale,
                                                          const float* __restrict__ bn_shift,
                                                          float* __restrict__ melbn /*[B,Fc,64]*/) {
    __shared__ float2 bufs[4][FFT_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float2* buf = bufs[wv];
    const float2* __restrict__ tw = g_fft_tw;
    const int brev6 = (int)(__brev((unsigned)lane) >> 26);
    const int blk = lane >> 4, r = lane & 15;

    // ---- per-lane constants
    float hw[16];            // Hann taps of the lane's 16 input samples
#pragma unroll
    for (int e = 0; e < 16; ++e) hw[e] = hann[(int)(__brev((unsigned)e) >> 28) * 64 + brev6];
    float2 t2[15];           // group 2: stage 5 -> t2[0], stage 6 -> t2[1..2], stage 7 -> t2[3..6], stage 8 -> t2[7..14]
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
#pragma unroll
        for (int jl = 0; jl < (1 << sg); ++jl) t2[(1 << sg) - 1 + jl] = tw[(16 << sg) + r + 16 * jl];
    float2 t9[4], t10[8];    // group 3: stage 9 twiddle per u, stage 10 per (u, pair)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        t9[u] = tw[256 + lane + 64 * u];
        t10[2 * u] = tw[512 + lane + 64 * u];
        t10[2 * u + 1] = tw[512 + lane + 64 * u + 256];
    }
    const int mst = (int)mel_start[lane], mln = (int)mel_len[lane];
    float mw[MW];
#pragma unroll
    for (int j = 0; j < MW; ++j) mw[j] = (j < mln) ? mel_wt[j * 64 + lane] : 0.f;
    const float bsc = bn_scale[lane], bsh = bn_shift[lane];

    const long long total = (long long)B * Fc;
    for (long long fid = (long long)blockIdx.x * 4 + wv; fid < total; fid += (long long)gridDim.x * 4) {
        const int b = (int)(fid / Fc), f = (int)(fid % Fc);
        const float* w = wave + (size_t)b * N;
        float2 v[16];
        // ---- group 1: bit-reversed points 16*lane + e  <-  sample brev4(e)*64 + brev6(lane), reflect padded
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = (int)(__brev((unsigned)e) >> 28) * 64 + brev6;
            int s = f * HOP - N_FFT / 2 + n;
            if (s < 0) s = -s;
            if (s >= N) s = 2 * (N - 1) - s;
            v[e] = make_float2(w[s] * hw[e], 0.f);
        }
#pragma unroll
        for (int st = 1; st <= 4; ++st) {
            const int half = 1 << (st - 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = k & (half - 1);
                const int i0 = ((k >> (st - 1)) << st) + j;
                bfly(v[i0], v[i0 + half], tw[half + j]);   // wave-uniform table entries
            }
        }
        // ---- exchange 1: (L1 = point >> 4, e = point & 15) kept at L1*17 + e
#pragma unroll
        for (int e = 0; e < 16; ++e) buf[lane * 17 + e] = v[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = buf[(blk * 16 + q) * 17 + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- group 2: stages 5-8 on points blk*256 + r + 16q
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const int hq = 1 << sg;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int jl = k & (hq - 1);
                const int q0 = ((k >> sg) << (sg + 1)) + jl;
                bfly(v[q0], v[q0 + hq], t2[hq - 1 + jl]);
            }
        }
        // ---- exchange 2: point p kept at p + (p >> 8) * 16
#pragma unroll
        for (int q = 0; q < 16; ++q) buf[blk * 272 + r + 16 * q] = v[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[qq * 4 + u] = buf[qq * 272 + lane + 64 * u];   // point lane + 64u + 256qq
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- group 3: stage 9 pairs (q' = 0,1), (2,3); stage 10 pairs (0,2), (1,3)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bfly(v[0 * 4 + u], v[1 * 4 + u], t9[u]);
            bfly(v[2 * 4 + u], v[3 * 4 + u], t9[u]);
            bfly(v[0 * 4 + u], v[2 * 4 + u], t10[2 * u]);
            bfly(v[1 * 4 + u], v[3 * 4 + u], t10[2 * u + 1]);
        }
        // ---- power of bins 0..512 -> LDS (float view of the wave's buffer)
        float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float2 x = v[qq * 4 + u];
                pw[qq * 256 + lane + 64 * u] = x.x * x.x + x.y * x.y;
            }
        if (lane == 0) pw[512] = v[8].x * v[8].x + v[8].y * v[8].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < MW; ++j) acc = fmaf(mw[j], pw[min(mst + j, 512)], acc);
        const float db = 10.f * log10f(fmaxf(acc, 1e-10f));
        melbn[((size_t)b * Fc + f) * 64 + lane] = db * bsc + bsh;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}


// ------------------------------------------------------------------------------------------------
// The same front end on a HALF-LENGTH transform (the form that runs).  A frame is real, so its 1024-point DFT is one
// 512-point complex FFT of z[n] = x[2n] + i x[2n+1] and a recombination,
//     X[k] = E[k] - i W^k O[k],   E = (Z[k] + conj Z[512-k]) / 2,   O = (Z[k] - conj Z[512-k]) / 2,   W = exp(-2 pi i / 1024),
// for k = 0 .. 511, and X[512] = Re Z[0] - Im Z[0]: 36 butterflies per lane instead of 80 and 72 LDS operations per
// frame and lane instead of 108 (the kernel is bound by those).  Same decimation-in-time butterflies in the same order,
// 3 + 3 + 3 stages in registers:
//   group 1 (stages 1-3):  lane L owns points 8L .. 8L+7 of the bit-reversed sequence = z[brev3(e)*64 + brev6(L)];
//   exchange 1:            64 x 8 transpose through a pitch-9 image;
//   group 2 (stages 4-6):  lane (blk, r) owns points blk*64 + r + 8q, q < 8;
//   exchange 2:            natural order, 8 points of padding per 64;
//   group 3 (stages 7-9):  lane owns points lane + 64u, u < 8 -> Z[lane + 64u];
//   recombination:         Z through LDS once more (a lane needs Z[512-k] for its eight k), |X|^2 of bins 0..512 to LDS
//                          for the mel filterbank, as before.
// Against the full-length form the spectrum differs by rounding only (~1e-7 relative; the log-mel tests hold the same
// 2e-3 dB).
// ------------------------------------------------------------------------------------------------
template <int MW>
__global__ __launch_bounds__(256, 2) void frontend_rfft_kernel(const float* __restrict__ wave, int B, int N, int Fc,
                                                               const float* __restrict__ hann,
                                                               const float* __restrict__ mel_start,
                                                               const float* __restrict__ mel_len,
                                                               const float* __restrict__ mel_wt /*[MELW][64]*/,
                                                               const float* __restrict__ bn_scale,
                                                               const float* __restrict__ bn_shift,
                                                               float* __restrict__ melbn /*[B,Fc,64]*/) {
    __shared__ float2 bufs[4][FFT_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float2* buf = bufs[wv];
    const float2* __restrict__ tw = g_fft_tw;
    const int brev6 = (int)(__brev((unsigned)lane) >> 26);
    const int blk = lane >> 3, r = lane & 7;

    // ---- per-lane constants
    float2 hw[8];            // Hann taps of the lane's 8 complex inputs (samples 2n, 2n+1)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int n = (int)(__brev((unsigned)e) >> 29) * 64 + brev6;
        hw[e] = make_float2(hann[2 * n], hann[2 * n + 1]);
    }
    float2 t2[7];            // group 2: stage 4 -> t2[0], stage 5 -> t2[1..2], stage 6 -> t2[3..6]
#pragma unroll
    for (int sg = 0; sg < 3; ++sg)
#pragma unroll
        for (int jl = 0; jl < (1 << sg); ++jl) t2[(1 << sg) - 1 + jl] = tw[(8 << sg) + r + 8 * jl];
    float2 t3[7];            // group 3: stage 7 -> t3[0], stage 8 -> t3[1..2], stage 9 -> t3[3..6]
    t3[0] = tw[64 + lane];
    t3[1] = tw[128 + lane];
    t3[2] = tw[128 + lane + 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) t3[3 + u] = tw[256 + lane + 64 * u];
    float2 tp[8];            // recombination: W^k, k = lane + 64u
#pragma unroll
    for (int u = 0; u < 8; ++u) tp[u] = tw[512 + lane + 64 * u];
    const int mst = (int)mel_start[lane], mln = (int)mel_len[lane];
    float mw[MW];
#pragma unroll
    for (int j = 0; j < MW; ++j) mw[j] = (j < mln) ? mel_wt[j * 64 + lane] : 0.f;
    const float bsc = bn_scale[lane], bsh = bn_shift[lane];
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    const long long total = (long long)B * Fc;
    for (long long fid = (long long)blockIdx.x * 4 + wv; fid < total; fid += (long long)gridDim.x * 4) {
        const int b = (int)(fid / Fc), f = (int)(fid % Fc);
        const float* w = wave + (size_t)b * N;
        float2 v[8];
        // ---- group 1: bit-reversed points 8*lane + e  <-  z[n], n = brev3(e)*64 + brev6(lane), samples reflect padded.
        // All but the first and last two frames of a clip lie inside it: there a lane's pair (x[2n], x[2n+1]) is one
        // 8-byte load at a fixed offset from the frame's first sample (the reflection arithmetic and 64-bit addresses of
        // sixteen separate loads were a quarter of the loop's instructions).
        const long long first = (long long)f * HOP - N_FFT / 2;
        const float* fp = w + first;
        if (first >= 0 && first + N_FFT <= N && ((uintptr_t)fp & 7) == 0) {
            const float2* zp = reinterpret_cast<const float2*>(fp) + brev6;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 x = zp[(int)(__brev((unsigned)e) >> 29) * 64];
                v[e] = make_float2(x.x * hw[e].x, x.y * hw[e].y);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = (int)(__brev((unsigned)e) >> 29) * 64 + brev6;
                int s0 = f * HOP - N_FFT / 2 + 2 * n, s1 = s0 + 1;
                if (s0 < 0) s0 = -s0;
                if (s0 >= N) s0 = 2 * (N - 1) - s0;
                if (s1 < 0) s1 = -s1;
                if (s1 >= N) s1 = 2 * (N - 1) - s1;
                v[e] = make_float2(w[s0] * hw[e].x, w[s1] * hw[e].y);
            }
        }
#pragma unroll
        for (int st = 1; st <= 3; ++st) {
            const int half = 1 << (st - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = k & (half - 1);
                const int i0 = ((k >> (st - 1)) << st) + j;
                bfly(v[i0], v[i0 + half], tw[half + j]);   // wave-uniform table entries
            }
        }
        // ---- exchange 1: (L1 = point >> 3, e = point & 7) kept at L1*9 + e
#pragma unroll
        for (int e = 0; e < 8; ++e) buf[lane * 9 + e] = v[e];
        wave_sync();
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = buf[(blk * 8 + q) * 9 + r];
        wave_sync();
        // ---- group 2: stages 4-6 on points blk*64 + r + 8q
#pragma unroll
        for (int sg = 0; sg < 3; ++sg) {
            const int hq = 1 << sg;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int jl = k & (hq - 1);
                const int q0 = ((k >> sg) << (sg + 1)) + jl;
                bfly(v[q0], v[q0 + hq], t2[hq - 1 + jl]);
            }
        }
        // ---- exchange 2: point p kept at p + (p >> 6) * 8
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[blk * 72 + r + 8 * q] = v[q];
        wave_sync();
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = buf[lane + 72 * u];       // point lane + 64u
        wave_sync();
        // ---- group 3: stage 7 pairs (u, u+1), stage 8 (u, u+2), stage 9 (u, u+4)
#pragma unroll
        for (int u = 0; u < 8; u += 2) bfly(v[u], v[u + 1], t3[0]);
#pragma unroll
        for (int base = 0; base < 8; base += 4)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) bfly(v[base + bb], v[base + bb + 2], t3[1 + bb]);
#pragma unroll
        for (int u = 0; u < 4; ++u) bfly(v[u], v[u + 4], t3[3 + u]);
        // ---- recombination: Z[k] (k = lane + 64u) with conj Z[512 - k]; |X|^2 of bins 0..512 -> LDS
#pragma unroll
        for (int u = 0; u < 8; ++u) buf[lane + 64 * u] = v[u];
        wave_sync();
        float* pw = reinterpret_cast<float*>(buf + 512);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = lane + 64 * u;
            const float2 z = buf[(512 - k) & 511];
            const float ex = 0.5f * (v[u].x + z.x), ey = 0.5f * (v[u].y - z.y);
            const float ox = 0.5f * (v[u].x - z.x), oy = 0.5f * (v[u].y + z.y);
            const float wr = tp[u].x, wi = tp[u].y;
            const float xr = ex + (wr * oy + wi * ox), xi = ey - (wr * ox - wi * oy);
            pw[k] = xr * xr + xi * xi;
        }
        if (lane == 0) { const float ny = v[0].x - v[0].y; pw[512] = ny * ny; }
        wave_sync();
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < MW; ++j) acc = fmaf(mw[j], pw[min(mst + j, 512)], acc);
        const float db = 10.f * log10f(fmaxf(acc, 1e-10f));
        melbn[((size_t)b * Fc + f) * 64 + lane] = db * bsc + bsh;
        wave_sync();
    }
}

static int g_frontend_full_fft = 0;      // (debug) 1: the full-length transform of rounds 1-2
void frontend_set_variant(int full_fft) { g_frontend_full_fft = full_fft; }

// mel+bn0 of the first Fc frames of every clip: melbn [B, Fc, 64]; params are the packed fp32 front-end tables
int frontend(const float* wave, int B, int samples, int Fc, const float* hann, const float* mel_start,
             const float* mel_len, const float* mel_wt, const float* bn_scale, const float* bn_shift, float* melbn,
             hipStream_t st, int max_band) {
    static PerDeviceOnce once;
    static hipError_t init_err = hipSuccess;
    once([&] {   // the twiddle table lives in a __device__ array; complete before any stream reads it
        hipLaunchKernelGGL(fft_twiddle_kernel, dim3(4), dim3(256), 0, st);
        init_err = hipGetLastError();
        if (init_err == hipSuccess) init_err = hipStreamSynchronize(st);
    });
    if (init_err != hipSuccess) { set_error("htsat fft_twiddle_kernel: %s", hipGetErrorString(init_err)); return (int)init_err; }
    const long long fblocks = ((long long)B * Fc + 3) / 4;
    const dim3 grid((unsigned)(fblocks < 512 ? fblocks : 512));
    if (g_frontend_full_fft) {
        if (max_band <= 16)
            hipLaunchKernelGGL(frontend_kernel<16>, grid, dim3(256), 0, st, wave, B, samples, Fc, hann, mel_start, mel_len, mel_wt,
                               bn_scale, bn_shift, melbn);
        else
            hipLaunchKernelGGL(frontend_kernel<MELW>, grid, dim3(256), 0, st, wave, B, samples, Fc, hann, mel_start, mel_len, mel_wt,
                               bn_scale, bn_shift, melbn);
    } else if (max_band <= 16)
        hipLaunchKernelGGL(frontend_rfft_kernel<16>, dim3((unsigned)(fblocks < 768 ? fblocks : 768)), dim3(256), 0, st, wave, B, samples, Fc, hann, mel_start, mel_len, mel_wt,
                           bn_scale, bn_shift, melbn);
    else
        hipLaunchKernelGGL(frontend_rfft_kernel<MELW>, grid, dim3(256), 0, st, wave, B, samples, Fc, hann, mel_start, mel_len, mel_wt,
                           bn_scale, bn_shift, melbn);
    WISE_LAUNCH_CHECK("htsat frontend_kernel");
    return WISE_OK;
}
}  // namespace htsat
}  // namespace wise
