// MS-CLAP version '2022' audio encoder: PANNs Cnn14 + msclap Projection (gfx950).
//
// Stands behind  self.model.clap.audio_encoder(preprocessed_audio)[0]  of the reference's
// src/feature/microsoft_clap.py:49 when the feature id is microsoft/clap/2022/... (:20-31 accepts every key of
// msclap's CLAP.model_name; '2022' builds AudioEncoder('Cnn14', 2048, 1024)).  CPU restatement: oracle/cnn14_ref.py.
//
//   front end     htsat_frontend.hip's kernel (STFT power -> sparse mel -> dB -> folded bn0); the 2022 config differs
//                 from the 2023 one only in the filterbank (fmax 14000), which the packer builds       fp32 [B, T, 64]
//   block 1       conv 1 (1 -> 64) + conv 2 (64 -> 64) + pooling in ONE kernel (conv_block1_kernel below): the 1.57 GB tensor
//                 between the two convolutions (128 clips x 10 s) exists only as a 33 KiB image in LDS; conv 1 on the matrix
//                 cores with split-bf16 operands, conv 2's weights resident in registers                  bf16 NHWC, pooled
//   conv3x3       the other ten convolutions as implicit GEMMs on the matrix cores (gemm_bf16.hip:
//                 no im2col buffer — a K-tile is 32 or 64 channels of one tap, gathered by the LDS-DMA);
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

#include <mutex>
#include <type_traits>

#include "common.h"
#include "transformer.h"

namespace wise {
namespace cnn14 {
constexpr int N_FFT = 1024, HOP = 320, MELW = htsat::FRONT_MELW;
constexpr int NBLK = 6, EMB = 2048, OUT = 1024, MIN_FRAMES = 32;
constexpr int CH[NBLK] = {64, 128, 256, 512, 1024, 2048};

// ------------------------------------------------------------------------------------------------
// Block 1 as ONE kernel: conv 1 (1 -> 64), BatchNorm, ReLU, conv 2 (64 -> 64), BatchNorm, ReLU, 2x2 average pooling.
//
// Why: as separate kernels (a VALU first convolution + conv3x3_kernel<.., POOL> on its output; round-2 history, 0.60 + 1.25 ms at
// 128 clips x 10 s) block 1 was 22 % of the forward and neither half was bound by arithmetic.  The first convolution was
// bound by the 1.57 GB it WROTE — 2.5 TB/s sustained whether VALU or matrix cores computed it — and the second by staging that
// tensor back in: nine taps x 32 KiB of A tile per 256 rows is 13.8 GB of L2 -> LDS traffic per launch, the chip's LDS-DMA
// ceiling (K = 576 leaves nothing to amortise it over).  Here the tensor between the two convolutions exists only as a
// ring of image rows in LDS (0.94 ms for both; what bounds it now is latency per wave — two waves per SIMD, registers full):
//   tile      one POOLED row g2 = (clip, t2): 32 pooling windows = 128 conv-2 outputs x 64 channels, four waves as 2 x 2
//             (wm: 16 windows, wn: 32 channels); a workgroup owns a contiguous RUN of pooled rows and walks it in order;
//   image     conv 1's output for the four input rows 2 t2 - 1 .. 2 t2 + 2 (conv 2's neighbourhood of the two rows it
//             pools), 64 cells x 64 channels bf16 each, rows outside the clip zero (conv 2's padding).  Consecutive tiles of
//             a clip share two of their four rows, so the rows live in a RING of six slots (slot = (row + 1) mod 6): a tile
//             reads four and, meanwhile, the two rows the next tile adds are built into the other two (waves 0, 1 one row,
//             waves 2, 3 the other, half a row each) — every conv-1 row is computed once (a first form with a fresh four-row
//             image per tile built each row twice: 1.21 -> 0.94 ms).  A run's first tile and every clip's first tile build
//             all four rows.  Per 16 cells one MFMA per 16 channels with split-bf16 operands, K = 27 of 32 = x_hi w_hi +
//             x_lo w_hi + x_hi w_lo (fp32-grade products: the dropped term is 2^-16), the nine taps gathered by each lane
//             from log-mel rows that arrived by LDS-DMA during the tap loop;
//   conv 2    its 64 x 576 weights live in REGISTERS for the life of the wave (36 fragments of the wave's 32 channels), so
//             the nine-tap loop has no staging, no barrier and no LDS write: per tap 8 fragment reads of the image at rows
//             shifted by the tap and 16 MFMAs; the ring position t2 mod 3 is a compile-time parameter of the loop (three
//             copies), so every read is a per-lane base + an immediate;
//   pooling   the four members of a window are the four row tiles of one lane (as in conv3x3_kernel<POOL>): three adds.
// One barrier per tile.  A cell is 128 B with its 16-byte chunks XOR-swizzled by (cell >> 1) & 7: conv 2 reads cells
// 2 l + const (stride two), eight consecutive lanes hit eight different chunk positions.  The image has zero border
// columns (cells 0 and 65), so the tap loop needs no masking; its 72 fragment addresses are 4 per-lane bases (column
// offset -1 .. 2) plus compile-time row offsets.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short bf16_bits(float v) { return (unsigned short)f32_to_bf16(v); }
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// the 8 k-slots of lane group kq (k = 8 kq .. 8 kq + 7) of the 27-deep operand [a(9), b(9), c(9)], zero beyond
__device__ __forceinline__ bf16x8 slots27(const unsigned short (&a)[9], const unsigned short (&b)[9],
                                          const unsigned short (&c)[9], int kq) {
    unsigned short k[8];
    if (kq == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = a[e];
    } else if (kq == 1) {
        k[0] = a[8];
#pragma unroll
        for (int e = 1; e < 8; ++e) k[e] = b[e - 1];
    } else if (kq == 2) {
        k[0] = b[7]; k[1] = b[8];
#pragma unroll
        for (int e = 2; e < 8; ++e) k[e] = c[e - 2];
    } else {
        k[0] = c[6]; k[1] = c[7]; k[2] = c[8];
#pragma unroll
        for (int e = 3; e < 8; ++e) k[e] = 0;
    }
    uint4 u;
    u.x = k[0] | ((unsigned)k[1] << 16); u.y = k[2] | ((unsigned)k[3] << 16);
    u.z = k[4] | ((unsigned)k[5] << 16); u.w = k[6] | ((unsigned)k[7] << 16);
    return __builtin_bit_cast(bf16x8, u);
}

// one image: 4 input rows x 66 cells (columns -1 .. 64: the two border cells stay zero, conv 2's padding in f, so the
// nine-tap loop needs no masking) x 64 channels bf16
constexpr int B1_ROW = 66 * 128, B1_IMG = 6 * B1_ROW;   // a ring of six rows (four read by a tile, two being built)
#ifdef WISE_DEBUG_KNOBS
static int g_b1_ablate = 0;   // host-side: which timing-only instantiation of block 1 to launch (wise_debug_set_cnn14)
#endif
constexpr int B1_MEL = B1_IMG + 4096;          // behind the ring and conv 1's fragments: [2 buffers][4 waves][1 KiB] of log-mel rows

// 16 bytes per lane from global memory straight into LDS (base wave-uniform, lane i lands at base + 16 i)
__device__ __forceinline__ void b1_glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// byte offset of 16-byte chunk `chunk` of cell c (0 .. 65 = column c - 1) within an image row
__device__ __forceinline__ int b1_cell(int c, int chunk) { return c * 128 + ((chunk ^ ((c >> 1) & 7)) << 4); }

// ABL (timing-only instantiations, reachable from the debug library alone): 1 = images built once per workgroup, 2 = no tap loop
template <int ABL>
__global__ __launch_bounds__(256, 2) void conv_block1_kernel(const float* __restrict__ mel /*[B,T,64]*/,
                                                             const float* __restrict__ w0 /*[64][9]*/,
                                                             const float* __restrict__ s0 /*[64]*/,
                                                             const bf16_t* __restrict__ Wt /*[64][9*64]*/,
                                                             const float* __restrict__ bias2 /*[64]*/, int T, int rows2,
                                                             int chunk, bf16_t* __restrict__ out /*[rows2*32, 64]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // ring of six image rows | conv 1 fragments | log-mel rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, kq = lane >> 4;
    const int T2 = T >> 1;

    // conv 2: this wave's 32 channels x 576, as MFMA A-operand fragments, in registers
    bf16x8 wreg[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int sk = 0; sk < 2; ++sk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wreg[tap][sk][j] = *reinterpret_cast<const bf16x8*>(Wt + (size_t)(wn * 32 + j * 16 + l15) * 576 + tap * 64 +
                                                                    sk * 32 + kq * 8);
    // conv 1: channel j*16 + l15, operand [w_hi, w_hi, w_lo] (pairs with [x_hi, x_lo, x_hi]); the four fragments are the
    // same for every wave and are parked in LDS behind the images (registers are full of conv 2's weights)
    bf16x8* wf0s = reinterpret_cast<bf16x8*>(smem + B1_IMG);
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned short hi[9], lo[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float v = w0[(j * 16 + l15) * 9 + tap];
                hi[tap] = bf16_bits(v);
                lo[tap] = bf16_bits(v - bf16_val(hi[tap]));
            }
            wf0s[j * 64 + lane] = slots27(hi, hi, lo, kq);
        }
    }
    __syncthreads();

    // The three log-mel rows one conv-1 row t_in needs (t_in - 1 .. t_in + 1, clamped into the clip: rows outside it are
    // masked where they are used), by LDS-DMA into the wave's own 1 KiB, requested a tap loop ahead of their use.
    auto fetch_mel = [&](int b, int t_in, int buf) {
        int t = t_in - 1 + min(lane >> 4, 2);
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        b1_glds16(mel + ((size_t)b * T + t) * 64 + (lane & 15) * 4, smem + B1_MEL + (buf * 4 + wave) * 1024);
    };
    // conv 1 + BatchNorm + ReLU of input row t_in of clip b, cells u0*16 .. (u0 + NU)*16 - 1, into ring slot `slot`
    // (t_in outside the clip: zeros — conv 2's padding).  The wave's log-mel rows are in its buffer `buf`.
    auto build_row = [&](int t_in, int slot, int u0, int nu, int buf) {
        const bool row_in = (unsigned)t_in < (unsigned)T;          // wave-uniform
        const float* mrow = reinterpret_cast<const float*>(smem + B1_MEL + (buf * 4 + wave) * 1024) + 64;   // row t_in
        unsigned char* irow = smem + slot * B1_ROW;
        bf16x8 wf0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf0[j] = wf0s[j * 64 + lane];
        for (int u = u0; u < u0 + nu; ++u) {
            const int f = u * 16 + l15;
            bf16x8 af0 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
            if (row_in) {
                unsigned short xh[9], xl[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    const bool in = (unsigned)(t_in + dy) < (unsigned)T && (unsigned)(f + dx) < 64u;
                    const float v = in ? mrow[dy * 64 + f + dx] : 0.f;
                    xh[tap] = bf16_bits(v);
                    xl[tap] = bf16_bits(v - bf16_val(xh[tap]));
                }
                af0 = slots27(xh, xl, xh, kq);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 pk = make_uint2(0u, 0u);
                if (row_in) {
                    const float4 sh = *reinterpret_cast<const float4*>(s0 + j * 16 + kq * 4);
                    const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[j], af0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    pk.x = pack_bf16x2(fmaxf(a[0] + sh.x, 0.f), fmaxf(a[1] + sh.y, 0.f));
                    pk.y = pack_bf16x2(fmaxf(a[2] + sh.z, 0.f), fmaxf(a[3] + sh.w, 0.f));
                }
                // channels j*16 + kq*4 .. +3 of column f (cell f + 1): chunk j*2 + (kq >> 1), second half of it when kq is odd
                *reinterpret_cast<uint2*>(irow + b1_cell(f + 1, j * 2 + (kq >> 1)) + (kq & 1) * 8) = pk;
            }
        }
    };

    // the border cells of the six ring rows: zero once, never written again
    if (threadIdx.x < 96) {
        const int sl = threadIdx.x >> 4, side = (threadIdx.x >> 3) & 1, ch = threadIdx.x & 7;
        *reinterpret_cast<uint4*>(smem + sl * B1_ROW + (side ? 65 : 0) * 128 + ch * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // fragment addresses of the tap loop: column 2 f2 + e (e = -1 .. 2) -> cell 2 f2 + e + 1, k-half sk
    // (the second k-half is chunk 4 + kq = kq ^ 4: the same address with bit 6 flipped)
    int abase[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) abase[e] = b1_cell(2 * (wm * 16 + l15) + e, kq);
    // A tile's 16 bytes per lane leave for HBM at the top of the NEXT iteration: the wait in front of the row build (vmcnt
    // counts stores too) then finds them a whole tap loop old instead of stalling on stores it has just issued.  Until
    // then they sit in the wave's other log-mel buffer, each lane in its own 16 bytes — registers are full.
    auto flush = [&](int g_done, int buf) {
        const uint4 v = *reinterpret_cast<const uint4*>(smem + B1_MEL + (buf * 4 + wave) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the next fetch's LDS-DMA may land on it
        bf16_t* o = out + ((size_t)g_done * 32 + wm * 16 + l15) * 64 + wn * 32 + kq * 4;
        *reinterpret_cast<uint2*>(o) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(o + 16) = make_uint2(v.z, v.w);
    };
    // the nine-tap loop of one tile; PH = t2 mod 3 fixes the ring slots of its four rows at compile time
    f32x4 acc[4][2];
    auto taps = [&](auto ph) {
        constexpr int PH = decltype(ph)::value;
        // 18 steps (tap, k-half), the A fragments of step n + 1 requested before the MFMAs of step n
        auto load_af = [&](int step, bf16x8 (&af)[4]) {
            const int tap = step >> 1, sk = step & 1;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)            // member i of window wm*16 + l15: input cell (2 t2 + (i >> 1), 2 f2 + (i & 1))
                af[i] = *reinterpret_cast<const bf16x8*>(smem + ((2 * PH + 1 + (i >> 1) + dy) % 6) * B1_ROW +
                                                         (abase[(i & 1) + dx + 1] ^ (sk << 6)));
        };
        bf16x8 afa[4], afb[4];
        load_af(0, afa);
#pragma unroll
        for (int step = 0; step < (ABL == 2 ? 0 : 18); ++step) {
            const int tap = step >> 1, sk = step & 1;
            if (step + 1 < 18) { if (step & 1) load_af(step + 1, afa); else load_af(step + 1, afb); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[tap][sk][j], (step & 1) ? afb[i] : afa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep later steps' reads where they are: registers are full
        }
    };

    // this workgroup's pooled rows: a contiguous run, walked in order so that consecutive tiles of a clip share image rows
    const int r0 = blockIdx.x * chunk, r1 = min(rows2, r0 + chunk);
    int b = r0 / T2, t2 = r0 - b * T2, par = 0;
    bool cold = true;
    for (int g2 = r0; g2 < r1; ++g2) {
        const int ph = t2 % 3;
        if (g2 != r0) flush(g2 - 1, par);
        if (cold) {
            // all four rows of this tile, wave w row 2 t2 - 1 + w (ring slot (2 ph + w) mod 6)
            fetch_mel(b, 2 * t2 - 1 + wave, par);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                 // the previous tile's readers are done with the ring
            build_row(2 * t2 - 1 + wave, (2 * ph + wave) % 6, 0, 4, par);
            __syncthreads();
            cold = false;
        }
        // the next tile of the same clip needs two more rows, 2 t2 + 3 and 2 t2 + 4: waves 0, 1 build halves of the first,
        // waves 2, 3 of the second, into the two ring slots this tile does not read
        const bool more = t2 + 1 < T2 && g2 + 1 < r1;
        const int t_new = 2 * t2 + 3 + (wave >> 1);
        if (more) fetch_mel(b, t_new, par);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ph == 0) taps(std::integral_constant<int, 0>{});
        else if (ph == 1) taps(std::integral_constant<int, 1>{});
        else taps(std::integral_constant<int, 2>{});
        {
            uint4 pk;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 bj = *reinterpret_cast<const float4*>(bias2 + wn * 32 + j * 16 + kq * 4);
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v0 += fmaxf(acc[i][j][0] + bj.x, 0.f); v1 += fmaxf(acc[i][j][1] + bj.y, 0.f);
                    v2 += fmaxf(acc[i][j][2] + bj.z, 0.f); v3 += fmaxf(acc[i][j][3] + bj.w, 0.f);
                }
                (j ? pk.z : pk.x) = pack_bf16x2(0.25f * v0, 0.25f * v1);
                (j ? pk.w : pk.y) = pack_bf16x2(0.25f * v2, 0.25f * v3);
            }
            *reinterpret_cast<uint4*>(smem + B1_MEL + ((par ^ 1) * 4 + wave) * 1024 + lane * 16) = pk;   // this lane's own slot
        }
        if (more) {
            if (ABL != 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's log-mel rows have landed
                build_row(t_new, (2 * ph + 4 + (wave >> 1)) % 6, (wave & 1) * 2, 2, par);
            }
        } else {
            cold = true;                                     // the next tile starts a clip (or the run ends)
        }
        __syncthreads();      // the new rows are complete; every wave is done with this tile's rows
        par ^= 1;
        if (++t2 == T2) { t2 = 0; ++b; }
    }
    if (r1 > r0) flush(r1 - 1, par);
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
    // the largest tensor that reaches HBM is block 2's first convolution: B x T/2 x 32 cells x 128 channels (block 1's
    // unpooled tensors stay on chip); rows padded to the 256-row tile
    const size_t rows2 = ((size_t)B * (w.T / 2) * 32 + 255) / 256 * 256;
    size_t off = 0;
    w.zeros = off; off += 256;
    w.mel = off; off += up256((size_t)B * w.T * 64 * 4);
    w.a = off; off += up256(rows2 * 128 * 2);
    w.b = off; off += up256(rows2 * 128 * 2);
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
    {   // block 1: both convolutions and the pooling in one kernel (the tensor between them never leaves the CU)
        const int rows2 = B * (T / 2);
        const size_t lds = B1_MEL + 8192;        // six image rows + conv 1's four weight fragments + the log-mel rows in flight
        static PerDeviceOnce attr_b1;
        attr_b1([&] {
            raise_lds_limit(reinterpret_cast<const void*>(conv_block1_kernel<0>), (int)lds);
#ifdef WISE_DEBUG_KNOBS
            raise_lds_limit(reinterpret_cast<const void*>(conv_block1_kernel<1>), (int)lds);
            raise_lds_limit(reinterpret_cast<const void*>(conv_block1_kernel<2>), (int)lds);
#endif
        });
        auto kern = conv_block1_kernel<0>;
#ifdef WISE_DEBUG_KNOBS
        if (g_b1_ablate == 1) kern = conv_block1_kernel<1>;
        if (g_b1_ablate == 2) kern = conv_block1_kernel<2>;
#endif
        // two persistent workgroups per CU, each a contiguous run of pooled rows (consecutive rows of a clip share image rows)
        const int nblk = rows2 < 512 ? rows2 : 512, chunk = (rows2 + nblk - 1) / nblk;
        hipLaunchKernelGGL(kern, dim3((rows2 + chunk - 1) / chunk), dim3(256), lds, st, mel, pf + o.c0_w, pf + o.c0_b,
                           wb + o.cw[0][1], pf + o.cb[0][1], T, rows2, chunk, cur);
        WISE_LAUNCH_CHECK("cnn14 conv_block1_kernel");
        T /= 2; F /= 2;
    }
    for (int i = 1; i < NBLK; ++i) {
        const int cin = CH[i - 1], cout = CH[i];
        for (int j = 0; j < 2; ++j) {
            // the block's second convolution carries the 2x2 average pooling that follows it (blocks 2-5)
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

#ifdef WISE_DEBUG_KNOBS
extern "C" int wise_debug_set_cnn14(int block1_ablate) {
    wise::cnn14::g_b1_ablate = block1_ablate;
    return 0;
}
#endif

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
