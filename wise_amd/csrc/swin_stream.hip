// HP-1 audio: norm1 + QKV projection + window attention of a Swin block of MS-CLAP's HTSAT (stages 2 and 3) as ONE kernel.
//
//   o[rows of a window] = softmax( (q k^T) / sqrt(24) + relative-position bias (+ shift mask) ) v,   q | k | v = LN1(x) W_qkv^T + b
//
// behind `self.clap.get_audio_embeddings(...)` (src/feature/microsoft_clap.py:49-50; msclap HTSAT WindowAttention inside
// SwinTransformerBlock).  The three-kernel form (LayerNorm, QKV GEMM, swin_attention_kernel) writes the normalised rows and the
// 3C-wide qkv rows to HBM and reads them back (stage 3: 2 x 25 + 2 x 75 MB per block, stage 2: 2 x 50 + 2 x 151 MB at 128
// clips) — after the one-kernel MLP (mlp_stream.hip) the largest remaining item of the tower's excess traffic, and 39 % of
// its time.  Here a workgroup owns TWO windows (4 waves, one per SIMD; C = 192: FOUR windows, 8 waves — one wave's softmax under
// the other's MFMAs: stage 2 one batch at a time -4 %); a wave = 32 of a window's 64 tokens:
//   prologue  the wave gathers its rows of x by the window / cyclic-shift mapping, normalises them (two-pass statistics in
//             registers) and keeps them as MFMA operand fragments for the whole kernel (in the accumulator file: 96 registers);
//   steps     the weights arrive as a stream of 48-column blocks — the q, then k, then v columns of a PAIR of heads (head dim 24:
//             48 columns are three 16-column MFMA tiles, no padding) — through a three-slot LDS ring by LDS-DMA, fragments
//             LDS -> registers through a ring of registers, exactly as in mlp_stream.hip;
//   per pair  q stays in registers as packed bf16; k goes to LDS as ready-made MFMA operands (both waves of a window need all 64
//             keys), v as a row-major image read back transposed (ds_read_b64_tr_b16); then S^T = K Q^T per head (the 24 head
//             dims of a head are picked out of the pair's three tiles by lane selects, 8 zero slots fill the K = 32 MFMA),
//             bias, mask, softmax and O^T = V^T P^T statement for statement as swin_attention_kernel (htsat.hip), and the wave
//             writes its 32 rows x 24 columns of o.
// The packer stores W_qkv as that stream in the qkv slot (same size) and the bias in step order (htsat.py: swin_qkv_stream).
// Roofline: MFMA for the projection (2 * 3C * C flop per row), HBM for x in / o out (4 + 2 bytes per element).
#include <utility>
#include "gemm_w4.h"
#include "transformer.h"

namespace wise {
namespace swin_stream {

using w4::lds_cptr;
using w4::lds_void;
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
using short4v = __attribute__((ext_vector_type(4))) short;

// accumulate with the B operand (the rows' fragments) in the accumulator file.  s_nop 1 in the text: the compiler may place a
// register copy of its own (v_accvgpr_write / _read, a reloaded spill) right in front of the statement and does not know the
// statement is an MFMA that reads it (mlp_stream.hip's wrong rows); two idle issue cycles in front of a 16-cycle MFMA cost nothing
__device__ __forceinline__ void mfma_zero_ba(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_ba(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
template <int... I, typename Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& fn) { (fn(std::integral_constant<int, I>{}), ...); }
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) { static_for_impl(std::make_integer_sequence<int, N>{}, fn); }

constexpr int VRS = 144;     // bytes per key row of the v image (48 head dims of a pair + room for the second head's 16-wide tiles)

// NWIN windows per workgroup, two waves each (NWIN = 2: one wave per SIMD; NWIN = 4: two per SIMD — one wave's softmax runs under
// the other's MFMAs, and a step's weight fragments serve twice the rows)
template <int C, int NWIN>
__global__ __launch_bounds__(128 * NWIN, 1) void swin_qkv_attn_kernel(const float* __restrict__ x, const float* __restrict__ lnw,
                                                               const float* __restrict__ lnb, float eps,
                                                               const bf16_t* __restrict__ ws, const float* __restrict__ bq,
                                                               const float* __restrict__ relb, bf16_t* __restrict__ o,
                                                               int H, int shift) {
    using namespace w4;
    constexpr int HEADS = C / 24, NP = HEADS / 2, NSTEP = 3 * NP, KS = C / 32;
    constexpr int NF = 3 * KS, STEP = NF * 1024;             // fragments (1 KiB) and bytes of a step: 48 weight rows x C
    constexpr int WAVES = 2 * NWIN;
    constexpr int PW = (NF + WAVES - 1) / WAVES, NW = PW + 1; // DMAs per wave and step
    constexpr int BSL = WAVES * 256;                          // a bias slot: every wave's own 256-byte copy
    constexpr int RB = 3 * STEP, KV = RB + 3 * BSL;           // bias ring, then the k / v images
    constexpr int KIMG = 8 * 1024, VIMG = 64 * VRS, WIN = KIMG + VIMG;   // per window: [head of pair][key tile][1 KiB], [key][VRS]
    static_assert(KV + NWIN * WIN <= 160 * 1024 && HEADS % 2 == 0, "shape");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int wsel = wave >> 1, half = wave & 1;              // which of the workgroup's two windows, which half of its tokens
    const int W = H, nwx = W >> 3, nwin = (H >> 3) * nwx;
    const long long wing = (long long)blockIdx.x * NWIN + wsel;
    const int b = (int)(wing / nwin), win = (int)(wing % nwin);
    const int wy = win / nwx, wx = win % nwx;
    // token p (0..63) of this window -> row in the original layout, and its shift-region id (as swin_attention_kernel)
    auto row_of = [&](int p) {
        const int ys = wy * 8 + (p >> 3), xs = wx * 8 + (p & 7);
        int y = ys + shift, xx = xs + shift;
        if (y >= H) y -= H;
        if (xx >= W) xx -= W;
        return y * W + xx;
    };
    auto region_of = [&](int p) {
        const int ys = wy * 8 + (p >> 3), xs = wx * 8 + (p & 7);
        const int ry = (ys >= H - 8) + (ys >= H - 4), rx = (xs >= W - 8) + (xs >= W - 4);
        return ry * 3 + rx;
    };
    const size_t img0 = (size_t)b * H * W;
    size_t grow[2];                                           // global rows of the wave's two row tiles (this lane's row of each)
#pragma unroll
    for (int i = 0; i < 2; ++i) grow[i] = img0 + row_of(half * 32 + i * 16 + l15);

    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)bq, 0, 3 * C * 4, 0x00020000);   // reads past the end return 0
    // piece k (k < PW: one 1-KiB fragment; k == PW: the step's biases) of step t's request
    auto request_piece = [&](int t, int k) {
        if (k < PW) {
            const int u = (k * WAVES + wave) % NF;
            dma16(rS, (t % 3) * STEP + u * 1024, lane * 16, t * STEP + u * 1024);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(uintptr_t)(RB + (t % 3) * BSL + wave * 256), 4, lane * 4, t * 192, 0, 0);
        }
    };
    auto request = [&](int t) {
#pragma unroll
        for (int k = 0; k <= PW; ++k) request_piece(t, k);
    };

    // ---- prologue: gather, LayerNorm, operand fragments: lane (l15, g) holds LN(x)[row][32 ks + 8 g .. + 7]
    bf16x8 af[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* xr = x + grow[i] * C + g * 8;
        float4 v[KS][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            v[ks][0] = *reinterpret_cast<const float4*>(xr + ks * 32);
            v[ks][1] = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
        }
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            sm += ((v[ks][0].x + v[ks][0].y) + (v[ks][0].z + v[ks][0].w)) + ((v[ks][1].x + v[ks][1].y) + (v[ks][1].z + v[ks][1].w));
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float mean = sm * (1.0f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v[ks][t].x -= mean; v[ks][t].y -= mean; v[ks][t].z -= mean; v[ks][t].w -= mean;
                q = fmaf(v[ks][t].x, v[ks][t].x, q); q = fmaf(v[ks][t].y, v[ks][t].y, q);
                q = fmaf(v[ks][t].z, v[ks][t].z, q); q = fmaf(v[ks][t].w, v[ks][t].w, q);
            }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / (float)C) + eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 w0 = *reinterpret_cast<const float4*>(lnw + ks * 32 + g * 8), w1 = *reinterpret_cast<const float4*>(lnw + ks * 32 + g * 8 + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(lnb + ks * 32 + g * 8), c1 = *reinterpret_cast<const float4*>(lnb + ks * 32 + g * 8 + 4);
            union { unsigned u[4]; bf16x8 f; } cv;
            cv.u[0] = pack_bf16x2(fmaf(v[ks][0].x * rstd, w0.x, c0.x), fmaf(v[ks][0].y * rstd, w0.y, c0.y));
            cv.u[1] = pack_bf16x2(fmaf(v[ks][0].z * rstd, w0.z, c0.z), fmaf(v[ks][0].w * rstd, w0.w, c0.w));
            cv.u[2] = pack_bf16x2(fmaf(v[ks][1].x * rstd, w1.x, c1.x), fmaf(v[ks][1].y * rstd, w1.y, c1.y));
            cv.u[3] = pack_bf16x2(fmaf(v[ks][1].z * rstd, w1.z, c1.z), fmaf(v[ks][1].w * rstd, w1.w, c1.w));
            af[i][ks] = cv.f;
        }
    }
    request(0);
    request(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");     // step 0 has landed (step 1 may fly)
    __builtin_amdgcn_s_barrier();

    const unsigned kimg = KV + wsel * WIN, vimg = kimg + KIMG;   // this window's images
    unsigned qp[2][3][2];                                           // q of the current pair: [row tile][tile of 16 columns][2 x bf16x2]
    int regq[2] = {0, 0}, regk[4][4];
    if (shift > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) regq[i] = region_of(half * 32 + i * 16 + l15);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) regk[kt][r] = region_of(kt * 16 + g * 4 + r);
    }

    // the relative-position bias rows of a pair's two heads, this lane's (query, 4 keys) entries: asked for one step ahead of the
    // attention that adds them (read where they are used, each head's 8 loads exposed a global round trip: 16 - 32 us per launch)
    float4 relv[2][2][4];
    auto bias_request = [&](int pair) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    relv[hh][qt][kt] = *reinterpret_cast<const float4*>(relb + (size_t)(pair * 2 + hh) * 4096 + (half * 32 + qt * 16 + l15) * 64 + kt * 16 + g * 4);
    };
    // One quarter of the attention of head pair `pair` (its q in qp, its k / v in the window's images): head hh of the pair,
    // the wave's query tile qt.  A piece starts with LDS reads and ends with global stores — memory operations, which keep their
    // place among the volatile MFMA statements — so the four pieces of a pair can be written BETWEEN the MFMAs of the next
    // step's projection and stay there: the softmax's vector work runs while the matrix pipe works on those.
    auto attention_piece = [&](int pair, auto hh_c, auto qt_c) {
        constexpr int hh = decltype(hh_c)::value, qt = decltype(qt_c)::value;
        const int h = pair * 2 + hh;
        // q as the B operand: 8 slots = the head's dims out of the pair's tiles (head 0: tile 0 and the low half of tile 1, head
        // 1: tile 2 and the high half of tile 1), the rest zero — the k operands were written with the same selection
        bf16x8 qf;
        {
            union { unsigned u[4]; bf16x8 f; } cv;
            const bool mid = hh == 0 ? g < 2 : g >= 2;
            cv.u[0] = qp[qt][hh == 0 ? 0 : 2][0]; cv.u[1] = qp[qt][hh == 0 ? 0 : 2][1];
            cv.u[2] = mid ? qp[qt][1][0] : 0u; cv.u[3] = mid ? qp[qt][1][1] : 0u;
            qf = cv.f;
        }
        f32x4 sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bf16x8 kf = *reinterpret_cast<lds_frag_ptr>((lds_cptr)(uintptr_t)(kimg + (hh * 4 + kt) * 1024 + lane * 16));
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
            sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, c, 0, 0, 0);
        }
        const float scale = 0.20412414523193154f;  // 24^-0.5
        const float LOG2E = 1.4426950408889634f;
        float mx = -INFINITY;
        float s2[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float4 bb = relv[hh][qt][kt];
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv = sacc[kt][r] * scale + bv[r];
                if (shift > 0 && regk[kt][r] != regq[qt]) sv += -100.f;
                sv *= LOG2E;
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
        f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // O^T += V^T P^T over two K-steps of 32 key slots (slot order the same on both sides)
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
                // transposed read of the row-major v image: the 16 lanes of group g fetch a 4-key x 16-dim block
                const unsigned blk = vimg + (ks * 32 + g * 4 + (l15 >> 2)) * VRS + (hh * 24 + dt * 16 + (l15 & 3) * 4) * 2;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)blk);
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(blk + 16 * VRS));
                union { short8 s; bf16x8 f; } cv;
                cv.s = short8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cv.f, pf, oacc[dt], 0, 0, 0);
            }
        }
        // o[row][24 h + d], d = 16 dt + 4 g + r < 24
        bf16_t* orow = o + grow[qt] * C + h * 24;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int dh = dt * 16 + g * 4;
            if (dh < 24) {
                uint2 pk2;
                pk2.x = pack_bf16x2(oacc[dt][0] * linv, oacc[dt][1] * linv);
                pk2.y = pack_bf16x2(oacc[dt][2] * linv, oacc[dt][3] * linv);
                *reinterpret_cast<uint2*>(orow + dh) = pk2;
            }
        }
    };
    auto attention = [&](int pair) {
        attention_piece(pair, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        attention_piece(pair, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        attention_piece(pair, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        attention_piece(pair, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    };

    for (int s = 0; s < NSTEP; ++s) {
        const int pair = s / 3, typ = s - pair * 3;
        const bool attn = typ == 0 && s > 0;                      // the previous pair's attention rides in this step (its k / v images are complete)
        if (typ == 2) bias_request(pair);                         // used by the attention of `pair` in the next iteration
        const bool req = s + 2 < NSTEP;                           // step s + 2's request rides between this step's MFMAs, a piece at a time
                                                                  // (a burst of PW + 1 LDS-DMA instructions costs a lone wave ~100 cycles each)
        // ---- the step's 48 columns: tiles [16 j ..][rows] over K = C, fragments through a ring of registers
        const unsigned sb = (s % 3) * STEP;
        const lds_cptr base = (lds_cptr)(uintptr_t)(sb + lane * 16);
        f32x4 a1[2][3];
        {
            constexpr int D = 6;
            auto frag = [&](int f) { return *reinterpret_cast<lds_frag_ptr>(base + ((f % 3) * KS + f / 3) * 1024); };   // walk (ks, j)
            bf16x8 wr[D];
            static_for<D - 1>([&](auto d_c) { constexpr int d = decltype(d_c)::value; wr[d] = frag(d); });
            static_for<NF>([&](auto f_c) {
                constexpr int f = decltype(f_c)::value, j = f % 3, ks = f / 3;
                if constexpr (f + D - 1 < NF) wr[(f + D - 1) % D] = frag(f + D - 1);
                static_for<2>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    if constexpr (ks == 0) mfma_zero_ba(a1[i][j], wr[f % D], af[i][0]);
                    else mfma_ba(a1[i][j], wr[f % D], af[i][ks]);
                });
                if constexpr (f % (NF / (PW + 1)) == 0 && f / (NF / (PW + 1)) <= PW) {
                    if (req) request_piece(s + 2, f / (NF / (PW + 1)));
                }
                if constexpr ((f + 1) % (NF / 4) == 0 && (f + 1) / (NF / 4) <= 4) {
                    constexpr int piece = (f + 1) / (NF / 4) - 1;
                    if (attn) attention_piece(pair - 1, std::integral_constant<int, piece / 2>{}, std::integral_constant<int, piece % 2>{});
                }
            });
        }
        f32x4 bv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            bv[j] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
                (lds_cptr)(uintptr_t)(RB + (s % 3) * BSL + wave * 256 + (j * 16 + g * 4) * 4));
        mfma_retire();
        unsigned pk[2][3][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                pin_v(a1[i][j]);
                pk[i][j][0] = pack_bf16x2(a1[i][j][0] + bv[j][0], a1[i][j][1] + bv[j][1]);
                pk[i][j][1] = pack_bf16x2(a1[i][j][2] + bv[j][2], a1[i][j][3] + bv[j][3]);
            }
        if (typ == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { qp[i][j][0] = pk[i][j][0]; qp[i][j][1] = pk[i][j][1]; }
        } else if (typ == 1) {
            // k as ready-made A operands: [head of pair][key tile = 2 half + i][lane]: the same slot selection as q
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool mid = hh == 0 ? g < 2 : g >= 2;
                    w4::u32x4_t v4 = {pk[i][hh == 0 ? 0 : 2][0], pk[i][hh == 0 ? 0 : 2][1], mid ? pk[i][1][0] : 0u, mid ? pk[i][1][1] : 0u};
                    W4_LDS(w4::u32x4_t, kimg + (hh * 4 + half * 2 + i) * 1024 + lane * 16) = v4;
                }
        } else {
            // v row-major: key = 32 half + 16 i + l15, 48 dims of the pair; the lane's 4 consecutive dims of each tile
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    W4_LDS(w4::u32x2_t, vimg + (half * 32 + i * 16 + l15) * VRS + (j * 16 + g * 4) * 2) = w4::u32x2_t{pk[i][j][0], pk[i][j][1]};
        }
        // every wave is through with slot sb and with its image writes; the DMAs of step s + 1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 2 < NSTEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    attention(NP - 1);
}

static PerDeviceOnce g_once;

}  // namespace swin_stream

bool swin_qkv_attn_ok(int C, int H) { return (C == 192 || C == 384) && H >= 8 && H % 8 == 0; }

// ws: W_qkv [3C, C] as the kernel's stream, bq: the qkv bias in step order (wise_amd/feature/htsat.py:swin_qkv_stream)
int swin_qkv_attn(const float* x, const float* lnw, const float* lnb, float eps, const bf16_t* ws, const float* bq,
                  const float* relb, bf16_t* o, int B, int H, int C, int shift, hipStream_t st) {
    using namespace swin_stream;
    WISE_CHECK_ARG(x && lnw && lnb && ws && bq && relb && o && B > 0, "swin_qkv_attn: bad argument");
    WISE_CHECK_ARG(swin_qkv_attn_ok(C, H) && (shift == 0 || shift == 4), "swin_qkv_attn: C = 192 or 384, H %% 8 == 0, shift 0 or 4 (C=%d, H=%d, shift=%d)", C, H, shift);
    const long long nwin = (long long)B * (H / 8) * (H / 8);
    WISE_CHECK_ARG(nwin % 2 == 0, "swin_qkv_attn: an even number of windows (%lld)", nwin);
    constexpr int WINB = 8 * 1024 + 64 * VRS;
    constexpr int L384 = 3 * 36 * 1024 + 3 * 4 * 256 + 2 * WINB, L192 = 3 * 18 * 1024 + 3 * 4 * 256 + 2 * WINB, L192W = 3 * 18 * 1024 + 3 * 8 * 256 + 4 * WINB;
    g_once([&] {
        raise_lds_limit(reinterpret_cast<const void*>(swin_qkv_attn_kernel<384, 2>), L384);
        raise_lds_limit(reinterpret_cast<const void*>(swin_qkv_attn_kernel<192, 2>), L192);
        raise_lds_limit(reinterpret_cast<const void*>(swin_qkv_attn_kernel<192, 4>), L192W);
    });
    ProfScope prof(PROF_GEMM, 2.0 * (double)B * H * H * C * 3.0 * C, st);
    if (C == 384)
        hipLaunchKernelGGL((swin_qkv_attn_kernel<384, 2>), dim3((unsigned)(nwin / 2)), dim3(256), (size_t)L384, st, x, lnw, lnb, eps, ws, bq, relb, o, H, shift);
    else if (nwin % 4 == 0)
        hipLaunchKernelGGL((swin_qkv_attn_kernel<192, 4>), dim3((unsigned)(nwin / 4)), dim3(512), (size_t)L192W, st, x, lnw, lnb, eps, ws, bq, relb, o, H, shift);
    else
        hipLaunchKernelGGL((swin_qkv_attn_kernel<192, 2>), dim3((unsigned)(nwin / 2)), dim3(256), (size_t)L192, st, x, lnw, lnb, eps, ws, bq, relb, o, H, shift);
    WISE_LAUNCH_CHECK("swin_qkv_attn_kernel");
    return WISE_OK;
}

}  // namespace wise

extern "C" int wise_swin_qkv_attn(const float* x, const float* lnw, const float* lnb, float eps, const uint16_t* ws, const float* bq,
                                  const float* relb, uint16_t* o, int B, int H, int C, int shift, void* stream) {
    return wise::swin_qkv_attn(x, lnw, lnb, eps, ws, bq, relb, o, B, H, C, shift, (hipStream_t)stream);
}
