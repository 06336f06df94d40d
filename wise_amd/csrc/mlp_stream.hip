// HP-1 audio: the MLP half of a Swin block of MS-CLAP's HTSAT (stages 2 and 3) as ONE kernel.
//
//   x += fc2( GELU( fc1( h ) ) ),   h = LayerNorm(x) as bf16 [M, C],  fc1: C -> 4C,  fc2: 4C -> C
//
// behind `self.clap.get_audio_embeddings(...)` (src/feature/microsoft_clap.py:49-50; msclap HTSAT SwinTransformerBlock.mlp).
// The two-GEMM form writes the 4C-wide hidden activations to HBM as bf16 and reads them back: 2 x 100 MB per block of stage
// 3 and 2 x 200 MB per block of stage 2 at 128 clips — the largest item of the tower's excess HBM traffic (VERDICT r03 item
// 4).  Here a wave keeps its rows' h fragments in REGISTERS for the whole kernel, walks the hidden dimension in steps of
// 32, and the hidden values never leave the register file:
//
//   step s:  H_s [rows, 32]  = h [rows, C] @ W1[32 s .. 32 s + 31, :]^T        (phase 1, 2 * C/32 MFMAs per row tile)
//            G_s             = bf16( GELU( H_s + b1 ) )                          (the accumulators ARE the next MFMA's operand)
//            acc [rows, C]  += G_s @ W2[:, 32 s .. 32 s + 31]^T                 (phase 2, C/16 MFMAs per row tile)
//
// software-pipelined over the steps (iteration s: phase 1 of step s + 1, the GELU stages of step s between the MFMAs, phase 2
// of step s - 1), output accumulators in AGPRs.  Every wave needs every weight fragment: the weights travel global -> LDS
// once per workgroup (LDS-DMA into three-slot rings, what an iteration requests is used two iterations later) and LDS ->
// registers once per wave (ds_read_b128 into a ring of fragment registers, each fragment feeding RT MFMAs).  The packer stores
// fc1 / fc2 as ONE stream in the order the kernel consumes it (include/wise_hip.h, wise_mlp_stream): a step is C/8 KiB
// contiguous, a fragment 1 KiB with lane l's 16 bytes at 16 l — the DMA is lane-linear, the reads conflict-free, no swizzle.
//   C = 192: four waves x 2 row tiles (128 rows per workgroup); C = 384: eight waves x 1 row tile.  Each instantiation stays
//   inside the register file it is given WITHOUT compiler spills into the other half of the file: a build of the C = 192 form
//   for two workgroups per CU (256 registers per wave: 24 values parked in AGPRs by the compiler, its copies next to the
//   inline-assembly MFMAs whose hazards it does not know) returned garbage rows; tools/kernel_resources.py must show
//   a96 / a192 and no scratch for these kernels.
//
// Measured (profiles/r04_mlp_stream_study.txt; 128 clips): stage 2 (M = 131072) 174 -> 114 us per block, stage 3 (M = 32768)
// 102 -> 97 us; the tower one batch at a time 3.55 -> 3.40 ms, with two batches in flight 3.33 -> 3.21 ms (38.4 k -> 40.0 k
// clips/s): the default (wise_htsat_forward2 flags bit 1).  What bounds it is in the study: a lone wave's instruction stream
// is additive here (MFMA ~22 cycles each + 16 per fragment read + the GELU's ~72 per value + 60 - 180 per DMA), and with two
// waves per SIMD the fragment reads (one per MFMA at one row tile per wave) fill the LDS pipe.
// Roofline: MFMA (HBM: h read once, x read and written once: 10 bytes per element instead of 26).
#include <utility>
#include "gemm_w4.h"
#include "transformer.h"

namespace wise {

using w4::lds_cptr;
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;

// first MFMA of a chain: C = 0 (no zero-initialised accumulator registers)
__device__ __forceinline__ void mfma16v_zero(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}

template <int... I, typename Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& fn) { (fn(std::integral_constant<int, I>{}), ...); }
// fn(integral_constant<int, 0>) ... fn(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) { static_for_impl(std::make_integer_sequence<int, N>{}, fn); }
// chunk k of nch sits right behind MFMA (k + 1) * nm / (nch + 1) of nm (evenly spread, none in front of the first): which one behind m?
constexpr int chunk_at(int m, int nm, int nch) {
    for (int k = 0; k < nch; ++k)
        if (nm > 0 && (k + 1) * nm / (nch + 1) == m) return k;
    return -1;
}

// (An s_nop 1 in front of every MFMA statement — swin_stream.hip's guard against a compiler register copy placed right before a
//  statement it does not know to be an MFMA — was measured on the 8-wave form: 97.5 -> 106 us per launch, the LDS-bound loop does
//  not hide it.  These kernels instead stay inside their half of the register file: tests/test_kernel_resources_cpu.py holds them to
//  their accumulators in the accumulator file and no scratch, tests/test_gpu_mlp_stream.py to the float64 reference.)
// phase-1 accumulators in the accumulator file (two waves per SIMD with two row tiles each: 128 + 128 registers per wave, and
// the arithmetic needs the VGPR half): the GELU's first stage copies a tile out (the compiler's own v_accvgpr_read)
__device__ __forceinline__ void mfma16a_zero(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "v"(b));
}

// One group of four hidden values (an accumulator tile's four registers of a lane) through bias + GELU in six stages; the
// stages of several groups are issued BETWEEN the MFMAs of the neighbouring steps (a lone wave per SIMD has nobody else to
// fill the VALU's slots while the matrix pipe runs, and nobody to fill the matrix pipe while it evaluates 16 - 32 GELUs).
struct GeluGroup {
    f32x4 h, u;
};
template <int STAGE>
__device__ __forceinline__ void gelu_stage(GeluGroup& q, const f32x4& a, const f32x4& b) {
    if constexpr (STAGE == 0) {
        q.h = f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
    } else if constexpr (STAGE == 1) {
        q.u = f32x4{fminf(q.h[0] * q.h[0], 64.f), fminf(q.h[1] * q.h[1], 64.f), fminf(q.h[2] * q.h[2], 64.f), fminf(q.h[3] * q.h[3], 64.f)};
    } else if constexpr (STAGE == 2) {       // act_gelu's odd polynomial (gemm_shared.h), the same coefficients
#pragma unroll
        for (int r = 0; r < 4; ++r) q.u[r] = fmaf(q.u[r], fmaf(q.u[r], 0.0010142630198970437f, -0.10677572339773178f), -2.301121234893799f);
    } else if constexpr (STAGE == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) q.u[r] = __builtin_amdgcn_exp2f(q.h[r] * q.u[r]);
    } else if constexpr (STAGE == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) q.u[r] = __builtin_amdgcn_rcpf(1.f + q.u[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) q.h[r] *= q.u[r];
    }
}

// LN: the kernel reads the fp32 rows themselves and applies norm2 on the way into its operand registers (h, lnw, lnb: the
// LayerNorm's weight and bias; two-pass statistics in registers, the four lanes that share a row exchanging two sums) — the
// LayerNorm launch, its write of h and this kernel's read of it are gone.  !LN: h = the LayerNorm'd rows as bf16.
template <int C, int RT, int WAVES, bool LN>
__global__ __launch_bounds__(64 * WAVES, 1) void mlp_stream_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ ws,
                                                            const float* __restrict__ b1, const float* __restrict__ b2,
                                                            float* x, const float* __restrict__ lnw,
                                                            const float* __restrict__ lnb, float eps) {
    using namespace w4;
    constexpr int KS = C / 32, NJ2 = C / 16, F = 4 * C, NSTEP = F / 32;
    constexpr int W1F = 2 * KS, NF = W1F + NJ2;          // 1-KiB fragments of a step in the stream: fc1's, then fc2's
    constexpr int STEP = NF * 1024;
    constexpr int W1B = W1F * 1024, W2B = NJ2 * 1024;    // ring slots: three of each, and three of 1 KiB for the fc1 biases
    constexpr int BSL = WAVES * 256;                      // a bias slot: every wave's own 256-byte copy
    constexpr int R1 = 0, R2 = 3 * W1B, RB = R2 + 3 * W2B;
    // DMAs per wave and iteration (where the fragments do not divide among the waves, the last round wraps: a few fragments
    // are fetched twice, same bytes to the same place)
    constexpr int P1W = (W1F + WAVES - 1) / WAVES, P2W = (NJ2 + WAVES - 1) / WAVES, NITER = P1W + P2W + 1;
    static_assert(RB + 3 * BSL <= 160 * 1024 && RT * NJ2 <= 64, "shape");
    constexpr bool A1A = WAVES == 8 && RT == 2;          // phase-1 accumulators in the accumulator file
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const size_t row0 = ((size_t)blockIdx.x * WAVES + wave) * (RT * 16);

    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, F * 4, 0x00020000);   // reads past the end return 0
    auto req_w1 = [&](int t) {      // fc1 fragments of step t -> slot t % 3 of ring 1
        const unsigned slot = R1 + (t % 3) * W1B;
#pragma unroll
        for (int k = 0; k < P1W; ++k) { const int u = (k * WAVES + wave) % W1F; dma16(rS, slot + u * 1024, lane * 16, t * STEP + u * 1024); }
    };
    auto req_w2 = [&](int t) {
        const unsigned slot = R2 + (t % 3) * W2B;
#pragma unroll
        for (int k = 0; k < P2W; ++k) { const int u = (k * WAVES + wave) % NJ2; dma16(rS, slot + u * 1024, lane * 16, t * STEP + W1B + u * 1024); }
    };
    auto req_b = [&](int t) {       // the step's 32 fc1 biases (a wave's own 256-byte copy: 64 lanes x 4 bytes)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(uintptr_t)(RB + (t % 3) * BSL + wave * 256), 4, lane * 4, t * 128, 0, 0);
    };

    // the wave's rows of h as MFMA operands: lane (l15, g) holds h[row0 + 16 i + l15][32 ks + 8 g .. + 7]
    bf16x8 af[RT][KS];
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float* xr = x + (row0 + i * 16 + l15) * C + g * 8;
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
    } else {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                af[i][ks] = *reinterpret_cast<const bf16x8*>(h + (row0 + i * 16 + l15) * C + ks * 32 + g * 8);
    }
    req_w1(0); req_w1(1); req_b(0);

    f32x4 acc[RT][NJ2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NJ2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the steps, one barrier per iteration.  Iteration s runs
    //     part A:  phase 1 of step s + 1 (MFMAs)   with the GELU stages of step s, row tiles 0 .. RT/2 - 1, between them
    //     part B:  phase 2 of step s - 1 (MFMAs)   with the GELU stages of step s, row tiles RT/2 .. RT - 1
    // and requests fc1(s + 3), fc2(s + 1), bias(s + 2): what an iteration requests has landed by the top of iteration s + 2
    // (the counted wait in front of a barrier leaves only the iteration's own DMAs in flight).
    f32x4 a1[2][RT][2];      // phase-1 accumulators of steps of either parity
    bf16x8 hf[2][RT];        // GELU outputs as phase-2 operands, either parity
    auto iteration = [&](int s, auto par_c, auto p1_c, auto ge_c, auto p2_c) {
        constexpr int PAR = decltype(par_c)::value;              // s & 1
        constexpr bool DO_P1 = decltype(p1_c)::value, DO_G = decltype(ge_c)::value, DO_P2 = decltype(p2_c)::value;
        if (s + 3 < NSTEP) req_w1(s + 3);
        if (s + 1 < NSTEP) req_w2(s + 1);
        if (s + 2 < NSTEP) req_b(s + 2);
        constexpr int NG = RT;                                   // GELU groups per half: the step's 2 RT (row tile, 16 hidden) tiles in two halves
        constexpr int NCH = 6 * NG;                              // stage-major chunks of a half
        GeluGroup q[2 * NG];
        f32x4 bv[2];
        if constexpr (DO_G) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bv[j] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
                    (lds_cptr)(uintptr_t)(RB + (s % 3) * BSL + wave * 256 + (j * 16 + g * 4) * 4));
        }
        auto chunk = [&](auto c_c, auto half_c) {
            constexpr int c = decltype(c_c)::value;
            constexpr int stage = c / NG, grp = decltype(half_c)::value * NG + c % NG, i = grp / 2, j = grp % 2;
            // empty volatile statements on either side: the stage can be computed neither earlier nor later than HERE in the
            // order of the volatile statements, i.e. between the MFMAs it is written between (plain arithmetic would all be
            // hoisted to the top of the iteration: it depends on nothing the MFMAs produce)
            if constexpr (stage == 0) { if constexpr (A1A) asm volatile("" : "+a"(a1[PAR][i][j])); else asm volatile("" : "+v"(a1[PAR][i][j])); }
            else asm volatile("" : "+v"(q[grp].h), "+v"(q[grp].u));
            gelu_stage<stage>(q[grp], a1[PAR][i][j], bv[j]);
            if constexpr (stage == 0) asm volatile("" : "+v"(q[grp].h));
            else asm volatile("" : "+v"(q[grp].h), "+v"(q[grp].u));
            if constexpr (stage == 5 && j == 1) {
                union { unsigned u[4]; bf16x8 f; } cv;
                cv.u[0] = pack_bf16x2(q[grp - 1].h[0], q[grp - 1].h[1]); cv.u[1] = pack_bf16x2(q[grp - 1].h[2], q[grp - 1].h[3]);
                cv.u[2] = pack_bf16x2(q[grp].h[0], q[grp].h[1]); cv.u[3] = pack_bf16x2(q[grp].h[2], q[grp].h[3]);
                hf[PAR][i] = cv.f;
            }
        };
        // ---- the iteration's MFMAs in one stream: phase 1 of step s + 1 (fragment f < FA: fc1's (j, ks)), then phase 2 of step
        //      s - 1 (fc2's jn).  A fragment feeds RT MFMAs (32 - 64 cycles of the matrix pipe), an LDS read takes longer than
        //      that to come back: the fragments are read D - 1 ahead into a ring of registers.  The GELU chunks of step s go
        //      between the MFMAs, evenly spread (none in front of the first MFMA).
        {
            constexpr int FA = DO_P1 ? W1F : 0, FB = DO_P2 ? NJ2 : 0, NFR = FA + FB, NMT = NFR * RT;
            constexpr int D = (WAVES == 8 && RT == 2) ? 4 : 6, NCHT = DO_G ? 2 * NCH : 0;
            const lds_cptr base1 = (lds_cptr)(uintptr_t)(R1 + ((s + 1) % 3) * W1B + lane * 16);
            const lds_cptr base2 = (lds_cptr)(uintptr_t)(R2 + ((s + 2) % 3) * W2B + lane * 16);      // step s - 1
            // phase 1 walks (ks, j): the four accumulator tiles of a step take turns, a dependent MFMA is four issues away
            auto frag = [&](int f) {
                return *reinterpret_cast<lds_frag_ptr>(f < FA ? base1 + ((f % 2) * KS + f / 2) * 1024 : base2 + (f - FA) * 1024);
            };
            static_assert(NCHT <= NMT || NMT == 0, "at most one GELU chunk behind an MFMA");
            auto chunks_behind = [&](auto m_c) {      // the GELU chunk whose place is right behind MFMA m, if any
                constexpr int k = chunk_at(decltype(m_c)::value, NMT, NCHT);
                if constexpr (k >= 0) {
                    chunk(std::integral_constant<int, k % NCH>{}, std::integral_constant<int, k / NCH>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            bf16x8 wr[D];
            static_for<D - 1>([&](auto d_c) {
                constexpr int d = decltype(d_c)::value;
                if constexpr (d < NFR) wr[d] = frag(d);
            });
            if constexpr (NMT == 0)
                static_for<NCHT>([&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    chunk(std::integral_constant<int, k % NCH>{}, std::integral_constant<int, k / NCH>{});
                });
            static_for<NFR>([&](auto f_c) {
                constexpr int f = decltype(f_c)::value;
                if constexpr (f + D - 1 < NFR) wr[(f + D - 1) % D] = frag(f + D - 1);
                static_for<RT>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    if constexpr (f < FA) {
                        constexpr int j = f % 2, ks = f / 2;
                        if constexpr (A1A) {
                            if constexpr (ks == 0) mfma16a_zero(a1[PAR ^ 1][i][j], wr[f % D], af[i][0]);
                            else mfma16a(a1[PAR ^ 1][i][j], wr[f % D], af[i][ks]);
                        } else {
                            if constexpr (ks == 0) mfma16v_zero(a1[PAR ^ 1][i][j], wr[f % D], af[i][0]);
                            else mfma16v(a1[PAR ^ 1][i][j], wr[f % D], af[i][ks]);
                        }
                    } else {
                        mfma16a(acc[i][f - FA], wr[f % D], hf[PAR ^ 1][i]);
                    }
                    chunks_behind(std::integral_constant<int, f * RT + i>{});
                });
            });
        }
        // every wave is through with the slots this iteration read; what earlier iterations requested has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 3 < NSTEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NITER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    using T = std::true_type; using Fa = std::false_type;
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    iteration(-1, P1{}, T{}, Fa{}, Fa{});
    iteration(0, P0{}, T{}, T{}, Fa{});
    for (int s = 1; s < NSTEP - 1; s += 2) {       // NSTEP is even: s = 1 .. NSTEP - 2 in pairs
        iteration(s, P1{}, T{}, T{}, T{});
        iteration(s + 1, P0{}, T{}, T{}, T{});
    }
    iteration(NSTEP - 1, P1{}, Fa{}, T{}, T{});
    iteration(NSTEP, P0{}, Fa{}, Fa{}, T{});
    mfma_retire();
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NJ2; ++j) pin_a(acc[i][j]);

    // ---- epilogue: x += acc + b2.  A lane holds 4 consecutive columns of row 16 i + l15: 16-byte accesses, 64 bytes per row
    //      and instruction; a row tile's loads all in flight before its first add (the h fragments' registers are free).
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        float* xr = x + (row0 + i * 16 + l15) * C + g * 4;
        float4 r[NJ2];
#pragma unroll
        for (int j = 0; j < NJ2; ++j) r[j] = *reinterpret_cast<const float4*>(xr + j * 16);
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            const float4 b = *reinterpret_cast<const float4*>(b2 + j * 16 + g * 4);
            const f32x4 a = acc[i][j];
            r[j].x += a[0] + b.x; r[j].y += a[1] + b.y; r[j].z += a[2] + b.z; r[j].w += a[3] + b.w;
            *reinterpret_cast<float4*>(xr + j * 16) = r[j];
        }
    }
}

static PerDeviceOnce g_mlp_stream_once;

bool mlp_stream_ok(int M, int C) { return (C == 384 || C == 192) && M > 0 && M % 128 == 0; }

// ws: fc1 [4C, C] and fc2 [C, 4C] as one stream (wise_hip.h wise_mlp_stream; wise_amd/feature/htsat.py:mlp_stream_weights)
// h != null: the LayerNorm'd rows as bf16.  h == null: lnw / lnb / eps given, the kernel normalises the fp32 rows itself.
int mlp_stream(const bf16_t* h, const bf16_t* ws, const float* b1, const float* b2, float* x, int M, int C, hipStream_t st,
               const float* lnw, const float* lnb, float eps) {
    WISE_CHECK_ARG(ws && b1 && b2 && x && (h || (lnw && lnb)), "mlp_stream: null pointer");
    WISE_CHECK_ARG(mlp_stream_ok(M, C), "mlp_stream: C = 384 or 192, M %% 128 == 0 (M=%d, C=%d)", M, C);
    constexpr int L384 = 3 * (2 * 12 + 24) * 1024 + 3 * 8 * 256, L192 = 3 * (2 * 6 + 12) * 1024 + 3 * 8 * 256;
    g_mlp_stream_once([&] {
        raise_lds_limit(reinterpret_cast<const void*>(mlp_stream_kernel<384, 1, 8, false>), L384);
        raise_lds_limit(reinterpret_cast<const void*>(mlp_stream_kernel<192, 2, 4, false>), L192);
        raise_lds_limit(reinterpret_cast<const void*>(mlp_stream_kernel<384, 1, 8, true>), L384);
        raise_lds_limit(reinterpret_cast<const void*>(mlp_stream_kernel<192, 2, 4, true>), L192);
    });
    ProfScope prof(PROF_GEMM, 2.0 * (double)M * C * 4.0 * C * 2.0, st);
    const dim3 grid((unsigned)(M / 128));
    if (C == 384 && h)
        hipLaunchKernelGGL((mlp_stream_kernel<384, 1, 8, false>), grid, dim3(512), (size_t)L384, st, h, ws, b1, b2, x, lnw, lnb, eps);
    else if (C == 384)
        hipLaunchKernelGGL((mlp_stream_kernel<384, 1, 8, true>), grid, dim3(512), (size_t)L384, st, h, ws, b1, b2, x, lnw, lnb, eps);
    else if (h)
        hipLaunchKernelGGL((mlp_stream_kernel<192, 2, 4, false>), grid, dim3(256), (size_t)L192, st, h, ws, b1, b2, x, lnw, lnb, eps);
    else
        hipLaunchKernelGGL((mlp_stream_kernel<192, 2, 4, true>), grid, dim3(256), (size_t)L192, st, h, ws, b1, b2, x, lnw, lnb, eps);
    WISE_LAUNCH_CHECK("mlp_stream_kernel");
    return WISE_OK;
}

}  // namespace wise

extern "C" int wise_mlp_stream(const uint16_t* h, const uint16_t* ws, const float* b1, const float* b2, float* x, int M, int C,
                               void* stream) {
    WISE_CHECK_ARG(h, "mlp_stream: null pointer");
    return wise::mlp_stream(h, ws, b1, b2, x, M, C, (hipStream_t)stream, nullptr, nullptr, 0.f);
}
extern "C" int wise_mlp_stream_ln(const float* lnw, const float* lnb, float eps, const uint16_t* ws, const float* b1, const float* b2,
                                  float* x, int M, int C, void* stream) {
    WISE_CHECK_ARG(lnw && lnb, "mlp_stream_ln: null pointer");
    return wise::mlp_stream(nullptr, ws, b1, b2, x, M, C, (hipStream_t)stream, lnw, lnb, eps);
}
