// LAB ONLY (tools/gemm_lab.hip): a persistent GEMM with two accumulator sets whose epilogue runs between the MFMAs of the
// next tile.  It is not part of libwise_hip.so.  What it showed (MI355X, ViT-B/32 fc1 12800x3072x768, QuickGELU): 62.9 us
// against 64.6 us for the product's choice — the drain's ~960 VALU instructions per tile are not free, because one wave
// per SIMD issues one instruction per four cycles whatever its type, and an MFMA of 16 cycles leaves two to three such
// slots, which the loop's own LDS reads, DMAs and scalar work already use.  Its scalar operands reach inline-assembly
// DMAs without the compiler's hazard padding, so with SGPR spills in the function it is not safe to ship either.
#pragma once
#include "../wise_amd/csrc/gemm_w4.h"

namespace wise {

// ------------------------------------------------------------------------------------------------------------------
// Persistent form with TWO accumulator sets, 128 x 256 tiles (MI = 4, NJ = 8; 3 + 2 slots; any M % 128 == 0): the whole
// epilogue of a tile — bias, activation, rounding, transposition through a wave-private LDS image, stores — runs between
// the MFMAs of the NEXT tile, one or two vector instructions per MFMA gap.  gemm_w4p_kernel leaves the packing step
// exposed (VALU work without memory traffic: 1.9k cycles per 160 x 256 tile, 6.2k with QuickGELU because a wave alone on
// its SIMD issues a VALU instruction every 4 cycles and a transcendental every 8); here a 16-cycle MFMA gap has 8 cycles of
// the wave's issue to spare, which is two plain VALU instructions or one transcendental, and the schedule below gives every
// gap at most that.
//   * 32 accumulator tiles per set: the set being accumulated (acc) and the set being drained (accD) are the two halves of
//     the AGPR file.  At a tile's end acc is copied to accD (128 v_accvgpr_mov: ~0.5k cycles, the only exposed epilogue
//     work besides the retire wait) and the next tile's first MFMA phase starts acc from C = 0.
//   * Drain program (QuickGELU / GELU family: 20 gaps per accumulator tile; none / ReLU: 6): x = a + bias (a gap per value:
//     the AGPR read and the add), the activation's argument (two values per gap), exp2, + 1, rcp (one value per gap),
//     x * r (two per gap), the two conversions, the 8-byte LDS write.  After the eighth tile of a 16-row tile, four
//     ds_read_b128 / buffer_store pairs take it out row-major (two alternating images, so a row tile may leave while the
//     next one is being written).  640 + 24 gaps of a tile's 64 * K/64: K >= 704 with an activation, K >= 256 without.
//   * The gap a snippet belongs to is fixed with scheduling barriers; the snippets themselves are plain C++ (the compiler
//     pads the wait states behind transcendentals where a consumer would follow directly — the schedule never lets it).
// Counted waits: a step's barrier allows the DMAs of its phase A plus the stores the drain put into that phase (a
// compile-time count per unrolled step).
// ------------------------------------------------------------------------------------------------------------------
namespace w4 {
constexpr bool sigmoid_like(int mode) { return mode == EPI_QUICKGELU || mode == EPI_GELU || mode == EPI_GELU_TANH; }
constexpr int q_gaps_per_tile(int mode) { return sigmoid_like(mode) ? 20 : 6; }
constexpr int q_drain_gaps(int mode) { return 32 * q_gaps_per_tile(mode) + 24; }
// row tile i (0..3) is complete after gap 8*(i+1)*GPT - 1; its four pieces are read at +2, +6, +10, +14 and stored at
// +5, +9, +13, +17 behind that
constexpr bool q_is_piece_read(int q, int gpt, int* i, int* p) {
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k)
            if (q == 8 * (r + 1) * gpt + 2 + 4 * k) { *i = r; *p = k; return true; }
    return false;
}
constexpr bool q_is_piece_store(int q, int gpt, int* i, int* p) {
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k)
            if (q == 8 * (r + 1) * gpt + 5 + 4 * k) { *i = r; *p = k; return true; }
    return false;
}
// LDS-DMA and register loads the compiler does not see as memory operations: it then neither orders the kernel's own LDS
// reads behind "possibly aliasing" DMAs with s_waitcnt vmcnt(0) nor counts them; the kernel's counted waits do both.
// M0 <- LDS base, one wait state, the load.  (Scalar operands must come from SALU results or kernel arguments: the
// hazard recogniser does not pad a VALU-written SGPR in front of inline assembly.)
__device__ __forceinline__ void dma16_quiet(rsrc_words_t rsrc, unsigned lds_off, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_off), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void load16_quiet(f32x4& dst, rsrc_words_t rsrc, int voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ float agpr_read(float a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
__device__ __forceinline__ rsrc_words_t rsrc_words(const void* p) {
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    return rsrc_words_t{(unsigned)b, (unsigned)(b >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
}
template <class Fn, int... Q>
__device__ __forceinline__ void each_constant(Fn&& f, std::integer_sequence<int, Q...>) { (f(Q), ...); }
constexpr int q_stores_in(int q0, int q1, int gpt) {
    int n = 0, a = 0, b = 0;
    for (int q = q0; q < q1; ++q) n += q_is_piece_store(q, gpt, &a, &b) ? 1 : 0;
    return n;
}
}  // namespace w4

template <int MODE>
__global__ __launch_bounds__(256, 1) void gemm_w4q_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                          const float* __restrict__ bias, int M, int N, int K,
                                                          bf16_t* __restrict__ out) {
    using namespace w4;
    static_assert(bf16_out(MODE), "bf16 outputs");
    constexpr int MI = 4, NJ = 8, BMB = 128, BNB = 256, NM = MI * NJ /* MFMAs per phase */;
    constexpr int ASZ = BMB * 128, WSZ = BNB * 128, WBASE = 3 * ASZ, SCR = WBASE + 2 * WSZ, IMG = 16 * 272;
    constexpr int PA = BMB / 32, PW = BNB / 32;
    constexpr int GPT = q_gaps_per_tile(MODE), QD = q_drain_gaps(MODE);
    constexpr int NU = (QD + 2 * NM - 1) / (2 * NM);          // K-steps the drain spans (unrolled)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_m = M / BMB, tiles_n = N / BNB, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int T = (ntiles - (int)blockIdx.x + G - 1) / G;

    const int ldb = K * 2;
    const int voff = (lane >> 3) * ldb + (((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned fa0 = (wm * MI * 16 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fa1 = fa0 ^ 64;
    const unsigned fw0 = WBASE + (wn * NJ * 16 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fw1 = fw0 ^ 64;
    const int nk = K / 64;

    int tm, tn;
    tile_coords_v(blockIdx.x, ntiles, tiles_m, tiles_n, 8, &tm, &tn);
    int m0 = tm * BMB, n0 = tn * BNB;
    const bf16_t* tA = A + (size_t)m0 * K;          // this tile's operand rows
    const bf16_t* tW = Wt + (size_t)n0 * K;
    auto stage = [&](const bf16_t* src, unsigned slot, auto n_c) {
        const rsrc_words_t r = rsrc_words(src);
#pragma unroll
        for (int d = 0; d < decltype(n_c)::value; ++d) dma16_quiet(r, slot + (d * 4 + wave) * 1024, voff, (d * 4 + wave) * 8 * ldb);
    };
    stage(tA, 0, std::integral_constant<int, PA>{});
    stage(tW, WBASE, std::integral_constant<int, PW>{});
    stage(tA + 64, ASZ, std::integral_constant<int, PA>{});
    stage(tW + 64, WBASE + WSZ, std::integral_constant<int, PW>{});
    wait_vmcnt<PA + PW>();
    __builtin_amdgcn_s_barrier();

    f32x4 acc[MI][NJ], accD[MI][NJ];
    bf16x8 af0[MI], wf0[NJ], af1[MI], wf1[NJ];
    {
        lds_cptr pw = (lds_cptr)(uintptr_t)fw0, pa = (lds_cptr)(uintptr_t)fa0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf0[j] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + j * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + i * 2048);
    }
    unsigned oA0 = 0, oA1 = ASZ, oA2 = 2 * ASZ;
    unsigned oW0 = 0, oW1 = WSZ, oW2 = 0;

    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    const int vst = (g * N + l15 * 8) * 2;
    int sstD = 0;                                                  // ((m0 + wm*64) * N + n0 + wn*128) * 2 of the tile being drained
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
    const unsigned scr = SCR + wave * (2 * IMG);
    const lds_ptr scw = (lds_ptr)(uintptr_t)(scr + l15 * 272 + g * 8);     // image writes: + (i & 1) * IMG + j * 32
    const lds_cptr scrd = (lds_cptr)(uintptr_t)(scr + g * 272 + l15 * 16); // image reads:  + (i & 1) * IMG + p * 4 * 272
    f32x4 bvD[NJ];                    // bias of the tile being drained
    const rsrc_words_t rB = rsrc_words(bias);
    const int bvoff = (wn * 128 + g * 4) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) bvD[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    u32x2_t pk = {0u, 0u};
    u32x4_t piece[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

    // what gap q of the drain does (q = MFMA index since the start of the tile); everything constant-folds after unrolling
    auto drain_gap = [&](int q) {
#ifdef W4Q_NODRAIN
        if (M != 1) return;
#endif
        const int e = q / GPT, gq = q % GPT;
        if (e < 32) {
            const int i = e >> 3, j = e & 7;
            if (sigmoid_like(MODE)) {
                switch (gq) {
                    case 0: x0 = agpr_read(accD[i][j][0]) + bvD[j][0]; break;
                    case 1: x1 = agpr_read(accD[i][j][1]) + bvD[j][1]; break;
                    case 2: x2 = agpr_read(accD[i][j][2]) + bvD[j][2]; break;
                    case 3: x3 = agpr_read(accD[i][j][3]) + bvD[j][3]; break;
                    case 4: t0 = act_arg<MODE>(x0); t1 = act_arg<MODE>(x1); break;
                    case 5: t2 = act_arg<MODE>(x2); t3 = act_arg<MODE>(x3); break;
                    case 6: t0 = __builtin_amdgcn_exp2f(t0); break;
                    case 7: t1 = __builtin_amdgcn_exp2f(t1); break;
                    case 8: t2 = __builtin_amdgcn_exp2f(t2); break;
                    case 9: t3 = __builtin_amdgcn_exp2f(t3); break;
                    case 10: t0 = 1.f + t0; t1 = 1.f + t1; break;
                    case 11: t2 = 1.f + t2; t3 = 1.f + t3; break;
                    case 12: t0 = __builtin_amdgcn_rcpf(t0); break;
                    case 13: t1 = __builtin_amdgcn_rcpf(t1); break;
                    case 14: t2 = __builtin_amdgcn_rcpf(t2); break;
                    case 15: t3 = __builtin_amdgcn_rcpf(t3); break;
                    case 16: x0 *= t0; x1 *= t1; break;
                    case 17: x2 *= t2; x3 *= t3; break;
                    case 18: pk = u32x2_t{pack_bf16x2(x0, x1), pack_bf16x2(x2, x3)}; break;
                    default: *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(scw + (i & 1) * IMG + j * 32) = pk; break;
                }
            } else {
                switch (gq) {
                    case 0: x0 = act_apply<MODE>(agpr_read(accD[i][j][0]) + bvD[j][0]); break;
                    case 1: x1 = act_apply<MODE>(agpr_read(accD[i][j][1]) + bvD[j][1]); break;
                    case 2: x2 = act_apply<MODE>(agpr_read(accD[i][j][2]) + bvD[j][2]); break;
                    case 3: x3 = act_apply<MODE>(agpr_read(accD[i][j][3]) + bvD[j][3]); break;
                    case 4: pk = u32x2_t{pack_bf16x2(x0, x1), pack_bf16x2(x2, x3)}; break;
                    default: *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(scw + (i & 1) * IMG + j * 32) = pk; break;
                }
            }
        }
        int ri = 0, rp = 0;
        if (q_is_piece_read(q, GPT, &ri, &rp))
            piece[rp & 1] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(scrd + (ri & 1) * IMG + rp * 4 * 272);
#ifndef W4Q_NOSTORE
        if (q_is_piece_store(q, GPT, &ri, &rp))
#else
        if (q_is_piece_store(q, GPT, &ri, &rp) && M == 1)
#endif
            __builtin_amdgcn_raw_buffer_store_b128(piece[rp & 1], rO, vst, sstD + (ri * 16 + rp * 4) * N * 2, 2);
    };

    // One phase: 32 MFMAs (INIT: the tile's first, C = 0) with the fragment reads of the next phase and the DMAs of the
    // phase between them; Q0 >= 0: gap m carries drain gap Q0 + m, fenced so that it stays in its gap.
    auto phase = [&](const bf16x8 (&af)[MI], const bf16x8 (&wf)[NJ], bf16x8 (&afn)[MI], bf16x8 (&wfn)[NJ], lds_cptr pw,
                     lds_cptr pa, auto init_c, auto nd_c, rsrc_words_t rq, unsigned dslot, auto q0_c) {
        constexpr bool INIT = decltype(init_c)::value;
        constexpr int ND = decltype(nd_c)::value, Q0 = decltype(q0_c)::value;
        constexpr int NR = MI + NJ;
        constexpr int MPR = (NM * 3 / 4) / NR;                 // reads in front of MFMAs 0, 2, 4, ...
        constexpr int MPD = (NM * 3 / 4) / ND;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m % MPR == 0 && m / MPR < NR) {
                const int r = m / MPR;
                if (r < NJ) wfn[r] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + r * 2048);
                else afn[r - NJ] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + (r - NJ) * 2048);
            }
            if (m >= 1 && (m - 1) % MPD == 0 && (m - 1) / MPD < ND) {
                const int u = ((m - 1) / MPD) * 4 + wave;
                dma16_quiet(rq, dslot + u * 1024, voff, u * 8 * ldb);     // (the K-step is in rq's base: no per-step scalar offsets)
            }
            if (INIT) mfma16a_init(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
            else mfma16a(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
            if constexpr (Q0 >= 0) {
                if (Q0 + m < QD) {
                    drain_gap(Q0 + m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // K-step S of the current tile (S a compile-time constant in the drain's steps, -1 in the rolled rest); qA / qW: where
    // the request two steps down the stream reads; BIAS: the tile's last step also fetches its bias (the drain set's
    // registers are free by then), in front of the phase-A requests so that the step's counted wait covers it
    const bf16_t* qA = tA + 128;
    const bf16_t* qW = tW + 128;
    auto step = [&](auto init_c, auto s_c, auto bias_c) {
        const rsrc_words_t rqA = rsrc_words(qA), rqW = rsrc_words(qW);
        constexpr int S = decltype(s_c)::value;
        constexpr int QA = S >= 0 ? S * 2 * NM : -1, QB = S >= 0 ? S * 2 * NM + NM : -1;
#ifndef W4Q_NOSTORE
        constexpr int STA = S >= 0 ? q_stores_in(S * 2 * NM, S * 2 * NM + NM, GPT) : 0;     // stores of this step's phase A
#else
        constexpr int STA = 0;
#endif
        if constexpr (decltype(bias_c)::value) {
            if (bias) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) load16_quiet(bvD[j], rB, bvoff, (n0 + j * 16) * 4);
            }
        }
        phase(af0, wf0, af1, wf1, (lds_cptr)(uintptr_t)(oW0 + fw1), (lds_cptr)(uintptr_t)(oA0 + fa1), init_c,
              std::integral_constant<int, PA>{}, rqA, oA2, std::integral_constant<int, QA>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<PA + STA>();
        __builtin_amdgcn_s_barrier();
        phase(af1, wf1, af0, wf0, (lds_cptr)(uintptr_t)(oW1 + fw0), (lds_cptr)(uintptr_t)(oA1 + fa0), std::false_type{},
              std::integral_constant<int, PW>{}, rqW, WBASE + oW2, std::integral_constant<int, QB>{});
        const unsigned a0 = oA0;
        oA0 = oA1; oA1 = oA2; oA2 = a0;
        oW0 = oW1; oW1 = oW2; oW2 = oW0;
    };
    oW2 = oW0;

    using F = std::false_type; using Tt = std::true_type;
    using Roll = std::integral_constant<int, -1>;

    for (int t = 0; t < T; ++t) {
        int m0n = m0, n0n = n0;
        if (t + 1 < T) {
            tile_coords_v((int)blockIdx.x + (t + 1) * G, ntiles, tiles_m, tiles_n, 8, &tm, &tn);
            m0n = tm * BMB; n0n = tn * BNB;
        }
        const bf16_t* tAn = A + (size_t)m0n * K;
        const bf16_t* tWn = Wt + (size_t)n0n * K;
        // after step s the request stream moves on 64 columns, or, when step s+1 is the one that first reads the next
        // tile (s + 3 == nk), to that tile's rows
        auto advance = [&](int s) {
            const bool turn = s + 3 == nk;
            qA = turn ? tAn : qA + 64;
            qW = turn ? tWn : qW + 64;
        };
        int s0 = 0;
        W4P_STAMP(t, 0);
        if (t > 0) {
            // the drain's steps, unrolled (NU of them, none of them the tile's last)
            auto drain_step = [&](auto s_c) {
                constexpr int S = decltype(s_c)::value;
                if constexpr (S < NU) {
                    if constexpr (S == 0) step(Tt{}, s_c, F{});
                    else step(F{}, s_c, F{});
                    advance(S);
                }
            };
            drain_step(std::integral_constant<int, 0>{}); drain_step(std::integral_constant<int, 1>{});
            drain_step(std::integral_constant<int, 2>{}); drain_step(std::integral_constant<int, 3>{});
            drain_step(std::integral_constant<int, 4>{}); drain_step(std::integral_constant<int, 5>{});
            drain_step(std::integral_constant<int, 6>{}); drain_step(std::integral_constant<int, 7>{});
            drain_step(std::integral_constant<int, 8>{}); drain_step(std::integral_constant<int, 9>{});
            drain_step(std::integral_constant<int, 10>{});
            static_assert(NU <= 11, "drain steps");
            s0 = NU;
        } else {
            step(Tt{}, Roll{}, F{});
            advance(0);
            s0 = 1;
        }
        W4P_STAMP(t, 1);
        for (int s = s0; s < nk - 1; ++s) { step(F{}, Roll{}, F{}); advance(s); }
        step(F{}, Roll{}, Tt{});
        advance(nk - 1);
        W4P_STAMP(t, 2);
        mfma_retire();
        // the finished tile becomes the drain set
#pragma unroll
        for (int m = 0; m < MI * NJ; ++m) pin_a(acc[m / NJ][m % NJ]);
#pragma unroll
        for (int m = 0; m < MI * NJ; ++m) { accD[m / NJ][m % NJ] = acc[m / NJ][m % NJ]; pin_a(accD[m / NJ][m % NJ]); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) pin_v(bvD[j]);
        sstD = ((m0 + wm * MI * 16) * N + n0 + wn * 128) * 2;
        m0 = m0n; n0 = n0n;
        W4P_STAMP(t, 3);
    }
    // the last tile's drain, with no MFMAs to sit between (a fold over the gap numbers: every index a constant)
    wait_vmcnt<0>();
    each_constant(drain_gap, std::make_integer_sequence<int, QD>{});
}

template <int MODE>
static void launch_w4q(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, void* out, int num_cus,
                       hipStream_t st) {
    if constexpr (bf16_out(MODE)) {
        auto kern = gemm_w4q_kernel<MODE>;
        constexpr int LDS = 3 * 128 * 128 + 2 * 256 * 128 + 4 * 2 * 16 * 272;
        static std::once_flag attr_set;
        std::call_once(attr_set, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        });
        const int ntiles = (M / 128) * (N / 256);
        const int grid = ntiles < num_cus ? ntiles : num_cus;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, A, Wt, bias, M, N, K, reinterpret_cast<bf16_t*>(out));
    }
}

constexpr bool w4q_shape_ok(int M, int N, int K, int mode) {
    return M % 128 == 0 && N % 256 == 0 && K % 64 == 0 && (K / 64 - 1) * 64 >= w4::q_drain_gaps(mode) && K / 64 >= 4;
}


}  // namespace wise
