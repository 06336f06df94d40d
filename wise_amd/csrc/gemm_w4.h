// One-wave-per-SIMD bf16 GEMM for gfx950:  C[M,N] = epilogue( A[M,K] @ Wt[N,K]^T + bias[N] ).
//
// Stands behind the four nn.Linear calls per transformer block of open_clip's VisionTransformer.forward as reached from
// src/feature/mlfoundation_openclip.py:99, like every kernel of gemm_bf16.hip.  Roofline: MFMA.
//
// Why another structure.  The ping-pong kernel (gemm_bf16.hip, gemm_pp_kernel) runs eight waves as two groups that
// alternate a load part and a 32-MFMA cluster between two barriers per 32-deep K-tile; its stamps (profiles/
// r02_gemm_study_stamps.txt) show the matrix pipe of a SIMD busy for 2 x 512 of every ~2150 cycles: barriers, counted
// waits and fragment reads that nothing overlaps.  Here a workgroup is FOUR waves, one per SIMD, each owning a
// (MI*16) x (NJ*16) block of the output tile in MI*NJ accumulator tiles (256 accumulator registers at 8 x 8: the whole
// AGPR half of the wave's 512 registers; tiles past the 64th live in VGPRs), and each wave runs ONE software-pipelined instruction stream per 64-deep
// K-step — 2*MI*NJ MFMAs with the step's 2*(MI+NJ) fragment reads and its LDS-DMA instructions issued BETWEEN them — and
// ONE barrier per K-step:
//
//   step s = [ phase A: MFMAs on k-half 0 of step s   ||  read k-half 1 of step s   ||  LDS-DMA of A(s+2) ]
//            s_waitcnt lgkmcnt(0), counted vmcnt, s_barrier
//            [ phase B: MFMAs on k-half 1 of step s   ||  read k-half 0 of step s+1 ||  LDS-DMA of W(s+2) ]
//
// LDS holds SA slots for the activation tile and SW slots for the weight tile of a K-step ([rows][64 k] bf16, 128-byte
// rows, 16-byte chunks XOR-swizzled by row through the DMA's per-lane SOURCE address); operand X(s) sits in slot
// s mod SX.  An operand with THREE slots is requested two steps ahead in phase A (X(s+2) overwrites X(s-1), last read in
// phase A of step s-1, in front of barrier s-1); an operand with TWO slots in phase B (X(s+2) overwrites X(s), last read
// in phase A of step s: every wave retires those reads, lgkmcnt(0), in front of barrier s).  RAW: before barrier s a wave
// waits for its own DMAs of step s+1 (all but the operand requested in phase A of this step), behind which phase B reads
// them.  Every operand is requested at least one whole step (16*MI*NJ/8 MFMA slots, ~1 us) before the barrier that needs
// it.  256 x 256 tiles: 3 + 2 slots of 32 KiB = the whole 160 KiB; 320 x 256: 2 + 2 (144 KiB); 320 x 192: 2 + 3.
//
// The DMAs are buffer loads (buffer_load_dwordx4 ... lds): a lane's voffset never changes, the tile row and the K-step
// are a scalar offset, so a DMA costs no vector instruction.  The product is computed transposed (weights as the MFMA A
// operand) so that a lane holds 4 consecutive output columns; both epilogues leave through wave-private LDS images as
// whole row segments (bf16: 256 B, fp32: 512 B per row and instruction).
#pragma once
#include <type_traits>
#include "gemm_shared.h"

namespace wise {
namespace w4 {

// In-kernel s_memtime stamps of block 0 and of the last block (tools/gemm_lab.hip builds with -DW4_STAMPS): slot 0 kernel
// entry, 1 first operands landed, 2 main loop done, 3 epilogue done.  Compiled out of the product.
#ifdef W4_STAMPS
__device__ unsigned long long g_w4_stamps[2][4];
#define W4_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                                \
            g_w4_stamps[blockIdx.x == 0 ? 0 : 1][k] = __builtin_amdgcn_s_memtime();                              \
    } while (0)
#else
#define W4_STAMP(k) do {} while (0)
#endif


constexpr int lds_bytes(int MI, int NJ, int SA, int SW) { return SA * (2 * MI * 16 * 128) + SW * (2 * NJ * 16 * 128); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define W4_LDS(T, off) (*reinterpret_cast<__attribute__((address_space(3))) T*>((uintptr_t)(off)))

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_off, int voff, int soff) {
    // LDS destination = lds_off + lane * 16 (wave-uniform base; linear)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(uintptr_t)lds_off, 16, voff, soff, 0, 0);
}

// one operand tile of ROWS x 64 bf16 -> one slot; 4 waves, wave-instruction u = t*4 + wave covers rows 8u .. 8u+7
template <int ROWS>
__device__ __forceinline__ void stage_operand(__amdgpu_buffer_rsrc_t rsrc, unsigned slot_off, int voff, int ld_bytes,
                                              int k_bytes, int wave) {
#pragma unroll
    for (int t = 0; t < ROWS / 32; ++t) {
        const int u = t * 4 + wave;
        dma16(rsrc, slot_off + u * 1024, voff, u * 8 * ld_bytes + k_bytes);
    }
}

// One MFMA as a volatile asm statement: the accumulator is ONE tied operand (no accumulator copies, whatever the
// register allocator makes of 256 live accumulator registers), and volatile statements keep their source order relative
// to each other and to every memory operation — so the interleave of MFMAs, fragment reads and DMAs below is exactly the
// one written (the compiler still counts lgkmcnt for the fragment registers the statement reads).  Accumulator tiles
// 0..63 are AGPRs ("a"), any further ones VGPRs ("v").
__device__ __forceinline__ void mfma16a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16v(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// bf16 epilogue of RI row tiles (RI <= 4) x NJ*16 columns: wave-private image of 16*RI rows, then a row-major walk in
// which every global store instruction writes whole row segments (16 bytes per lane, consecutive lanes consecutive)
template <int MODE, int MI, int NJ, int RI>
__device__ __forceinline__ void epi_bf16_pass(const f32x4 (&acc)[MI][NJ], int i0, const float4 (&bv)[NJ], bf16_t* __restrict__ out,
                                              int N, int row0, int col0, int lane, unsigned my) {
    constexpr int RS = NJ * 32 + 16, CPR = NJ * 2;   // image row bytes, 16-byte chunks per row
    static_assert((RI * 16 * CPR) % 64 == 0, "walk covers the piece in whole wave instructions");
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ii = 0; ii < RI; ++ii) {
            const f32x4 a = acc[i0 + ii][j];
            const float v0 = act_apply<MODE>(a[0] + bv[j].x), v1 = act_apply<MODE>(a[1] + bv[j].y),
                        v2 = act_apply<MODE>(a[2] + bv[j].z), v3 = act_apply<MODE>(a[3] + bv[j].w);
            W4_LDS(u32x2_t, my + (ii * 16 + l15) * RS + (j * 16 + g * 4) * 2) = u32x2_t{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < RI * 16 * CPR / 64; ++t) {
        const int idx = t * 64 + lane, row = idx / CPR, c = idx % CPR;
        const u32x4_t v = W4_LDS(const u32x4_t, my + row * RS + c * 16);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + c * 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fp32 epilogue (EPI_RESID: out += ..., EPI_F32: out = ...) of RI row tiles x NJ*16 columns; `res` holds the piece's
// residual values in walk order (loaded by resid_load a pass earlier, so their latency hides under the previous pass)
template <int NJ, int RI>
__device__ __forceinline__ void resid_load(const float* __restrict__ out, int N, int row0, int col0, int lane,
                                           float4 (&res)[RI * 16 * NJ * 4 / 64]) {
    constexpr int CPR = NJ * 4;
#pragma unroll
    for (int t = 0; t < RI * 16 * CPR / 64; ++t) {
        const int idx = t * 64 + lane, row = idx / CPR, c = idx % CPR;
        res[t] = *reinterpret_cast<const float4*>(out + (size_t)(row0 + row) * N + col0 + c * 4);
    }
}

template <int MODE, int MI, int NJ, int RI>
__device__ __forceinline__ void epi_f32_pass(const f32x4 (&acc)[MI][NJ], int i0, const float4 (&bv)[NJ], float* __restrict__ out,
                                             int N, int row0, int col0, int lane, unsigned my,
                                             const float4 (&res)[RI * 16 * NJ * 4 / 64]) {
    constexpr int RS = NJ * 64 + 16, CPR = NJ * 4;
    static_assert((RI * 16 * CPR) % 64 == 0, "walk covers the piece in whole wave instructions");
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ii = 0; ii < RI; ++ii) {
            const f32x4 a = acc[i0 + ii][j];
            W4_LDS(f32x4, my + (ii * 16 + l15) * RS + (j * 16 + g * 4) * 4) =
                f32x4{a[0] + bv[j].x, a[1] + bv[j].y, a[2] + bv[j].z, a[3] + bv[j].w};
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < RI * 16 * CPR / 64; ++t) {
        const int idx = t * 64 + lane, row = idx / CPR, c = idx % CPR;
        const f32x4 l = W4_LDS(const f32x4, my + row * RS + c * 16);
        float4 v = make_float4(l[0], l[1], l[2], l[3]);
        if (MODE == EPI_RESID) { v.x += res[t].x; v.y += res[t].y; v.z += res[t].z; v.w += res[t].w; }
        *reinterpret_cast<float4*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + c * 4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace w4

// Block tile (32*MI) x (32*NJ), waves 2 x 2, wave tile (16*MI) x (16*NJ); SA / SW = LDS slots of the activation / weight
// operand (3 + 2, 2 + 3 or 2 + 2).  M % (32*MI) == 0, N % (32*NJ) == 0, K % 64 == 0, K >= 192.
template <int MODE, int MI, int NJ, int SA, int SW>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                         const float* __restrict__ bias, int M, int N, int K,
                                                         void* __restrict__ out) {
    using namespace w4;
    static_assert((SA == 3 && SW == 2) || (SA == 2 && SW == 3) || (SA == 2 && SW == 2), "slot plan");
    static_assert(lds_bytes(MI, NJ, SA, SW) <= 160 * 1024, "LDS");
    constexpr int BMB = 2 * MI * 16, BNB = 2 * NJ * 16;
    constexpr int ASZ = BMB * 128, WSZ = BNB * 128, WBASE = SA * ASZ;   // slot sizes; the W slots sit behind the A slots
    constexpr int PA = BMB / 32, PW = BNB / 32;       // DMAs per wave and operand
    // which operand is requested early (phase A of step s for step s+2): the one with three slots
    constexpr bool A_EARLY = SA == 3, W_EARLY = SW == 3;
    constexpr int NDMA_A = A_EARLY ? PA : (W_EARLY ? PW : 0);            // DMAs in phase A
    constexpr int NDMA_B = A_EARLY ? PW : (W_EARLY ? PA : PA + PW);      // DMAs in phase B
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    W4_STAMP(0);
    int tm, tn;
    tile_coords(M / BMB, N / BNB, 4, &tm, &tn);
    const int m0 = tm * BMB, n0 = tn * BNB;

    const int ldb = K * 2;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (size_t)n0 * K), 0, 0x7fffffff, 0x00020000);
    const int voff = (lane >> 3) * ldb + (((lane & 7) ^ (lane >> 3)) << 4);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a slot: row (base + l15), chunk (khalf*4 + g) ^ (l15 & 7)
    const unsigned fa0 = (wm * MI * 16 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fa1 = fa0 ^ 64;
    const unsigned fw0 = WBASE + (wn * NJ * 16 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fw1 = fw0 ^ 64;

    const int nk = K / 64;
    // prologue: steps 0 and 1 -> slots 0 and 1 of each operand
    stage_operand<BMB>(rA, 0, voff, ldb, 0, wave);
    stage_operand<BNB>(rW, WBASE, voff, ldb, 0, wave);
    stage_operand<BMB>(rA, ASZ, voff, ldb, 128, wave);
    stage_operand<BNB>(rW, WBASE + WSZ, voff, ldb, 128, wave);
    wait_vmcnt<PA + PW>();
    __builtin_amdgcn_s_barrier();
    W4_STAMP(1);

    bf16x8 af0[MI], wf0[NJ], af1[MI], wf1[NJ];
    {
        lds_cptr pw = (lds_cptr)(uintptr_t)fw0, pa = (lds_cptr)(uintptr_t)fa0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf0[j] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + j * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + i * 2048);
    }

    // slot byte offsets of A(s), A(s+1), A(s+2) and W(s), W(s+1), W(s+2) (relative to the operand's first slot)
    unsigned oA0 = 0, oA1 = ASZ, oA2 = (SA == 3) ? 2 * ASZ : 0;
    unsigned oW0 = 0, oW1 = WSZ, oW2 = (SW == 3) ? 2 * WSZ : 0;
    int kb = 0;                          // byte offset of K-step s inside a row

    // One phase: NM = MI*NJ MFMAs on (af, wf) in i-major order, with the phase's fragment reads (first the NJ weight
    // fragments, then the MI activation fragments: the order the next phase needs them) and DMAs placed between them:
    // read r in front of MFMA r*MPR, DMA d in front of MFMA 1 + d*MPD (DMAs 0..NDA-1 fetch the activation tile into
    // slot dA, the rest the weight tile into slot dW).
    auto phase = [&](const bf16x8 (&af)[MI], const bf16x8 (&wf)[NJ], bf16x8 (&afn)[MI], bf16x8 (&wfn)[NJ], auto read_c,
                     lds_cptr pw, lds_cptr pa, auto nda_c, auto ndw_c, unsigned dA, unsigned dW, int dkb) {
        constexpr bool READ = decltype(read_c)::value;
        constexpr int NDA = decltype(nda_c)::value, NDW = decltype(ndw_c)::value, ND = NDA + NDW;
        constexpr int NM = MI * NJ, NR = MI + NJ;
        constexpr int MPR = (NM * 3 / 4) / NR;                 // reads spread over the first three quarters of the phase
        constexpr int MPD = ND > 0 ? ((NM * 3 / 4) / (ND > 0 ? ND : 1) > 0 ? (NM * 3 / 4) / (ND > 0 ? ND : 1) : 1) : 1;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (READ && m % MPR == 0 && m / MPR < NR) {
                const int r = m / MPR;
                if (r < NJ) wfn[r] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + r * 2048);
                else afn[r - NJ] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + (r - NJ) * 2048);
            }
            if (ND > 0 && m >= 1 && (m - 1) % MPD == 0 && (m - 1) / MPD < ND) {
                const int d = (m - 1) / MPD;
                if (d < NDA) { const int u = d * 4 + wave; dma16(rA, dA + u * 1024, voff, u * 8 * ldb + dkb); }
                else { const int u = (d - NDA) * 4 + wave; dma16(rW, WBASE + dW + u * 1024, voff, u * 8 * ldb + dkb); }
            }
            if (m < 64) mfma16a(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
            else mfma16v(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
        }
    };

    // one K-step; DMA: request step s+2; NEXT: read k-half 0 of step s+1; DRAIN: the barrier waits for every DMA
    auto step = [&](auto dma_c, auto next_c, auto drain_c) {
        constexpr bool DMA = decltype(dma_c)::value, DRAIN = decltype(drain_c)::value;
        using Z = std::integral_constant<int, 0>;
        // ---- phase A: k-half 0 of step s; read k-half 1 of step s; request the three-slot operand of step s+2
        phase(af0, wf0, af1, wf1, std::true_type{}, (lds_cptr)(uintptr_t)(oW0 + fw1), (lds_cptr)(uintptr_t)(oA0 + fa1),
              std::integral_constant<int, (DMA && A_EARLY) ? PA : 0>{}, std::integral_constant<int, (DMA && W_EARLY) ? PW : 0>{},
              oA2, oW2, kb + 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (DRAIN) wait_vmcnt<0>(); else wait_vmcnt<NDMA_A>();
        __builtin_amdgcn_s_barrier();
        // ---- phase B: k-half 1 of step s; read k-half 0 of step s+1; request the two-slot operand(s) of step s+2
        phase(af1, wf1, af0, wf0, next_c, (lds_cptr)(uintptr_t)(oW1 + fw0), (lds_cptr)(uintptr_t)(oA1 + fa0),
              std::integral_constant<int, (DMA && !A_EARLY) ? PA : 0>{}, std::integral_constant<int, (DMA && !W_EARLY) ? PW : 0>{},
              oA2, oW2, kb + 256);
        (void)Z{};
        const unsigned a0 = oA0, w0 = oW0;
        oA0 = oA1; oA1 = oA2; oA2 = (SA == 3) ? a0 : oA0;     // two slots: step s+3 goes where step s+1 sits
        oW0 = oW1; oW1 = oW2; oW2 = (SW == 3) ? w0 : oW0;
        kb += 128;
    };
    // with two slots X(s+2) shares the slot of X(s): oX2 must name it
    if (SA == 2) oA2 = oA0;
    if (SW == 2) oW2 = oW0;
    using T = std::true_type; using F = std::false_type;
    for (int s = 0; s < nk - 2; ++s) step(T{}, T{}, F{});
    step(F{}, T{}, T{});     // s = nk-2: nothing left to request; step nk-1 must have landed
    step(F{}, F{}, T{});     // s = nk-1
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the epilogue reads the accumulators

    W4_STAMP(2);
    // ---- epilogue: the slots are dead once every wave is past its last fragment read
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int row0 = m0 + wm * MI * 16, col0 = n0 + wn * NJ * 16;
    float4 bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        bv[j] = bias ? *reinterpret_cast<const float4*>(bias + col0 + j * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned my = wave * 20480;    // wave-private scratch inside the dead slots
    if constexpr (bf16_out(MODE)) {
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
        // the tiles that live in VGPRs (rows 64/NJ and up) leave first
        if constexpr (MI % 4 != 0) epi_bf16_pass<MODE, MI, NJ, MI % 4>(acc, MI - MI % 4, bv, o, N, row0, col0, lane, my);
#pragma unroll
        for (int i0 = (MI / 4 - 1) * 4; i0 >= 0; i0 -= 4) epi_bf16_pass<MODE, MI, NJ, 4>(acc, i0, bv, o, N, row0, col0, lane, my);
    } else {
        float* o = reinterpret_cast<float*>(out);
        constexpr int RP = NJ == 8 ? 2 : 1;            // row tiles per pass
        constexpr int NP = (MI + RP - 1) / RP, NRES = RP * 16 * NJ * 4 / 64;
        static_assert(MI % RP == 0 || RP == 2, "passes");
        float4 res[2][NRES];
        auto load = [&](int p, float4 (&r)[NRES]) {
            if (p * RP + RP <= MI) resid_load<NJ, RP>(o, N, row0 + p * RP * 16, col0, lane, r);
            else if constexpr (RP == 2) {
                float4 (&h)[NRES / 2] = reinterpret_cast<float4 (&)[NRES / 2]>(r);
                resid_load<NJ, 1>(o, N, row0 + p * RP * 16, col0, lane, h);
            }
        };
        if (MODE == EPI_RESID) load(0, res[0]);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (MODE == EPI_RESID && p + 1 < NP) load(p + 1, res[(p + 1) & 1]);
            if (p * RP + RP <= MI) epi_f32_pass<MODE, MI, NJ, RP>(acc, p * RP, bv, o, N, row0, col0, lane, my, res[p & 1]);
            else if constexpr (RP == 2)
                epi_f32_pass<MODE, MI, NJ, 1>(acc, p * RP, bv, o, N, row0, col0, lane, my,
                                              reinterpret_cast<const float4 (&)[NRES / 2]>(res[p & 1]));
        }
    }
    W4_STAMP(3);
}

template <int MODE, int MI, int NJ, int SA, int SW>
static void launch_w4(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, void* out, hipStream_t st) {
    auto kern = gemm_w4_kernel<MODE, MI, NJ, SA, SW>;
    constexpr int LDS = w4::lds_bytes(MI, NJ, SA, SW) > 4 * 20480 ? w4::lds_bytes(MI, NJ, SA, SW) : 4 * 20480;
    static std::once_flag attr_set;
    std::call_once(attr_set, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    const int grid = (M / (32 * MI)) * (N / (32 * NJ));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, A, Wt, bias, M, N, K, out);
}

constexpr bool w4_shape_ok(int M, int N, int K, int MI, int NJ = 8) {
    return M % (32 * MI) == 0 && N % (32 * NJ) == 0 && K % 64 == 0 && K >= 192;
}

}  // namespace wise
