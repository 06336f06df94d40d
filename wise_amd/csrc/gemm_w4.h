// One-wave-per-SIMD bf16 GEMM for gfx950:  C[M,N] = epilogue( A[M,K] @ Wt[N,K]^T + bias[N] ).
//
// Stands behind the four nn.Linear calls per transformer block of open_clip's VisionTransformer.forward as reached from
// src/feature/mlfoundation_openclip.py:99, like every kernel of gemm_bf16.hip.  Roofline: MFMA.
//
// Why another structure.  The ping-pong kernel (gemm_bf16.hip, gemm_pp_kernel) runs eight waves as two groups that
// alternate a load part and a 32-MFMA cluster between two barriers per 32-deep K-tile; its stamps (profiles/
// r02_gemm_study_stamps.txt) show the matrix pipe of a SIMD busy for 2 x 512 of every ~2150 cycles: barriers, counted
// waits and fragment reads that nothing overlaps.  Here a workgroup is FOUR waves, one per SIMD, each owning a
// (MI*16) x 128 block of the output tile in 8*MI accumulator tiles (256 accumulator registers at MI = 8: the whole
// AGPR half of the wave's 512 registers), and each wave runs ONE software-pipelined instruction stream per 64-deep
// K-step — 16*MI MFMAs with the step's 2*(MI+8) fragment reads and its LDS-DMA instructions issued BETWEEN them — and
// ONE barrier per K-step:
//
//   step s = [ phase A: MFMAs on k-half 0 of step s   ||  read k-half 1 of step s   ||  LDS-DMA of A(s+2) ]
//            s_waitcnt lgkmcnt(0), counted vmcnt, s_barrier
//            [ phase B: MFMAs on k-half 1 of step s   ||  read k-half 0 of step s+1 ||  LDS-DMA of W(s+2) ]
//
// LDS is a ring of FIVE operand slots of 32 KiB ([256 rows][64 k] bf16, 128-byte rows, 16-byte chunks XOR-swizzled by
// row through the DMA's per-lane SOURCE address): operands enter in the order A(0), W(0), A(1), W(1), ... at slot
// q mod 5.  RAW: a wave waits for its own DMAs of A(s+1) and W(s+1) (everything but the youngest operand) before the
// barrier of step s, behind which phase B reads them.  WAR: A(s+2) overwrites the slot of W(s-1), last read in phase A
// of step s-1, in front of barrier s-1; W(s+2) overwrites the slot of A(s), last read in phase A of step s — every wave
// retires those reads (lgkmcnt(0)) in front of barrier s.  An operand is requested at least one whole step (16*MI MFMA
// slots, ~1 us) before the barrier that needs it.
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

constexpr int SLOT = 32768, NSLOT = 5, NJ = 8;
constexpr int LDS_BYTES = SLOT * NSLOT;   // 160 KiB: the whole LDS of a CU

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define W4_LDS(T, off) (*reinterpret_cast<__attribute__((address_space(3))) T*>((uintptr_t)(off)))

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_off, int voff, int soff) {
    // LDS destination = lds_off + lane * 16 (wave-uniform base; linear)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(uintptr_t)lds_off, 16, voff, soff, 0, 0);
}

// one operand tile of ROWS x 64 bf16 -> one ring slot; 4 waves, wave-instruction u = t*4 + wave covers rows 8u .. 8u+7
template <int ROWS>
__device__ __forceinline__ void stage_operand(__amdgpu_buffer_rsrc_t rsrc, unsigned slot_off, int voff, int ld_bytes,
                                              int k_bytes, int wave) {
#pragma unroll
    for (int t = 0; t < ROWS / 32; ++t) {
        const int u = t * 4 + wave;
        dma16(rsrc, slot_off + u * 1024, voff, u * 8 * ld_bytes + k_bytes);
    }
}

__device__ __forceinline__ bf16x8 lds_read16(unsigned off) {
    return W4_LDS(const bf16x8, off);
}

__device__ __forceinline__ unsigned next_slot2(unsigned off) {   // two slots on, modulo the ring
    off += 2 * SLOT;
    return off >= (unsigned)LDS_BYTES ? off - LDS_BYTES : off;
}

// bf16 epilogue of RI row tiles (RI <= 4) x 128 columns: wave-private image of 16*RI rows x 272 B
template <int MODE, int MI, int RI>
__device__ __forceinline__ void epi_bf16_pass(const f32x4 (&acc)[MI][NJ], int i0, const float4 (&bv)[NJ], bf16_t* __restrict__ out,
                                              int N, int row0, int col0, int lane, unsigned my) {
    constexpr int RS = 272;
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
    for (int t = 0; t < RI * 4; ++t) {
        const int row = t * 4 + g;
        const u32x4_t v = W4_LDS(const u32x4_t, my + row * RS + l15 * 16);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + l15 * 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fp32 epilogue (EPI_RESID: out += ..., EPI_F32: out = ...) of RI row tiles (RI <= 2) x 128 columns: image of 16*RI rows
// x 528 B; `res` holds the pass's residual values in walk order (loaded by resid_load a pass earlier)
template <int RI>
__device__ __forceinline__ void resid_load(const float* __restrict__ out, int N, int row0, int col0, int lane, float4 (&res)[16]) {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int t = 0; t < RI * 8; ++t)
        res[t] = *reinterpret_cast<const float4*>(out + (size_t)(row0 + t * 2 + h) * N + col0 + c * 4);
}

template <int MODE, int MI, int RI>
__device__ __forceinline__ void epi_f32_pass(const f32x4 (&acc)[MI][NJ], int i0, const float4 (&bv)[NJ], float* __restrict__ out,
                                             int N, int row0, int col0, int lane, unsigned my, const float4 (&res)[16]) {
    constexpr int RS = 528;
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
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int t = 0; t < RI * 8; ++t) {
        const int row = t * 2 + h;
        const f32x4 l = W4_LDS(const f32x4, my + row * RS + c * 16);
        float4 v = make_float4(l[0], l[1], l[2], l[3]);
        if (MODE == EPI_RESID) { v.x += res[t].x; v.y += res[t].y; v.z += res[t].z; v.w += res[t].w; }
        *reinterpret_cast<float4*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + c * 4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One MFMA as a volatile asm statement: the accumulator is ONE tied AGPR operand (no accumulator copies, whatever the
// register allocator makes of 256 live accumulator registers), and volatile statements keep their source order relative
// to each other and to every memory operation — so the interleave of MFMAs, fragment reads and DMAs below is exactly the
// one written (the compiler still counts lgkmcnt for the fragment registers the statement reads).
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

}  // namespace w4

// MI = row tiles per wave: 8 -> 256 x 256 block tile, 5 -> 160 x 256.  M % (32*MI) == 0, N % 256 == 0, K % 64 == 0, K >= 192.
template <int MODE, int MI>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                         const float* __restrict__ bias, int M, int N, int K,
                                                         void* __restrict__ out) {
    using namespace w4;
    constexpr int BMB = 2 * MI * 16, BNB = 256;
    constexpr int PA = BMB / 32, PW = BNB / 32;     // DMAs per wave and operand
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
    const unsigned fw0 = (wn * 128 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fw1 = fw0 ^ 64;

    const int nk = K / 64;
    // prologue: A(0), W(0), A(1), W(1) -> slots 0..3
    stage_operand<BMB>(rA, 0 * SLOT, voff, ldb, 0, wave);
    stage_operand<BNB>(rW, 1 * SLOT, voff, ldb, 0, wave);
    stage_operand<BMB>(rA, 2 * SLOT, voff, ldb, 128, wave);
    stage_operand<BNB>(rW, 3 * SLOT, voff, ldb, 128, wave);
    wait_vmcnt<PA + PW>();
    __builtin_amdgcn_s_barrier();
    W4_STAMP(1);

    typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
    bf16x8 af0[MI], wf0[NJ], af1[MI], wf1[NJ];
    {
        lds_cptr pw = (lds_cptr)(uintptr_t)(1 * SLOT + fw0), pa = (lds_cptr)(uintptr_t)(0 * SLOT + fa0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf0[j] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + j * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + i * 2048);
    }

    unsigned sA = 0, sW = SLOT;          // slots of A(s), W(s); A(s+1) = next_slot2(sA), A(s+2) = next of that; W(s+2) -> sA
    int kb = 0;                          // byte offset of K-step s inside a row

    // One phase: NM = MI*NJ MFMAs on (af, wf) in i-major order, with the phase's fragment reads (first the NJ weight
    // fragments, then the MI activation fragments: the order the next phase needs them) and DMAs placed between them:
    // read r in front of MFMA r*MPR, DMA d in front of MFMA DMA0 + d*MPD.
    auto phase = [&](const bf16x8 (&af)[MI], const bf16x8 (&wf)[NJ], bf16x8 (&afn)[MI], bf16x8 (&wfn)[NJ], auto read_c,
                     lds_cptr pw, lds_cptr pa, auto ndma_c, __amdgpu_buffer_rsrc_t rsrc, unsigned dslot, int dkb) {
        constexpr bool READ = decltype(read_c)::value;
        constexpr int NDMA = decltype(ndma_c)::value;
        constexpr int NM = MI * NJ, NR = MI + NJ;
        constexpr int MPR = (NM * 3 / 4) / NR;                 // reads spread over the first three quarters of the phase
        constexpr int MPD = NDMA > 0 ? (NM / 2) / (NDMA > 0 ? NDMA : 1) : 1, DMA0 = 1;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (READ && m % MPR == 0 && m / MPR < NR) {
                const int r = m / MPR;
                if (r < NJ) wfn[r] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + r * 2048);
                else afn[r - NJ] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + (r - NJ) * 2048);
            }
            if (NDMA > 0 && m >= DMA0 && (m - DMA0) % MPD == 0 && (m - DMA0) / MPD < NDMA) {
                const int u = ((m - DMA0) / MPD) * 4 + wave;
                dma16(rsrc, dslot + u * 1024, voff, u * 8 * ldb + dkb);
            }
            mfma16(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
        }
    };

    // one K-step; DMA: request A(s+2) / W(s+2); NEXT: read k-half 0 of step s+1; DRAIN: the barrier waits for every DMA
    auto step = [&](auto dma_c, auto next_c, auto drain_c) {
        constexpr bool DMA = decltype(dma_c)::value, DRAIN = decltype(drain_c)::value;
        const unsigned sA1 = next_slot2(sA), sW1 = next_slot2(sW), sA2 = next_slot2(sA1);
        // ---- phase A: k-half 0 of step s; read k-half 1 of step s; request A(s+2)
        phase(af0, wf0, af1, wf1, std::true_type{}, (lds_cptr)(uintptr_t)(sW + fw1), (lds_cptr)(uintptr_t)(sA + fa1),
              std::integral_constant<int, DMA ? PA : 0>{}, rA, sA2, kb + 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (DRAIN) wait_vmcnt<0>(); else wait_vmcnt<PA>();
        __builtin_amdgcn_s_barrier();
        // ---- phase B: k-half 1 of step s; read k-half 0 of step s+1; request W(s+2) into the slot of A(s)
        phase(af1, wf1, af0, wf0, next_c, (lds_cptr)(uintptr_t)(sW1 + fw0), (lds_cptr)(uintptr_t)(sA1 + fa0),
              std::integral_constant<int, DMA ? PW : 0>{}, rW, sA, kb + 256);
        sA = sA1; sW = sW1; kb += 128;
    };
    using T = std::true_type; using F = std::false_type;
    for (int s = 0; s < nk - 2; ++s) step(T{}, T{}, F{});
    step(F{}, T{}, T{});     // s = nk-2: nothing left to request; A(nk-1), W(nk-1) must have landed
    step(F{}, F{}, T{});     // s = nk-1
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the epilogue reads the accumulators

    W4_STAMP(2);
    // ---- epilogue: the ring is dead once every wave is past its last fragment read
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int row0 = m0 + wm * MI * 16, col0 = n0 + wn * 128;
    float4 bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        bv[j] = bias ? *reinterpret_cast<const float4*>(bias + col0 + j * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned my = wave * 20480;    // wave-private scratch inside the dead ring
    if constexpr (bf16_out(MODE)) {
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
#pragma unroll
        for (int i0 = 0; i0 + 4 <= MI; i0 += 4) epi_bf16_pass<MODE, MI, 4>(acc, i0, bv, o, N, row0, col0, lane, my);
        if constexpr (MI % 4 != 0) epi_bf16_pass<MODE, MI, MI % 4>(acc, MI - MI % 4, bv, o, N, row0, col0, lane, my);
    } else {
        float* o = reinterpret_cast<float*>(out);
        float4 res[2][16];
        constexpr int NP = (MI + 1) / 2;
        if (MODE == EPI_RESID) resid_load<2>(o, N, row0, col0, lane, res[0]);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (MODE == EPI_RESID && p + 1 < NP) {
                if ((p + 1) * 2 + 2 <= MI) resid_load<2>(o, N, row0 + (p + 1) * 32, col0, lane, res[(p + 1) & 1]);
                else resid_load<1>(o, N, row0 + (p + 1) * 32, col0, lane, res[(p + 1) & 1]);
            }
            if (p * 2 + 2 <= MI) epi_f32_pass<MODE, MI, 2>(acc, p * 2, bv, o, N, row0, col0, lane, my, res[p & 1]);
            else epi_f32_pass<MODE, MI, 1>(acc, p * 2, bv, o, N, row0, col0, lane, my, res[p & 1]);
        }
    }
    W4_STAMP(3);
}

template <int MODE, int MI>
static void launch_w4(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, void* out, hipStream_t st) {
    auto kern = gemm_w4_kernel<MODE, MI>;
    static std::once_flag attr_set;
    std::call_once(attr_set, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES);
    });
    const int grid = (M / (32 * MI)) * (N / 256);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), w4::LDS_BYTES, st, A, Wt, bias, M, N, K, out);
}

constexpr bool w4_shape_ok(int M, int N, int K, int MI) { return M % (32 * MI) == 0 && N % 256 == 0 && K % 64 == 0 && K >= 192; }

}  // namespace wise
