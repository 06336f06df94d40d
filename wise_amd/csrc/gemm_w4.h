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

// LayerNorm folded into the GEMMs on either side of it (round 4, vit.hip "fold mode").  y = LN(x) W^T + b with
// LN(x) = (x - mean) rstd gamma + beta is rstd * (x W''^T) + b'' when W'' = gamma W minus its row mean over K (the centring
// moves into the weights: sum_k (x_k - mean) w_k = sum_k x_k (w_k - mean_k w)) and b'' = b + W beta — both made by the packer.
//   FOLD = 1 (bf16 outputs: QKV, fc1): the operand is bf16(x) as it stands and the epilogue is fma(acc, rstd[row], b''[col]).
//   FOLD = 2 (EPI_RESID: out-projection, fc2): x += ... on a residual stream stored as hi + lo (two bf16 arrays; hi is the next
//             GEMM's operand as it stands), and per-row partial sums of x and x^2 over aligned 64-column groups, reduced in ONE fixed tree
//             whatever the tile shape (wave parts are 64 or 128 columns wide: NJ = 4 or 8), so a row's statistics — and with
//             them every bit downstream — do not depend on the batch it sits in; the LAST workgroup of a row stripe to finish
//             (arrival counter) adds the N/64 partials in order and writes rstd = rsqrt(var + eps) for the stripe's rows.
// No LayerNorm launch is left between the two GEMMs, and the fp32 rows are not read a second time.
struct FoldArgs {
    // FOLD 1: stats = the [M] row scales (rstd).  FOLD 2: stats = ONE region laid out as rstd_out [M] floats, then one arrival
    // counter (int) per row stripe of the launch ((M/128 + 1) of them rounded up to 64: zero on entry, zero on exit), then
    // the partial sums [M][N/G][2] floats, G = 64 or 32 columns per group (group32: widths that are not a multiple of 128
    // columns, i.e. tiles with 96-column wave parts — MS-CLAP's HTSAT stages; a model uses ONE group size for a given N
    // whatever tile a batch size selects); hcopy = the [M, N] bf16 copy of the updated rows.  (Few kernel arguments on
    // purpose: they stay live in scalar registers across a loop whose inline-assembly loads need theirs.)
    // FOLD 2 keeps the residual stream itself as TWO bf16 arrays, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significand
    // bits, 4 bytes per element like the fp32 rows it replaces): hcopy = hi [M, N] — which IS the next GEMM's operand, so the
    // fold writes nothing extra — and lo = hcopy + lo_off elements.  The kernel's `out` argument is not used then.
    float* stats = nullptr;
    bf16_t* hcopy = nullptr;
    float eps = 1e-5f;
    int group32 = 0;
    int lo_off = 0;
};
__host__ __device__ inline size_t fold_count_slots(int M) { return (size_t)((M / 128 + 1 + 63) / 64 * 64); }
__host__ __device__ inline size_t fold_stats_bytes(int M, int N) { return (size_t)M * 4 + fold_count_slots(M) * 4 + (size_t)M * (N / 32) * 8; }

namespace w4 {

// In-kernel s_memtime stamps of block 0 and of the last block (tools/gemm_lab.hip builds with -DW4_STAMPS): slot 0 kernel
// entry, 1 first operands landed, 2 main loop done, 3 epilogue done.  Compiled out of the product.
#ifdef W4_STAMPS
__device__ unsigned long long g_w4p_stamps[4][4];
#define W4P_STAMP(t, k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (t) < 4) g_w4p_stamps[t][k] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_w4_stamps[2][4];
#define W4_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                                \
            g_w4_stamps[blockIdx.x == 0 ? 0 : 1][k] = __builtin_amdgcn_s_memtime();                              \
    } while (0)
#else
#define W4_STAMP(k) do {} while (0)
#define W4P_STAMP(t, k) do {} while (0)
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
// The compiler does not know these statements are MFMAs, so it pads no wait states between one and a read of its result,
// and it may move such a read (a plain register copy) up to right behind the statement.  After the loop: idle long enough
// for the last MFMA to retire, then pass every accumulator through a volatile statement — volatile statements keep their
// order, so every later read of the accumulators stays behind the idle slots.
// a 16-byte buffer load whose destination is an AGPR tuple (gfx90a+: vector memory instructions may target AGPRs): the
// residual prefetch parks values in the accumulator file's spare registers.  The compiler does not count it in vmcnt: the
// kernel's own counted waits include it, and s_waitcnt vmcnt(0) stands in front of the first use.
typedef unsigned rsrc_words_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load16_to_agpr(f32x4& dst, rsrc_words_t rsrc, int voff, int soff) {
    // (readfirstlane: under scalar-register pressure the allocator has handed this "s" operand a vector register)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(dst) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
__device__ __forceinline__ void mfma_retire() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ void pin_a(f32x4& acc) { asm volatile("" : "+a"(acc)); }
__device__ __forceinline__ void pin_v(f32x4& acc) { asm volatile("" : "+v"(acc)); }

// ---- the residual prefetch registers --------------------------------------------------------------------------------
// The 160 x 256 residual kernel parks the fp32 values its tile will be added to in a[160:255], the 96 accumulator-file
// registers its 40 accumulator tiles leave free, by buffer loads issued between the MFMAs of the last three K-steps.  The
// registers are NAMED in the assembly text and listed as clobbers of every MFMA statement of such a kernel, so the register
// allocator never places anything that lives across the loop there and never moves them.  (Round 3 handed the loads
// compiler-allocated "=a" tuples: the allocator was free to copy or re-home a tuple before its load had landed — it did,
// with the extra live values of the LayerNorm-fold epilogue, and silently returned stale residual rows.)
#define W4_HI_AGPRS "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
__device__ __forceinline__ void mfma16a_hi(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b) : W4_HI_AGPRS);
}
// tuple t (0..23) <- 16 bytes at rsrc + voff + soff.  Not counted by the compiler in vmcnt: the kernel's counted waits include it.
// s_nop 4: under scalar-register pressure the compiler keeps the row offsets in vector registers and the scalar operand comes
// out of a v_readfirstlane right in front of the statement — a VALU-writes-SGPR / VMEM-reads-it hazard (5 wait states) that
// nobody pads for an inline-assembly consumer: tuple 9 was loaded from tuple 8's address.
__device__ __forceinline__ void rpre_load_hilo(int t, rsrc_words_t rsrc, int voff, int soff_hi, int soff_lo);
__device__ __forceinline__ void rpre_load(int t, rsrc_words_t rsrc, int voff, int soff) {
    const int so = __builtin_amdgcn_readfirstlane(soff);
    switch (t) {
        case 0: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[160:163], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 1: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[164:167], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 2: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[168:171], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 3: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[172:175], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 4: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[176:179], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 5: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[180:183], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 6: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[184:187], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 7: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[188:191], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 8: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[192:195], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 9: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[196:199], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 10: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[200:203], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 11: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[204:207], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 12: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[208:211], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 13: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[212:215], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 14: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[216:219], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 15: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[220:223], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 16: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[224:227], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 17: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[228:231], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 18: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[232:235], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 19: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[236:239], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 20: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[240:243], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 21: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[244:247], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 22: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[248:251], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        case 23: asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[252:255], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(so) : "memory"); break;
        default: break;
    }
}
// the hi + lo form: 8 bytes of hi into the tuple's first two registers, 8 bytes of lo into the other two (TWO loads in vmcnt)
__device__ __forceinline__ void rpre_load_hilo(int t, rsrc_words_t rsrc, int voff, int soff_hi, int soff_lo) {
    const int sh = __builtin_amdgcn_readfirstlane(soff_hi), sl = __builtin_amdgcn_readfirstlane(soff_lo);
    switch (t) {
        case 0: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[160:161], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[162:163], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 1: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[164:165], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[166:167], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 2: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[168:169], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[170:171], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 3: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[172:173], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[174:175], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 4: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[176:177], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[178:179], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 5: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[180:181], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[182:183], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 6: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[184:185], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[186:187], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 7: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[188:189], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[190:191], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 8: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[192:193], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[194:195], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 9: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[196:197], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[198:199], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 10: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[200:201], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[202:203], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 11: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[204:205], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[206:207], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 12: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[208:209], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[210:211], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 13: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[212:213], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[214:215], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 14: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[216:217], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[218:219], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 15: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[220:221], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[222:223], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 16: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[224:225], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[226:227], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 17: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[228:229], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[230:231], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 18: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[232:233], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[234:235], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 19: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[236:237], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[238:239], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 20: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[240:241], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[242:243], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 21: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[244:245], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[246:247], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 22: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[248:249], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[250:251], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        case 23: asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[252:253], %0, %1, %2 offen\n\tbuffer_load_dwordx2 a[254:255], %0, %1, %3 offen" ::"v"(voff), "s"(rsrc), "s"(sh), "s"(sl) : "memory"); break;
        default: break;
    }
}
// tuple t after the wait that covers its load.  Every read lists the whole range as clobbered: nothing the allocator wants to
// keep (a spilled VGPR, say) may sit in a register that a LATER read still has to fetch — it took a[196:199] for a spill
// between two reads when only the wait carried the list.
__device__ __forceinline__ float4 rpre_read(int t) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    switch (t) {
        case 0: asm volatile("v_accvgpr_read_b32 %0, a160\n\tv_accvgpr_read_b32 %1, a161\n\tv_accvgpr_read_b32 %2, a162\n\tv_accvgpr_read_b32 %3, a163" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 1: asm volatile("v_accvgpr_read_b32 %0, a164\n\tv_accvgpr_read_b32 %1, a165\n\tv_accvgpr_read_b32 %2, a166\n\tv_accvgpr_read_b32 %3, a167" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 2: asm volatile("v_accvgpr_read_b32 %0, a168\n\tv_accvgpr_read_b32 %1, a169\n\tv_accvgpr_read_b32 %2, a170\n\tv_accvgpr_read_b32 %3, a171" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 3: asm volatile("v_accvgpr_read_b32 %0, a172\n\tv_accvgpr_read_b32 %1, a173\n\tv_accvgpr_read_b32 %2, a174\n\tv_accvgpr_read_b32 %3, a175" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 4: asm volatile("v_accvgpr_read_b32 %0, a176\n\tv_accvgpr_read_b32 %1, a177\n\tv_accvgpr_read_b32 %2, a178\n\tv_accvgpr_read_b32 %3, a179" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 5: asm volatile("v_accvgpr_read_b32 %0, a180\n\tv_accvgpr_read_b32 %1, a181\n\tv_accvgpr_read_b32 %2, a182\n\tv_accvgpr_read_b32 %3, a183" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 6: asm volatile("v_accvgpr_read_b32 %0, a184\n\tv_accvgpr_read_b32 %1, a185\n\tv_accvgpr_read_b32 %2, a186\n\tv_accvgpr_read_b32 %3, a187" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 7: asm volatile("v_accvgpr_read_b32 %0, a188\n\tv_accvgpr_read_b32 %1, a189\n\tv_accvgpr_read_b32 %2, a190\n\tv_accvgpr_read_b32 %3, a191" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 8: asm volatile("v_accvgpr_read_b32 %0, a192\n\tv_accvgpr_read_b32 %1, a193\n\tv_accvgpr_read_b32 %2, a194\n\tv_accvgpr_read_b32 %3, a195" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 9: asm volatile("v_accvgpr_read_b32 %0, a196\n\tv_accvgpr_read_b32 %1, a197\n\tv_accvgpr_read_b32 %2, a198\n\tv_accvgpr_read_b32 %3, a199" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 10: asm volatile("v_accvgpr_read_b32 %0, a200\n\tv_accvgpr_read_b32 %1, a201\n\tv_accvgpr_read_b32 %2, a202\n\tv_accvgpr_read_b32 %3, a203" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 11: asm volatile("v_accvgpr_read_b32 %0, a204\n\tv_accvgpr_read_b32 %1, a205\n\tv_accvgpr_read_b32 %2, a206\n\tv_accvgpr_read_b32 %3, a207" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 12: asm volatile("v_accvgpr_read_b32 %0, a208\n\tv_accvgpr_read_b32 %1, a209\n\tv_accvgpr_read_b32 %2, a210\n\tv_accvgpr_read_b32 %3, a211" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 13: asm volatile("v_accvgpr_read_b32 %0, a212\n\tv_accvgpr_read_b32 %1, a213\n\tv_accvgpr_read_b32 %2, a214\n\tv_accvgpr_read_b32 %3, a215" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 14: asm volatile("v_accvgpr_read_b32 %0, a216\n\tv_accvgpr_read_b32 %1, a217\n\tv_accvgpr_read_b32 %2, a218\n\tv_accvgpr_read_b32 %3, a219" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 15: asm volatile("v_accvgpr_read_b32 %0, a220\n\tv_accvgpr_read_b32 %1, a221\n\tv_accvgpr_read_b32 %2, a222\n\tv_accvgpr_read_b32 %3, a223" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 16: asm volatile("v_accvgpr_read_b32 %0, a224\n\tv_accvgpr_read_b32 %1, a225\n\tv_accvgpr_read_b32 %2, a226\n\tv_accvgpr_read_b32 %3, a227" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 17: asm volatile("v_accvgpr_read_b32 %0, a228\n\tv_accvgpr_read_b32 %1, a229\n\tv_accvgpr_read_b32 %2, a230\n\tv_accvgpr_read_b32 %3, a231" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 18: asm volatile("v_accvgpr_read_b32 %0, a232\n\tv_accvgpr_read_b32 %1, a233\n\tv_accvgpr_read_b32 %2, a234\n\tv_accvgpr_read_b32 %3, a235" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 19: asm volatile("v_accvgpr_read_b32 %0, a236\n\tv_accvgpr_read_b32 %1, a237\n\tv_accvgpr_read_b32 %2, a238\n\tv_accvgpr_read_b32 %3, a239" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 20: asm volatile("v_accvgpr_read_b32 %0, a240\n\tv_accvgpr_read_b32 %1, a241\n\tv_accvgpr_read_b32 %2, a242\n\tv_accvgpr_read_b32 %3, a243" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 21: asm volatile("v_accvgpr_read_b32 %0, a244\n\tv_accvgpr_read_b32 %1, a245\n\tv_accvgpr_read_b32 %2, a246\n\tv_accvgpr_read_b32 %3, a247" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 22: asm volatile("v_accvgpr_read_b32 %0, a248\n\tv_accvgpr_read_b32 %1, a249\n\tv_accvgpr_read_b32 %2, a250\n\tv_accvgpr_read_b32 %3, a251" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        case 23: asm volatile("v_accvgpr_read_b32 %0, a252\n\tv_accvgpr_read_b32 %1, a253\n\tv_accvgpr_read_b32 %2, a254\n\tv_accvgpr_read_b32 %3, a255" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : : W4_HI_AGPRS); break;
        default: break;
    }
    return r;
}

// v + (v of the lane the DPP control selects within its row of 16)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// bf16 epilogue of RI row tiles (RI <= 4) x NJ*16 columns: wave-private image of 16*RI rows, then a row-major walk in
// which every global store instruction writes whole row segments (16 bytes per lane, consecutive lanes consecutive)
template <int MODE, int MI, int NJ, int RI, int FOLD = 0>
__device__ __forceinline__ void epi_bf16_pass(const f32x4 (&acc)[MI][NJ], int i0, const float4 (&bv)[NJ], bf16_t* __restrict__ out,
                                              int N, int row0, int col0, int lane, unsigned my,
                                              const float* __restrict__ rstd = nullptr) {
    constexpr int RS = NJ * 32 + 16, CPR = NJ * 2;   // image row bytes, 16-byte chunks per row
    static_assert((RI * 16 * CPR) % 64 == 0, "walk covers the piece in whole wave instructions");
    const int l15 = lane & 15, g = lane >> 4;
    float rs[RI];
    if constexpr (FOLD == 1) {
#pragma unroll
        for (int ii = 0; ii < RI; ++ii) rs[ii] = rstd[row0 + (i0 + ii) * 16 + l15];
    }
#pragma unroll
    for (int j = 0; j < NJ; j += 2)
#pragma unroll
        for (int ii = 0; ii < RI; ++ii) {          // eight values at a time, stage by stage (act_apply_n)
            const f32x4 a = acc[i0 + ii][j], b = acc[i0 + ii][j + 1];
            float x[8];
            if constexpr (FOLD == 1) {
                const float r = rs[ii];
                x[0] = fmaf(a[0], r, bv[j].x); x[1] = fmaf(a[1], r, bv[j].y); x[2] = fmaf(a[2], r, bv[j].z); x[3] = fmaf(a[3], r, bv[j].w);
                x[4] = fmaf(b[0], r, bv[j + 1].x); x[5] = fmaf(b[1], r, bv[j + 1].y); x[6] = fmaf(b[2], r, bv[j + 1].z); x[7] = fmaf(b[3], r, bv[j + 1].w);
            } else {
                x[0] = a[0] + bv[j].x; x[1] = a[1] + bv[j].y; x[2] = a[2] + bv[j].z; x[3] = a[3] + bv[j].w;
                x[4] = b[0] + bv[j + 1].x; x[5] = b[1] + bv[j + 1].y; x[6] = b[2] + bv[j + 1].z; x[7] = b[3] + bv[j + 1].w;
            }
            act_apply_n<MODE, 8>(x);
            W4_LDS(u32x2_t, my + (ii * 16 + l15) * RS + (j * 16 + g * 4) * 2) = u32x2_t{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])};
            W4_LDS(u32x2_t, my + (ii * 16 + l15) * RS + ((j + 1) * 16 + g * 4) * 2) = u32x2_t{pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])};
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < RI * 16 * CPR / 64; ++t) {
        const int idx = t * 64 + lane, row = idx / CPR, c = idx % CPR;
        const u32x4_t v = W4_LDS(const u32x4_t, my + row * RS + c * 16);
#ifdef W4_PLAIN_STORES
        *reinterpret_cast<u32x4_t*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + c * 8) = v;
#else
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + c * 8));
#endif
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

// hi + lo residual rows: raw = {hi(c0,c1), hi(c2,c3), lo(c0,c1), lo(c2,c3)} (two bf16 per dword, first element low)
__device__ __forceinline__ float4 hilo_join(const float4 rawf) {
    const unsigned h0 = __float_as_uint(rawf.x), h1 = __float_as_uint(rawf.y), l0 = __float_as_uint(rawf.z), l1 = __float_as_uint(rawf.w);
    return make_float4(__uint_as_float(h0 << 16) + __uint_as_float(l0 << 16),
                       __uint_as_float(h0 & 0xffff0000u) + __uint_as_float(l0 & 0xffff0000u),
                       __uint_as_float(h1 << 16) + __uint_as_float(l1 << 16),
                       __uint_as_float(h1 & 0xffff0000u) + __uint_as_float(l1 & 0xffff0000u));
}
__device__ __forceinline__ void hilo_split(const float4 v, u32x2_t& hi, u32x2_t& lo) {
    hi = u32x2_t{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    lo = u32x2_t{pack_bf16x2(v.x - __uint_as_float(hi[0] << 16), v.y - __uint_as_float(hi[0] & 0xffff0000u)),
                 pack_bf16x2(v.z - __uint_as_float(hi[1] << 16), v.w - __uint_as_float(hi[1] & 0xffff0000u))};
}
template <int NJ, int RI>
__device__ __forceinline__ void resid_load_hilo(const bf16_t* __restrict__ hi, int lo_off, int N, int row0, int col0, int lane,
                                                float4 (&res)[RI * 16 * NJ * 4 / 64]) {
    constexpr int CPR = NJ * 4;
#pragma unroll
    for (int t = 0; t < RI * 16 * CPR / 64; ++t) {
        const int idx = t * 64 + lane, row = idx / CPR, c = idx % CPR;
        const bf16_t* p = hi + (size_t)(row0 + row) * N + col0 + c * 4;
        const u32x2_t h = *reinterpret_cast<const u32x2_t*>(p), l = *reinterpret_cast<const u32x2_t*>(p + lo_off);
        res[t] = make_float4(__uint_as_float(h[0]), __uint_as_float(h[1]), __uint_as_float(l[0]), __uint_as_float(l[1]));
    }
}

// `res_at(t)`: the residual values of walk step t (an array filled by resid_load a pass earlier, or — the 160 x 256 residual
// kernel — the prefetch registers, fetched one tuple at a time right where it is added, so that no copy of them lives in VGPRs)
template <int MODE, int MI, int NJ, int RI, int FOLD = 0, typename RES>
__device__ __forceinline__ void epi_f32_pass(const f32x4 (&acc)[MI][NJ], int i0, const float4 (&bv)[NJ], float* __restrict__ out,
                                             int N, int row0, int col0, int lane, unsigned my,
                                             RES&& res_at, bf16_t* __restrict__ hcopy = nullptr,
                                             float* __restrict__ part = nullptr, int group32 = 0, int lo_off = 0) {
    static_assert(FOLD != 2 || NJ == 4 || NJ == 6 || NJ == 8, "fold statistics: wave parts of 64, 96 or 128 columns");
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
        if (MODE == EPI_RESID) {
            const float4 r = FOLD == 2 ? hilo_join(res_at(t)) : res_at(t);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if constexpr (FOLD != 2) *reinterpret_cast<float4*>(out + (size_t)(row0 + i0 * 16 + row) * N + col0 + c * 4) = v;
        if constexpr (FOLD == 2) {
            // the new rows as hi + lo (hi is the next GEMM's operand), and the row's sums over this aligned 64-column group: 4
            // values in the lane, then the lanes of the group by an xor butterfly — one tree for every tile shape
            const size_t grow = (size_t)(row0 + i0 * 16 + row);
            u32x2_t hi2, lo2;
            hilo_split(v, hi2, lo2);
            *reinterpret_cast<u32x2_t*>(hcopy + grow * N + col0 + c * 4) = hi2;
            *reinterpret_cast<u32x2_t*>(hcopy + lo_off + grow * N + col0 + c * 4) = lo2;
            float s1 = ((v.x + v.y) + v.z) + v.w;
            float s2 = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
            // pairs, quads (quad permutes), the two quads of a half (row_half_mirror), the two halves (row_mirror): the xor
            // butterfly's tree on the VALU's data-parallel-primitive path — no LDS round trips
            s1 = dpp_add<0xB1>(s1); s2 = dpp_add<0xB1>(s2);
            s1 = dpp_add<0x4E>(s1); s2 = dpp_add<0x4E>(s2);
            s1 = dpp_add<0x141>(s1); s2 = dpp_add<0x141>(s2);
            if (NJ != 6 && !group32) { s1 = dpp_add<0x140>(s1); s2 = dpp_add<0x140>(s2); }     // 64-column groups: the two halves
            const int gsh = (NJ == 6 || group32) ? 5 : 6;
            if ((c & ((1 << (gsh - 2)) - 1)) == 0) {
                // agent-scope stores: written through to where every XCD reads the same value — no cache write-back fence later
                const int np = N >> gsh;
                float* pp = part + (grow * np + ((col0 + c * 4) >> gsh)) * 2;
                __hip_atomic_store(pp, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pp + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace w4

// Block tile (32*MI) x (32*NJ), waves 2 x 2, wave tile (16*MI) x (16*NJ); SA / SW = LDS slots of the activation / weight
// operand (3 + 2, 2 + 3 or 2 + 2).  M % (32*MI) == 0, N % (32*NJ) == 0, K % 64 == 0, K >= 192.
template <int MODE, int MI, int NJ, int SA, int SW, int OCC = 1, int FOLD = 0>
__global__ __launch_bounds__(256, OCC) void gemm_w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                         const float* __restrict__ bias, int M, int N, int K,
                                                         void* __restrict__ out, const FoldArgs fold) {
    using namespace w4;
    static_assert(FOLD == 0 || (FOLD == 1 && bf16_out(MODE)) || (FOLD == 2 && (MODE == EPI_RESID || MODE == EPI_F32)),
                  "fold: consumer / producer (EPI_F32: the stream's first rows, nothing to add to)");
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

#ifdef W4_NOREAD
#pragma unroll
    for (int j = 0; j < NJ; ++j) wf1[j] = wf0[j];
#pragma unroll
    for (int i = 0; i < MI; ++i) af1[i] = af0[i];
#endif
    // slot byte offsets of A(s), A(s+1), A(s+2) and W(s), W(s+1), W(s+2) (relative to the operand's first slot)
    unsigned oA0 = 0, oA1 = ASZ, oA2 = (SA == 3) ? 2 * ASZ : 0;
    unsigned oW0 = 0, oW1 = WSZ, oW2 = (SW == 3) ? 2 * WSZ : 0;
    int kb = 0;                          // byte offset of K-step s inside a row

    // Residual prefetch (EPI_RESID, 160 x 256 tiles): the fp32 values this tile will be added to are requested during the
    // last three K-steps — four 16-byte loads at the end of each of their six phases — into the 96 accumulator-file
    // registers the 40 accumulator tiles leave free: the epilogue's passes 0 (rows 0-31 of the wave) and 2 (rows 64-79).
    // The stamps of the plain epilogue show it as one HBM-bound burst of 8 bytes per output element (4 read + 4 written)
    // that nothing overlaps; with the reads under the loop it is the 4 written, plus pass 1's reads.
    constexpr bool RESPRE = MODE == EPI_RESID && MI == 5 && NJ == 8;
    rsrc_words_t rO = {0u, 0u, 0u, 0u};
    int rvoff = 0, rsoff = 0;
    constexpr int RBYTES = FOLD == 2 ? 2 : 4;       // bytes per residual element and array: hi / lo bf16, or fp32
    constexpr int RLOADS = FOLD == 2 ? 2 : 1;       // vector-memory loads per prefetched tuple
    if constexpr (RESPRE) {
        const unsigned long long ob = (unsigned long long)(uintptr_t)(FOLD == 2 ? (void*)fold.hcopy : out);
        rO = rsrc_words_t{(unsigned)ob, (unsigned)(ob >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
        rvoff = ((lane >> 5) * N + (lane & 31) * 4) * RBYTES;                  // walk order of epi_f32_pass / resid_load
        rsoff = ((m0 + wm * MI * 16) * N + n0 + wn * NJ * 16) * RBYTES;
    }

    // One phase: NM = MI*NJ MFMAs on (af, wf) in i-major order, with the phase's fragment reads (first the NJ weight
    // fragments, then the MI activation fragments: the order the next phase needs them) and DMAs placed between them:
    // read r in front of MFMA r*MPR, DMA d in front of MFMA 1 + d*MPD (DMAs 0..NDA-1 fetch the activation tile into
    // slot dA, the rest the weight tile into slot dW).
    auto phase = [&](const bf16x8 (&af)[MI], const bf16x8 (&wf)[NJ], bf16x8 (&afn)[MI], bf16x8 (&wfn)[NJ], auto read_c,
                     lds_cptr pw, lds_cptr pa, auto nda_c, auto ndw_c, unsigned dA, unsigned dW, int dkb, auto rb_c) {
#ifdef W4_NOREAD      // (lab timing knobs: W4_NOREAD / W4_NODMA / W4_NOBAR take one ingredient out of the loop; results are wrong)
        constexpr bool READ = false;
#else
        constexpr bool READ = decltype(read_c)::value;
#endif
        constexpr int RB = decltype(rb_c)::value;                // >= 0: residual loads RB .. RB+3 behind this phase's DMAs
#ifdef W4_NODMA
        constexpr int NDA = 0, NDW = 0, ND = 0;
#else
        constexpr int NDA = decltype(nda_c)::value, NDW = decltype(ndw_c)::value, ND = NDA + NDW;
#endif
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
            if constexpr (RESPRE && RB >= 0) {
                if (m >= NM - 8 && (m - (NM - 8)) % 2 == 0) {
                    const int t = RB + (m - (NM - 8)) / 2;           // 0..15: pass 0 (rows 2t, 2t+1); 16..23: pass 2 (rows 64 + ...)
                    const int row = t < 16 ? t * 2 : 64 + (t - 16) * 2;
                    if constexpr (FOLD == 2) rpre_load_hilo(t, rO, rvoff, rsoff + row * N * 2, rsoff + row * N * 2 + fold.lo_off * 2);
                    else rpre_load(t, rO, rvoff, rsoff + row * N * 4);
                }
            }
            if constexpr (RESPRE) mfma16a_hi(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
            else if (m < 64) mfma16a(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
            else mfma16v(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
        }
    };

    // one K-step; DMA: request step s+2; NEXT: read k-half 0 of step s+1; VM: the counted wait in front of the barrier
    // (-1: none); RA / RB: first residual load of phase A / B (-1: none)
    auto step = [&](auto dma_c, auto next_c, auto vm_c, auto ra_c, auto rb_c) {
        constexpr bool DMA = decltype(dma_c)::value;
        constexpr int VM = decltype(vm_c)::value;
        // ---- phase A: k-half 0 of step s; read k-half 1 of step s; request the three-slot operand of step s+2
        phase(af0, wf0, af1, wf1, std::true_type{}, (lds_cptr)(uintptr_t)(oW0 + fw1), (lds_cptr)(uintptr_t)(oA0 + fa1),
              std::integral_constant<int, (DMA && A_EARLY) ? PA : 0>{}, std::integral_constant<int, (DMA && W_EARLY) ? PW : 0>{},
              oA2, oW2, kb + 256, ra_c);
#ifndef W4_NOBAR
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (VM >= 0) wait_vmcnt<VM>();
        __builtin_amdgcn_s_barrier();
#endif
        // ---- phase B: k-half 1 of step s; read k-half 0 of step s+1; request the two-slot operand(s) of step s+2
        phase(af1, wf1, af0, wf0, next_c, (lds_cptr)(uintptr_t)(oW1 + fw0), (lds_cptr)(uintptr_t)(oA1 + fa0),
              std::integral_constant<int, (DMA && !A_EARLY) ? PA : 0>{}, std::integral_constant<int, (DMA && !W_EARLY) ? PW : 0>{},
              oA2, oW2, kb + 256, rb_c);
        const unsigned a0 = oA0, w0 = oW0;
        oA0 = oA1; oA1 = oA2; oA2 = (SA == 3) ? a0 : oA0;     // two slots: step s+3 goes where step s+1 sits
        oW0 = oW1; oW1 = oW2; oW2 = (SW == 3) ? w0 : oW0;
        kb += 128;
    };
    // with two slots X(s+2) shares the slot of X(s): oX2 must name it
    if (SA == 2) oA2 = oA0;
    if (SW == 2) oW2 = oW0;
    using T = std::true_type; using F = std::false_type;
    using N1 = std::integral_constant<int, -1>;
    using VMS = std::integral_constant<int, NDMA_A>;      // steady state: the DMAs of phase A may still be in flight
    using VM0 = std::integral_constant<int, 0>;
    if constexpr (RESPRE) {
        // the last three steps carry the residual loads (4 behind the DMAs of every phase).  Counted waits: step nk-3 its
        // phase-A DMAs + 4 loads; step nk-2 needs W(nk-1), requested in phase B of step nk-3 — the 4 loads issued behind
        // those DMAs and the 4 of its own phase A may still fly; step nk-1 needs nothing new.
        for (int s = 0; s < nk - 3; ++s) step(T{}, T{}, VMS{}, N1{}, N1{});
        step(T{}, T{}, std::integral_constant<int, NDMA_A + 4 * RLOADS>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        step(F{}, T{}, std::integral_constant<int, 8 * RLOADS>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 12>{});
        step(F{}, F{}, N1{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 20>{});
    } else {
        for (int s = 0; s < nk - 2; ++s) step(T{}, T{}, VMS{}, N1{}, N1{});
        step(F{}, T{}, VM0{}, N1{}, N1{});     // s = nk-2: nothing left to request; step nk-1 must have landed
        step(F{}, F{}, VM0{}, N1{}, N1{});     // s = nk-1
    }
    mfma_retire();
#pragma unroll
    for (int m = 0; m < MI * NJ; ++m) { if (m < 64) pin_a(acc[m / NJ][m % NJ]); else pin_v(acc[m / NJ][m % NJ]); }

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
        if constexpr (MI % 4 != 0) epi_bf16_pass<MODE, MI, NJ, MI % 4, FOLD>(acc, MI - MI % 4, bv, o, N, row0, col0, lane, my, fold.stats);
#pragma unroll
        for (int i0 = (MI / 4 - 1) * 4; i0 >= 0; i0 -= 4) epi_bf16_pass<MODE, MI, NJ, 4, FOLD>(acc, i0, bv, o, N, row0, col0, lane, my, fold.stats);
    } else {
        float* o = reinterpret_cast<float*>(out);
        constexpr int RP = NJ == 8 ? 2 : 1;            // row tiles per pass
        constexpr int NP = (MI + RP - 1) / RP, NRES = RP * 16 * NJ * 4 / 64;
        static_assert(MI % RP == 0 || RP == 2, "passes");
        float4 res[2][NRES];
        auto load = [&](int p, float4 (&r)[NRES]) {
            if (p * RP + RP <= MI) {
                if constexpr (FOLD == 2) resid_load_hilo<NJ, RP>(fold.hcopy, fold.lo_off, N, row0 + p * RP * 16, col0, lane, r);
                else resid_load<NJ, RP>(o, N, row0 + p * RP * 16, col0, lane, r);
            } else if constexpr (RP == 2) {
                float4 (&h)[NRES / 2] = reinterpret_cast<float4 (&)[NRES / 2]>(r);
                if constexpr (FOLD == 2) resid_load_hilo<NJ, 1>(fold.hcopy, fold.lo_off, N, row0 + p * RP * 16, col0, lane, h);
                else resid_load<NJ, 1>(o, N, row0 + p * RP * 16, col0, lane, h);
            }
        };
        if constexpr (RESPRE) {
            // passes 0 and 2 come out of the prefetch registers (everything requested has landed behind this wait);
            // pass 1 is loaded here, under pass 0's transposition
            load(1, res[1]);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES * RLOADS) : "memory", W4_HI_AGPRS);       // all but pass 1's loads
            epi_f32_pass<MODE, MI, NJ, 2, FOLD>(acc, 0, bv, o, N, row0, col0, lane, my, [](int t) { return rpre_read(t); }, fold.hcopy, fold.stats + M + fold_count_slots(M), fold.group32, fold.lo_off);
            epi_f32_pass<MODE, MI, NJ, 2, FOLD>(acc, 2, bv, o, N, row0, col0, lane, my, [&](int t) { return res[1][t]; }, fold.hcopy, fold.stats + M + fold_count_slots(M), fold.group32, fold.lo_off);
            epi_f32_pass<MODE, MI, NJ, 1, FOLD>(acc, 4, bv, o, N, row0, col0, lane, my, [](int t) { return rpre_read(16 + t); }, fold.hcopy, fold.stats + M + fold_count_slots(M), fold.group32, fold.lo_off);
        } else {
            if (MODE == EPI_RESID) load(0, res[0]);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (MODE == EPI_RESID && p + 1 < NP) load(p + 1, res[(p + 1) & 1]);
                const float4 (&rp)[NRES] = res[p & 1];
                if (p * RP + RP <= MI) epi_f32_pass<MODE, MI, NJ, RP, FOLD>(acc, p * RP, bv, o, N, row0, col0, lane, my, [&](int t) { return rp[t]; }, fold.hcopy, fold.stats + M + fold_count_slots(M), fold.group32, fold.lo_off);
                else if constexpr (RP == 2)
                    epi_f32_pass<MODE, MI, NJ, 1, FOLD>(acc, p * RP, bv, o, N, row0, col0, lane, my, [&](int t) { return rp[t]; }, fold.hcopy, fold.stats + M + fold_count_slots(M), fold.group32, fold.lo_off);
            }
        }
    }
    if constexpr (FOLD == 2) {
        // The stripe's statistics: every workgroup's partial sums are agent-scope stores (above); once they are acknowledged
        // (vmcnt 0) it bumps the stripe's counter, and the workgroup that finds all the others there reads the N/64 partials of
        // each row with agent-scope loads, adds them in column order and writes rstd.  No release / acquire FENCES: on this
        // chip they write back and invalidate a whole XCD's L2 (measured: +50 us per launch with 240 workgroups doing so).
        // The counter goes back to zero for the next launch.  (The flag sits at LDS offset 0: the kernel owns the whole
        // allocation by raw offsets, and every wave is past its use of the scratch there at the first barrier.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            int* count = reinterpret_cast<int*>(fold.stats + (size_t)M) + tm;
            const int seen = __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = seen == N / BNB - 1;
            W4_LDS(int, 0) = last;
            if (last) __hip_atomic_store(count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (W4_LDS(const int, 0)) {
            const int np = N >> ((NJ == 6 || fold.group32) ? 5 : 6);
            const float inv_n = 1.0f / (float)N;
            for (int r = threadIdx.x; r < BMB; r += 256) {
                const float* pr = fold.stats + (size_t)M + fold_count_slots(M) + (size_t)(m0 + r) * np * 2;
                float s1 = 0.f, s2 = 0.f;
                const unsigned long long* pq = reinterpret_cast<const unsigned long long*>(pr);
                for (int base = 0; base < np; base += 16) {       // 16 (sum, sum of squares) pairs per trip, all loads in flight
                    unsigned long long v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        v[k] = (base + k < np) ? __hip_atomic_load(pq + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (base + k < np) { s1 += __uint_as_float((unsigned)v[k]); s2 += __uint_as_float((unsigned)(v[k] >> 32)); }
                }
                const float mean = s1 * inv_n;
                const float var = fmaxf(fmaf(-mean, mean, s2 * inv_n), 0.f);
                fold.stats[m0 + r] = rsqrtf(var + fold.eps);
            }
        }
    }
    W4_STAMP(3);
}

// OCC = 2: two workgroups per CU (each still one wave per SIMD): their LDS (<= 80 KiB each) and registers (<= 256 per
// wave) must allow it; one workgroup's prologue and epilogue then run under the other's loop.
template <int MODE, int MI, int NJ, int SA, int SW, int OCC = 1, int FOLD = 0>
static void launch_w4(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, void* out, hipStream_t st,
                      const FoldArgs fold = FoldArgs()) {
    auto kern = gemm_w4_kernel<MODE, MI, NJ, SA, SW, OCC, FOLD>;
    static_assert(OCC == 1 || w4::lds_bytes(MI, NJ, SA, SW) <= 80 * 1024, "two workgroups per CU: 80 KiB of LDS each");
    constexpr int LDS = w4::lds_bytes(MI, NJ, SA, SW) > 4 * 20480 ? w4::lds_bytes(MI, NJ, SA, SW) : 4 * 20480;
    static PerDeviceOnce attr_set;
    attr_set([&] {
        raise_lds_limit(reinterpret_cast<const void*>(kern), LDS);
    });
    const int grid = (M / (32 * MI)) * (N / (32 * NJ));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, A, Wt, bias, M, N, K, out, fold);
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent form for bf16 outputs, 160 x 256 tiles (MI = 5, NJ = 8; 3 + 2 slots): a workgroup per CU walks tiles
// b, b + G, b + 2G, ... and runs their K-steps as ONE stream.  The stamps of the plain kernel (tools/gemm_lab.hip) show a
// tile's life as ~3.5k cycles waiting for its first operands, ~20k in the loop and 9-25k in an epilogue that is one
// HBM-bound store burst (every CU stores its C tile at the same moment) plus, with an activation, VALU work at half rate
// (one wave per SIMD).  Here
//   * the last two steps of a tile request steps 0 and 1 of the workgroup's NEXT tile (same slots, same counts), and the
//     last phase reads their first fragments: no prologue between tiles;
//   * after the last MFMA the accumulators are biased, activated and packed to bf16 into 80 VGPRs (the only exposed part
//     of the epilogue: VALU work without memory traffic), and the first MFMA phase of the next tile starts from C = 0;
//   * the packed tile then leaves DURING the next tile's first five K-steps: per step one 16-row tile goes through a
//     4-KiB wave-private LDS image — eight ds_write_b64 between the MFMAs of phase A, four ds_read_b128 + global_store
//     pairs between those of phase B — so the C traffic is spread over the loop instead of bursting behind it.
// Stores sit in phase B only, so the counted vmcnt in front of a barrier (the DMAs of phase A) is the plain kernel's.
// K % 64 == 0, K >= 512 (five drain steps + the two that request the next tile); M % 160 == 0, N % 256 == 0.
// ------------------------------------------------------------------------------------------------------------------
namespace w4 {

// a tile's first MFMA per accumulator: C = 0, no accumulator initialisation pass.  (Starting from the bias instead would
// save the epilogue's adds but round differently from every other kernel of the library, which add the bias to the
// finished sum: the same row must give the same bits whatever kernel its batch size selects.)
__device__ __forceinline__ void mfma16a_init(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ void tile_coords_v(int vb, int nwg, int tiles_m, int tiles_n, int group_m, int* tm, int* tn) {
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = group_m * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * group_m;
    const int gsize = min(tiles_m - first_m, group_m);
    const int in_group = bid - gid * per_group;
    *tm = first_m + in_group % gsize;
    *tn = in_group / gsize;
}

}  // namespace w4

template <int MODE, int FOLD = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                          const float* __restrict__ bias, int M, int N, int K,
                                                          bf16_t* __restrict__ out, const FoldArgs fold) {
    using namespace w4;
    static_assert(bf16_out(MODE), "packed drain: bf16 outputs");
    static_assert(FOLD == 0 || FOLD == 1, "persistent form: the consumer side of the LayerNorm fold");
    constexpr int MI = 5, NJ = 8, BMB = 160, BNB = 256;
    constexpr int ASZ = BMB * 128, WSZ = BNB * 128, WBASE = 3 * ASZ, SCR = WBASE + 2 * WSZ, SCRW = 16 * 272;
    constexpr int PA = BMB / 32, PW = BNB / 32;
    constexpr int NDRAIN = MI;          // K-steps over which a packed tile leaves (one row tile each)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_m = M / BMB, tiles_n = N / BNB, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int T = (ntiles - (int)blockIdx.x + G - 1) / G;     // this workgroup's tiles: blockIdx.x + r * G

    const int ldb = K * 2;
    const int voff = (lane >> 3) * ldb + (((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned fa0 = (wm * MI * 16 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fa1 = fa0 ^ 64;
    const unsigned fw0 = WBASE + (wn * NJ * 16 + l15) * 128 + ((g ^ (l15 & 7)) << 4), fw1 = fw0 ^ 64;
    const unsigned scr = SCR + wave * SCRW;
    const int nk = K / 64;

    int tm, tn;
    tile_coords_v(blockIdx.x, ntiles, tiles_m, tiles_n, 4, &tm, &tn);
    int m0 = tm * BMB, n0 = tn * BNB;
    auto rsrc_of = [&](const bf16_t* base, int row) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)row * K), 0, 0x7fffffff, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rA = rsrc_of(A, m0), rW = rsrc_of(Wt, n0);

    // prologue of the FIRST tile: steps 0 and 1
    stage_operand<BMB>(rA, 0, voff, ldb, 0, wave);
    stage_operand<BNB>(rW, WBASE, voff, ldb, 0, wave);
    stage_operand<BMB>(rA, ASZ, voff, ldb, 128, wave);
    stage_operand<BNB>(rW, WBASE + WSZ, voff, ldb, 128, wave);
    wait_vmcnt<PA + PW>();
    __builtin_amdgcn_s_barrier();

    f32x4 acc[MI][NJ];
    bf16x8 af0[MI], wf0[NJ], af1[MI], wf1[NJ];
    {
        lds_cptr pw = (lds_cptr)(uintptr_t)fw0, pa = (lds_cptr)(uintptr_t)fa0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf0[j] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + j * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + i * 2048);
    }
    unsigned oA0 = 0, oA1 = ASZ, oA2 = 2 * ASZ;       // slots of A(g), A(g+1), A(g+2) over the stream of steps g
    unsigned oW0 = 0, oW1 = WSZ, oW2 = 0;

    u32x2_t packed[MI][NJ];          // the previous tile, activated / rounded: lane holds (row i*16+l15, cols j*16+g*4..+3)
    // the packed tile's way out: buffer stores, a lane's voffset fixed (row g of a 4-row piece, its 16 bytes of the row),
    // piece and tile position in the scalar offset
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    const int vst = (g * N + l15 * 8) * 2;
    int sst = 0;                                                    // ((pm0 + wm*80) * N + pn0 + wn*128) * 2
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
    const lds_ptr scw = (lds_ptr)(uintptr_t)(scr + l15 * 272 + g * 8);   // image writes: + j*32
    const lds_cptr scrd = (lds_cptr)(uintptr_t)(scr + g * 272 + l15 * 16); // image reads: + t*4*272
    auto load_bias = [&](int ncol, f32x4 (&b)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float4 v = bias ? *reinterpret_cast<const float4*>(bias + ncol + wn * 128 + j * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[j] = f32x4{v.x, v.y, v.z, v.w};
        }
    };
    f32x4 bv[NJ];                    // the bias of the tile in flight (loaded in front of its last two steps)
    float rs[MI];                    // FOLD: rstd of the lane's five rows of that tile (loaded with the bias)

    // One phase: 40 MFMAs (INIT: the tile's first, C = 0) with the fragment reads of the next phase, the DMAs of the
    // phase, and DW / DR = the drain's LDS writes / read + store pairs of row tile DI of the packed tile between them.
    auto phase = [&](const bf16x8 (&af)[MI], const bf16x8 (&wf)[NJ], bf16x8 (&afn)[MI], bf16x8 (&wfn)[NJ], lds_cptr pw,
                     lds_cptr pa, auto init_c, auto nd_c, __amdgpu_buffer_rsrc_t rq, unsigned dslot, int dkb,
                     auto di_c, auto dw_c, auto dr_c) {
        constexpr bool INIT = decltype(init_c)::value, DW = decltype(dw_c)::value, DR = decltype(dr_c)::value;
        constexpr int ND = decltype(nd_c)::value, DI = decltype(di_c)::value;
        constexpr int NM = MI * NJ, NR = MI + NJ;
        constexpr int MPR = (NM * 3 / 4) / NR;                 // reads in front of MFMAs 0, 2, 4, ...
        constexpr int MPD = (NM * 3 / 4) / ND;
        u32x4_t piece[4];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m % MPR == 0 && m / MPR < NR) {
                const int r = m / MPR;
                if (r < NJ) wfn[r] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pw + r * 2048);
                else afn[r - NJ] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(pa + (r - NJ) * 2048);
            }
            if (m >= 1 && (m - 1) % MPD == 0 && (m - 1) / MPD < ND) {
                const int u = ((m - 1) / MPD) * 4 + wave;
                dma16(rq, dslot + u * 1024, voff, u * 8 * ldb + dkb);
            }
            if (DW && m % 4 == 3 && m / 4 < NJ)                 // eight 8-byte writes: the row tile's image, 272-byte rows
                *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(scw + (m / 4) * 32) = packed[DI][m / 4];
            if (DR && m % 8 == 3 && m / 8 < 4)                  // four 16-byte reads ...
                piece[m / 8] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(scrd + (m / 8) * 4 * 272);
            if (DR && m % 8 == 1 && m >= 9)                     // ... each stored six MFMAs later (m = 9: piece 0, ..., 33: piece 3)
                __builtin_amdgcn_raw_buffer_store_b128(piece[m / 8 - 1], rO, vst, sst + (DI * 16 + (m / 8 - 1) * 4) * N * 2, 2);
            if (INIT) mfma16a_init(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
            else mfma16a(acc[m / NJ][m % NJ], wf[m % NJ], af[m / NJ]);
        }
    };

    // K-step of the current tile; rqA / rqW, kq: where the request two steps down the stream goes (this tile, or the first
    // steps of the next one).  DI >= 0: row tile DI of the packed tile leaves in this step.
    auto step = [&](auto init_c, auto di_c, __amdgpu_buffer_rsrc_t rqA, __amdgpu_buffer_rsrc_t rqW, int kq) {
        constexpr int DI = decltype(di_c)::value;
        using DIc = std::integral_constant<int, DI < 0 ? 0 : DI>;
        using DOc = std::integral_constant<bool, (DI >= 0)>;
        phase(af0, wf0, af1, wf1, (lds_cptr)(uintptr_t)(oW0 + fw1), (lds_cptr)(uintptr_t)(oA0 + fa1), init_c,
              std::integral_constant<int, PA>{}, rqA, oA2, kq, DIc{}, DOc{}, std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<PA>();
        __builtin_amdgcn_s_barrier();
        phase(af1, wf1, af0, wf0, (lds_cptr)(uintptr_t)(oW1 + fw0), (lds_cptr)(uintptr_t)(oA1 + fa0), std::false_type{},
              std::integral_constant<int, PW>{}, rqW, WBASE + oW2, kq, DIc{}, std::false_type{}, DOc{});
        const unsigned a0 = oA0;
        oA0 = oA1; oA1 = oA2; oA2 = a0;
        oW0 = oW1; oW1 = oW2; oW2 = oW0;
    };
    // (two W slots: W(g+2) shares the slot of W(g))
    oW2 = oW0;

    for (int t = 0; t < T; ++t) {
        // the next tile of this workgroup (the last tile requests its own first steps again: valid addresses, dead slots)
        int m0n = m0, n0n = n0;
        if (t + 1 < T) {
            tile_coords_v((int)blockIdx.x + (t + 1) * G, ntiles, tiles_m, tiles_n, 4, &tm, &tn);
            m0n = tm * BMB; n0n = tn * BNB;
        }
        const __amdgpu_buffer_rsrc_t rAn = rsrc_of(A, m0n), rWn = rsrc_of(Wt, n0n);
        using F = std::false_type; using Tt = std::true_type;
        using NoDrain = std::integral_constant<int, -1>;
        W4P_STAMP(t, 0);
        if (t > 0) {
            step(Tt{}, std::integral_constant<int, 0>{}, rA, rW, 2 * 128);
            step(F{}, std::integral_constant<int, 1>{}, rA, rW, 3 * 128);
            step(F{}, std::integral_constant<int, 2>{}, rA, rW, 4 * 128);
            step(F{}, std::integral_constant<int, 3>{}, rA, rW, 5 * 128);
            step(F{}, std::integral_constant<int, 4>{}, rA, rW, 6 * 128);
        } else {
            step(Tt{}, NoDrain{}, rA, rW, 2 * 128);
            for (int s = 1; s < NDRAIN; ++s) step(F{}, NoDrain{}, rA, rW, (s + 2) * 128);
        }
        W4P_STAMP(t, 1);
        for (int s = NDRAIN; s < nk - 2; ++s) step(F{}, NoDrain{}, rA, rW, (s + 2) * 128);
        // the tile's bias: requested in front of the last two steps, so that the wait in front of its first use leaves the
        // DMAs of those steps — the next tile's first operands — in flight
        load_bias(n0, bv);
        if constexpr (FOLD == 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i) rs[i] = fold.stats[m0 + wm * 80 + i * 16 + l15];
        }
        step(F{}, NoDrain{}, rAn, rWn, 0);          // s = nk-2: step 0 of the next tile
        step(F{}, NoDrain{}, rAn, rWn, 128);        // s = nk-1: step 1 of the next tile
        W4P_STAMP(t, 2);
        mfma_retire();
#pragma unroll
        for (int m = 0; m < MI * NJ; ++m) pin_a(acc[m / NJ][m % NJ]);

        // bias, activation, round: the tile becomes 80 packed registers
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {       // eight values at a time, stage by stage (act_apply_n)
                float x[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (FOLD == 1) { x[r] = fmaf(acc[i][j][r], rs[i], bv[j][r]); x[4 + r] = fmaf(acc[i][j + 1][r], rs[i], bv[j + 1][r]); }
                    else { x[r] = acc[i][j][r] + bv[j][r]; x[4 + r] = acc[i][j + 1][r] + bv[j + 1][r]; }
                }
                act_apply_n<MODE, 8>(x);
                packed[i][j] = u32x2_t{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])};
                packed[i][j + 1] = u32x2_t{pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])};
            }
        sst = ((m0 + wm * 80) * N + n0 + wn * 128) * 2;
        m0 = m0n; n0 = n0n; rA = rAn; rW = rWn;
        W4P_STAMP(t, 3);
    }
    // the last tile leaves without a loop to hide behind
    wait_vmcnt<0>();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(scw + j * 32) = packed[i][j];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(scrd + t4 * 4 * 272);
            __builtin_amdgcn_raw_buffer_store_b128(v, rO, vst, sst + (i * 16 + t4 * 4) * N * 2, 2);
        }
    }
}

template <int MODE, int FOLD = 0>
static void launch_w4p(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, void* out, int num_cus,
                       hipStream_t st, const FoldArgs fold = FoldArgs()) {
    if constexpr (bf16_out(MODE)) {
        auto kern = gemm_w4p_kernel<MODE, FOLD>;
        constexpr int LDS = 3 * 160 * 128 + 2 * 256 * 128 + 4 * 16 * 272;
        static PerDeviceOnce attr_set;
        attr_set([&] {
            raise_lds_limit(reinterpret_cast<const void*>(kern), LDS);
        });
        const int ntiles = (M / 160) * (N / 256);
        const int grid = ntiles < num_cus ? ntiles : num_cus;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, A, Wt, bias, M, N, K, reinterpret_cast<bf16_t*>(out), fold);
    }
}

constexpr bool w4p_shape_ok(int M, int N, int K) { return M % 160 == 0 && N % 256 == 0 && K % 64 == 0 && K >= 512; }

constexpr bool w4_shape_ok(int M, int N, int K, int MI, int NJ = 8) {
    return M % (32 * MI) == 0 && N % (32 * NJ) == 0 && K % 64 == 0 && K >= 192;
}

}  // namespace wise
