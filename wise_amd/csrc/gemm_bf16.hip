// bf16 MFMA GEMM with fused epilogues for the ViT / HTSAT blocks (gfx950).
//
//   C[M,N] = epilogue( A[M,K] @ Wt[N,K]^T + bias[N] )      A, Wt bf16 row-major, fp32 accumulate
//
// Stands behind the four nn.Linear calls per transformer block that open_clip's
// VisionTransformer.forward issues from src/feature/mlfoundation_openclip.py:99.
// Roofline: MFMA-bound (2*M*N*K flop).
//
// Structure: 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 MFMA
// 16x16x32 accumulators.  Both operands go HBM/L2 -> LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4), double-buffered, one barrier per K-step.  The LDS image is
// lane-linear (a DMA's destination is base + lane*16) and XOR-swizzled through the per-lane SOURCE
// address: 16-byte chunk c of row r is stored at chunk c ^ (r & 7), which makes the ds_read_b128
// fragment reads conflict-free.  The product is computed transposed (weights as the MFMA A
// operand, activations as B) so that each lane ends up holding 4 consecutive output columns of one
// row: 8-byte bf16 / 16-byte fp32 epilogue accesses, and the bias is one float4 per lane.
#include "common.h"

namespace wise {

enum : int { EPI_BF16 = 0, EPI_QUICKGELU = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_F32 = 4 };

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per buffer

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// stage one 128x64 bf16 tile (rows row0.., k from k0) into a 16 KiB LDS tile
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int ld, int row0, int k0,
                                           unsigned char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = (t * 4 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);  // logical chunk held at physical chunk (lane&7)
        const bf16_t* src = G + (size_t)(row0 + r) * ld + k0 + c * 8;
        glds16(src, lds_tile + (t * 4 + wave) * 1024);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* lds_tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

__device__ __forceinline__ float act_quickgelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A,
                                                           const bf16_t* __restrict__ Wt,
                                                           const float* __restrict__ bias, int M, int N, int K,
                                                           void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // smem: [buf][A tile | W tile]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles
    const int tiles_n = N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage_tile(A, K, m0, 0, smem, wave, lane);
    stage_tile(Wt, K, n0, 0, smem + TILE_BYTES, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __syncthreads();  // waits vmcnt(0): tile kt landed; everyone done reading buffer cur^1
        if (kt + 1 < nk) {
            unsigned char* nb = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile(A, K, m0, (kt + 1) * BK, nb, wave, lane);
            stage_tile(Wt, K, n0, (kt + 1) * BK, nb + TILE_BYTES, wave, lane);
        }
        const unsigned char* At = smem + cur * 2 * TILE_BYTES;
        const unsigned char* Bt = At + TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = s * 4 + (lane >> 4);
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = lds_frag(At, wm * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = lds_frag(Bt, wn * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane&15)][n0 + wn*64 + j*16 + (lane>>4)*4 + r]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z,
                  v3 = acc[i][j][3] + bv.w;
            const size_t off = (size_t)m * N + n;
            if (MODE == EPI_RESID) {
                float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + off);
                float4 x = *p;
                x.x += v0; x.y += v1; x.z += v2; x.w += v3;
                *p = x;
            } else if (MODE == EPI_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + off) = make_float4(v0, v1, v2, v3);
            } else {
                if (MODE == EPI_QUICKGELU) {
                    v0 = act_quickgelu(v0); v1 = act_quickgelu(v1); v2 = act_quickgelu(v2); v3 = act_quickgelu(v3);
                } else if (MODE == EPI_GELU) {
                    v0 = act_gelu(v0); v1 = act_gelu(v1); v2 = act_gelu(v2); v3 = act_gelu(v3);
                }
                uint2 pk;
                pk.x = pack_bf16x2(v0, v1);
                pk.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + off) = pk;
            }
        }
    }
}

template <int MODE>
static void launch_gemm(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, void* out,
                        hipStream_t st) {
    auto kern = gemm_bf16_kernel<MODE>;
    const size_t lds = 4 * TILE_BYTES;  // 64 KiB
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    const int grid = (M / BM) * (N / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, A, Wt, bias, M, N, K, out);
}

int gemm_bf16(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, int mode, void* out,
              hipStream_t st) {
    WISE_CHECK_ARG(A && Wt && out, "gemm_bf16: null pointer");
    WISE_CHECK_ARG(M > 0 && M % BM == 0 && N > 0 && N % BN == 0 && K > 0 && K % BK == 0,
                   "gemm_bf16: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, BM, BN, BK);
    switch (mode) {
        case EPI_BF16: launch_gemm<EPI_BF16>(A, Wt, bias, M, N, K, out, st); break;
        case EPI_QUICKGELU: launch_gemm<EPI_QUICKGELU>(A, Wt, bias, M, N, K, out, st); break;
        case EPI_GELU: launch_gemm<EPI_GELU>(A, Wt, bias, M, N, K, out, st); break;
        case EPI_RESID: launch_gemm<EPI_RESID>(A, Wt, bias, M, N, K, out, st); break;
        case EPI_F32: launch_gemm<EPI_F32>(A, Wt, bias, M, N, K, out, st); break;
        default: set_error("gemm_bf16: unknown mode %d", mode); return WISE_E_INVALID;
    }
    WISE_LAUNCH_CHECK("gemm_bf16_kernel");
    return WISE_OK;
}

}  // namespace wise

extern "C" int wise_gemm_bf16(const uint16_t* A, const uint16_t* Wt, const float* bias, int M, int N, int K, int mode,
                              void* out, void* stream) {
    return wise::gemm_bf16(A, Wt, bias, M, N, K, mode, out, (hipStream_t)stream);
}
