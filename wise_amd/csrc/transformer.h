// Pieces of the pre-LN transformer shared by the image tower (vit.hip) and the text tower (text.hip).
#pragma once
#include "common.h"

namespace wise {

int gemm_bf16(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, int mode, void* out,
              hipStream_t st);
// the same when only the first m_valid of the M rows carry data (a single text query: 77 of 256): skinny problems go
// through a split-K pair of kernels instead of leaving most of the chip idle
// Split-K partials live in the CALLER's workspace: `sk` points at sk_bytes of device scratch private to the calling
// engine (and stream); nullptr / too small = the ordinary kernels.  gemm_splitk_bytes says what a call would use (0: it
// would not split), so that a tower sizes its workspace from its own GEMM shapes and the choice depends on the shape alone.
size_t gemm_splitk_bytes(int M, int m_valid, int N, int K);
int gemm_bf16_rows(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int m_valid, int N, int K, int mode,
                   void* out, hipStream_t st, float* sk = nullptr, size_t sk_bytes = 0);
// relu(conv3x3(X [B, T, F, Cin] bf16 NHWC, zero padding 1) + bias) as an implicit GEMM (Wt [Cout, 9*Cin] with
// k = (kh*3 + kw)*Cin + c): out [ceil256(B*T*F), Cout] bf16, or with pool its 2x2 average pooling (floor),
// out [B*(T/2)*(F/2), Cout], fused; zeros = 16 bytes of zeros on the device
int conv3x3_bf16(const bf16_t* X, const bf16_t* Wt, const float* bias, const bf16_t* zeros, int B, int T, int F, int Cin,
                 int Cout, bool pool, bf16_t* out, hipStream_t st);
// x[M,N] += A @ Wt^T + bias, then h = LN(x) (bf16) and, for post-LN blocks, x = LN(x): the residual GEMM of a block and the
// LayerNorm behind it; skinny calls fuse the split-K reduction with the LayerNorm (one launch less, same bits)
int gemm_resid_ln_rows(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int m_valid, int ln_rows, int N, int K,
                       float* x, const float* ln_w, const float* ln_b, float eps, bool post_ln, bf16_t* h, hipStream_t st,
                       float* sk = nullptr, size_t sk_bytes = 0);
// The LayerNorm fold (gemm_w4.h FoldArgs): out = act(rstd[row] * (A @ Wt^T) + bias) as bf16, and the residual GEMM over a
// residual stream kept as hi + lo (two bf16 arrays, lo = hi + lo_off elements: x = hi + lo; hi is the next such call's operand)
// that also writes the rows' rstd for it.  stats: gemm_fold_stats_bytes(M, N) bytes — rstd [M] first (what
// gemm_fold_bf16 takes), then the arrival counters (fold_count_slots(M) ints at byte offset 4 M: zero on entry), then partial
// sums.  group32: statistics kept per 32 columns instead of 64 (any width that is a multiple of 192 as well as of 128: ONE
// choice per model and width, so that every tile gives a row the same bits)
bool gemm_fold_shape_ok(int M, int N, int K);
int gemm_fold_bf16(const bf16_t* A, const bf16_t* Wt, const float* bias, const float* rstd, int M, int N, int K, int mode,
                   bf16_t* out, hipStream_t st);
size_t gemm_fold_stats_bytes(int M, int N);
int gemm_fold_resid(const bf16_t* A, const bf16_t* Wt, const float* bias, int M, int N, int K, bf16_t* hi, long long lo_off,
                    float* stats, float eps, hipStream_t st, int group32 = 0, bool accumulate = true /* false: x = ..., no rows read */);
size_t gemm_fold_counters_bytes(int M);
// hint for the tile heuristic: the calling THREAD is about to enqueue GEMMs on several streams that overlap in time
// (thread-local: two host threads driving two engines do not see each other's hint)
void gemm_set_overlapped(bool on);
int layernorm_f32_bf16(const float* x, const float* w, const float* b, int rows, int W, float eps, bf16_t* y,
                       hipStream_t st);
// head dim dh = 64, or 80 without a mask (ViT-H/14); causal = query t attends keys <= t (CLIP text tower)
// lens != null (dh 64, no causal mask): sequence b has lens[b] keys, the rest of its T rows are padding
int attention_bf16(const bf16_t* qkv, int B, int T, int H, bf16_t* o, hipStream_t st, bool causal, int dh = 64,
                   const int* lens = nullptr);
// LayerNorm with two outputs: xo fp32 (may alias x) and y bf16 — the post-LN blocks of the BERT-family text towers
int layernorm_f32_dual(const float* x, const float* w, const float* b, int rows, int W, float eps, float* xo, bf16_t* y,
                       hipStream_t st);

// where one residual block's weights sit: bf16 blob (in_proj [3W,W], out_proj [W,W], c_fc [F,W], c_proj [W,F])
// and fp32 blob (ln_1 w,b, in_proj bias, out_proj bias, ln_2 w,b, c_fc bias, c_proj bias); element offsets
struct BlockWeights {
    const bf16_t* wb; size_t per_layer_b, in_proj, out_proj, c_fc, c_proj;
    const float* pf; size_t per_layer_f, ln1_w, ln1_b, in_b, out_b, ln2_w, ln2_b, fc_b, proj_b;
};
int transformer_blocks(const BlockWeights& bw, int L, int W, int H, int F, int act, int batch, int T, bool causal,
                       float* x, bf16_t* h, bf16_t* qkv, bf16_t* a, hipStream_t st, float eps = 1e-5f,
                       bool skinny = false /*text towers: calls of <= 128 rows may take the split-K kernels*/,
                       float* sk = nullptr, size_t sk_bytes = 0 /*their partials: transformer_splitk_bytes() of workspace*/);
// split-K scratch a skinny transformer_blocks call of these dimensions uses (0 when it would not split)
size_t transformer_splitk_bytes(int W, int F, int batch, int T);
int l2norm_rows(const float* e, int rows, int D, float* out, hipStream_t st);
// LayerNorm of row pos[b] (or 0) of every sequence -> hb bf16 [batch, W]
int pooled_ln(const float* x, const float* ln_w, const float* ln_b, int batch, int T, int W, const int* pos,
              bf16_t* hb, hipStream_t st, float eps = 1e-5f);
int pooled_head(const float* x, const float* ln_w, const float* ln_b, const bf16_t* projT, int batch, int T, int W,
                int D, const int* pos, bf16_t* hb, float* e, float* out, hipStream_t st, float eps = 1e-5f,
                const float* proj_bias = nullptr /*[D]: text_projection as a Linear with bias (SigLIP)*/);

// x[M,C] += fc2(GELU(fc1(h))) in one kernel, the hidden activations never leaving the register file (mlp_stream.hip): C = 384
// (M % 128 == 0) or 192 (M % 256 == 0); ws = fc1 and fc2 as one stream in the kernel's order (wise_hip.h wise_mlp_stream)
bool mlp_stream_ok(int M, int C);
int mlp_stream(const bf16_t* h /* or null: the kernel applies LayerNorm(lnw, lnb, eps) to x itself */, const bf16_t* ws, const float* b1,
               const float* b2, float* x, int M, int C, hipStream_t st, const float* lnw = nullptr, const float* lnb = nullptr,
               float eps = 1e-5f);

// norm1 + QKV projection + window attention of a Swin block (HTSAT stages 2, 3) in one kernel (swin_stream.hip): o [B*H*H, C] bf16
bool swin_qkv_attn_ok(int C, int H);
int swin_qkv_attn(const float* x, const float* lnw, const float* lnb, float eps, const bf16_t* ws, const float* bq,
                  const float* relb, bf16_t* o, int B, int H, int C, int shift, hipStream_t st);

// msclap Projection head (htsat.hip): lat bf16 [Bp, d_in] -> out fp32 [B, 1024], L2-normalised
int clap_projection(const bf16_t* lat, const bf16_t* W1, const bf16_t* W2, const float* lw, const float* lb, int B,
                    int d_in, float* e, bf16_t* g, float* out, hipStream_t st);

namespace htsat {
void frontend_set_variant(int full_fft);   // (debug) 1: the full-length FFT kernel of rounds 1-2
// mel + bn0 of the first Fc frames of every clip (htsat_frontend.hip): melbn fp32 [B, Fc, 64]
int frontend(const float* wave, int B, int samples, int Fc, const float* hann, const float* mel_start,
             const float* mel_len, const float* mel_wt, const float* bn_scale, const float* bn_shift, float* melbn,
             hipStream_t st, int max_band = 36 /* FFT bins of the widest mel band in the table: 16 for fmax 8000, 35 for 14000 */);
constexpr int FRONT_MELW = 36;   // most FFT bins a mel band may span in the sparse table (2023 config: 16, 2022: 35)
}  // namespace htsat

}  // namespace wise
