// OpenCLIP text towers that wrap a Hugging Face encoder (`HFTextEncoder`): XLM-RoBERTa — the text side of
// xlm-roberta-large-ViT-H-14/frozen_laion5b_s13b_b90k, the reference's default feature id (extract-features.py:192).
//
// Replaces `self.model.encode_text(tokens)` + L2 normalise at src/feature/mlfoundation_openclip.py:103-108 for that
// model (arithmetic: open_clip 2.24.0 hf_model.py `HFTextEncoder.forward` with pooler 'mean_pooler' and proj 'mlp' over
// transformers' XLMRobertaModel; restated in oracle/xlmr_text_ref.py, pinned to transformers):
//   mask = tokens != pad ; position = cumsum(mask) * mask + pad                      (right-padded batches)
//   x = LayerNorm(word_emb[tokens] + pos_emb[position] + type_emb[0])              [B, T, W] fp32
//   L POST-LN blocks: x = LN(x + Wo attn(x W_qkv^T + b) + bo) ; x = LN(x + W2 gelu(W1 x + b1) + b2),
//     attention bidirectional, padded keys masked
//   pooled = sum_t mask x / sum_t mask ; out = normalize( W_p2 gelu(W_p1 pooled) )   (both projections without bias)
// The same encoder family with other switches is MS-CLAP 2022's caption encoder (msclap TextEncoder over bert-base-uncased,
// reference call site src/feature/microsoft_clap.py:53-58): absolute positions 0..T-1 (pos_mode 1), LayerNorm eps 1e-12,
// the [CLS] row as the pooled vector (pool 1) and msclap's Projection as the head (head 1: e1 = W1 x, e2 = W2 gelu(e1),
// LayerNorm(e1 + e2), L2 normalise — clap_projection of htsat.hip); restated in oracle/clap_bert_ref.py.
// GEMMs, attention and LayerNorm are the image tower's kernels (vit.hip, gemm_bf16.hip): attention takes the
// per-sequence key count, LayerNorm writes the fp32 residual stream and the bf16 GEMM operand in one pass.
#include <algorithm>

#include "transformer.h"

namespace wise {

// x[row, :] = word[tokens[row]] + pos[position(row)] + type0 ; lens[b] = number of non-pad tokens.  Wave per row.
__global__ __launch_bounds__(256) void xlmr_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type0,
                                                         int B, int T, int W, int vocab, int max_pos, int pad,
                                                         int pos_abs, float* __restrict__ x, int* __restrict__ lens) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int b = row / T, t = row - b * T;
    int id = tokens[row];
    const bool live = id != pad;
    // cumsum(mask)[t] and sum(mask): T <= 128, two tokens per lane
    int upto = 0, all = 0;
    for (int tt = lane; tt < T; tt += 64) {
        const int m = tokens[b * T + tt] != pad ? 1 : 0;
        all += m;
        upto += (tt <= t) ? m : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { upto += __shfl_xor(upto, off, 64); all += __shfl_xor(all, off, 64); }
    int p = pos_abs ? t : (live ? upto + pad : pad);   // BERT: arange(T); RoBERTa: cumsum(mask) * mask + pad
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // ids are validated on the host; stay in bounds regardless
    const float4* e = reinterpret_cast<const float4*>(word + (size_t)id * W);
    const float4* q = reinterpret_cast<const float4*>(pos + (size_t)p * W);
    const float4* ty = reinterpret_cast<const float4*>(type0);
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * W);
    for (int c = lane; c < (W >> 2); c += 64) {
        const float4 a = e[c], d = q[c], f = ty[c];
        xr[c] = make_float4(a.x + d.x + f.x, a.y + d.y + f.y, a.z + d.z + f.z, a.w + d.w + f.w);
    }
    if (t == 0 && lane == 0) lens[b] = all;
}

// pooled[b, :] = sum_{t < lens[b]} x[b, t, :] / lens[b]  -> bf16 ; block per sequence, thread per 4 channels
// (first_only: the first row alone — BERT's [CLS] pooling)
__global__ __launch_bounds__(256) void xlmr_meanpool_kernel(const float* __restrict__ x, const int* __restrict__ lens, int T,
                                                            int W, int first_only, bf16_t* __restrict__ pooled) {
    const int b = blockIdx.x;
    const int n = first_only ? 1 : max(1, min(T, lens[b]));
    const float inv = 1.f / (float)n;
    for (int c = threadIdx.x; c < (W >> 2); c += 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < n; ++t) {
            const float4 v = reinterpret_cast<const float4*>(x + ((size_t)b * T + t) * W)[c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        uint2 pk;
        pk.x = pack_bf16x2(s.x * inv, s.y * inv);
        pk.y = pack_bf16x2(s.z * inv, s.w * inv);
        reinterpret_cast<uint2*>(pooled + (size_t)b * W)[c] = pk;
    }
}

struct XlmrDims {
    int T, V, P, W, L, H, F, Hd, D, pad, pos_abs, pool, head;
    float eps;
};
static int xlmr_dims(const wise_xlmr_config* c, XlmrDims* d) {
    WISE_CHECK_ARG(c, "xlmr: null config");
    d->T = c->context; d->V = c->vocab; d->P = c->max_positions; d->W = c->width; d->L = c->layers; d->H = c->heads;
    d->F = c->mlp; d->Hd = c->proj_hidden; d->D = c->embed_dim; d->pad = c->pad_id;
    d->pos_abs = c->pos_mode; d->pool = c->pool; d->head = c->head;
    d->eps = c->eps_e12 > 0 ? (float)((double)c->eps_e12 * 1e-12) : 1e-5f;
    WISE_CHECK_ARG((d->pos_abs == 0 || d->pos_abs == 1) && (d->pool == 0 || d->pool == 1) && (d->head == 0 || d->head == 1) &&
                       c->eps_e12 >= 0, "xlmr: pos_mode %d / pool %d / head %d / eps_e12 %d", c->pos_mode, c->pool, c->head, c->eps_e12);
    WISE_CHECK_ARG(d->head == 0 || (d->D == 1024 && d->Hd == d->D), "xlmr: the msclap projection head is W -> 1024 -> 1024 (got %d, %d)",
                   d->Hd, d->D);
    WISE_CHECK_ARG(d->T >= 1 && d->T <= 128, "xlmr: context %d must be in [1,128]", d->T);
    WISE_CHECK_ARG(d->V >= 4 && d->pad >= 0 && d->pad < d->V, "xlmr: vocab %d / pad id %d", d->V, d->pad);
    WISE_CHECK_ARG(d->P >= (c->pos_mode ? d->T : d->T + d->pad + 1), "xlmr: %d position rows cannot hold %d tokens (pad id %d)", d->P, d->T, d->pad);
    WISE_CHECK_ARG(d->W > 128 && d->W % 128 == 0 && d->H * 64 == d->W && d->W <= 4096,
                   "xlmr: width %d must be heads*64 and a multiple of 128", d->W);
    WISE_CHECK_ARG(d->F > 0 && d->F % 128 == 0 && d->Hd > 0 && d->Hd % 32 == 0, "xlmr: mlp %d must be a multiple of 128, projection hidden %d of 32",
                   d->F, d->Hd);
    WISE_CHECK_ARG(d->D > 0 && d->D % 4 == 0 && d->L >= 0, "xlmr: bad dims");
    return WISE_OK;
}

struct XlmrOffsets {
    size_t per_layer_b, qkv, out, fc1, fc2, proj1, proj2, total_b;                          // bf16 blob
    size_t word, pos, type0, eln_w, eln_b, layer0_f, per_layer_f, qkv_b, out_b, ln1_w, ln1_b, fc1_b, fc2_b, ln2_w, ln2_b,
        head_ln, total_f;                                                                   // fp32 blob
};
static XlmrOffsets xlmr_offsets(const XlmrDims& d) {
    XlmrOffsets o;
    const size_t W = d.W, F = d.F;
    o.qkv = 0; o.out = 3 * W * W; o.fc1 = o.out + W * W; o.fc2 = o.fc1 + F * W;
    o.per_layer_b = o.fc2 + W * F;
    o.proj1 = o.per_layer_b * d.L; o.proj2 = o.proj1 + (size_t)d.Hd * W;
    o.total_b = o.proj2 + (size_t)d.D * d.Hd;
    o.word = 0; o.pos = (size_t)d.V * W; o.type0 = o.pos + (size_t)d.P * W; o.eln_w = o.type0 + W; o.eln_b = o.eln_w + W;
    o.layer0_f = o.eln_b + W;
    o.qkv_b = 0; o.out_b = 3 * W; o.ln1_w = 4 * W; o.ln1_b = 5 * W; o.fc1_b = 6 * W; o.fc2_b = o.fc1_b + F;
    o.ln2_w = o.fc2_b + W; o.ln2_b = o.ln2_w + W;
    o.per_layer_f = o.ln2_b + W;
    o.head_ln = o.layer0_f + o.per_layer_f * d.L;                                           // head 1: LayerNorm w, b [D]
    o.total_f = o.head_ln + (d.head == 1 ? 2 * (size_t)d.D : 0);
    return o;
}

struct XlmrWs {
    size_t x, h, qkv, a, lens, sk, sk_bytes, total;
    int M, Mp;
};
static XlmrWs xlmr_ws(const XlmrDims& d, int B) {
    XlmrWs w;
    w.M = B * d.T; w.Mp = (w.M + 255) / 256 * 256;
    const size_t Bp = (size_t)(B + 255) / 256 * 256;
    size_t off = 0;
    w.x = off; off += align_up((size_t)w.Mp * d.W * 4, 256);
    w.h = off; off += align_up(std::max((size_t)w.Mp, Bp) * d.W * 2, 256);                  // also the pooled rows [Bp, W]
    w.qkv = off; off += align_up(std::max((size_t)w.Mp * 3 * d.W * 2, Bp * d.D * 4), 256); // also the projected rows fp32
    w.a = off; off += align_up(std::max((size_t)w.Mp * d.F * 2, Bp * d.Hd * 2), 256);       // also the projection's hidden rows
    w.lens = off; off += align_up((size_t)B * 4, 256);
    // split-K partials of a skinny call (one query) and of the head's two GEMMs, private to this workspace
    w.sk_bytes = transformer_splitk_bytes(d.W, d.F, B, d.T);
    w.sk_bytes = std::max(w.sk_bytes, gemm_splitk_bytes(256, 1, d.Hd, d.W));     // (any batch <= 128 splits alike)
    w.sk_bytes = std::max(w.sk_bytes, gemm_splitk_bytes(256, 1, d.D, d.Hd));
    w.sk = off; off += align_up(w.sk_bytes, 256);
    w.total = off;
    return w;
}

}  // namespace wise

using namespace wise;

extern "C" int wise_xlmr_layout(const wise_xlmr_config* cfg, int64_t* wb_elems, int64_t* pf_elems) {
    XlmrDims d;
    int rc = xlmr_dims(cfg, &d);
    if (rc) return rc;
    const XlmrOffsets o = xlmr_offsets(d);
    if (wb_elems) *wb_elems = (int64_t)o.total_b;
    if (pf_elems) *pf_elems = (int64_t)o.total_f;
    return WISE_OK;
}

extern "C" size_t wise_xlmr_workspace_bytes(const wise_xlmr_config* cfg, int batch) {
    XlmrDims d;
    if (xlmr_dims(cfg, &d) || batch < 1) return 0;
    return xlmr_ws(d, batch).total;
}

extern "C" int wise_xlmr_forward(const wise_xlmr_config* cfg, const uint16_t* wb, const float* pf, const int32_t* tokens,
                                 int batch, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    XlmrDims d;
    int rc = xlmr_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(wb && pf && tokens && out, "xlmr_forward: null pointer");
    WISE_CHECK_ARG(batch >= 1 && batch <= (1 << 20), "xlmr_forward: batch=%d", batch);
    const XlmrWs ws = xlmr_ws(d, batch);
    if (!workspace || workspace_bytes < ws.total) {
        set_error("xlmr_forward: workspace %zu < %zu bytes", workspace_bytes, ws.total);
        return WISE_E_WORKSPACE;
    }
    WISE_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wb & 15) == 0 && ((uintptr_t)pf & 15) == 0,
                   "xlmr_forward: workspace must be 256-byte and weight blobs 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const XlmrOffsets o = xlmr_offsets(d);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    float* x = reinterpret_cast<float*>(wsb + ws.x);
    bf16_t* h = reinterpret_cast<bf16_t*>(wsb + ws.h);
    bf16_t* qkv = reinterpret_cast<bf16_t*>(wsb + ws.qkv);
    bf16_t* a = reinterpret_cast<bf16_t*>(wsb + ws.a);
    int* lens = reinterpret_cast<int*>(wsb + ws.lens);
    float* sk = reinterpret_cast<float*>(wsb + ws.sk);
    const size_t skb = ws.sk_bytes;
    const int M = ws.M, Mp = ws.Mp, W = d.W;
    const float eps = d.eps;                                  // layer_norm_eps: 1e-5 XLM-RoBERTa, 1e-12 BERT

    hipLaunchKernelGGL(xlmr_embed_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tokens, pf + o.word, pf + o.pos, pf + o.type0,
                       batch, d.T, W, d.V, d.P, d.pad, d.pos_abs, x, lens);
    WISE_LAUNCH_CHECK("xlmr_embed_kernel");
    if ((rc = layernorm_f32_dual(x, pf + o.eln_w, pf + o.eln_b, M, W, eps, x, h, st))) return rc;
    for (int l = 0; l < d.L; ++l) {
        const bf16_t* lw = wb + o.per_layer_b * l;
        const float* lp = pf + o.layer0_f + o.per_layer_f * l;
        if ((rc = gemm_bf16_rows(h, lw + o.qkv, lp + o.qkv_b, Mp, M, 3 * W, W, 0, qkv, st, sk, skb))) return rc;
        if ((rc = attention_bf16(qkv, batch, d.T, d.H, h, st, false, 64, lens))) return rc;
        // x = LN(x + out(attention)) and x = LN(x + fc2(gelu(fc1 x))): the residual GEMM and the post-LN behind it as one call
        // (a single query: split-K partials + ONE kernel for reduction, residual and LayerNorm)
        if ((rc = gemm_resid_ln_rows(h, lw + o.out, lp + o.out_b, Mp, M, M, W, W, x, lp + o.ln1_w, lp + o.ln1_b, eps, true, h, st, sk, skb))) return rc;
        if ((rc = gemm_bf16_rows(h, lw + o.fc1, lp + o.fc1_b, Mp, M, d.F, W, 2, a, st, sk, skb))) return rc;
        if ((rc = gemm_resid_ln_rows(a, lw + o.fc2, lp + o.fc2_b, Mp, M, M, W, d.F, x, lp + o.ln2_w, lp + o.ln2_b, eps, true, h, st, sk, skb))) return rc;
    }
    // mean over the sequence's own tokens -> MLP projection (no biases) -> L2 normalise
    const int Bp = (batch + 255) / 256 * 256;
    if (d.head == 1) {   // rows past the batch feed the head's 128-row GEMM tiles: keep them finite
        hipError_t me = hipMemsetAsync(h, 0, (size_t)Bp * W * 2, st);
        if (me != hipSuccess) { set_error("xlmr_forward: hipMemsetAsync: %s", hipGetErrorString(me)); return (int)me; }
    }
    hipLaunchKernelGGL(xlmr_meanpool_kernel, dim3(batch), dim3(256), 0, st, x, lens, d.T, W, d.pool, h);
    WISE_LAUNCH_CHECK("xlmr_meanpool_kernel");
    float* e = reinterpret_cast<float*>(qkv);
    if (d.head == 1)
        return clap_projection(h, wb + o.proj1, wb + o.proj2, pf + o.head_ln, pf + o.head_ln + d.D, batch, W, e, a, out, st);
    if ((rc = gemm_bf16_rows(h, wb + o.proj1, nullptr, Bp, batch, d.Hd, W, 2, a, st, sk, skb))) return rc;
    if ((rc = gemm_bf16_rows(a, wb + o.proj2, nullptr, Bp, batch, d.D, d.Hd, 4, e, st, sk, skb))) return rc;
    return l2norm_rows(e, batch, d.D, out, st);
}

// parity tap: the residual stream x [batch*context, W] after a forward with the same batch
extern "C" int wise_xlmr_tap_residual(const wise_xlmr_config* cfg, int batch, const void* workspace, float* dst, void* stream) {
    XlmrDims d;
    int rc = xlmr_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(workspace && dst && batch >= 1, "xlmr_tap_residual: bad argument");
    const XlmrWs ws = xlmr_ws(d, batch);
    hipError_t e = hipMemcpyAsync(dst, reinterpret_cast<const unsigned char*>(workspace) + ws.x, (size_t)ws.M * d.W * 4,
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("xlmr_tap_residual: %s", hipGetErrorString(e)); return (int)e; }
    return WISE_OK;
}
