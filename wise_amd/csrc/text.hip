// OpenCLIP text tower (`encode_text`) on gfx950 — the query side of HP-2.
//
// Replaces `self.model.encode_text(tokens)` + L2 normalise at src/feature/mlfoundation_openclip.py:103-108
// (arithmetic: open_clip 2.24.0 `CLIP.encode_text` / `TextTransformer`; the same computation as transformers'
// CLIPTextModelWithProjection, which is what the oracle is pinned against):
//   x = token_embedding[tokens] + positional_embedding            [B, 77, W] fp32
//   L pre-LN residual blocks with a causal attention mask          (shared with the image tower: vit.hip)
//   ln_final on the end-of-text row only: row argmax(tokens[b])   (LayerNorm is per row, so normalising just
//   @ text_projection [W, D] ; L2 normalise                         the pooled row is the same arithmetic)
// Tokenising is host work (wise_amd/feature/clip_tokenizer.py); this file takes int32 token ids.
//
// The same pipeline with three switches is the MS-CLAP caption encoder (src/feature/microsoft_clap.py:53-58:
// msclap 1.3.3 `TextEncoder` over a GPT-2 base): activation gelu_new (act = 2), pooling at the last token that is
// not the pad id 0 (pool = 1), and the msclap `Projection` head instead of a linear projection (head = 1).
#include <algorithm>

#include "transformer.h"

namespace wise {

// x[b*T + t, :] = tok_emb[tokens[b,t], :] + pos_emb[t, :] ; also eot[b] = first argmax_t tokens[b,t].
// One wave per token row; wave 0 of each sequence's first row finds the end-of-text position.
__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb, int B, int T, int W,
                                                         int vocab, int pool, float* __restrict__ x,
                                                         int* __restrict__ eot) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int b = row / T, t = row - b * T;
    int id = tokens[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // ids are validated on the host; stay in bounds regardless
    const float4* e = reinterpret_cast<const float4*>(tok_emb + (size_t)id * W);
    const float4* p = reinterpret_cast<const float4*>(pos_emb + (size_t)t * W);
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * W);
    for (int c = lane; c < (W >> 2); c += 64) {
        const float4 a = e[c], q = p[c];
        xr[c] = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
    }
    if (t == 0 && pool == 2) {
        if (lane == 0) eot[b] = T - 1;                 // SigLIP: the last position of the (padded) context
    } else if (t == 0 && pool == 1) {
        // msclap: sequence_lengths = ne(input_ids, 0).sum(-1) - 1 (right-padded with id 0)
        int cnt = 0;
        for (int tt = lane; tt < T; tt += 64) cnt += tokens[b * T + tt] != 0 ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (lane == 0) eot[b] = cnt > 0 ? cnt - 1 : T - 1;  // torch indexes -1 from the end when nothing is set
    } else if (t == 0) {
        // torch.argmax returns the first maximal index; T <= 128: two candidates per lane
        int best = -1, best_t = 0;
        for (int tt = lane; tt < T; tt += 64) {
            const int v = tokens[b * T + tt];
            if (v > best) { best = v; best_t = tt; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int ov = __shfl_xor(best, off, 64), ot = __shfl_xor(best_t, off, 64);
            if (ov > best || (ov == best && ot < best_t)) { best = ov; best_t = ot; }
        }
        if (lane == 0) eot[b] = best_t;
    }
}

struct TextDims {
    int T, V, W, L, H, F, D, act, pool, head, no_causal;
    float eps;
};
static int text_dims(const wise_text_config* c, TextDims* d) {
    WISE_CHECK_ARG(c, "text: null config");
    d->T = c->context; d->V = c->vocab; d->W = c->width; d->L = c->layers; d->H = c->heads; d->F = c->mlp;
    d->D = c->embed_dim;
    WISE_CHECK_ARG(d->T >= 1 && d->T <= 128, "text: context %d must be in [1,128]", d->T);
    WISE_CHECK_ARG(d->V >= 2, "text: vocab %d", d->V);
    WISE_CHECK_ARG(d->W > 0 && d->W % 128 == 0 && d->H * 64 == d->W && d->W <= 4096,
                   "text: width %d must be heads*64 and a multiple of 128", d->W);
    WISE_CHECK_ARG(d->F > 0 && d->F % 128 == 0, "text: mlp %d must be a multiple of 128", d->F);
    WISE_CHECK_ARG(d->D > 0 && d->D % 4 == 0 && d->L >= 0, "text: bad dims");
    WISE_CHECK_ARG(c->act >= 0 && c->act <= 2, "text: act must be 0 (quick_gelu), 1 (gelu) or 2 (gelu_new)");
    WISE_CHECK_ARG(c->pool >= 0 && c->pool <= 2 && c->head >= 0 && c->head <= 2, "text: pool/head must be 0, 1 or 2");
    WISE_CHECK_ARG(c->head != 1 || d->D == 1024, "text: the msclap Projection head is 1024 wide");
    WISE_CHECK_ARG((c->no_causal == 0 || c->no_causal == 1) && (c->eps_e6 == 0 || c->eps_e6 == 1), "text: no_causal / eps_e6 must be 0 or 1");
    d->act = c->act; d->pool = c->pool; d->head = c->head; d->no_causal = c->no_causal; d->eps = c->eps_e6 ? 1e-6f : 1e-5f;
    return WISE_OK;
}

struct TextOffsets {
    size_t layer0_b, per_layer_b, in_proj, out_proj, c_fc, c_proj, projT, proj2, total_b;              // bf16 blob
    size_t tok, pos, layer0_f, per_layer_f, ln1_w, ln1_b, in_b, out_b, ln2_w, ln2_b, fc_b, proj_b, lnf_w, lnf_b,
        pj_lw, pj_lb, total_f;                                                                         // fp32 blob
};
static TextOffsets text_offsets(const TextDims& d) {
    TextOffsets o;
    const size_t W = d.W, F = d.F;
    o.layer0_b = 0;
    o.in_proj = 0; o.out_proj = 3 * W * W; o.c_fc = o.out_proj + W * W; o.c_proj = o.c_fc + F * W;
    o.per_layer_b = o.c_proj + W * F;
    o.projT = o.per_layer_b * d.L;
    o.proj2 = o.projT + (size_t)d.D * W;                       // head 1: W2 [D,D] after W1 [D,W]
    o.total_b = o.proj2 + (d.head == 1 ? (size_t)d.D * d.D : 0);
    o.tok = 0; o.pos = (size_t)d.V * W; o.layer0_f = o.pos + (size_t)d.T * W;
    o.ln1_w = 0; o.ln1_b = W; o.in_b = 2 * W; o.out_b = 5 * W; o.ln2_w = 6 * W; o.ln2_b = 7 * W; o.fc_b = 8 * W;
    o.proj_b = o.fc_b + F;
    o.per_layer_f = o.proj_b + W;
    o.lnf_w = o.layer0_f + o.per_layer_f * d.L; o.lnf_b = o.lnf_w + W;
    o.pj_lw = o.lnf_b + W; o.pj_lb = o.pj_lw + d.D;            // head 1: LayerNorm of the Projection
    o.total_f = d.head == 1 ? o.pj_lb + d.D : (d.head == 2 ? o.pj_lw + d.D : o.pj_lw);   // head 2: the projection's bias [D]
    return o;
}

struct TextWs {
    size_t x, h, qkv, a, eot, sk, sk_bytes, total;
    int M, Mp;
};
static TextWs text_ws(const TextDims& d, int B) {
    TextWs w;
    w.M = B * d.T; w.Mp = (w.M + 255) / 256 * 256;
    const size_t Bp = (size_t)(B + 255) / 256 * 256;
    size_t off = 0;
    w.x = off; off += align_up((size_t)w.Mp * d.W * 4, 256);
    // h also holds the pooled rows [Bp, W]; qkv also holds the projected rows fp32 [Bp, D]
    w.h = off; off += align_up(std::max((size_t)w.Mp, Bp) * d.W * 2, 256);
    w.qkv = off; off += align_up(std::max((size_t)w.Mp * 3 * d.W * 2, Bp * d.D * 4), 256);
    w.a = off; off += align_up(std::max((size_t)w.Mp * d.F * 2, Bp * d.D * 2), 256);   // also the head's bf16 scratch
    w.eot = off; off += align_up((size_t)B * 4, 256);
    // split-K partials of a skinny call (one query: 77 rows), private to this workspace: graph- and stream-safe
    w.sk_bytes = transformer_splitk_bytes(d.W, d.F, B, d.T);
    w.sk = off; off += align_up(w.sk_bytes, 256);
    w.total = off;
    return w;
}

}  // namespace wise

using namespace wise;

extern "C" int wise_text_layout(const wise_text_config* cfg, int64_t* wb_elems, int64_t* pf_elems) {
    TextDims d;
    int rc = text_dims(cfg, &d);
    if (rc) return rc;
    const TextOffsets o = text_offsets(d);
    if (wb_elems) *wb_elems = (int64_t)o.total_b;
    if (pf_elems) *pf_elems = (int64_t)o.total_f;
    return WISE_OK;
}

extern "C" size_t wise_text_workspace_bytes(const wise_text_config* cfg, int batch) {
    TextDims d;
    if (text_dims(cfg, &d) || batch < 1) return 0;
    return text_ws(d, batch).total;
}

extern "C" int wise_text_forward(const wise_text_config* cfg, const uint16_t* wb, const float* pf, const int32_t* tokens,
                                 int batch, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    TextDims d;
    int rc = text_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(wb && pf && tokens && out, "text_forward: null pointer");
    WISE_CHECK_ARG(batch >= 1 && batch <= (1 << 20), "text_forward: batch=%d", batch);
    const TextWs ws = text_ws(d, batch);
    if (!workspace || workspace_bytes < ws.total) {
        set_error("text_forward: workspace %zu < %zu bytes", workspace_bytes, ws.total);
        return WISE_E_WORKSPACE;
    }
    WISE_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wb & 15) == 0 && ((uintptr_t)pf & 15) == 0,
                   "text_forward: workspace must be 256-byte and weight blobs 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const TextOffsets o = text_offsets(d);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    float* x = reinterpret_cast<float*>(wsb + ws.x);
    bf16_t* h = reinterpret_cast<bf16_t*>(wsb + ws.h);
    bf16_t* qkv = reinterpret_cast<bf16_t*>(wsb + ws.qkv);
    bf16_t* a = reinterpret_cast<bf16_t*>(wsb + ws.a);
    int* eot = reinterpret_cast<int*>(wsb + ws.eot);

    hipLaunchKernelGGL(text_embed_kernel, dim3((ws.M + 3) / 4), dim3(256), 0, st, tokens, pf + o.tok, pf + o.pos, batch,
                       d.T, d.W, d.V, d.pool, x, eot);
    WISE_LAUNCH_CHECK("text_embed_kernel");
    const BlockWeights bw = {wb + o.layer0_b, o.per_layer_b, o.in_proj, o.out_proj, o.c_fc, o.c_proj,
                             pf + o.layer0_f, o.per_layer_f, o.ln1_w, o.ln1_b, o.in_b, o.out_b, o.ln2_w, o.ln2_b,
                             o.fc_b, o.proj_b};
    if ((rc = transformer_blocks(bw, d.L, d.W, d.H, d.F, d.act, batch, d.T, d.no_causal == 0, x, h, qkv, a, st, d.eps, true,
                                 reinterpret_cast<float*>(wsb + ws.sk), ws.sk_bytes))) return rc;
    if (d.head == 0 || d.head == 2)
        return pooled_head(x, pf + o.lnf_w, pf + o.lnf_b, wb + o.projT, batch, d.T, d.W, d.D, eot, h,
                           reinterpret_cast<float*>(qkv), out, st, d.eps, d.head == 2 ? pf + o.pj_lw : nullptr);
    // msclap: ln_f on the pooled row -> Projection(W1, GELU, W2, LayerNorm(e1 + e2)) -> L2 normalise
    if ((rc = pooled_ln(x, pf + o.lnf_w, pf + o.lnf_b, batch, d.T, d.W, eot, h, st))) return rc;
    return clap_projection(h, wb + o.projT, wb + o.proj2, pf + o.pj_lw, pf + o.pj_lb, batch, d.W,
                           reinterpret_cast<float*>(qkv), a, out, st);
}

// parity tap: the residual stream x [batch*context, W] after a forward with the same batch
extern "C" int wise_text_tap_residual(const wise_text_config* cfg, int batch, const void* workspace, float* dst,
                                      void* stream) {
    TextDims d;
    int rc = text_dims(cfg, &d);
    if (rc) return rc;
    WISE_CHECK_ARG(workspace && dst && batch >= 1, "text_tap_residual: bad argument");
    const TextWs ws = text_ws(d, batch);
    hipError_t e = hipMemcpyAsync(dst, reinterpret_cast<const unsigned char*>(workspace) + ws.x,
                                  (size_t)ws.M * d.W * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("text_tap_residual: %s", hipGetErrorString(e)); return (int)e; }
    return WISE_OK;
}
