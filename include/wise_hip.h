/*
 * wise_hip.h — C ABI of libwise_hip.so: the MI355X (gfx950) hot paths of WISE.
 *
 * The reference (ox-vgg/wise) has no FFI of its own: its hot paths disappear into three Python
 * packages (open_clip, msclap, faiss).  Each entry point below replaces ONE such call and names the
 * reference call site it stands behind.  Everything is plain pointers + sizes: device pointers are
 * raw HBM addresses (any allocator: hipMalloc, torch), `stream` is a hipStream_t passed as void*.
 * No call allocates, frees or synchronises; all scratch comes from the caller's workspace, so every
 * entry point is hipGraph-capturable.
 *
 * Return value: 0 on success, otherwise a WISE_E_* code (<0) or a hipError_t (>0).
 * wise_last_error() returns a thread-local human-readable message for the last failure.
 */
#ifndef WISE_HIP_H
#define WISE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WISE_OK 0
#define WISE_E_INVALID (-1)   /* bad argument (shape, alignment, null pointer)          */
#define WISE_E_WORKSPACE (-2) /* workspace smaller than *_workspace_bytes() asks for    */
#define WISE_E_UNSUPPORTED (-3)

const char* wise_last_error(void);
/* ABI version of this header; bumped on any signature change (2: per-index counters for the two-stage search,
 * the shadow's error norm, wise_build_flags; 3: wise_vit_config.arch, wise_text_config.no_causal / eps_e6 — the SigLIP
 * towers — and the wise_xlmr_* entry points; 4: wise_xlmr_config.pos_mode / pool / head / eps_e12 — MS-CLAP 2022's BERT
 * caption encoder — and the wise_cnn14_* entry points; 5: wise_ip_shadow_i8 / wise_ip_topk_shadow8_f32 (int8 shadow,
 * norms[4]), wise_ip_topk_shadow_workspace_bytes depends on nq and returns 0 under 2^18 rows, two-stage k up to 1024; and,
 * added within 5: wise_vit_config.ln_fold, the wise_gemm_fold_* entry points, wise_attention_oproj_fold, wise_htsat_forward2,
 * wise_mlp_stream, wise_mlp_stream_ln, wise_swin_qkv_attn, the wise_ivf_* build entry points). */
int wise_abi_version(void);
/* Host-side hint for the GEMM tile heuristic (no device work), local to the CALLING THREAD: on != 0 while this thread
 * enqueues batches that will run beside another stream's (two batches in flight); tilings that measured slower there
 * are then avoided for the shapes the one-wave-per-SIMD kernels do not take.  wise_vit_forward sets it itself for its
 * half batches; the engines' forward_pipelined bracket their calls with it.  Results never depend on it. */
void wise_overlap_hint(int on);
/* The compiler flags the device code of this library was built with (wise_amd/build.py).  The product kernels must
 * be built without packed f32 VALU math ("-fno-slp-vectorize ... -packed-fp32-ops": DESIGN.md section 4, a gfx950
 * wait-state hazard); __graft_entry__.smoke() and the tests assert it on the library that is actually loaded. */
const char* wise_build_flags(void);
/* 1 when a HIP device of arch gfx950 is visible to the calling process, else 0. */
int wise_device_ok(void);

/* Measurement aid (bench.py): between wise_prof_begin and wise_prof_end every bf16 GEMM launch
 * (class 0, work = flop) and every IP-scan launch (class 1, work = algorithmic bytes N*d*4) is
 * bracketed by a HIP event pair recorded on the launch stream.  wise_prof_end synchronises and
 * fills three arrays of length 2: summed milliseconds, launch counts, summed work.  Not thread-safe. */
int wise_prof_begin(int capacity);
int wise_prof_end(double* ms_sum, int64_t* launches, double* work_sum);

/* ------------------------------------------------------------------------------------------------
 * HP-2  brute-force inner-product top-k  (replaces faiss IndexIDMap{IndexFlatIP}::search,
 *       reference call sites src/index/feature_search_index.py:113 and api/routes.py:1407)
 *
 * X   [N,d]  fp32 row-major database rows resident in HBM (what faiss stores: :47-52,:81)
 * Q   [nq,d] fp32 queries (device)
 * ids [N]    int64 external ids (IndexIDMap, :52,:81) or NULL => id = id_base + row
 * outD [nq,k] fp32 descending; outI [nq,k] int64; tail padded with (-3.4028235e38, -1) when N < k
 *            (faiss semantics relied on by search.py:142-143, routes.py:1411)
 * Ties: the row with the lower position wins (faiss leaves tie order unspecified).
 * Limits: d % 4 == 0, 4 <= d <= 2048, 1 <= k <= 2048, 1 <= nq <= 1024, X 16-byte aligned.
 * Batches: nq < 8 (or k > 16, d > 512, d % 32 != 0) runs the single-pass VALU scan per group of up to 4
 * queries; nq >= 8 runs passes of 32 (or 64) queries on the matrix cores, one pass over X per pass: for k <= 12
 * candidates are ranked by split-bf16 products (error <= 2^-16 sum|x_c q_c|) and the best 16 per query are
 * re-scored in fp32, for 12 < k <= 16 the products themselves are fp32.
 * Every returned score is an fp32 dot product (different summation orders between paths: scores agree to ~1e-6
 * relative, ids agree wherever neighbouring scores differ by more than that).
 * ---------------------------------------------------------------------------------------------- */
size_t wise_ip_topk_workspace_bytes(int64_t N, int d, int nq, int k);
int wise_ip_topk_f32(const float* X, int64_t N, int d, const float* Q, int nq, int k,
                     const int64_t* ids, int64_t id_base, float* outD, int64_t* outI,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The same search in two stages over a bf16 shadow copy of the rows (half the bytes of X per query), exact by
 * construction; first of all for ONE query, the reference's call shape (feature_search_index.py:113):
 *   wise_ip_shadow_bf16   Xb [N,d] bf16 (round-to-nearest-even copy of X) and two device floats norms[0] = max_r |X[r,:]|,
 *       norms[1] = max_r |X[r,:] - Xb[r,:]| (the largest rounding residual: a score computed from Xb differs from the
 *       exact one by at most eps = |q| norms[1] + f32 accumulation slack — Cauchy-Schwarz, no assumption on the data).
 *   wise_ip_topk_shadow_f32, one query (N >= 2^18; smaller indexes are answered by wise_ip_topk_f32 itself):
 *       (1) a sample of 64K rows (128 evenly spaced chunks) of Xb gives s_A, the k-th best approximate score seen;
 *           the exact k-th best score of the index is then >= s_A - eps, and a row of the exact top-k scores
 *           >= s_A - 2 eps approximately; (2) one pass over Xb collects EVERY row with approximate score >= s_A - 2 eps
 *           (typically 1-3 thousand of 10M; up to 262144); (3) the collected scores themselves give a sharper bound
 *           (their k-th largest slice maximum L: keep what reaches L - 2 eps, up to 16384 rows); (4) the scores of what
 *           is left are recomputed from the fp32 rows and the k best are written.  Nothing is certified after the fact
 *           and nothing depends on how the data is distributed (runs of near-duplicates — consecutive video frames —
 *           only make the lists longer).  (5) If a list overflows, the fp32 scan queued behind answers instead (it
 *           returns at once otherwise).
 *       All on the stream, no host round trip.
 * Any k <= 1024 for one query (the k WISE sends: REST `end` = 20, api/routes.py:1171,1407; evaluation k = 100 and
 * --topk 1000, docs/Search-Index-Evaluation.md:109, docs/Retrieval-Evaluation.md:39): the selections are radix selections,
 * and for k > 64 the collect pass runs in two ranges with the threshold tightened in between.
 * Two or more queries (k <= 12, d = 256 or 512; three or more with k <= 128 for d = 256 ... 1024, the fp32 VALU scan as the
 * gated fallback) run 128 / 64 / 32 queries at a time on the matrix cores in the same threshold form: the bf16 rows are MFMA
 * operands as loaded, every (query, row) that could matter is collected, re-scored in fp32 and the k best selected per
 * query; if any query's list overflows, the scan of the fp32 rows queued behind and gated redoes the pass.  Otherwise: one
 * query at a time.
 * counters (device, two int32, may be NULL): [0] += queries answered from the shadow, [1] += queries handed to the fp32
 * scan.  They belong to the caller (one pair per index); the library keeps no process-wide state for this.
 * Same outputs, ties and padding as wise_ip_topk_f32.  Limits: d % 8 == 0, 8 <= d <= 1024, k <= 1024, N >= 1, nq <= 1024. */
int wise_ip_shadow_bf16(const float* X, int64_t N, int d, uint16_t* Xb, float* norms /*[2]*/, void* stream);
size_t wise_ip_topk_shadow_workspace_bytes(int64_t N, int d, int nq, int k);
int wise_ip_topk_shadow_f32(const float* X, const uint16_t* Xb, const float* norms /*[2]*/, int64_t N, int d, const float* Q,
                            int nq, int k, const int64_t* ids, int64_t id_base, float* outD, int64_t* outI,
                            int32_t* counters, void* workspace, size_t workspace_bytes, void* stream);

/* The same two-stage search over an INT8 shadow: a quarter of the bytes of X per query (N (d + 4) instead of 4 N d).
 *   wise_ip_shadow_i8   Xq [N,d] int8 = round(X[r,:] / scales[r]), scales[r] = max|X[r,:]| / 127, and four device floats:
 *       norms[0] = max_r |scales[r] Xq[r,:]|, norms[1] = the score error bound per unit |q| (the largest quantisation
 *       residual max_r |X[r,:] - scales[r] Xq[r,:]|, in norms[2], plus sqrt(d) 1.6e-5 norms[0] for the query, which the
 *       scans take as two int8 pieces); no assumption on the data: Cauchy-Schwarz, as for the bf16 shadow.
 *   wise_ip_topk_shadow8_f32   one query at a time in the threshold form of wise_ip_topk_shadow_f32 — sample, threshold,
 *       collect, refine, exact re-scoring from the fp32 rows, gated fp32 scan — with the two scans on v_dot4_i32_i8
 *       (integer sums: exact).  The bound is ~4x the bf16 shadow's, so a few hundred rows are re-scored instead of a
 *       few dozen; the results are the same bits as wise_ip_topk_f32's.  Workspace: wise_ip_topk_shadow_workspace_bytes.
 * Limits: d % 16 == 0, 16 <= d <= 1024, k <= 1024, nq <= 1024 (answered one by one). */
int wise_ip_shadow_i8(const float* X, int64_t N, int d, int8_t* Xq, float* scales /*[N]*/, float* norms /*[4]*/, void* stream);
int wise_ip_topk_shadow8_f32(const float* X, const int8_t* Xq, const float* scales, const float* norms /*[4]*/, int64_t N, int d,
                             const float* Q, int nq, int k, const int64_t* ids, int64_t id_base, float* outD, int64_t* outI,
                             int32_t* counters, void* workspace, size_t workspace_bytes, void* stream);

/* IndexIVFFlat search, second stage (the first stage — the `nprobe` nearest centroids of each query — is
 * wise_ip_topk_f32 over the centroid table): scan the probed inverted lists and keep the k best.
 * Replaces faiss IndexIVFFlat::search as reached through self.index.search at
 * src/index/feature_search_index.py:113 for indexes built at :53-76 (nprobe set at api/routes.py:899-902).
 *   X        [N,d] fp32 rows grouped by list (list l occupies rows list_off[l] .. list_off[l+1]-1)
 *   list_off [nlist+1] int64;  ids [N] int64 external ids in the same order (NULL => the row number)
 *   probes   [nq,nprobe] int64 list numbers per query, entries < 0 are skipped
 * outD/outI as wise_ip_topk_f32 (descending scores, (-3.4028235e38, -1) padding).
 * Ties: the row that comes first in X wins. */
size_t wise_ivf_scan_workspace_bytes(int nq, int nprobe, int k);
int wise_ivf_scan_f32(const float* X, int64_t N, int d, const int64_t* list_off, int nlist, const int64_t* ids,
                      const float* Q, int nq, const int64_t* probes, int nprobe, int k, float* outD, int64_t* outI,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Dense scores [nq, N] = Q x X^T in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain per
 * score).  The coarse stage of IndexIVFFlat when nprobe is a sizeable part of nlist — the reference's nprobe = 1024
 * (config.py:19, api/routes.py:899-902) over 10 round(sqrt(N)) cells: all centroid scores, then wise_select_topk_f32.
 * X [N,d], Q [nq,d] fp32, 16-byte aligned, d % 4 == 0. */
int wise_ip_scores_f32(const float* X, int64_t N, int d, const float* Q, int nq, float* scores, void* stream);
/* Indices of the k largest entries of each row of scores [rows, n] fp32 (ties: lower index), written in ascending
 * index order, -1 padding when n < k.  The second half of the coarse stage (scores from wise_ip_scores_f32).  NaN
 * scores are not supported. */
int wise_select_topk_f32(const float* scores, int rows, int n, int k, int64_t* out, void* stream);
/* (ABI 5) Building an IndexIVFFlat on the device without a vendor kernel: the bookkeeping of `index.train(features)` (spherical
 * k-means) and of `index.add_with_ids(...)` / the grouping by list (src/index/feature_search_index.py:53-76; faiss Clustering,
 * IndexIVF.add).  All deterministic: the same inputs give the same bits run after run.
 *   wise_ivf_argmax: out[r] = argmax_c scores[r, c] (ties: the lowest c; a row of NaNs: 0).
 *   wise_ivf_group: order [n] = the row numbers grouped by list, STABLE (rows of a list in ascending row number), list_off
 *     [nlist + 1] their offsets, counts [nlist] (or null); assign values in [0, nlist), nlist <= 2^24; workspace of
 *     wise_ivf_group_workspace_bytes(n, nlist) bytes.
 *   wise_ivf_list_sums: sums[c, :] = the sum of x[order[i], :] over list c's entries, added in that order (d % 4 == 0).
 *   wise_ivf_normalize_rows: out[r, :] = in[r, :] / max(||in[r, :]||, 1e-20) (may be in place).
 *   wise_ivf_reseed: sums[empty[e], :] = sums[donor[e], :] * (1 + 1e-3 sign(.)).
 *   wise_ivf_gather_rows / _i64: out[i] = x[idx[i]] (rows of d floats, d % 4 == 0 / int64 scalars).
 *   wise_ivf_expand_lists: out[list_off[c] .. list_off[c + 1]) = c. */
int wise_ivf_argmax(const float* scores, int rows, int n, int64_t* out, void* stream);
size_t wise_ivf_group_workspace_bytes(int64_t n, int nlist);
int wise_ivf_group(const int64_t* assign, int64_t n, int nlist, int64_t* order, int64_t* list_off, int64_t* counts, void* workspace,
                   size_t workspace_bytes, void* stream);
int wise_ivf_list_sums(const float* x, const int64_t* order, const int64_t* list_off, int nlist, int d, float* sums, void* stream);
int wise_ivf_normalize_rows(const float* in, int rows, int d, float* out, void* stream);
int wise_ivf_reseed(float* sums, const int64_t* empty, const int64_t* donor, int n_empty, int d, void* stream);
int wise_ivf_gather_rows(const float* x, const int64_t* idx, int64_t n, int d, float* out, void* stream);
int wise_ivf_gather_i64(const int64_t* a, const int64_t* idx, int64_t n, int64_t* out, void* stream);
int wise_ivf_expand_lists(const int64_t* list_off, int nlist, int64_t* out, void* stream);

/* Merge `parts` partial top-k lists (e.g. one per GPU after the RCCL all-gather) into one.
 * inD [parts,nq,k] fp32, inI [parts,nq,k] int64 (entries with id -1 are padding) -> outD/outI [nq,k].
 * Ties: lower part index first, then the order within the part.  k <= 2048, parts*k <= 65536. */
int wise_topk_merge(const float* inD, const int64_t* inI, int parts, int nq, int k, float* outD,
                    int64_t* outI, void* stream);

/* IndexIDMap::reconstruct_batch (api/routes.py:1078): out[i,:] = X[row_of(ids_query[i]),:].
 * ids==NULL => row = id - id_base.  With ids given, a linear search kernel maps id -> row
 * (faiss does the same without a direct map).  Missing ids give a row of NaN. */
int wise_reconstruct_batch(const float* X, int64_t N, int d, const int64_t* ids, int64_t id_base,
                           const int64_t* query_ids, int n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HP-1  OpenCLIP VisionTransformer image tower (replaces model.encode_image + L2 normalise,
 *       reference call site src/feature/mlfoundation_openclip.py:99-100)
 *
 * Weights are two caller-owned device blobs in the fixed layout wise_vit_layout() reports:
 *   wb  bf16 : conv1 [W, Kp] (K = 3*P*P zero-padded to Kp = roundup(K,64)), then per layer
 *              in_proj [3W,W], out_proj [W,W], c_fc [F,W], c_proj [W,F], then proj^T [D,W]
 *   pf  fp32 : class_embedding [W], positional_embedding [T,W], ln_pre w,b [W],[W], then per layer
 *              ln_1 w,b, in_proj_bias [3W], out_proj bias [W], ln_2 w,b, c_fc bias [F], c_proj bias [W],
 *              then ln_post w,b
 * (names are open_clip 2.24.0 state-dict keys under `visual.`; see wise_amd/feature/vit_weights.py)
 * ---------------------------------------------------------------------------------------------- */
typedef struct wise_vit_config {
    int32_t image_size; /* S: 224 */
    int32_t patch;      /* P: 32 (ViT-B/32), 14 (ViT-L/14), 16 */
    int32_t width;      /* W: 768 / 1024 ; multiple of 128, head dim must be 64 */
    int32_t layers;     /* L */
    int32_t heads;      /* H = W/64 */
    int32_t mlp;        /* F: 4W */
    int32_t embed_dim;  /* D: 512 / 768 */
    int32_t act;        /* 0 = QuickGELU x*sigmoid(1.702x) (openai tags), 1 = erf GELU, 2 = GELU tanh form */
    int32_t arch;       /* 0 = open_clip VisionTransformer (class token, ln_pre, ln_post on the class row, linear projection);
                           1 = timm ViT as open_clip's SigLIP towers use it (`TimmModel`, pool 'map'): no class token, no
                           ln_pre, LayerNorm eps 1e-6, final norm over all tokens, attention-pool head (one latent query,
                           projection, y + mlp(norm(y))), no projection (embed_dim == width); u8 input is normalised with
                           mean = std = 0.5.  Weight layout of arch 1: wise_amd/feature/siglip.py:pack_siglip_vision */
    int32_t ln_fold;    /* (ABI 5) 0 = LayerNorm kernels between the GEMMs; 1 (arch 0 only) = the blocks' LayerNorms folded into
                           the GEMMs around them: the residual stream is kept as bf16 hi + lo, the residual GEMMs also emit the
                           rows' 1/sqrt(var + eps), the QKV / fc1 GEMMs take hi as their operand and scale their rows by it.  The blob layout is the same, but the packer must then store
                           in_proj / c_fc as gamma-scaled, row-centred weights and their biases as b + W beta
                           (wise_amd/feature/vit.py:pack_weights); ln_1 / ln_2 slots are not read.  Same results within the
                           bf16 path's tolerance (not bit-equal to ln_fold = 0).
                           2 = 1 with a block's attention, out-projection, residual add and row statistics in ONE kernel
                           (wise_attention_oproj_fold: up to 64 tokens, 12 heads of 64 — ViT-B/32); bit-equal to ln_fold = 1. */
} wise_vit_config;

#define WISE_VIT_IN_F32 0  /* images [B,3,S,S] fp32, already normalised (preprocess_image output) */
#define WISE_VIT_IN_U8 1   /* images [B,3,S,S] uint8 0..255; (x/255-mean)/std fused in the patch gather */

/* element counts of the two blobs and the byte offset table (for packers); returns 0 or WISE_E_* */
int wise_vit_layout(const wise_vit_config* cfg, int64_t* wb_elems, int64_t* pf_elems);
size_t wise_vit_workspace_bytes(const wise_vit_config* cfg, int batch);
/* out [B,D] fp32, rows L2-normalised without epsilon (mlfoundation_openclip.py:100). */
int wise_vit_forward(const wise_vit_config* cfg, const uint16_t* wb, const float* pf,
                     const void* images, int in_kind, int batch, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same forward kept entirely on `stream` (wise_vit_forward splits a batch of >= 64 images into two halves on two
 * internal streams).  For callers that pipeline whole batches themselves — two in flight, each with its own stream and
 * workspace; workspace_bytes as reported by wise_vit_workspace_bytes. */
int wise_vit_forward_single(const wise_vit_config* cfg, const uint16_t* wb, const float* pf, const void* images,
                            int in_kind, int batch, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* Debug/parity tap: copy the residual stream x [B*T, W] fp32 as it stands after `after_layer`
 * blocks (0 = after ln_pre) from the workspace of the LAST forward into dst. */
int wise_vit_tap_residual(const wise_vit_config* cfg, int batch, const void* workspace, float* dst,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * HP-1 audio  MS-CLAP (version 2023) HTSAT audio encoder + projection (replaces
 *       self.model.clap.audio_encoder(x)[0] + L2 normalise, reference call site
 *       src/feature/microsoft_clap.py:49-50).  The architecture is fixed (msclap config_2023:
 *       n_fft 1024, hop 320, 64 mel bands 50..8000 Hz at sr 44100, Swin depths 2-2-6-2, dims 96..768,
 *       heads 4..32, window 8, projection 768->1024), so there is no config struct.
 * wave [B, samples] fp32 (what preprocess_audio yields after the reshape at microsoft_clap.py:47-48);
 * only the first 1024 STFT frames (6.8 s at hop 320) enter the model.  out [B,1024] fp32, unit rows.
 * Weight blobs: layout in wise_amd/feature/htsat.py:pack_htsat_weights (msclap state-dict keys).
 * ---------------------------------------------------------------------------------------------- */
int wise_htsat_layout(int64_t* wb_elems, int64_t* pf_elems);
size_t wise_htsat_workspace_bytes(int batch, int samples);
int wise_htsat_forward(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                       float* out, void* workspace, size_t workspace_bytes, void* stream);
/* (ABI 5) wise_htsat_forward with flags.  bit 0: the LayerNorms of stages 2 - 4 folded into the GEMMs around them (the
 * residual stream of those stages as bf16 hi + lo; see wise_gemm_fold_resid) — the packer must then store the folded qkv / fc1
 * weights and biases of those stages (wise_amd/feature/htsat.py:pack_htsat_weights(fold=True)).  bit 1: the MLP of every block
 * of stages 2 and 3 through wise_mlp_stream (one kernel, hidden activations in registers) — the packer must then store those
 * blocks' fc1 + fc2 slots as that kernel's weight stream (pack_htsat_weights(mlp_stream=True)).  bit 2: norm1 + QKV projection
 * + window attention of every block of stages 2 and 3 through wise_swin_qkv_attn — the packer must then store those blocks'
 * qkv weight and bias slots as that kernel's stream (pack_htsat_weights(attn_stream=True)).  Bit 0 excludes bits 1 and 2.
 * flags 0 = wise_htsat_forward. */
int wise_htsat_forward2(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                        float* out, void* workspace, size_t workspace_bytes, int flags, void* stream);
/* parity tap after a forward with the same (batch, samples): what 0 = BatchNorm'd log-mel
 * [B*frames,64], 1 = residual stream (fp32 rows), 2 = the last stage's residual stream of a flags-bit-0 forward (hi + lo,
 * returned as fp32); copies `count` floats. */
int wise_htsat_tap(int what, const void* workspace, int batch, int samples, float* dst, int64_t count,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * HP-1 audio  MS-CLAP (version 2022) audio encoder: PANNs Cnn14 + projection — the same reference call site
 *       (src/feature/microsoft_clap.py:49-50) when the feature id's version token is '2022'
 *       (:20-31: every key of msclap's CLAP.model_name is accepted).  Fixed architecture (msclap config_2022:
 *       n_fft 1024, hop 320, 64 mel bands 50..14000 Hz at sr 44100; six ConvBlocks 64..2048 channels of two
 *       3x3 convolutions + BatchNorm + ReLU, 2x2 average pooling after the first five; mean over mel, max + mean over
 *       time; fc1 2048->2048 + ReLU; projection 2048->1024).  wave [B, samples] fp32, any length of at least 32 STFT
 *       frames (the whole clip enters the model).  out [B,1024] fp32, unit rows.
 * Weight blobs: layout in wise_amd/feature/cnn14.py:pack_cnn14_weights (msclap state-dict keys; BatchNorm folded).
 * ---------------------------------------------------------------------------------------------- */
int wise_cnn14_layout(int64_t* wb_elems, int64_t* pf_elems);
size_t wise_cnn14_workspace_bytes(int batch, int samples);
int wise_cnn14_forward(const uint16_t* wb, const float* pf, const float* wave, int batch, int samples,
                       float* out, void* workspace, size_t workspace_bytes, void* stream);
/* parity tap after a forward with the same (batch, samples): what 0 = BatchNorm'd log-mel fp32 [B*frames*64],
 * 1 = pooled latent bf16 [B*2048], 2 = fc1 output bf16 [B*2048]; copies `bytes` bytes. */
int wise_cnn14_tap(int what, const void* workspace, int batch, int samples, void* dst, int64_t bytes,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * HP-2 query side: OpenCLIP text tower — replaces `self.model.encode_text(tokens)` + L2
 *       normalise, reference call site src/feature/mlfoundation_openclip.py:103-108 (reached from
 *       FeatureSearchIndex.search, src/index/feature_search_index.py:112).  Token ids in, unit vectors out;
 *       tokenising stays on the host (wise_amd/feature/clip_tokenizer.py).
 *
 *   wb  bf16 : per layer in_proj [3W,W], out_proj [W,W], c_fc [F,W], c_proj [W,F]; then text_projection^T [D,W]
 *   pf  fp32 : token_embedding [V,W], positional_embedding [T,W]; per layer ln_1 w,b, in_proj_bias [3W],
 *              out_proj bias [W], ln_2 w,b, c_fc bias [F], c_proj bias [W]; then ln_final w,b
 * (open_clip 2.24.0 state-dict keys without the `visual.` prefix; see wise_amd/feature/text.py)
 * tokens int32 [batch, context] (device): <start_of_text> ... <end_of_text> 0 0 ...; the pooled row is
 * argmax(tokens[b]) as in open_clip (the end-of-text id is the largest id of the vocabulary).
 * The MS-CLAP caption encoder (src/feature/microsoft_clap.py:53-58: GPT-2 base + msclap Projection) is the same
 * pipeline with act = 2, pool = 1, head = 1; then wb ends with W1 [1024,W], W2 [1024,1024] and pf with the
 * Projection's LayerNorm w,b [1024] after ln_f.
 * ---------------------------------------------------------------------------------------------- */
typedef struct wise_text_config {
    int32_t context;    /* T: 77 */
    int32_t vocab;      /* V: 49408 */
    int32_t width;      /* W: 512 (ViT-B/32), 768 (ViT-L/14); multiple of 128, head dim 64 */
    int32_t layers;     /* L: 12 */
    int32_t heads;      /* H = W/64 */
    int32_t mlp;        /* F = 4W */
    int32_t embed_dim;  /* D: 512 / 768 (1024 with head = 1) */
    int32_t act;        /* 0 = QuickGELU, 1 = erf GELU, 2 = gelu_new (tanh form, GPT-2) */
    int32_t pool;       /* pooled row: 0 = first argmax of the ids (open_clip), 1 = last id != 0 (msclap, pad id 0),
                           2 = the last position of the context (open_clip pool_type 'last': SigLIP) */
    int32_t head;       /* 0 = ln_final + linear projection; 1 = ln_f + msclap Projection (W1, GELU, W2, LayerNorm);
                           2 = ln_final + Linear WITH bias (SigLIP: pf ends with the bias [D] after ln_final w,b) */
    int32_t no_causal;  /* 0 = causal mask (CLIP, GPT-2); 1 = none (open_clip no_causal_mask: SigLIP) */
    int32_t eps_e6;     /* LayerNorm epsilon: 0 = 1e-5, 1 = 1e-6 (SigLIP norm_kwargs) */
} wise_text_config;
int wise_text_layout(const wise_text_config* cfg, int64_t* wb_elems, int64_t* pf_elems);
size_t wise_text_workspace_bytes(const wise_text_config* cfg, int batch);
int wise_text_forward(const wise_text_config* cfg, const uint16_t* wb, const float* pf, const int32_t* tokens,
                      int batch, float* out /* [batch, D] fp32, L2-normalised */, void* workspace,
                      size_t workspace_bytes, void* stream);
/* parity tap: residual stream fp32 [batch*context, W] of the last forward with the same batch */
int wise_text_tap_residual(const wise_text_config* cfg, int batch, const void* workspace, float* dst, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HP-2 query side for open_clip models whose text tower wraps a Hugging Face encoder (`HFTextEncoder`):
 *       XLM-RoBERTa with the mean pooler and the two-layer MLP projection — the text tower of
 *       xlm-roberta-large-ViT-H-14/frozen_laion5b_s13b_b90k, the reference's DEFAULT feature id
 *       (extract-features.py:192; reached from src/feature/mlfoundation_openclip.py:103-108 like wise_text_forward).
 *       tokens int32 [batch, context], right-padded with pad_id (the HF tokenizer's padding='max_length').
 * Weight layout (host packer: wise_amd/feature/xlmr_text.py):
 *   wb bf16: per layer  W_qkv [3W,W] (query|key|value), W_out [W,W], W_fc1 [F,W], W_fc2 [W,F];  then the projection
 *            W_p1 [Hd,W], W_p2 [D,Hd]  (open_clip 'mlp' proj, Hd = (W + D)/2, no biases)
 *   pf fp32: word_emb [V,W], pos_emb [P,W], type_emb row 0 [W], embedding LayerNorm w,b;  per layer  b_qkv [3W],
 *            b_out [W], attention-output LayerNorm w,b, b_fc1 [F], b_fc2 [W], output LayerNorm w,b
 * ---------------------------------------------------------------------------------------------- */
typedef struct wise_xlmr_config {
    int32_t context;        /* T: 77 (open_clip's context_length) */
    int32_t vocab;          /* V: 250002 */
    int32_t max_positions;  /* P: 514 */
    int32_t width;          /* W: 1024; multiple of 128, head dim 64 */
    int32_t layers;         /* L: 24 */
    int32_t heads;          /* H = W/64 */
    int32_t mlp;            /* F: 4096 */
    int32_t proj_hidden;    /* Hd: (W + D)/2 */
    int32_t embed_dim;      /* D: 1024 */
    int32_t pad_id;         /* 1 (XLM-R <pad>); 0 for BERT's [PAD] */
    /* ABI 4: the BERT-family switches (all 0 = XLM-RoBERTa under open_clip's HFTextEncoder) */
    int32_t pos_mode;       /* 0: position = cumsum(mask)*mask + pad (RoBERTa); 1: position = 0..T-1 (BERT) */
    int32_t pool;           /* 0: mean over the sequence's own tokens; 1: the first row ([CLS]) */
    int32_t head;           /* 0: W -> Hd -> D, GELU between, no biases (open_clip 'mlp'); 1: msclap Projection
                             *    (e1 = W1 x, e2 = W2 gelu(e1), LayerNorm(e1 + e2); Hd = D = 1024; the fp32 blob then
                             *    ends with that LayerNorm's w, b [D]) — MS-CLAP 2022's caption encoder */
    int32_t eps_e12;        /* LayerNorm eps in units of 1e-12 (BERT: 1); 0 = 1e-5 */
} wise_xlmr_config;
int wise_xlmr_layout(const wise_xlmr_config* cfg, int64_t* wb_elems, int64_t* pf_elems);
size_t wise_xlmr_workspace_bytes(const wise_xlmr_config* cfg, int batch);
int wise_xlmr_forward(const wise_xlmr_config* cfg, const uint16_t* wb, const float* pf, const int32_t* tokens,
                      int batch, float* out /* [batch, D] fp32, L2-normalised */, void* workspace,
                      size_t workspace_bytes, void* stream);
/* parity tap: residual stream fp32 [batch*context, W] of the last forward with the same batch */
int wise_xlmr_tap_residual(const wise_xlmr_config* cfg, int batch, const void* workspace, float* dst, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HP-1  image transform on the GPU: replaces the per-frame CPU loop of
 *       MlfoundationOpenClip.preprocess_image, src/feature/mlfoundation_openclip.py:81-90, for uint8 frames
 *       [n,3,H,W] (the decoder's output, src/dataloader/dataset.py:298):
 *       to_pil_image -> Resize(S, BICUBIC, shorter side) -> CenterCrop(S)   => uint8 [n,3,S,S],
 *       bit-identical to Pillow's 8-bit antialiased resampler; ToTensor + Normalize happen inside
 *       wise_vit_forward(in_kind = WISE_VIT_IN_U8).
 *
 * A plan belongs to one (H, W, S).  plan_init and tables are host-only (no GPU needed); the caller copies the
 * `table_bytes` blob to the device once and passes it to every wise_preproc_u8 call for that geometry.
 * Limits: H, W <= 16384; S % 4 == 0; the staged input of one output tile must fit 64 KiB of LDS
 * (downscale factors up to ~20x).
 * ---------------------------------------------------------------------------------------------- */
typedef struct wise_preproc_plan {
    int32_t H, W, S;            /* input frame size, output edge */
    int32_t new_w, new_h;       /* size after Resize(S) (torchvision: long side = int(S*long/short)) */
    int32_t left, top;          /* CenterCrop origin inside the resized image */
    int32_t tile;               /* output tile edge of one workgroup */
    int32_t ndh, ndv;           /* dwords of padded taps per output column / row */
    int32_t max_cols4, max_rows4; /* staged input rectangle of the worst tile: column dwords, rows / 4 */
    int32_t lds_bytes;
    int32_t reserved;           /* bit 0 (set by the _init functions): 0 = Resize(shorter side) + CenterCrop, 1 = squash.  bit 1 (set by
                                   them too): the table blob carries the integer matrix-core form's tables and wise_preproc_u8
                                   uses that kernel; a caller may clear it (dot-product kernel) or set bit 2 with it (four waves per
                                   tile instead of one).  The three kernels return the same bytes. */
    uint64_t table_bytes;       /* size of the tap-table blob */
} wise_preproc_plan;
int wise_preproc_plan_init(int H, int W, int S, wise_preproc_plan* plan);
/* the same for open_clip's resize_mode 'squash' (the SigLIP models' preprocess_cfg): Resize((S, S), BICUBIC) without
 * regard to the aspect ratio, no crop — new_w = new_h = S, left = top = 0; same tables, same kernel */
int wise_preproc_plan_init_squash(int H, int W, int S, wise_preproc_plan* plan);
int wise_preproc_tables(const wise_preproc_plan* plan, void* host_tables);
/* frames uint8 [n,3,H,W] (device) -> out uint8 [n,3,S,S] (device, 4-byte aligned). */
int wise_preproc_u8(const wise_preproc_plan* plan, const void* dev_tables, const uint8_t* frames, int n,
                    uint8_t* out, void* stream);
/* Pillow's tap table of one axis (first input index, count, 22-bit fixed-point coefficients [out][ksize]);
 * host-only, exported so that the tables can be pinned against the oracle without a GPU. */
int wise_preproc_taps(int in_size, int out_size, int* ksize, int* first, int* count, int* coef,
                      int coef_capacity);

/* Building blocks, exported so the parity tests can pin each kernel separately. */
/* C[M,N] = epilogue(A[M,K] bf16 @ Wt[N,K]^T bf16 + bias[N]) ; M%128==0 rows must be readable
 * (callers pad), N%4==0, K%32==0.
 * mode: 0 -> out_bf16 = acc+bias ; 1 -> out_bf16 = quickgelu(acc+bias) ; 2 -> out_bf16 = gelu(acc+bias)
 *       3 -> resid_f32 += acc+bias (in place, fp32 residual stream) ; 4 -> out_f32 = acc+bias
 *       5 -> out_bf16 = gelu_tanh(acc+bias) ; 6 -> out_bf16 = relu(acc+bias) */
int wise_gemm_bf16(const uint16_t* A, const uint16_t* Wt, const float* bias, int M, int N, int K,
                   int mode, void* out, void* stream);
/* (ABI 5) The two GEMMs of wise_vit_config.ln_fold = 1: a LayerNorm between a residual GEMM and the next linear layer is
 * carried by the GEMMs themselves (open_clip's ln_1 / ln_2 as reached from src/feature/mlfoundation_openclip.py:99).
 *   wise_gemm_fold_resid: x += A[M,K] @ Wt[N,K]^T + bias on a residual stream kept as TWO bf16 arrays, x = hi + lo
 *     (hi = bf16(x), lo = bf16(x - hi), lo = hi + lo_off elements, lo_off >= M*N: 16 significand bits in the 4 bytes per
 *     element an fp32 stream takes — and hi is the next GEMM's operand as it stands); stats = one region of
 *     wise_gemm_fold_stats_bytes(M, N) bytes whose first M floats become rstd[r] = 1 / sqrt(var(x[r,:]) + eps); behind them
 *     wise_gemm_fold_counters_bytes(M) bytes of arrival counters (at byte offset wise_gemm_fold_counters_offset(M)) that must
 *     be ZERO on entry and are zero again on exit, then scratch (partial sums per 64 columns; per 32 with group32 != 0, which
 *     widths that are multiples of 192 but not of 128 need).  The statistics are reduced in one fixed tree, so a row's rstd
 *     does not depend on M or on the tile chosen (for one value of group32).
 *   wise_gemm_fold_bf16: out[M,N] bf16 = act(rstd[row] * (A @ Wt^T) + bias[col]), mode 0 / 1 / 2 / 5 as wise_gemm_bf16; with
 *     A = h, Wt = gamma-scaled row-centred weights and bias = b + W beta this is act(Linear(LayerNorm(x))).
 * M % 128 == 0, N % 128 == 0 (or N % 192 == 0 with group32), K % 64 == 0, K >= 192.  group32: bit 0 = statistics per 32
 * columns; bit 1 = the rows START the stream (x = A @ Wt^T + bias: nothing is read from hi / lo). */
size_t wise_gemm_fold_stats_bytes(int M, int N);
size_t wise_gemm_fold_counters_offset(int M);
size_t wise_gemm_fold_counters_bytes(int M);
int wise_gemm_fold_bf16(const uint16_t* A, const uint16_t* Wt, const float* bias, const float* rstd, int M, int N, int K,
                        int mode, uint16_t* out, void* stream);
int wise_gemm_fold_resid(const uint16_t* A, const uint16_t* Wt, const float* bias, int M, int N, int K, uint16_t* hi,
                         int64_t lo_off, float* stats, float eps, int group32, void* stream);
/* (ABI 5) wise_attention_bf16 followed by wise_gemm_fold_resid (N = K = 64 H) as ONE kernel, for sequences of up to 64
 * tokens and H = 12: per frame b, rows b*T .. b*T+T-1 of the hi + lo stream become x + softmax(q k^T / 8) v @ Wt^T + bias and
 * rstd[row] = 1 / sqrt(var(x[row,:]) + eps) — bit for bit what the two calls produce (same MFMA order, same reduction tree),
 * without the attention output's round trip through HBM, the arrival counters or the scratch region.  qkv [B*T, 3*64*H]
 * as wise_attention_bf16 takes it; Wt [64H, 64H], bias [64H]; rows past B*T are not touched.  Stands behind
 * open_clip's ResidualAttentionBlock.attention + ls_1 + residual as reached from src/feature/mlfoundation_openclip.py:99. */
int wise_attention_oproj_fold(const uint16_t* qkv, int B, int T, int H, const uint16_t* Wt, const float* bias, uint16_t* hi,
                              int64_t lo_off, float* rstd, float eps, void* stream);
/* (ABI 5) x[M,C] fp32 += fc2(GELU(fc1(h))) — erf GELU, h [M,C] bf16 (the LayerNorm'd rows), fc1: C -> 4C, fc2: 4C -> C — as ONE
 * kernel: the 4C-wide hidden activations stay in registers (rounded to bf16 where the two-GEMM form stores them).  C = 384
 * with M % 128 == 0, or C = 192 with M % 256 == 0: the MLP of a Swin block of HTSAT's stages 3 and 2 (msclap HTSAT
 * SwinTransformerBlock.mlp as reached from src/feature/microsoft_clap.py:49-50).  b1 [4C], b2 [C] fp32.  ws = both weight
 * matrices as ONE bf16 stream of 8 C^2 elements in the order the kernel consumes them: for every step s of 32 hidden units
 * (s < C/8) first the 2 * C/32 fc1 fragments (j = 0, 1; ks < C/32) and then the C/16 fc2 fragments (jn < C/16), a fragment
 * being 64 lanes x 8 elements with lane = 16 g + l:
 *   fc1 fragment (j, ks):  W1[32 s + 16 j + l][32 ks + 8 g + e],                        e < 8
 *   fc2 fragment (jn):     W2[16 jn + l][32 s + 4 g + e] for e < 4,  W2[16 jn + l][32 s + 16 + 4 g + (e - 4)] for e >= 4
 * (wise_amd/feature/htsat.py:mlp_stream_weights builds it). */
int wise_mlp_stream(const uint16_t* h, const uint16_t* ws, const float* b1, const float* b2, float* x, int M, int C,
                    void* stream);
/* the same with h = LayerNorm(x; lnw, lnb, eps) computed inside the kernel from the fp32 rows (norm2 of the block: two-pass
 * statistics in registers) — no LayerNorm launch and no h array; what wise_htsat_forward2 flags bit 1 runs. */
int wise_mlp_stream_ln(const float* lnw, const float* lnb, float eps, const uint16_t* ws, const float* b1, const float* b2,
                       float* x, int M, int C, void* stream);
/* (ABI 5) norm1 + QKV projection + window attention of a Swin block of HTSAT's stages 2 / 3 (C = 192 / 384: 8 / 16 heads of 24,
 * windows of 8 x 8 tokens on an H x H grid, cyclic shift 0 or 4) as ONE kernel: o [B*H*H, C] bf16 (original token order) =
 * softmax(q k^T / sqrt(24) + relb[head] (+ the shift mask)) v with q | k | v = LayerNorm(x; lnw, lnb, eps) W_qkv^T + b — what
 * wise_layernorm_f32_bf16 + wise_gemm_bf16 + the tower's window-attention kernel compute, without the normalised rows and the
 * qkv rows ever reaching HBM (msclap HTSAT WindowAttention as reached from src/feature/microsoft_clap.py:49-50).
 * x [B*H*H, C] fp32; relb [heads][64][64] fp32 (the expanded relative-position bias); B * (H/8)^2 even.
 * ws = W_qkv [3C, C] as a stream of 3 * C/48 steps, step s = 3 p + t (p: pair of heads, t: 0 q, 1 k, 2 v) holding weight rows
 * t*C + 48 p .. + 47 as 3 * C/32 fragments (j < 3, ks < C/32) of [lane = 16 g + l][8]: W[t*C + 48 p + 16 j + l][32 ks + 8 g + e];
 * bq = the qkv bias in the same step order, 48 floats per step (wise_amd/feature/htsat.py:swin_qkv_stream builds both). */
int wise_swin_qkv_attn(const float* x, const float* lnw, const float* lnb, float eps, const uint16_t* ws, const float* bq,
                       const float* relb, uint16_t* o, int B, int H, int C, int shift, void* stream);
/* out[ceil256(B*T*F), cout] bf16 = relu(conv3x3(x [B,T,F,cin] bf16, position-major ("NHWC"), stride 1, zero padding 1)
 * + bias[cout]); with pool != 0 its 2x2 average pooling (floor) instead, out [B*(T/2)*(F/2), cout], computed in the same
 * kernel (the unpooled tensor is never written).  wt [cout, 9*cin] bf16 with k = (kh*3 + kw)*cin + c (BatchNorm folded
 * by the caller); cin, cout multiples of 64; zeros = at least 16 bytes of zeros on the device (what every out-of-image
 * tap reads).  The implicit GEMM behind the ConvBlocks of wise_cnn14_forward
 * (torch: F.avg_pool2d(F.relu(bn(F.conv2d(x, w, padding=1))), 2)). */
int wise_conv3x3_relu_bf16(const uint16_t* x, const uint16_t* wt, const float* bias, const uint16_t* zeros, int batch,
                           int T, int F, int cin, int cout, int pool, uint16_t* out, void* stream);
/* LayerNorm fused into the GEMM's A operand: out_bf16[M,N] = epi( LN(x_f32[M,K]; lnw, lnb, eps) @ Wt[N,K]^T + bias ),
 * for K in {96, 192} (HTSAT stages 1-2), N % 8 == 0, M % 128 == 0, bf16 output modes (0, 1, 2, 5). */
int wise_gemm_ln_bf16(const float* x, const float* lnw, const float* lnb, const uint16_t* Wt, const float* bias,
                      int M, int N, int K, float eps, int mode, uint16_t* out, void* stream);
/* The MLP of a C = 96 Swin block in one kernel (HTSAT stage 1): x_f32[M,96] += fc2(gelu(fc1(LN(x; lnw, lnb, eps)))),
 * W1 [384,96], b1 [384], W2 [96,384], b2 [96]; M % 128 == 0.  The 384-wide hidden layer never reaches HBM. */
int wise_mlp96_fused(float* x, const float* lnw, const float* lnb, const uint16_t* W1, const float* b1,
                     const uint16_t* W2, const float* b2, int M, float eps, void* stream);
/* y_bf16[r,:] = (x[r,:]-mean)/sqrt(var+eps)*w+b over W for r < rows. */
int wise_layernorm_f32_bf16(const float* x, const float* w, const float* b, int rows, int W,
                            float eps, uint16_t* y, void* stream);
/* qkv [B*T, 3*H*64] bf16 (q|k|v packed like in_proj) -> o [B*T, H*64] bf16 ; softmax(QK^T/8)V */
int wise_attention_bf16(const uint16_t* qkv, int B, int T, int H, uint16_t* o, void* stream);
/* the same at head dim dh = 64 or 80 (ViT-H/14: width 1280 over 16 heads): qkv [B*T, 3*H*dh], softmax(QK^T/sqrt(dh))V */
int wise_attention_dh_bf16(const uint16_t* qkv, int B, int T, int H, int dh, uint16_t* o, void* stream);
/* head dim 64, right-padded sequences: sequence b has lens[b] (device int32 [B], 1..T) keys; keys past it are masked for
 * every query of that sequence (the BERT-family text towers, wise_xlmr_forward) */
int wise_attention_lens_bf16(const uint16_t* qkv, int B, int T, int H, const int32_t* lens, uint16_t* o, void* stream);
/* head dim 64 with the causal mask of the text tower: query t attends keys <= t */
int wise_attention_causal_bf16(const uint16_t* qkv, int B, int T, int H, uint16_t* o, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WISE_HIP_H */
