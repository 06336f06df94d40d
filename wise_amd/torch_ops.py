"""`torch.ops.wise_hip.*` — the hot paths as PyTorch-ROCm custom operators (BASELINE.json north_star: "called from
Python via PyTorch-ROCm custom ops"; SURVEY.md §8(b), interface kind).

A thin, stateless layer over the C ABI of libwise_hip.so (include/wise_hip.h): tensors in, tensors out, one operator per
entry point a caller of the reference would reach through open_clip / msclap / faiss.  Each operator
  * is DEFINED with a schema in the `wise_hip` namespace (torch.library), so it shows up as torch.ops.wise_hip.<name>;
  * is IMPLEMENTED for the CUDA dispatch key only (on ROCm builds "CUDA" is HIP): device tensors are passed to the
    library as raw pointers on the current stream; scratch comes from torch's caching allocator;
  * has NO CPU kernel: called with CPU tensors the dispatcher raises — there is no fallback path;
  * has a fake (meta) implementation, so shapes and dtypes can be inferred without a device (tracing, tests).
The engines (wise_amd.feature.vit.VitEngine, ...; wise_amd.index.flat_ip.FlatIPIndex) call the same C ABI directly
and keep their workspaces between calls; the operators are for callers that want plain functional torch ops.

    import wise_amd.torch_ops            # registers the operators (idempotent)
    D, I = torch.ops.wise_hip.ip_topk(X, Q, 10, None, 1)
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import _lib

NAMESPACE = "wise_hip"
_LIBRARY = None

SCHEMAS = {
    # HP-2 (faiss IndexIDMap{IndexFlatIP}.search, src/index/feature_search_index.py:113, api/routes.py:1407)
    "ip_topk": "(Tensor X, Tensor Q, int k, Tensor? ids, int id_base) -> (Tensor, Tensor)",
    "ip_shadow_bf16": "(Tensor X) -> (Tensor, Tensor)",
    "ip_topk_shadow": "(Tensor X, Tensor Xb, Tensor norms, Tensor Q, int k, Tensor? ids, int id_base, "
                      "Tensor(a!) counters) -> (Tensor, Tensor)",
    "ip_shadow_i8": "(Tensor X) -> (Tensor, Tensor, Tensor)",
    "ip_topk_shadow8": "(Tensor X, Tensor Xq, Tensor scales, Tensor norms, Tensor Q, int k, Tensor? ids, int id_base, "
                       "Tensor(a!) counters) -> (Tensor, Tensor)",
    "topk_merge": "(Tensor Ds, Tensor Is, int k) -> (Tensor, Tensor)",
    "reconstruct_batch": "(Tensor X, Tensor? ids, int id_base, Tensor query_ids) -> Tensor",
    # IndexIVFFlat (src/index/feature_search_index.py:53-76, api/routes.py:899-902)
    "ip_scores": "(Tensor X, Tensor Q) -> Tensor",
    "select_topk": "(Tensor scores, int k) -> Tensor",
    "ivf_scan": "(Tensor X, Tensor list_off, Tensor? ids, Tensor Q, Tensor probes, int k) -> (Tensor, Tensor)",
    # HP-1 (open_clip encode_image / encode_text, msclap audio_encoder; src/feature/*.py)
    "vit_forward": "(Tensor images, Tensor wb, Tensor pf, int[] config) -> Tensor",
    "text_forward": "(Tensor tokens, Tensor wb, Tensor pf, int[] config) -> Tensor",
    "xlmr_forward": "(Tensor tokens, Tensor wb, Tensor pf, int[] config) -> Tensor",
    "htsat_forward": "(Tensor wave, Tensor wb, Tensor pf) -> Tensor",
    "cnn14_forward": "(Tensor wave, Tensor wb, Tensor pf) -> Tensor",
    "conv3x3_relu": "(Tensor x, Tensor wt, Tensor bias, bool pool) -> Tensor",
    "clip_preprocess_u8": "(Tensor frames, int size) -> Tensor",
}


def _check(rc: int, what: str):
    _lib.check(rc, what)


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("wise_hip operators run on HIP device tensors only")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


def _req(t, dtype, name: str, op: str, min_numel: int = 0):
    """tensors whose raw pointer goes to the C ABI as is: right dtype, contiguous, big enough — or an error, not garbage"""
    if t is None:
        return
    if t.dtype != dtype or not t.is_contiguous() or t.numel() < min_numel:
        raise ValueError(f"wise_hip::{op}: {name} must be a contiguous {dtype} tensor"
                         + (f" of at least {min_numel} elements" if min_numel else "")
                         + f" (got {t.dtype}, contiguous={t.is_contiguous()}, numel={t.numel()})")


# ---------------------------------------------------------------------------------------------- HP-2
def _ip_topk(X, Q, k, ids, id_base):
    lib = _lib.lib()
    _dev(X, Q, ids)
    _req(ids, torch.int64, "ids", "ip_topk", X.shape[0])
    X, Q = _f32c(X), _f32c(Q)
    N, d = X.shape
    nq = Q.shape[0]
    D = torch.empty(nq, k, dtype=torch.float32, device=X.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=X.device)
    if nq == 0:
        return D, I
    need = lib.wise_ip_topk_workspace_bytes(N, d, nq, k)
    if need == 0:
        raise ValueError(f"wise_hip::ip_topk: unsupported shape N={N} d={d} nq={nq} k={k}")
    ws = torch.empty(need, dtype=torch.uint8, device=X.device)
    _check(lib.wise_ip_topk_f32(X.data_ptr(), N, d, Q.data_ptr(), nq, k, _lib.ptr(ids), id_base, D.data_ptr(),
                                I.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_ip_topk_f32")
    return D, I


def _ip_shadow_bf16(X):
    lib = _lib.lib()
    _dev(X)
    X = _f32c(X)
    Xb = torch.empty(X.shape, dtype=torch.int16, device=X.device)       # bf16 bit patterns
    norms = torch.zeros(2, dtype=torch.float32, device=X.device)
    _check(lib.wise_ip_shadow_bf16(X.data_ptr(), X.shape[0], X.shape[1], Xb.data_ptr(), norms.data_ptr(),
                                   _lib.stream_ptr()), "wise_ip_shadow_bf16")
    return Xb, norms


def _ip_topk_shadow(X, Xb, norms, Q, k, ids, id_base, counters):
    lib = _lib.lib()
    _dev(X, Xb, norms, Q, ids, counters)
    _req(Xb, torch.int16, "Xb (bf16 bit patterns)", "ip_topk_shadow", X.numel())
    _req(norms, torch.float32, "norms", "ip_topk_shadow", 2)
    _req(ids, torch.int64, "ids", "ip_topk_shadow", X.shape[0])
    X, Q = _f32c(X), _f32c(Q)
    N, d = X.shape
    nq = Q.shape[0]
    D = torch.empty(nq, k, dtype=torch.float32, device=X.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=X.device)
    if nq == 0:
        return D, I
    if counters.dtype != torch.int32 or counters.numel() < 2:
        raise ValueError("wise_hip::ip_topk_shadow: counters must be an int32 tensor of two elements")
    need = lib.wise_ip_topk_shadow_workspace_bytes(N, d, nq, k)
    if need == 0:
        raise ValueError(f"wise_hip::ip_topk_shadow: unsupported shape N={N} d={d} nq={nq} k={k}")
    ws = torch.empty(need, dtype=torch.uint8, device=X.device)
    _check(lib.wise_ip_topk_shadow_f32(X.data_ptr(), Xb.data_ptr(), norms.data_ptr(), N, d, Q.data_ptr(), nq, k,
                                       _lib.ptr(ids), id_base, D.data_ptr(), I.data_ptr(), counters.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_ip_topk_shadow_f32")
    return D, I


def _ip_shadow_i8(X):
    lib = _lib.lib()
    _dev(X)
    X = _f32c(X)
    Xq = torch.empty(X.shape, dtype=torch.int8, device=X.device)
    scales = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
    norms = torch.zeros(4, dtype=torch.float32, device=X.device)
    _check(lib.wise_ip_shadow_i8(X.data_ptr(), X.shape[0], X.shape[1], Xq.data_ptr(), scales.data_ptr(), norms.data_ptr(),
                                 _lib.stream_ptr()), "wise_ip_shadow_i8")
    return Xq, scales, norms


def _ip_topk_shadow8(X, Xq, scales, norms, Q, k, ids, id_base, counters):
    lib = _lib.lib()
    _dev(X, Xq, scales, norms, Q, ids, counters)
    _req(Xq, torch.int8, "Xq", "ip_topk_shadow8", X.numel())
    _req(scales, torch.float32, "scales", "ip_topk_shadow8", X.shape[0])
    _req(norms, torch.float32, "norms", "ip_topk_shadow8", 4)
    _req(ids, torch.int64, "ids", "ip_topk_shadow8", X.shape[0])
    X, Q = _f32c(X), _f32c(Q)
    N, d = X.shape
    nq = Q.shape[0]
    D = torch.empty(nq, k, dtype=torch.float32, device=X.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=X.device)
    if nq == 0:
        return D, I
    if counters.dtype != torch.int32 or counters.numel() < 2:
        raise ValueError("wise_hip::ip_topk_shadow8: counters must be an int32 tensor of two elements")
    need = max(lib.wise_ip_topk_shadow_workspace_bytes(N, d, nq, k), lib.wise_ip_topk_workspace_bytes(N, d, nq, k))
    if need == 0 or d % 16 != 0:
        raise ValueError(f"wise_hip::ip_topk_shadow8: unsupported shape N={N} d={d} nq={nq} k={k}")
    ws = torch.empty(need, dtype=torch.uint8, device=X.device)
    _check(lib.wise_ip_topk_shadow8_f32(X.data_ptr(), Xq.data_ptr(), scales.data_ptr(),
                                        norms.data_ptr(), N, d, Q.data_ptr(), nq, k, _lib.ptr(ids), id_base, D.data_ptr(),
                                        I.data_ptr(), counters.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
           "wise_ip_topk_shadow8_f32")
    return D, I


def _topk_merge(Ds, Is, k):
    lib = _lib.lib()
    _dev(Ds, Is)
    parts, nq, kk = Ds.shape
    D = torch.empty(nq, k, dtype=torch.float32, device=Ds.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=Ds.device)
    if k != kk:
        raise ValueError("wise_hip::topk_merge: the lists hold k entries each")
    _check(lib.wise_topk_merge(_f32c(Ds).data_ptr(), Is.contiguous().data_ptr(), parts, nq, kk, D.data_ptr(),
                               I.data_ptr(), _lib.stream_ptr()), "wise_topk_merge")
    return D, I


def _reconstruct_batch(X, ids, id_base, query_ids):
    lib = _lib.lib()
    _dev(X, ids, query_ids)
    _req(ids, torch.int64, "ids", "reconstruct_batch", X.shape[0])
    X = _f32c(X)
    q = query_ids.to(torch.int64).contiguous()
    out = torch.empty(q.numel(), X.shape[1], dtype=torch.float32, device=X.device)
    _check(lib.wise_reconstruct_batch(X.data_ptr(), X.shape[0], X.shape[1], _lib.ptr(ids), id_base, q.data_ptr(),
                                      q.numel(), out.data_ptr(), _lib.stream_ptr()), "wise_reconstruct_batch")
    return out


def _ip_scores(X, Q):
    lib = _lib.lib()
    _dev(X, Q)
    X, Q = _f32c(X), _f32c(Q)
    S = torch.empty(Q.shape[0], X.shape[0], dtype=torch.float32, device=X.device)
    _check(lib.wise_ip_scores_f32(X.data_ptr(), X.shape[0], X.shape[1], Q.data_ptr(), Q.shape[0], S.data_ptr(),
                                  _lib.stream_ptr()), "wise_ip_scores_f32")
    return S


def _select_topk(scores, k):
    lib = _lib.lib()
    _dev(scores)
    scores = _f32c(scores)
    out = torch.empty(scores.shape[0], k, dtype=torch.int64, device=scores.device)
    _check(lib.wise_select_topk_f32(scores.data_ptr(), scores.shape[0], scores.shape[1], k, out.data_ptr(),
                                    _lib.stream_ptr()), "wise_select_topk_f32")
    return out


def _ivf_scan(X, list_off, ids, Q, probes, k):
    lib = _lib.lib()
    _dev(X, list_off, ids, Q, probes)
    _req(list_off, torch.int64, "list_off", "ivf_scan", 2)
    _req(ids, torch.int64, "ids", "ivf_scan", X.shape[0])
    _req(probes, torch.int64, "probes", "ivf_scan")
    X, Q = _f32c(X), _f32c(Q)
    nq, nprobe = probes.shape
    D = torch.empty(nq, k, dtype=torch.float32, device=X.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=X.device)
    need = lib.wise_ivf_scan_workspace_bytes(nq, nprobe, k)
    if need == 0:
        raise ValueError(f"wise_hip::ivf_scan: unsupported shape nq={nq} nprobe={nprobe} k={k}")
    ws = torch.empty(need, dtype=torch.uint8, device=X.device)
    _check(lib.wise_ivf_scan_f32(X.data_ptr(), X.shape[0], X.shape[1], list_off.contiguous().data_ptr(),
                                 list_off.numel() - 1, _lib.ptr(ids), Q.data_ptr(), nq,
                                 probes.to(torch.int64).contiguous().data_ptr(), nprobe, k, D.data_ptr(), I.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_ivf_scan_f32")
    return D, I


# ---------------------------------------------------------------------------------------------- HP-1
def _vit_forward(images, wb, pf, config: List[int]):
    lib = _lib.lib()
    _dev(images, wb, pf)
    cfg = _lib.VitConfig(*config)
    if images.dtype == torch.uint8:
        kind = _lib.WISE_VIT_IN_U8
    elif images.dtype == torch.float32:
        kind = _lib.WISE_VIT_IN_F32
    else:
        raise ValueError("wise_hip::vit_forward: images must be float32 (normalised) or uint8")
    x = images.contiguous()
    B = x.shape[0]
    out = torch.empty(B, cfg.embed_dim, dtype=torch.float32, device=x.device)
    need = lib.wise_vit_workspace_bytes(C.byref(cfg), B)
    if need == 0:
        raise ValueError("wise_hip::vit_forward: bad config")
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    _check(lib.wise_vit_forward(C.byref(cfg), wb.data_ptr(), pf.data_ptr(), x.data_ptr(), kind, B, out.data_ptr(),
                                ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_vit_forward")
    return out


def _text_forward(tokens, wb, pf, config: List[int]):
    lib = _lib.lib()
    _dev(tokens, wb, pf)
    cfg = _lib.TextConfig(*config)
    t = tokens.to(torch.int32).contiguous()
    B = t.shape[0]
    out = torch.empty(B, cfg.embed_dim, dtype=torch.float32, device=t.device)
    need = lib.wise_text_workspace_bytes(C.byref(cfg), B)
    if need == 0:
        raise ValueError("wise_hip::text_forward: bad config")
    ws = torch.empty(need, dtype=torch.uint8, device=t.device)
    _check(lib.wise_text_forward(C.byref(cfg), wb.data_ptr(), pf.data_ptr(), t.data_ptr(), B, out.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_text_forward")
    return out


def _xlmr_forward(tokens, wb, pf, config: List[int]):
    lib = _lib.lib()
    _dev(tokens, wb, pf)
    cfg = _lib.XlmrConfig(*config)
    t = tokens.to(torch.int32).contiguous()
    B = t.shape[0]
    out = torch.empty(B, cfg.embed_dim, dtype=torch.float32, device=t.device)
    need = lib.wise_xlmr_workspace_bytes(C.byref(cfg), B)
    if need == 0:
        raise ValueError("wise_hip::xlmr_forward: bad config")
    ws = torch.empty(need, dtype=torch.uint8, device=t.device)
    _check(lib.wise_xlmr_forward(C.byref(cfg), wb.data_ptr(), pf.data_ptr(), t.data_ptr(), B, out.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_xlmr_forward")
    return out


def _htsat_forward(wave, wb, pf):
    lib = _lib.lib()
    _dev(wave, wb, pf)
    x = _f32c(wave)
    B, N = x.shape
    out = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
    need = lib.wise_htsat_workspace_bytes(B, N)
    if need == 0:
        raise ValueError("wise_hip::htsat_forward: bad shape")
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    _check(lib.wise_htsat_forward(wb.data_ptr(), pf.data_ptr(), x.data_ptr(), B, N, out.data_ptr(), ws.data_ptr(),
                                  ws.numel(), _lib.stream_ptr()), "wise_htsat_forward")
    return out


def _cnn14_forward(wave, wb, pf):
    lib = _lib.lib()
    _dev(wave, wb, pf)
    x = _f32c(wave)
    B, N = x.shape
    out = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
    need = lib.wise_cnn14_workspace_bytes(B, N)
    if need == 0:
        raise ValueError("wise_hip::cnn14_forward: bad shape (at least 32 STFT frames)")
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    _check(lib.wise_cnn14_forward(wb.data_ptr(), pf.data_ptr(), x.data_ptr(), B, N, out.data_ptr(), ws.data_ptr(),
                                  ws.numel(), _lib.stream_ptr()), "wise_cnn14_forward")
    return out


def _conv3x3_relu(x, wt, bias, pool):
    _dev(x, wt, bias)
    if x.dim() != 4 or x.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16 or wt.shape[1] != 9 * x.shape[3]:
        raise ValueError("wise_hip::conv3x3_relu: x [B,T,F,Cin] bf16, wt [Cout, 9*Cin] bf16, bias [Cout] fp32")
    from .feature.cnn14 import conv3x3_relu
    return conv3x3_relu(x, wt, _f32c(bias), bool(pool))


def _clip_preprocess_u8(frames, size):
    from .feature.preprocess import ClipPreprocessor

    _dev(frames)
    return ClipPreprocessor(int(size))(frames)


# ---------------------------------------------------------------------------------------------- fakes (shape inference)
def _fake_pair(nq, k, like):
    return like.new_empty((nq, k), dtype=torch.float32), like.new_empty((nq, k), dtype=torch.int64)


_IMPLS = {
    "ip_topk": (_ip_topk, lambda X, Q, k, ids, id_base: _fake_pair(Q.shape[0], k, X)),
    "ip_shadow_bf16": (_ip_shadow_bf16, lambda X: (X.new_empty(X.shape, dtype=torch.int16),
                                                   X.new_empty((2,), dtype=torch.float32))),
    "ip_topk_shadow": (_ip_topk_shadow, lambda X, Xb, norms, Q, k, ids, id_base, counters: _fake_pair(Q.shape[0], k, X)),
    "ip_shadow_i8": (_ip_shadow_i8, lambda X: (X.new_empty(X.shape, dtype=torch.int8), X.new_empty((X.shape[0],), dtype=torch.float32),
                                               X.new_empty((4,), dtype=torch.float32))),
    "ip_topk_shadow8": (_ip_topk_shadow8,
                        lambda X, Xq, scales, norms, Q, k, ids, id_base, counters: _fake_pair(Q.shape[0], k, X)),
    "topk_merge": (_topk_merge, lambda Ds, Is, k: _fake_pair(Ds.shape[1], k, Ds)),
    "reconstruct_batch": (_reconstruct_batch, lambda X, ids, id_base, q: X.new_empty((q.numel(), X.shape[1]),
                                                                                   dtype=torch.float32)),
    "ip_scores": (_ip_scores, lambda X, Q: X.new_empty((Q.shape[0], X.shape[0]), dtype=torch.float32)),
    "select_topk": (_select_topk, lambda s, k: s.new_empty((s.shape[0], k), dtype=torch.int64)),
    "ivf_scan": (_ivf_scan, lambda X, lo, ids, Q, probes, k: _fake_pair(Q.shape[0], k, X)),
    "vit_forward": (_vit_forward, lambda im, wb, pf, cfg: im.new_empty((im.shape[0], cfg[6]), dtype=torch.float32)),
    "text_forward": (_text_forward, lambda t, wb, pf, cfg: t.new_empty((t.shape[0], cfg[6]), dtype=torch.float32)),
    "xlmr_forward": (_xlmr_forward, lambda t, wb, pf, cfg: t.new_empty((t.shape[0], cfg[8]), dtype=torch.float32)),
    "htsat_forward": (_htsat_forward, lambda w, wb, pf: w.new_empty((w.shape[0], 1024), dtype=torch.float32)),
    "cnn14_forward": (_cnn14_forward, lambda w, wb, pf: w.new_empty((w.shape[0], 1024), dtype=torch.float32)),
    "conv3x3_relu": (_conv3x3_relu, lambda x, wt, b, pool: x.new_empty(
        (x.shape[0], x.shape[1] // 2 if pool else x.shape[1], x.shape[2] // 2 if pool else x.shape[2], wt.shape[0]))),
    "clip_preprocess_u8": (_clip_preprocess_u8, lambda f, s: f.new_empty((f.shape[0], 3, s, s), dtype=torch.uint8)),
}


def register() -> "torch.library.Library":
    """Define and implement the operators (once per process).  Needs no GPU: only calling an operator does."""
    global _LIBRARY
    if _LIBRARY is not None:
        return _LIBRARY
    lib = torch.library.Library(NAMESPACE, "DEF")
    for name, schema in SCHEMAS.items():
        lib.define(name + schema)
        impl, fake = _IMPLS[name]
        lib.impl(name, impl, "CUDA")          # HIP devices; deliberately no CPU / CompositeExplicitAutograd kernel
        torch.library.register_fake(f"{NAMESPACE}::{name}", fake, lib=lib)
    _LIBRARY = lib
    return lib


register()
