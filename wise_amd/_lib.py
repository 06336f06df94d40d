"""ctypes binding of libwise_hip.so (the C ABI declared in include/wise_hip.h).

This is the ONLY way the Python host side reaches the kernels; there is no CPU fallback.  Anything
that needs a kernel calls `lib()` and gets a RuntimeError if the library is missing or no gfx950
device is visible.  torch is used only to own device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / "lib" / "libwise_hip.so"
DEBUG_LIB_PATH = LIB_PATH.with_name("libwise_hip_debug.so")

WISE_OK = 0
WISE_VIT_IN_F32 = 0
WISE_VIT_IN_U8 = 1

EPI_BF16, EPI_QUICKGELU, EPI_GELU, EPI_RESID, EPI_F32 = range(5)


class VitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("image_size", "patch", "width", "layers", "heads", "mlp", "embed_dim", "act",
                                          "arch", "ln_fold")]


class TextConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("context", "vocab", "width", "layers", "heads", "mlp", "embed_dim", "act",
                                          "pool", "head", "no_causal", "eps_e6")]


class XlmrConfig(C.Structure):
    """wise_xlmr_config (include/wise_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in ("context", "vocab", "max_positions", "width", "layers", "heads", "mlp",
                                         "proj_hidden", "embed_dim", "pad_id", "pos_mode", "pool", "head", "eps_e12")]


# every symbol include/wise_hip.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
SIGNATURES = {
    "wise_last_error": (C.c_char_p, []),
    "wise_abi_version": (_i, []),
    "wise_overlap_hint": (None, [_i]),
    "wise_build_flags": (C.c_char_p, []),
    "wise_device_ok": (_i, []),
    "wise_prof_begin": (_i, [_i]),
    "wise_prof_end": (_i, [C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "wise_ip_topk_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "wise_ip_topk_f32": (_i, [_vp, _i64, _i, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "wise_ip_shadow_bf16": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "wise_ip_topk_shadow_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "wise_ip_topk_shadow_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wise_ip_shadow_i8": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "wise_ip_topk_shadow8_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wise_ivf_scan_workspace_bytes": (_sz, [_i, _i, _i]),
    "wise_ivf_scan_f32": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wise_ip_scores_f32": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp]),
    "wise_select_topk_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "wise_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "wise_reconstruct_batch": (_i, [_vp, _i64, _i, _vp, _i64, _vp, _i, _vp, _vp]),
    "wise_vit_layout": (_i, [C.POINTER(VitConfig), C.POINTER(_i64), C.POINTER(_i64)]),
    "wise_vit_workspace_bytes": (_sz, [C.POINTER(VitConfig), _i]),
    "wise_vit_forward": (_i, [C.POINTER(VitConfig), _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "wise_vit_forward_single": (_i, [C.POINTER(VitConfig), _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "wise_vit_tap_residual": (_i, [C.POINTER(VitConfig), _i, _vp, _vp, _vp]),
    "wise_htsat_layout": (_i, [C.POINTER(_i64), C.POINTER(_i64)]),
    "wise_htsat_workspace_bytes": (_sz, [_i, _i]),
    "wise_htsat_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "wise_htsat_forward2": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _i, _vp]),
    "wise_htsat_tap": (_i, [_i, _vp, _i, _i, _vp, _i64, _vp]),
    "wise_cnn14_layout": (_i, [C.POINTER(_i64), C.POINTER(_i64)]),
    "wise_cnn14_workspace_bytes": (_sz, [_i, _i]),
    "wise_cnn14_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "wise_cnn14_tap": (_i, [_i, _vp, _i, _i, _vp, _i64, _vp]),
    "wise_conv3x3_relu_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "wise_text_layout": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "wise_text_workspace_bytes": (_sz, [_vp, _i]),
    "wise_text_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "wise_text_tap_residual": (_i, [_vp, _i, _vp, _vp, _vp]),
    "wise_xlmr_layout": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "wise_xlmr_workspace_bytes": (_sz, [_vp, _i]),
    "wise_xlmr_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "wise_xlmr_tap_residual": (_i, [_vp, _i, _vp, _vp, _vp]),
    "wise_preproc_plan_init": (_i, [_i, _i, _i, _vp]),
    "wise_preproc_plan_init_squash": (_i, [_i, _i, _i, _vp]),
    "wise_preproc_tables": (_i, [_vp, _vp]),
    "wise_preproc_u8": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "wise_preproc_taps": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i]),
    "wise_gemm_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "wise_gemm_fold_stats_bytes": (C.c_size_t, [_i, _i]),
    "wise_gemm_fold_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "wise_gemm_fold_counters_offset": (C.c_size_t, [_i]),
    "wise_gemm_fold_counters_bytes": (C.c_size_t, [_i]),
    "wise_gemm_fold_resid": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _f, _i, _vp]),
    "wise_ivf_argmax": (_i, [_vp, _i, _i, _vp, _vp]),
    "wise_ivf_group_workspace_bytes": (_sz, [_i64, _i]),
    "wise_ivf_group": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wise_ivf_list_sums": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "wise_ivf_normalize_rows": (_i, [_vp, _i, _i, _vp, _vp]),
    "wise_ivf_reseed": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "wise_ivf_gather_rows": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "wise_ivf_gather_i64": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "wise_ivf_expand_lists": (_i, [_vp, _i, _vp, _vp]),
    "wise_swin_qkv_attn": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "wise_mlp_stream": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "wise_mlp_stream_ln": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "wise_mlp96_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "wise_gemm_ln_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "wise_layernorm_f32_bf16": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "wise_attention_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "wise_attention_oproj_fold": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _f, _vp]),
    "wise_attention_dh_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "wise_attention_lens_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "wise_attention_causal_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
}


def load(path: Path | None = None):
    """dlopen the library and attach prototypes.  No GPU needed for this step."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path) if path else LIB_PATH
    if path is None and os.environ.get("WISE_AMD_DEBUG_LIB") == "1":
        # developer opt-in (tools/): bind the whole package to the debug twin so that its wise_debug_* switches act on
        # the kernels the engines launch
        p = DEBUG_LIB_PATH
    if not p.exists():
        raise RuntimeError(
            f"{p} is missing: build it with `python -m wise_amd.build` (hipcc --offload-arch=gfx950). "
            "wise_amd has no CPU fallback.")
    # torch first: it brings its own libamdhip64, and the library must bind to THAT runtime (the one whose streams and
    # allocations it is handed) — dlopen before `import torch` would pull in the system copy and leave two HIP runtimes
    # in the process ("no ROCm-capable device" at the first launch)
    import torch  # noqa: F401
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = lib
    return lib


_DEBUG = None


def load_debug():
    """libwise_hip_debug.so: the same kernels built with their tuning / ablation switches live plus the hardware probes
    (wise_debug_*).  For tools/ and the neighbour tests only — nothing under wise_amd/ calls this."""
    global _DEBUG
    if _DEBUG is None:
        if not DEBUG_LIB_PATH.exists():
            raise RuntimeError(f"{DEBUG_LIB_PATH} is missing: build it with `python -m wise_amd.build`")
        import torch  # noqa: F401  (same reason as in load())
        d = C.CDLL(str(DEBUG_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(d, name)
            fn.restype = res
            fn.argtypes = args
        _DEBUG = d
    return _DEBUG


def lib():
    """Library handle for compute calls: additionally requires a visible gfx950 device."""
    import torch

    l = load()
    if not torch.cuda.is_available():
        raise RuntimeError("wise_amd: no HIP device visible (torch.cuda.is_available() is False); "
                           "the HIP path is the only path")
    return l


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().wise_last_error()
        raise RuntimeError(f"libwise_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor / None."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
