"""GPU image transform: uint8 frames [n,3,H,W] on the device -> uint8 [n,3,S,S].

Device-side replacement for the per-frame CPU loop of the reference's
`MlfoundationOpenClip.preprocess_image` (src/feature/mlfoundation_openclip.py:81-90):
`to_pil_image -> Resize(S, BICUBIC) -> CenterCrop(S)`; `ToTensor -> Normalize` are applied by the tower's
patch gather when it is fed uint8 (`WISE_VIT_IN_U8`).  Results are bit-identical to Pillow's.
Binds `wise_preproc_*` of include/wise_hip.h; there is no CPU path here (the CPU path is the reference's own
PIL loop, kept in mlfoundation_openclip.py for DataLoader workers).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _lib


class PreprocPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "W", "S", "new_w", "new_h", "left", "top", "tile", "ndh", "ndv",
                                          "max_cols4", "max_rows4", "lds_bytes", "reserved")] + \
               [("table_bytes", C.c_uint64)]


def make_plan(H: int, W: int, S: int, squash: bool = False) -> PreprocPlan:
    """Geometry + table sizes for one (H, W, S); host-only (works without a GPU).  squash: open_clip's resize_mode of the
    SigLIP models — Resize((S, S)) without regard to aspect and no crop — instead of Resize(shorter side) + CenterCrop."""
    lib = _lib.load()
    plan = PreprocPlan()
    init = lib.wise_preproc_plan_init_squash if squash else lib.wise_preproc_plan_init
    rc = init(int(H), int(W), int(S), C.byref(plan))
    if rc != 0:
        raise ValueError(lib.wise_last_error().decode())
    return plan


def plan_tables(plan: PreprocPlan) -> np.ndarray:
    """The int32 tap-table blob of a plan (host)."""
    lib = _lib.load()
    buf = np.empty(plan.table_bytes // 4, dtype=np.int32)
    rc = lib.wise_preproc_tables(C.byref(plan), buf.ctypes.data)
    if rc != 0:
        raise ValueError(lib.wise_last_error().decode())
    return buf


def pillow_taps(in_size: int, out_size: int):
    """(ksize, first[out], count[out], coef[out,ksize]) exactly as the library computes them (host-only)."""
    lib = _lib.load()
    scale = max(in_size / out_size, 1.0)
    cap = out_size * (int(np.ceil(2.0 * scale)) * 2 + 1)
    ks = C.c_int(0)
    first = np.empty(out_size, dtype=np.int32)
    count = np.empty(out_size, dtype=np.int32)
    coef = np.zeros(cap, dtype=np.int32)
    rc = lib.wise_preproc_taps(int(in_size), int(out_size), C.byref(ks), first.ctypes.data, count.ctypes.data,
                               coef.ctypes.data, cap)
    if rc != 0:
        raise ValueError(lib.wise_last_error().decode())
    return ks.value, first, count, coef[:out_size * ks.value].reshape(out_size, ks.value)


class ClipPreprocessor:
    """Callable: device uint8 [n,3,H,W] -> device uint8 [n,3,S,S].  Plans and device tables are cached per
    frame geometry (a video collection has a handful of distinct sizes)."""

    def __init__(self, size: int, device: str = "cuda", squash: bool = False, matrix_cores="auto"):
        """matrix_cores: which of the three kernels a geometry takes — they return the same bytes.  "auto" (default): measured
        once per (H, W) on the first call, on that call's own frames (the dot-product kernel, the integer matrix-core kernel with
        one wave per 32 x 32 tile, the same with four): small frames are served best by lone waves, 1080p by four, 720p by the
        dot products (tools/preproc_bench.py).  False: the dot-product kernel; True / 1: matrix cores, one wave; 4: four waves."""
        self.size = int(size)
        self.device = device
        self.squash = bool(squash)     # the SigLIP models' transform: Resize((S, S)), no crop
        self.matrix_cores = matrix_cores
        self.chosen: Dict[Tuple[int, int], str] = {}     # (H, W) -> the kernel in use
        self._plans: Dict[Tuple[int, int], Tuple[PreprocPlan, torch.Tensor]] = {}

    def _plan(self, H: int, W: int):
        key = (H, W)
        hit = self._plans.get(key)
        if hit is None:
            plan = make_plan(H, W, self.size, self.squash)
            tables = torch.from_numpy(plan_tables(plan)).to(self.device)
            offered = bool(plan.reserved & 2)    # wise_preproc_plan.reserved bit 1: the blob carries the matrix-core tables
            mc = self.matrix_cores
            if not offered or mc is False:
                plan.reserved &= ~6
                self.chosen[key] = "dot products"
            elif mc == 4:
                plan.reserved |= 4
                self.chosen[key] = "matrix cores, four waves per tile"
            elif mc == "auto":
                self.chosen[key] = None          # decided by the first call (it has frames to measure on)
            else:
                self.chosen[key] = "matrix cores, one wave per tile"
            hit = (plan, tables)
            self._plans[key] = hit
        return hit

    def _autotune(self, key, plan, tables, frames, out):
        """Time the three kernels on (up to 64 of) the caller's frames and keep the fastest for this geometry."""
        lib = _lib.lib()
        stream = torch.cuda.current_stream().cuda_stream
        n = min(int(frames.shape[0]), 64)
        base = plan.reserved & ~6
        best = None
        for name, bits in (("dot products", 0), ("matrix cores, one wave per tile", 2), ("matrix cores, four waves per tile", 6)):
            plan.reserved = base | bits
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(3):
                if rep == 1:
                    e0.record()
                rc = lib.wise_preproc_u8(C.byref(plan), tables.data_ptr(), frames.data_ptr(), n, out.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(lib.wise_last_error().decode())
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            if best is None or t < best[0]:
                best = (t, name, bits)
        plan.reserved = base | best[2]
        self.chosen[key] = best[1]

    def __call__(self, frames: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if not isinstance(frames, torch.Tensor) or frames.dtype != torch.uint8 or frames.dim() != 4 \
                or frames.shape[1] != 3:
            raise ValueError('GPU preprocess takes a uint8 tensor [n,3,H,W]')
        lib = _lib.lib()  # raises without a gfx950 device
        frames = frames.to(self.device).contiguous()
        n, _, H, W = frames.shape
        plan, tables = self._plan(H, W)
        S = self.size
        if out is None:
            out = torch.empty((n, 3, S, S), dtype=torch.uint8, device=self.device)
        if self.chosen.get((H, W), "") is None:
            self._autotune((H, W), plan, tables, frames, out)
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.wise_preproc_u8(C.byref(plan), tables.data_ptr(), frames.data_ptr(), n, out.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(lib.wise_last_error().decode())
        return out
