"""Streams that actually run side by side.

`torch.cuda.Stream()` hands out pooled streams in turn, and HIP places streams on a handful of hardware queues: two streams that
land on the SAME queue execute one after the other however independent their work is.  Which pairs collide depends on how
many streams the process took before (measured: tools/htsat_streams_probe.py — two MS-CLAP batches "in flight" took 3.50 ms
per step on one pair of consecutive pool streams and 3.27 ms on the next pair; bench.py's HTSAT leg lost its whole gain when an
unrelated leg took one more stream in front of it).  The engines that keep two batches in flight therefore take their
streams from here: a candidate is accepted once a short kernel on it completes while a spinning kernel occupies every stream
already chosen.  Stream plumbing only: nothing here computes."""
from __future__ import annotations

import time
from typing import List

import torch

_SPIN_CYCLES = None


def _spin_cycles(device) -> int:
    """torch.cuda._sleep cycles that last ~4 ms on this device (its tick is not specified: calibrated once)"""
    global _SPIN_CYCLES
    if _SPIN_CYCLES is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.device(device):
            torch.cuda._sleep(100_000)
            e0.record()
            torch.cuda._sleep(1_000_000)
            e1.record()
            torch.cuda.synchronize(device)
        ms = max(e0.elapsed_time(e1), 1e-3)
        _SPIN_CYCLES = int(min(max(4.0 / ms * 1_000_000, 10_000), 2_000_000_000))
    return _SPIN_CYCLES


def _runs_beside(cand: "torch.cuda.Stream", busy: "torch.cuda.Stream", device) -> bool:
    """does a small kernel on `cand` finish while `busy` spins?"""
    torch.cuda.synchronize(device)
    probe = torch.zeros(64, device=device)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(busy):
        torch.cuda._sleep(_spin_cycles(device))
        end = torch.cuda.Event()
        end.record(busy)
    with torch.cuda.stream(cand):
        probe.add_(1.0)
        done = torch.cuda.Event()
        done.record(cand)
    ok = False
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        if done.query():
            ok = not end.query()        # finished while the spinner was still running
            break
        if end.query():
            break
    torch.cuda.synchronize(device)
    return ok


def concurrent_streams(n: int, device, tries: int = 12, beside=()) -> List["torch.cuda.Stream"]:
    """n new streams of which every pair — and every pair with a stream of `beside` — was SEEN running concurrently (falls
    back to whatever torch hands out after `tries` rejected candidates: correctness never depends on it)."""
    device = torch.device(device)
    fixed = list(beside)
    got: List["torch.cuda.Stream"] = []
    rejected = 0
    while len(got) < n:
        s = torch.cuda.Stream(device=device)
        others = fixed + got
        if rejected >= tries or all(_runs_beside(s, g, device) and _runs_beside(g, s, device) for g in others):
            got.append(s)
        else:
            rejected += 1
    return got
