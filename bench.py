#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: frames/s embedded (OpenCLIP ViT-B/32, bs=256 per GPU)
and, in the `search` object of the same JSON line, queries/s over a 10M x 512 flat IP index.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" of the headline metric is one wise_vit_forward over a 256-frame batch
already resident in HBM (weak scaling: every rank embeds its own batch, no collective on the data
path).  A search step is one nq=1, k=10 scan of the index, whose 10M rows are sharded over the ranks
(strong scaling; RCCL all-gather of the per-shard top-k, then the merge kernel).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2516.6  # MI355X dense bf16 MFMA: 256 CUs x 2.4 GHz x 4096 FLOP/clk/CU (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak (same table)
METRIC = "frames/sec embedded (ViT-B/32 bs=256) + queries/sec over 10M×512 index, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--index-rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the CLAP-HTSAT and ViT-L/14 legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget per CPU baseline leg")
    ap.add_argument("--pmc-legs", action="store_true",
                    help="profiling aid for the PMC passes: only the headline shapes run (ViT-B/32 bs=256; 10M x 512 search at "
                         "nq=1, k=10 and nq=256, k=10; the fp32 scan), so that a per-kernel average of FETCH_SIZE / WRITE_SIZE is "
                         "the traffic of THOSE launches and not a mean over shard-sized and split launches too")
    ap.add_argument("--roofline-only", action="store_true",
                    help="profiling aid: run only the single-stream, event-bracketed ViT pass (the one `roofline` is "
                         "computed from) so that a rocprofv3 --stats of this command lists exactly those launches")
    return ap.parse_args()


def timed_region(fn, steps, world):
    """barrier + sync, EXACTLY `steps` calls, sync + barrier; returns MAX-over-ranks seconds."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def prof_pass(lib, fn, steps, capacity):
    """Same steps again with per-kernel HIP-event brackets (separate pass: the brackets cost a little)."""
    from wise_amd import _lib

    _lib.check(lib.wise_prof_begin(capacity), "prof_begin")
    for i in range(steps):
        fn(i)
    ms = (C.c_double * 2)()
    n = (C.c_int64 * 2)()
    work = (C.c_double * 2)()
    _lib.check(lib.wise_prof_end(ms, n, work), "prof_end")
    return [(ms[c], n[c], work[c]) for c in range(2)]


def load_pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary, if one exists (else None)."""
    p = ROOT / "profiles" / "pmc_traffic.json"
    if not p.exists():
        return None
    try:
        return json.loads(p.read_text()).get(kernel_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def traffic_fields(kernel_key):
    """`traffic` of a roofline object + where it comes from: the committed PMC summary, NOT counters of this run."""
    return {"traffic": load_pmc_traffic(kernel_key),
            "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the profile "
                              "sequence committed with this round, corrected per MI355X_MICROARCH.md; not measured in this run)"}


def cpu_baseline_vit(spec, sd, seconds):
    """SURVEY 8(d) cfg-1 stand-in, image half: the reference's CPU extraction path on synthetic decoded frames —
    uint8 [n,3,240,320] frames, 8 at a time exactly as extract-features.py:294,324-341 feeds them: per-frame PIL
    transform (preprocess_image, mlfoundation_openclip.py:81-90) then the fp32 forward of the oracle, all host cores."""
    from oracle import vit_ref
    from wise_amd.feature.mlfoundation_openclip import ClipImageTransform, to_pil_image

    tr = ClipImageTransform(spec.image_size)
    pool = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(64, 3, 240, 320), dtype=np.uint8))
    threads = min(os.cpu_count() or 1, 16)  # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)

    def chunk(c):
        frames = pool[(8 * c) % 64:(8 * c) % 64 + 8]
        x = torch.stack([tr(to_pil_image(f)) for f in frames])
        return vit_ref.vit_forward(sd, x, patch=spec.patch, heads=spec.heads, act=spec.act)

    with torch.no_grad():
        chunk(0)  # warm-up
        n, t_pre = 0, 0.0
        t0 = time.perf_counter()
        while True:
            chunk(n // 8)
            n += 8
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 600:      # cfg-1: 600 frames = 30 videos x 10 s x 2 fps (docs/Tests.md:17-18)
                break
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic uint8 240x320 frames in chunks of 8 (cfg-1 shape, extract-features.py:294): per-frame "
                      f"PIL transform + oracle/vit_ref.py fp32 forward, {threads} torch threads, {dt:.1f} s"}


def cpu_baseline_search(d, k, seconds, n_queries=20):
    """SURVEY 8(d) cfg-1 stand-in, search half: what faiss's IndexFlatIP does on the host for one query — a BLAS
    matrix-vector product over all rows on ALL host cores, then the k best (torch.mv + topk; the numpy restatement
    oracle/ip_topk_ref.py is the same arithmetic) — over the largest sample host RAM allows, 20 queries as cfg-1 asks.
    The one-core C heap loop (oracle/ip_topk_ref.c: faiss's small-nq code path without its SIMD) is timed beside it."""
    import psutil

    from oracle.build import build_oracle

    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    free = psutil.virtual_memory().available
    n = 10_000_000 if free > 3 * 10_000_000 * d * 4 else (4_000_000 if free > 3 * 4_000_000 * d * 4 else 1_000_000)
    g = torch.Generator().manual_seed(2)
    X = torch.empty(n, d)
    for s0 in range(0, n, 500_000):
        blk = torch.randn(min(500_000, n - s0), d, generator=g)
        X[s0:s0 + blk.shape[0]] = blk / blk.norm(dim=1, keepdim=True)
    Q = torch.randn(64, d, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    torch.topk(torch.mv(X, Q[0]), k)   # warm-up (page faults, BLAS threads)
    nq, t0 = 0, time.perf_counter()
    while True:
        D, I = torch.topk(torch.mv(X, Q[nq % 64]), k)
        nq += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or nq >= n_queries:
            break
    scale = 10_000_000 / n
    per_query_10m = dt / nq * scale
    out = {"value": round(1.0 / per_query_10m, 4), "unit": "queries/s", "cores": threads, "kind": "port",
           "sample": f"{nq} queries, torch.mv + topk (BLAS, {threads} threads) over {n} x {d} fp32 unit rows"
                     + ("" if n == 10_000_000 else f", time x{scale:g} for 10M rows") + f", {dt:.1f} s",
           "effective_gbs": round(10_000_000 * d * 4 / per_query_10m / 1e9, 1)}
    # one core, the sequential dot-product loop + heap (1M-row slice, x10)
    so = build_oracle()
    lib = C.CDLL(str(so))
    lib.wise_oracle_ip_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p]
    lib.wise_oracle_ip_topk.restype = None
    Xn, Qn = X[:1_000_000].numpy(), Q.numpy()
    Dn, In = np.empty((1, k), np.float32), np.empty((1, k), np.int64)
    m, t0 = 0, time.perf_counter()
    while m < 4 and time.perf_counter() - t0 < seconds / 3:
        lib.wise_oracle_ip_topk(Xn.ctypes.data, 1_000_000, d, Qn[m].ctypes.data, 1, k, None, 1, Dn.ctypes.data, In.ctypes.data)
        m += 1
    out["one_core_c_loop_queries_per_s"] = round(m / (time.perf_counter() - t0) / 10.0, 4)
    return out


def cpu_baseline_preprocess(S, seconds=3.0):
    """The reference's own CPU path for this step (mlfoundation_openclip.py:81-90): per-frame PIL resize + crop +
    ToTensor/Normalize, one core, as a DataLoader worker runs it."""
    from wise_amd.feature.mlfoundation_openclip import ClipImageTransform, to_pil_image
    tr = ClipImageTransform(S)
    frames = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (64, 3, 240, 320), dtype=np.uint8))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        tr(to_pil_image(frames[n % 64]))
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": f"{n} frames 240x320 through the PIL transform the reference calls (Pillow), {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one process per GPU with "
                  f"torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no HIP device visible; the HIP path is the only path", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box (WISE_BENCH_REHEARSAL=1): every rank uses device 0 and the ranks talk over gloo —
    # it walks the multi-rank code paths (sharding, barriers, max-over-ranks timing); its numbers mean nothing.
    rehearsal = os.environ.get("WISE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from wise_amd import _lib
    from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
    from wise_amd.index.flat_ip import FlatIPIndex
    from wise_amd.index.sharded import ShardedFlatIPIndex, shard_range

    lib = _lib.lib()
    dev_name = torch.cuda.get_device_name(local_rank)

    # ------------------------------------------------------------------ HP-1: ViT-B/32, bs=256/GPU
    spec = spec_for("ViT-B-32", "openai")
    sd = random_state_dict(spec, 0)
    eng = VitEngine(spec, sd, max_batch=args.batch)
    from wise_amd.feature.mlfoundation_openclip import CLIP_MEAN, CLIP_STD   # (the oracle is used by the cpu_baseline leg only)

    frames = torch.from_numpy(
        np.random.default_rng(1 + rank).integers(0, 256, size=(args.batch, 3, 224, 224), dtype=np.uint8))
    # what preprocess_image hands over: ToTensor + Normalize of the decoded frames, fp32 [B,3,224,224], resident
    x = ((frames.to(torch.float32) / 255.0 - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) /
         torch.tensor(CLIP_STD).view(1, 3, 1, 1)).cuda()
    # four DISTINCT resident batches, taken in turn: 4 x 154 MB do not fit the 256 MB memory-side cache, so no step finds
    # its input there (round 3 fed the same tensor every step, which flattered the patch gather)
    xs = [x] + [torch.roll(x, shifts=17 * (j + 1), dims=0).contiguous() for j in range(3)]
    out_holder = {}

    def vit_step(i):
        # one step = one batch of 256 frames through the tower.  Batches arrive in a stream (extract-features.py's
        # loop), so the engine keeps two of them in flight, each on its own stream and workspace
        # (VitEngine.forward_pipelined); every step's work is complete when the timed region's final sync returns.
        out_holder["h"] = eng.forward_pipelined(xs[i % 4])

    def vit_step_serial(i):
        out_holder["o"] = eng.forward(xs[i % 4])

    def vit_step_one_stream(i):   # every launch on the caller's stream (wise_vit_forward_single)
        out_holder["o"] = eng.forward(xs[i % 4], single_stream=True)

    if args.roofline_only:
        vit_step = vit_step_one_stream
    for i in range(args.warmup):
        vit_step(i)
    dt = timed_region(vit_step, args.steps, world)
    frames_per_s = world * args.batch * args.steps / dt
    if "h" in out_holder:
        out_holder["o"] = out_holder["h"].result()
    serial_dt = None
    if not args.roofline_only:   # the same K steps one batch at a time (wise_vit_forward: two half batches on two streams)
        for i in range(2):
            vit_step_serial(i)
        serial_dt = timed_region(vit_step_serial, args.steps, world)
    norms = out_holder["o"].norm(dim=1)
    assert bool(torch.isfinite(norms).all()) and abs(float(norms.mean()) - 1.0) < 1e-3, "embeddings not unit-norm"

    # Roofline pass: the same K steps again with HIP-event brackets around every GEMM launch.  The timed
    # region above overlaps two half-batches on two streams, which stretches every launch's wall time, so this
    # pass runs the forward on ONE stream: launch durations are then per-kernel and comparable with rocprofv3.
    n_gemm_per_fwd = 4 * (4 * spec.layers + 2)  # capacity (split launches, patch embed, projection)
    for i in range(2):
        vit_step_one_stream(i)
    prof = prof_pass(lib, vit_step_one_stream, args.steps, args.steps * n_gemm_per_fwd + 8)
    g_ms, g_n, g_flop = prof[0]
    gemm_tflops = (g_flop / g_n) / (g_ms / g_n * 1e-3) / 1e12 if g_n else 0.0
    roofline = {
        "kernel": "bf16 MFMA GEMM family (gemm_w4p_kernel persistent 160x256 for QKV / fc1 with the row scale of the folded "
                  "LayerNorm in the epilogue, gemm_w4_kernel 160x256 for the two residual GEMMs — which also emit bf16(x) and "
                  "the rows' statistics — and 224x192 for the patch embedding: one wave per SIMD, MFMA 16x16x32; "
                  "gemm_ring_kernel for the projection): every GEMM launch of the forward, HIP events on the launch "
                  "stream, single-stream pass",
        "bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
        "avg_launch_us": round(g_ms / max(g_n, 1) * 1e3, 2), "launches": int(g_n),
        "flop_per_launch_avg": g_flop / max(g_n, 1),
        **traffic_fields("gemm_bf16_kernel"),
        "end_to_end_tflops": round(frames_per_s / world * spec.flops_per_frame() / 1e12, 2),
        "end_to_end_frac": round(frames_per_s / world * spec.flops_per_frame() / 1e12 / PEAK_BF16_TFLOPS, 4),
        "note": ("with layernorm_fold the family's launches also carry the blocks' LayerNorm work (row statistics, the hi + lo "
                 "split of the residual stream) and the 24 LayerNorm launches per forward are gone: `frac` is not comparable with "
                 "the unfolded tower's (0.34-0.365 in round 3) — `end_to_end_frac` is (0.336 -> 0.353)") if eng.spec.ln_fold else "",
    }

    if args.roofline_only:
        if rank == 0:
            print(json.dumps({"roofline_only": True, "single_stream_frames_per_s": round(frames_per_s, 1),
                              "roofline": roofline}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    result = {
        "metric": METRIC, "value": round(frames_per_s, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "OpenCLIP ViT-B/32 image tower (encode_image + L2 normalise), bs=256 frames per GPU, "
                               "fp32 normalised frames resident in HBM -> [256,512] fp32 unit vectors; "
                               "seeded random weights (no checkpoints offline)",
                   "batches_in_flight": 2, "distinct_input_batches": 4,
                   "layernorm_fold": bool(eng.spec.ln_fold),
                   "one_batch_at_a_time_frames_per_s": (round(world * args.batch * args.steps / serial_dt, 1)
                                                        if serial_dt else None),
                   "global_batch": world * args.batch, "frames_per_gpu": args.batch, "parallelism": f"dp{world}",
                   "gflop_per_frame": round(spec.flops_per_frame() / 1e9, 4), "device": dev_name},
        "roofline": roofline,
    }

    # ------------------------------------------------------------------ HP-2: 10M x 512 flat IP, top-10
    if not args.no_search:
        N, d, k = args.index_rows, args.dim, args.topk
        lo, hi = shard_range(N, rank, world)
        n_loc = hi - lo
        X = torch.empty(n_loc, d, dtype=torch.float32, device="cuda")
        gen = torch.Generator(device="cuda").manual_seed(100 + rank)
        step_rows = 1_000_000
        for s in range(0, n_loc, step_rows):
            e = min(n_loc, s + step_rows)
            blk = torch.randn(e - s, d, generator=gen, device="cuda")
            X[s:e] = blk / blk.norm(dim=1, keepdim=True)
        local = FlatIPIndex(d).adopt(X, None, id_base=lo + 1)  # ids = global row + 1 (sqlite autoincrement)
        index = ShardedFlatIPIndex(local)
        Qh = np.random.default_rng(3).standard_normal((1000, d), dtype=np.float32)
        Qh /= np.linalg.norm(Qh, axis=1, keepdims=True)
        Q = torch.from_numpy(Qh).cuda()
        res = {}

        def search_step(i):
            j = i % 1000
            res["DI"] = index.search_device(Q[j:j + 1], k)

        s_steps = max(args.steps, 20)
        # the fp32 scan alone (what the two-stage search falls back to), for reference
        flat = getattr(index, "local", index)
        flat.shadow = False
        for i in range(3):
            search_step(i)
        qps_f32 = s_steps / timed_region(search_step, s_steps, world)
        flat.shadow = True
        for i in range(max(args.warmup, 3)):
            search_step(i)
        c0 = flat.shadow_counts()
        sdt = timed_region(search_step, s_steps, world)
        qps = s_steps / sdt
        c1 = flat.shadow_counts()
        shadow_stats = (c1[0] - c0[0], c1[1] - c0[1])
        D, I = res["DI"]
        assert bool((D[:, :-1] >= D[:, 1:]).all()) and bool((I >= 1).all())
        sprof = prof_pass(lib, search_step, s_steps, s_steps + 8)
        s_ms, s_n, s_bytes = sprof[1]
        scan_gbs = (s_bytes / s_n) / (s_ms / s_n * 1e-3) / 1e9 if s_n else 0.0

        # what the reference's caller sees (feature_search_index.py:113: numpy in, numpy out, one call per query): H2D of the
        # query, the search, D2H of (D, I), host sync — timed call by call on the host
        lat = []
        if world == 1:
            for i in range(5):
                flat.search(Qh[i:i + 1], k)
            for i in range(50):
                t0 = time.perf_counter()
                Dn, In = flat.search(Qh[100 + i:101 + i], k)
                lat.append(time.perf_counter() - t0)
            lat.sort()

        def search4(i):
            j = (4 * i) % 996
            res["DI4"] = index.search_device(Q[j:j + 4], k)

        def search32(i):  # the batched form: 32 queries share one pass over X (matrix cores)
            j = (32 * i) % 960
            res["DI32"] = index.search_device(Q[j:j + 32], k)

        sdt4 = sdt32 = float("nan")
        if not args.pmc_legs:
            for i in range(3):
                search4(i)
            sdt4 = timed_region(search4, s_steps, world)
            for i in range(3):
                search32(i)
            sdt32 = timed_region(search32, s_steps, world)

        def search256(i):  # cfg-3's nq=256 point: eight 32-query passes per call
            j = (256 * i) % 744
            res["DI256"] = index.search_device(Q[j:j + 256], k)

        for i in range(2):
            search256(i)
        s256 = max(4, s_steps // 5)
        sdt256 = timed_region(search256, s256, world)
        # the k the reference's server and evaluations send (REST `end` = 20: api/routes.py:1171,1407; k = 100:
        # docs/Search-Index-Evaluation.md:109; --topk 1000: docs/Retrieval-Evaluation.md:39), nq = 1, same index
        by_k = {}
        for kk in (() if args.pmc_legs else (20, 100, 1000)):
            def search_k(i, kk=kk):
                j = i % 1000
                res["DIk"] = index.search_device(Q[j:j + 1], kk)
            for i in range(3):
                search_k(i)
            c0k = flat.shadow_counts()
            dtk = timed_region(search_k, s_steps, world)
            c1k = flat.shadow_counts()
            by_k[f"k{kk}_queries_per_s"] = round(s_steps / dtk, 2)
            by_k[f"k{kk}_ms_per_query"] = round(dtk / s_steps * 1e3, 4)
            by_k[f"k{kk}_from_shadow_of_{s_steps}"] = int(c1k[0] - c0k[0])
        def search256_k100(i):   # the evaluation's k = 100 in batches (docs/Search-Index-Evaluation.md:109)
            j = (256 * i) % 744
            res["DI256k"] = index.search_device(Q[j:j + 256], 100)

        if not args.pmc_legs:
            for i in range(2):
                search256_k100(i)
            s256k = max(4, s_steps // 5)
            by_k["batched_nq256_k100_queries_per_s"] = round(256 * s256k / timed_region(search256_k100, s256k, world), 2)
        # ONE rank's share of this index at 8 GPUs (N / 8 rows, nq = 1, k = 10): the strong-scaling number one GPU can
        # measure — the scan shrinks 8-fold, the per-query fixed cost (sample, threshold, finish, gated launches) does not
        shard = None
        if world == 1 and n_loc >= 8 * (1 << 18) and not args.pmc_legs:
            n8 = n_loc // 8
            sh = FlatIPIndex(d).adopt(X[:n8], None, id_base=1)

            def search_shard(i):
                j = i % 1000
                res["DIs"] = sh.search_device(Q[j:j + 1], k)

            for i in range(5):
                search_shard(i)
            s8 = max(s_steps, 50)
            dts = timed_region(search_shard, s8, world)
            sprof8 = prof_pass(lib, search_shard, s8, s8 + 8)
            s8_ms, s8_n, _ = sprof8[1]
            Ds, Is = res["DIs"]
            sh.shadow = False
            Df, If = sh.search_device(Q[(s8 - 1) % 1000:(s8 - 1) % 1000 + 1], k)
            assert torch.equal(Is, If) and torch.equal(Ds, Df), "shard: two-stage != f32 scan"
            shard = {"rows": int(n8), "queries_per_s": round(s8 / dts, 1), "ms_per_query": round(dts / s8 * 1e3, 4),
                     "collect_kernel_us": round(s8_ms / max(s8_n, 1) * 1e3, 2),
                     "everything_else_us": round((dts / s8 - s8_ms / max(s8_n, 1) * 1e-3) * 1e6, 1),
                     "note": "one rank's rows of the 10M x 512 index at 8 GPUs, nq=1, k=10, ids and scores bit-equal to the "
                             "f32 scan of the same rows; the all-gather and merge of the 8-rank exchange are not in it"}
            del sh
        if not args.pmc_legs:
            D32, I32 = index.search_device(Q[:32], k)
            D1, I1 = index.search_device(Q[:1], k)
            # the two kernels sum the d products in different orders: same ids, scores to the tested 2e-5
            assert torch.equal(I32[:1], I1) and bool((D32[:1] - D1).abs().max() <= 2e-5), \
                "batched and single-query scans disagree"
        rows_per_rank = [n_loc]
        if world > 1:
            rows_per_rank = [None] * world
            dist.all_gather_object(rows_per_rank, n_loc)
        result["search"] = {
            "metric": "queries/sec over 10M×512 index (flat IP, top-10, nq=1 per call as the reference issues them)",
            "value": round(qps, 2), "unit": "queries/s", "ms_per_step": round(sdt / s_steps * 1e3, 4),
            "steps": s_steps, "scaling": "strong", "dtype": "f32",
            "config": {"workload": f"IndexFlatIP search, N={N} rows x d={d} fp32 unit rows resident in HBM "
                                   f"({N * d * 4 / 1e9:.2f} GB) with an int8 shadow copy, a scale per row "
                                   f"({N * (d + 4) / 1e9:.2f} GB), k={k}, nq=1; two-stage exact search: sample -> threshold "
                                   "-> every row that could belong to the top-k collected from the int8 rows (two int8 "
                                   "query pieces, v_dot4_i32_i8: exact integer sums) -> fp32 re-scoring (fp32 scan only "
                                   "if more than 16384 rows qualify); any k <= 1024; batches of queries use a bf16 "
                                   "shadow on the matrix cores (built on the first batched search)", "rows_per_gpu": rows_per_rank,
                       "parallelism": f"row-shard x{world} + ONE RCCL all-gather of the packed per-shard (score,id)[nq,k] lists",
                       "allgather_payload_bytes_per_rank_nq1": int(getattr(index, "last_exchange_bytes", 0)) if world > 1
                       else 0, "collectives_per_query": 1 if world > 1 else 0},
            "roofline": {"kernel": "ip_collect_i8_kernel<32> (the pass over the int8 rows of the two-stage exact search)", "bound": "hbm",
                         "achieved": round(scan_gbs, 1),
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(scan_gbs / PEAK_HBM_GBS, 4),
                         "avg_launch_us": round(s_ms / max(s_n, 1) * 1e3, 2), "launches": int(s_n),
                         "bytes_per_launch": s_bytes / max(s_n, 1),
                         "note": "bytes the kernel has to move: the int8 shadow rows and their scales, N*(d+4) per query; the "
                                 "fp32 rows (N*d*4, SURVEY 8(d)) are touched only for the collected candidates and by the "
                                 "fallback scan (fp32_scan_only_queries_per_s: 0.83 of HBM on those bytes)",
                         "fp32_rows_equivalent_gbs": round(n_loc * d * 4 / (s_ms / max(s_n, 1) * 1e-3) / 1e9, 1),
                         **traffic_fields("ip_collect_i8_kernel")},
            "single_query_latency_ms": ({"median": round(lat[len(lat) // 2] * 1e3, 4), "p90": round(lat[int(len(lat) * 0.9)] * 1e3, 4),
                                         "min": round(lat[0] * 1e3, 4),
                                         "note": "FlatIPIndex.search(): numpy [1,d] in, numpy (D, I) out, host-synchronous — the "
                                                 "reference's call shape (feature_search_index.py:113); `value` above is device "
                                                 "throughput with queries already resident and no per-query sync"} if lat else None),
            "two_stage": {"answered_from_the_shadow": int(shadow_stats[0]), "handed_to_fp32_scan": int(shadow_stats[1]),
                          "fp32_scan_only_queries_per_s": round(qps_f32, 2)},
            **by_k,
            "one_rank_of_8_shard": shard,
            "batched_nq4_queries_per_s": round(4 * s_steps / sdt4, 2),
            "batched_nq32_queries_per_s": round(32 * s_steps / sdt32, 2),
            "batched_nq32_ms_per_pass": round(sdt32 / s_steps * 1e3, 4),
            "batched_nq256_queries_per_s": round(256 * s256 / sdt256, 2),
            "batched_nq256_roofline": {"kernel": "ip_scan_shadow64_kernel<4,128,false> (stage 1 of the batched two-stage search)",
                                       "bound": "hbm",
                                       "achieved": round(N * d * 2 / world / (sdt256 / s256 / 2) / 1e9, 1),
                                       "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(N * d * 2 / world / (sdt256 / s256 / 2) / 1e9 / PEAK_HBM_GBS, 4),
                                       "note": "bf16 shadow rows (N*d*2 bytes) per pass of 128 queries (the query enters as one "
                                               "bf16 piece, its rounding carried in the error bound); whole call / 2 passes "
                                               "(sample, thresholds, collect, per-query refine + fp32 re-scoring + select, "
                                               "gated fallback launches), per GPU",
                                       **traffic_fields("ip_scan_shadow64_kernel")},
            "batched_nq32_roofline": {"kernel": "ip_scan_shadow64_kernel<4,64,false> (one pass, half its 64 query slots used)",
                                      "bound": "hbm",
                                      "achieved": round(N * d * 2 / world / (sdt32 / s_steps) / 1e9, 1),
                                      "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(N * d * 2 / world / (sdt32 / s_steps) / 1e9 / PEAK_HBM_GBS, 4),
                                      "note": "bf16 shadow rows (N*d*2 bytes) per call; whole call, per GPU",
                                      **traffic_fields("ip_scan_shadow64_kernel")},
        }
        # the regime the reference really indexes (2-fps frames of the same videos, extract-features.py:292-297,353): runs
        # of 20 near-duplicate rows (cosine >= 0.999).  Same N, same kernels; queries are noisy copies of indexed rows
        # (they have true neighbours) and random directions, alternating.  Rows are rewritten in place.
        del index, local
        torch.cuda.empty_cache()
        local = index = None
        if not args.pmc_legs:
            per = 20
            for s in range(0, n_loc, 1_000_000):
                e = min(n_loc, s + 1_000_000)
                items = (e - s + per - 1) // per
                base = torch.randn(items, 1, d, generator=gen, device="cuda")
                blk = (base / base.norm(dim=2, keepdim=True) + (0.03 / d ** 0.5) * torch.randn(items, per, d, generator=gen, device="cuda"))
                X[s:e] = (blk / blk.norm(dim=2, keepdim=True)).reshape(-1, d)[:e - s]
            local = FlatIPIndex(d).adopt(X, None, id_base=lo + 1)
            index = ShardedFlatIPIndex(local)
            gq = torch.Generator(device="cuda").manual_seed(9)          # the same queries on every rank
            Qc = torch.randn(200, d, generator=gq, device="cuda")
            Qc /= Qc.norm(dim=1, keepdim=True)
            if rank == 0:
                pick = torch.randint(0, n_loc, (100,), generator=gq, device="cuda")
                Qc[0::2] = torch.nn.functional.normalize(X[pick] + 0.02 * Qc[0::2], dim=1)
            if world > 1:
                dist.broadcast(Qc, src=0)

            def search_clustered(i):
                res["DIc"] = index.search_device(Qc[i % 200:i % 200 + 1], k)

            for i in range(3):
                search_clustered(i)
            cc0 = local.shadow_counts()
            cdt = timed_region(search_clustered, s_steps, world)
            cc1 = local.shadow_counts()
            Dc, Ic = index.search_device(Qc[:1], k)
            local.shadow = False
            Dcf, Icf = index.search_device(Qc[:1], k)
            assert torch.equal(Ic, Icf) and bool((Dc - Dcf).abs().max() <= 2e-6), "clustered rows: two-stage != f32 scan"
            result["search"]["clustered"] = {
                "queries_per_s": round(s_steps / cdt, 2), "ms_per_query": round(cdt / s_steps * 1e3, 4),
                "answered_from_the_shadow": int(cc1[0] - cc0[0]), "handed_to_fp32_scan": int(cc1[1] - cc0[1]),
                "workload": f"same N x d, rows in runs of {per} near-duplicates (cosine >= 0.999); queries alternate between "
                            f"noisy copies of indexed rows and random directions; ids and scores equal the f32 scan's"}
        del X, local, index
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ other BASELINE configs (short legs)
    if not args.no_extra and not args.pmc_legs:
        extra = {}
        # cfg-5: MS-CLAP HTSAT, 10-s clips @48 kHz, bs=128 per GPU (weak scaling, no collective)
        from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict

        ab = 128
        heng = HtsatEngine(random_htsat_state_dict(0), max_batch=ab, max_samples=480000)
        wav = 0.1 * torch.randn(ab, 480000, generator=torch.Generator(device="cuda").manual_seed(4 + rank),
                                device="cuda")
        hold = {}

        def clap_step_serial(i):
            hold["o"] = heng.forward(wav)

        def clap_step(i):  # two batches in flight, like the headline leg
            hold["p"] = heng.forward_pipelined(wav)

        for i in range(2):
            clap_step_serial(i)
        a_steps = max(4, min(args.steps, 30))
        adt_serial = timed_region(clap_step_serial, a_steps, world)
        for i in range(2):
            clap_step(i)
        torch.cuda.synchronize()
        adt = timed_region(clap_step, a_steps, world)
        assert abs(float(hold["o"].norm(dim=1).mean()) - 1.0) < 1e-3
        assert torch.equal(hold["p"].result(), hold["o"]), "HTSAT: in-flight and serial embeddings differ"
        # roofline of the leg.  Two figures, because the forward is two kinds of work:
        #  * the GEMM family (every wise_gemm_bf16 / gemm_ln / fused-MLP launch, HIP events on the launch stream) against
        #    the bf16 MFMA peak;
        #  * the whole forward against HBM: the bytes every kernel must move given the kernel boundaries as built
        #    (fp32 residual stream, bf16 qkv / attention / hidden activations, inputs and outputs of each launch
        #    counted once) — and, beside it, what a fully fused Swin block would move (x in, x out).
        hprof = prof_pass(lib, clap_step_serial, a_steps, a_steps * 200 + 8)
        hg_ms, hg_n, hg_flop = hprof[0]
        T0, blocks = ab * 4096, 0.0
        fused_ideal = 0.0
        for depth, C_ in ((2, 96), (2, 192), (6, 384), (2, 768)):
            T = T0 // (C_ // 96) ** 2
            x32, a16 = T * C_ * 4, T * C_ * 2
            ln_fused = False                # (LayerNorm inside the GEMM's A-tile build: only stage 1 used it, and stage 1 is now
            mlp_fused = C_ == 96            #  two kernels per block:) the attention half and the whole MLP, one kernel each
            mlp_stream = heng.mlp_stream and C_ in (192, 384)      # LayerNorm launch + wise_mlp_stream: the hidden rows never written
            attn_stream = heng.attn_stream and C_ in (192, 384)    # wise_swin_qkv_attn (x in, attention output out) + the projection GEMM
            attn_half = 2 * x32 if C_ == 96 else ((x32 + a16) + (a16 + 2 * x32) if attn_stream else
                                                  (x32 + a16 + a16 + 3 * a16) + (3 * a16 + a16) + (a16 + 2 * x32))
            mlp_half = 2 * x32 if mlp_fused else ((x32 + a16) + (a16 + 2 * x32) if mlp_stream else
                                                  ((x32 + 4 * a16 if ln_fused else x32 + a16 + a16 + 4 * a16) + 4 * a16 + 2 * x32))
            blocks += depth * (attn_half + mlp_half)
            fused_ideal += depth * 2 * x32
        front = ab * 480000 * 4 + ab * 1024 * 64 * 4 * 2 + ab * 4096 * 96 * 4
        hbm_bytes = blocks + front
        step_s = adt_serial / a_steps
        extra["clap_htsat"] = {"value": round(world * ab * a_steps / adt, 1), "unit": "clips/s",
                               "ms_per_step": round(adt / a_steps * 1e3, 3), "steps": a_steps,
                               "batches_in_flight": 2,
                               "one_batch_at_a_time_clips_per_s": round(world * ab * a_steps / adt_serial, 1),
                               "config": {"workload": "MS-CLAP 2023 HTSAT audio encoder + projection, 10-s clips "
                                                      "(480000 samples @48 kHz), bs=128 per GPU", "dtype": "bf16",
                                          "gflop_per_clip": 11.82, "one_kernel_mlp_stages_2_3": bool(heng.mlp_stream),
                                          "one_kernel_attention_stages_2_3": bool(heng.attn_stream)},
                               "tflops": round(world * ab * a_steps / adt * 11.82e9 / 1e12 / world, 2),
                               "roofline": {
                                   "kernel": "whole forward, one batch at a time (front end + 12 Swin blocks + head)",
                                   "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                                   "achieved": round(hbm_bytes / step_s / 1e9, 1),
                                   "frac": round(hbm_bytes / step_s / 1e9 / PEAK_HBM_GBS, 4),
                                   "bytes_per_forward_at_kernel_boundaries": hbm_bytes,
                                   "bytes_per_forward_if_each_block_were_one_kernel": fused_ideal + front,
                                   **traffic_fields("htsat_forward"),
                                   "gemm_family": {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_BF16_TFLOPS,
                                                   "achieved": round(hg_flop / max(hg_ms, 1e-9) / 1e9, 2),
                                                   "frac": round(hg_flop / max(hg_ms, 1e-9) / 1e9 / PEAK_BF16_TFLOPS, 4),
                                                   "launches_per_forward": int(hg_n // a_steps),
                                                   "ms_per_forward": round(hg_ms / a_steps, 3),
                                                   "share_of_forward": round(hg_ms / a_steps / (step_s * 1e3), 3)}}}
        del heng
        # MS-CLAP 2022's audio encoder (PANNs Cnn14), same clips, bs=128 per GPU: convolutions as implicit GEMMs
        from wise_amd.feature.cnn14 import Cnn14Engine, flops_per_clip, random_cnn14_state_dict

        cb = 128
        ceng = Cnn14Engine(random_cnn14_state_dict(0), max_batch=cb, max_samples=480000)
        cwav = wav[:cb]

        def cnn_step_serial(i):
            hold["co"] = ceng.forward(cwav)

        def cnn_step(i):
            hold["cp"] = ceng.forward_pipelined(cwav)

        for i in range(2):
            cnn_step_serial(i)
        c_steps = max(4, min(args.steps, 16))
        cdt_serial = timed_region(cnn_step_serial, c_steps, world)
        for i in range(2):
            cnn_step(i)
        torch.cuda.synchronize()
        cdt2 = timed_region(cnn_step, c_steps, world)
        assert abs(float(hold["co"].norm(dim=1).mean()) - 1.0) < 1e-3
        assert torch.equal(hold["cp"].result(), hold["co"]), "Cnn14: in-flight and serial embeddings differ"
        cprof = prof_pass(lib, cnn_step_serial, c_steps, c_steps * 40 + 8)
        cg_ms, cg_n, cg_flop = cprof[0]
        cfl = flops_per_clip(480000)
        # (one batch at a time is this encoder's faster mode: block 1's persistent workgroups hold every CU's registers, so a
        #  second batch beside it only interleaves — both figures are reported, `value` is the serial one)
        extra["clap_cnn14"] = {"value": round(world * cb * c_steps / cdt_serial, 1), "unit": "clips/s",
                               "ms_per_step": round(cdt_serial / c_steps * 1e3, 3), "steps": c_steps, "batches_in_flight": 1,
                               "two_batches_in_flight_clips_per_s": round(world * cb * c_steps / cdt2, 1),
                               "config": {"workload": "MS-CLAP 2022 Cnn14 audio encoder + projection, 10-s clips (480000 "
                                                      "samples @48 kHz), bs=128 per GPU", "dtype": "bf16",
                                          "gflop_per_clip": round(cfl / 1e9, 2)},
                               "tflops": round(cb * c_steps / cdt_serial * cfl / 1e12, 2),
                               "frac_of_bf16_peak": round(cb * c_steps / cdt_serial * cfl / 1e12 / PEAK_BF16_TFLOPS, 4),
                               "roofline": {"kernel": "the ten 3x3 convolutions of blocks 2-6 as implicit GEMMs + the head's GEMMs, "
                                                      "one batch at a time (HIP events on the launch stream; block 1 is its "
                                                      "own fused kernel and is not in this family)",
                                            "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_BF16_TFLOPS,
                                            "achieved": round(cg_flop / max(cg_ms, 1e-9) / 1e9, 2),
                                            "frac": round(cg_flop / max(cg_ms, 1e-9) / 1e9 / PEAK_BF16_TFLOPS, 4),
                                            "launches_per_forward": int(cg_n // c_steps),
                                            "ms_per_forward": round(cg_ms / c_steps, 3),
                                            "share_of_forward": round(cg_ms / c_steps / (cdt_serial / c_steps * 1e3), 3)}}
        hold.pop("co", None); hold.pop("cp", None)
        del ceng, cwav, wav
        torch.cuda.empty_cache()
        # cfg-4 (image half): ViT-L/14 at bs=256 per GPU; and ViT-H/14 (head width 80), the image tower of the reference's
        # default feature id (extract-features.py:192)
        from wise_amd.feature.siglip import SIGLIP_VISION, random_siglip_vision_state_dict
        for key, lname, ltag, label in (("vit_l14", "ViT-L-14", "openai", "ViT-L/14"),
                                        ("vit_h14", "ViT-H-14", "laion2b_s32b_b79k", "ViT-H/14"),
                                        ("siglip_l16_384", "ViT-L-16-SigLIP-384", "webli", "ViT-L/16 SigLIP 384 px (timm tower, "
                                         "attention-pool head; the video model of the reference's tests/test-kinetics-6.sh)")):
            if lname in SIGLIP_VISION:
                lspec = SIGLIP_VISION[lname]
                leng = VitEngine(lspec, random_siglip_vision_state_dict(lspec, 0), max_batch=args.batch)
                x_l = torch.randn(args.batch, 3, lspec.image_size, lspec.image_size, device="cuda",
                                  generator=torch.Generator(device="cuda").manual_seed(11 + rank)).clamp_(-1, 1)
            else:
                lspec = spec_for(lname, ltag)
                leng = VitEngine(lspec, random_state_dict(lspec, 0), max_batch=args.batch)
                x_l = x

            def l14_step(i):
                hold["l"] = leng.forward_pipelined(x_l)

            def l14_step_serial(i):
                hold["ls"] = leng.forward(x_l)

            l_steps = max(4, min(args.steps, 6))
            for i in range(2):
                l14_step_serial(i)
            ldt_serial = timed_region(l14_step_serial, l_steps, world)
            for i in range(2):
                l14_step(i)
            torch.cuda.synchronize()
            ldt = timed_region(l14_step, l_steps, world)
            assert torch.equal(hold["l"].result(), hold["ls"]), f"{label}: in-flight and serial embeddings differ"
            lfps = world * args.batch * l_steps / ldt
            extra[key] = {"value": round(lfps, 1), "unit": "frames/s", "ms_per_step": round(ldt / l_steps * 1e3, 3),
                          "steps": l_steps, "batches_in_flight": 2,
                          "one_batch_at_a_time_frames_per_s": round(world * args.batch * l_steps / ldt_serial, 1),
                          "config": {"workload": f"OpenCLIP {label} image tower, bs={args.batch} per GPU",
                                     "gflop_per_frame": round(lspec.flops_per_frame() / 1e9, 2)},
                          "tflops": round(lfps / world * lspec.flops_per_frame() / 1e12, 2),
                          "frac_of_bf16_peak": round(lfps / world * lspec.flops_per_frame() / 1e12 / PEAK_BF16_TFLOPS, 4)}
            hold.pop("l", None); hold.pop("ls", None)
            del leng, x_l
            torch.cuda.empty_cache()
        # f2: decoded uint8 frames [256,3,240,320] resident in HBM -> GPU transform -> ViT-B/32 (uint8 in)
        from wise_amd.feature.preprocess import ClipPreprocessor, make_plan

        fh, fw = 240, 320
        raw = torch.randint(0, 256, (args.batch, 3, fh, fw), dtype=torch.uint8, device="cuda",
                            generator=torch.Generator(device="cuda").manual_seed(7 + rank))
        pre = ClipPreprocessor(spec.image_size)
        crops = [torch.empty((args.batch, 3, spec.image_size, spec.image_size), dtype=torch.uint8, device="cuda")
                 for _ in range(3)]   # a crop buffer is reused only after the forward that read it has long finished
        crop = crops[0]

        crop_users = [None, None, None]

        def u8_step(i):
            c = crops[i % 3]
            if crop_users[i % 3] is not None:
                crop_users[i % 3].result()     # stream-order: the batch that read this buffer is done (no host sync)
            pre(raw, c)
            hold["uh"] = crop_users[i % 3] = eng.forward_pipelined(c)

        for i in range(3):
            u8_step(i)
        u_steps = max(5, min(args.steps, 40))
        udt = timed_region(u8_step, u_steps, world)
        hold["u"] = hold["uh"].result()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            pre(raw, crop)
        e1.record()
        torch.cuda.synchronize()
        pre_us = e0.elapsed_time(e1) / 20 * 1e3
        plan = make_plan(fh, fw, spec.image_size)
        sx, sy = fw / plan.new_w, fh / plan.new_h
        need_w = min(fw, int(spec.image_size * sx + 4 * max(sx, 1.0)) + 1)   # columns / rows the crop depends on
        need_h = min(fh, int(spec.image_size * sy + 4 * max(sy, 1.0)) + 1)
        pre_bytes = args.batch * 3 * (need_w * need_h + spec.image_size ** 2)
        extra["u8_frames_to_embeddings"] = {
            "value": round(world * args.batch * u_steps / udt, 1), "unit": "frames/s",
            "ms_per_step": round(udt / u_steps * 1e3, 3), "steps": u_steps,
            "config": {"workload": f"uint8 decoded frames [{args.batch},3,{fh},{fw}] resident in HBM -> Pillow-exact "
                                   f"bicubic resize + centre crop kernel -> ViT-B/32 with ToTensor/Normalize fused "
                                   f"into the patch gather"},
            "preprocess_kernel": {"kernel": "clip_resize_kernel", "avg_launch_us": round(pre_us, 1), "bound": "hbm",
                                  "bytes_per_launch": pre_bytes,
                                  "achieved": round(pre_bytes / pre_us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(pre_bytes / pre_us / 1e3 / PEAK_HBM_GBS, 4),
                                  **traffic_fields("clip_resize_kernel"),
                                  "frames_per_s": round(args.batch / pre_us * 1e6, 0)}}
        # The same path fed as the reference's callers feed it (extract-features.py:332-341 hands HOST tensors over,
        # mlfoundation_openclip.py:97,101 moves them in and the result out): pinned host uint8 frames -> H2D -> transform ->
        # tower -> D2H of the [256,512] embeddings.  Four distinct host batches in turn.  (a) the reference's synchronous call
        # shape, one batch at a time; (b) the same work with the copy of batch i+1 on a copy stream under the compute of batch i.
        hraw = [torch.randint(0, 256, (args.batch, 3, fh, fw), dtype=torch.uint8,
                              generator=torch.Generator().manual_seed(70 + j)).pin_memory() for j in range(4)]
        dbuf = [torch.empty_like(raw) for _ in range(3)]
        hout = [torch.empty(args.batch, spec.embed_dim, dtype=torch.float32).pin_memory() for _ in range(3)]
        frame_bytes = args.batch * 3 * fh * fw

        def host_sync_step(i):
            d = dbuf[0]
            d.copy_(hraw[i % 4], non_blocking=True)
            pre(d, crops[0])
            hout[0].copy_(eng.forward(crops[0]), non_blocking=False)      # .cpu(): blocks like the reference's call

        from wise_amd._streams import concurrent_streams
        eng.forward_pipelined(crops[0]).result()          # (the engine's two streams exist from here on)
        copy_s, out_s = concurrent_streams(2, "cuda", beside=[sl["stream"] for sl in eng._slots])   # streams that run BESIDE them
        h_users, h_events = [None] * 3, [None] * 3

        def host_async_step(i):
            j = i % 3
            if h_users[j] is not None:
                h_users[j].synchronize()              # slot j's previous embeddings are on the host: its buffers are free
            with torch.cuda.stream(copy_s):
                dbuf[j].copy_(hraw[i % 4], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_s)
            torch.cuda.current_stream().wait_event(ev)
            pre(dbuf[j], crops[j])
            pend = eng.forward_pipelined(crops[j])
            # the embeddings leave on a stream of their own, ordered behind THIS batch's forward only (as the plugin's
            # extract_image_features_async does): on the caller's stream the wait would hold back the next batch's transform
            # and forward until this one has finished — one batch in flight, not two
            with torch.cuda.stream(out_s):
                out = pend.result()
                hout[j].copy_(out, non_blocking=True)
                out.record_stream(out_s)
                done = torch.cuda.Event()
                done.record(out_s)
            h_users[j] = done

        for i in range(3):
            host_sync_step(i)
        hs_steps = max(5, min(args.steps, 20))
        hsdt = timed_region(host_sync_step, hs_steps, world)
        for i in range(4):
            host_async_step(i)
        torch.cuda.synchronize()
        hadt = timed_region(host_async_step, u_steps, world)
        assert abs(float(hout[(u_steps - 1) % 3].norm(dim=1).mean()) - 1.0) < 1e-3
        e0.record()
        for i in range(8):
            dbuf[i % 3].copy_(hraw[i % 4], non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        h2d_gbs = 8 * frame_bytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
        extra["host_frames_to_embeddings"] = {
            "value": round(world * args.batch * u_steps / hadt, 1), "unit": "frames/s",
            "ms_per_step": round(hadt / u_steps * 1e3, 3), "steps": u_steps,
            "pcie_gbs_achieved": round(frame_bytes * u_steps / hadt / 1e9, 2),
            "h2d_copy_alone_gbs": round(h2d_gbs, 2),
            "synchronous_call_shape": {"frames_per_s": round(world * args.batch * hs_steps / hsdt, 1),
                                       "ms_per_batch": round(hsdt / hs_steps * 1e3, 3),
                                       "pcie_gbs_achieved": round(frame_bytes * hs_steps / hsdt / 1e9, 2)},
            "config": {"workload": f"PINNED HOST uint8 frames [{args.batch},3,{fh},{fw}] (4 distinct batches in turn) -> H2D -> "
                                   f"clip_resize_kernel -> ViT-B/32 -> D2H of [{args.batch},{spec.embed_dim}] fp32; `value`: copies "
                                   f"on a copy stream, two batches in flight; `synchronous_call_shape`: copy in, transform, "
                                   f"forward, copy out, one batch at a time (ref:src/feature/mlfoundation_openclip.py:92-101)"}}
        del raw, crop, crops, hraw, dbuf, hout
        # f4: the query side — CLIP text tower (ViT-B/32 text), one query at a time and in batches of 256
        from wise_amd.feature.text import EOT_TOKEN, SOT_TOKEN, TextEngine, random_text_state_dict, text_spec_for

        tspec = text_spec_for("ViT-B-32", "openai")
        teng = TextEngine(tspec, random_text_state_dict(tspec, 0), max_batch=256)
        if world > 1:
            teng.graph_max_batch = 0   # no graph capture beside a live RCCL communicator; direct launches instead
        trng = np.random.default_rng(11 + rank)
        toks = np.zeros((256, tspec.context), dtype=np.int32)
        for i in range(256):
            k = int(trng.integers(3, 20))
            toks[i, 0] = SOT_TOKEN
            toks[i, 1:1 + k] = trng.integers(1, SOT_TOKEN, k)
            toks[i, 1 + k] = EOT_TOKEN
        toks = torch.from_numpy(toks).cuda()

        def text1(i):
            hold["t"] = teng.forward(toks[i % 256:i % 256 + 1])

        def text256(i):
            hold["t"] = teng.forward(toks)

        for i in range(3):
            text1(i); text256(i)
        t_steps = max(10, min(args.steps, 50))
        # (latency figure: the best of three timed regions — one host-side stall of tens of ms once turned 0.64 into 2.9)
        tdt1 = min(timed_region(text1, t_steps, world) for _ in range(3))
        tdt256 = timed_region(text256, t_steps, world)
        extra["clip_text_tower"] = {
            "value": round(world * 256 * t_steps / tdt256, 1), "unit": "queries/s (batches of 256)",
            "single_query_ms": round(tdt1 / t_steps * 1e3, 4), "ms_per_batch256": round(tdt256 / t_steps * 1e3, 3),
            "config": {"workload": "OpenCLIP ViT-B/32 text tower: token ids [n,77] resident in HBM -> unit vectors "
                                   "[n,512]; seeded weights", "gflop_per_query": round(tspec.flops_per_query() / 1e9, 3)},
            "tflops_batch256": round(256 * t_steps / tdt256 * tspec.flops_per_query() / 1e12, 2)}
        del teng, toks
        torch.cuda.empty_cache()
        # the query side of the reference's DEFAULT model pair (xlm-roberta-large-ViT-H-14, extract-features.py:192)
        from wise_amd.feature.xlmr_text import XLMR_SPECS, XlmrTextEngine, random_xlmr_state_dict
        xspec = XLMR_SPECS["xlm-roberta-large-ViT-H-14"]
        xeng = XlmrTextEngine(xspec, random_xlmr_state_dict(xspec, 0), max_batch=256)
        if world > 1:
            xeng.graph_max_batch = 0
        xt = np.full((256, xspec.context), xspec.pad_id, dtype=np.int32)
        for i in range(256):
            kx = int(trng.integers(3, 24))
            xt[i, 0] = 0
            xt[i, 1:1 + kx] = trng.integers(4, xspec.vocab, kx)
            xt[i, 1 + kx] = 2
        xt = torch.from_numpy(xt).cuda()

        def xtext1(i):
            hold["t"] = xeng.forward(xt[i % 256:i % 256 + 1])

        def xtext256(i):
            hold["t"] = xeng.forward(xt)

        for i in range(3):
            xtext1(i); xtext256(i)
        x_steps = max(5, min(args.steps, 20))
        xdt1 = min(timed_region(xtext1, x_steps, world) for _ in range(3))
        xdt256 = timed_region(xtext256, x_steps, world)
        extra["xlmr_text_tower"] = {
            "value": round(world * 256 * x_steps / xdt256, 1), "unit": "queries/s (batches of 256)",
            "single_query_ms": round(xdt1 / x_steps * 1e3, 4), "ms_per_batch256": round(xdt256 / x_steps * 1e3, 3),
            "config": {"workload": "XLM-RoBERTa-large text tower of xlm-roberta-large-ViT-H-14 (open_clip HFTextEncoder: mean "
                                   "pooler + MLP projection): token ids [n,77] resident in HBM -> unit vectors [n,1024]; seeded "
                                   "weights", "gflop_per_query": round(xspec.flops_per_query() / 1e9, 3)},
            "tflops_batch256": round(256 * x_steps / xdt256 * xspec.flops_per_query() / 1e12, 2)}
        del xeng, xt
        torch.cuda.empty_cache()
        # f4: IndexIVFFlat at the reference's geometry for 10M rows (nlist = 10 * round(sqrt(N)) = 31620, nprobe 32 =
        # the REST default, routes.py:902).  The lists are synthesised (equal sizes, rows = list direction + noise,
        # centroid = normalised list mean): training 31620 cells on 3.2M rows is hours of k-means and is not what
        # this leg times; recall and exactness are covered by tests/test_gpu_ivf.py.
        from wise_amd.index.ivf_flat import IVFFlatIPIndex

        nlist, per = 31620, max(1, (args.index_rows // 31620))
        n_ivf = nlist * per
        gen = torch.Generator(device="cuda").manual_seed(200 + rank)
        dirs = torch.randn(nlist, args.dim, generator=gen, device="cuda")
        Xi = torch.empty(n_ivf, args.dim, dtype=torch.float32, device="cuda")
        for s0 in range(0, nlist, 1024):
            e0 = min(nlist, s0 + 1024)
            blk = dirs[s0:e0, None, :] + 0.7 * torch.randn(e0 - s0, per, args.dim, generator=gen, device="cuda")
            Xi[s0 * per:e0 * per] = (blk / blk.norm(dim=2, keepdim=True)).reshape(-1, args.dim)
        cent = Xi.view(nlist, per, args.dim).mean(dim=1)
        cent = cent / cent.norm(dim=1, keepdim=True)
        ivf = IVFFlatIPIndex(args.dim, nlist)
        ivf.set_centroids(cent)
        ivf.adopt_lists(Xi, torch.arange(n_ivf, device="cuda", dtype=torch.int64) + 1,
                        torch.arange(nlist + 1, device="cuda", dtype=torch.int64) * per)
        Qi = Xi[torch.randint(0, n_ivf, (1000,), generator=gen, device="cuda")] + 0.05 * torch.randn(
            1000, args.dim, generator=gen, device="cuda")
        ivf_res = {}
        for nprobe in (32, 1024):
            ivf.nprobe = nprobe

            def ivf1(i):
                hold["ivf"] = ivf.search_device(Qi[i % 1000:i % 1000 + 1], args.topk)

            def ivf256(i):
                hold["ivf"] = ivf.search_device(Qi[:256], args.topk)

            for i in range(3):
                ivf1(i); ivf256(i)
            i_steps = max(10, min(args.steps, 50))
            # The FIRST timed region is the figure.  Two round-3 sequences (git: fe9ae72 r03_c_bench.json 9.36 ms/query,
            # cd12ecb r03_d 3.56 ms) had one stall of 0.2-0.5 s inside it — not reproduced by a replay of this leg alone
            # (tools/ivf_stall.py: 0.187 ms in every region, no allocator retry, no device allocation inside the regions), so
            # it is left over from the legs in front of it; a second region and the slowest host-side enqueue of the first are
            # reported beside the figure so that a stall shows up as what it is instead of being hidden by a minimum.
            enq = []

            def ivf1_timed(i):
                t0 = time.perf_counter()
                ivf1(i)
                enq.append(time.perf_counter() - t0)

            t1 = timed_region(ivf1_timed, i_steps, world)
            t1b = timed_region(ivf1, i_steps, world)
            t256 = timed_region(ivf256, i_steps, world)
            ivf_res[f"nprobe{nprobe}"] = {"single_query_ms": round(t1 / i_steps * 1e3, 4),
                                          "queries_per_s_nq1": round(i_steps / t1, 1),
                                          "second_region_single_query_ms": round(t1b / i_steps * 1e3, 4),
                                          "first_region_slowest_enqueue_ms": round(max(enq) * 1e3, 3),
                                          "queries_per_s_nq256": round(256 * i_steps / t256, 1)}
        extra["ivf_flat"] = {"value": ivf_res["nprobe32"]["queries_per_s_nq1"], "unit": "queries/s (nq=1, nprobe=32)",
                             "config": {"workload": f"IndexIVFFlat (inner product), {n_ivf} rows x d={args.dim} in "
                                                    f"{nlist} synthesised lists of {per} rows per GPU, top-{args.topk}; "
                                                    f"coarse top-nprobe over the centroids + list scan"},
                             **ivf_res}
        del ivf, Xi, dirs, cent
        result["extra"] = extra

    # ------------------------------------------------------------------ CPU baselines (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_vit(spec, sd, args.cpu_seconds)
        if "search" in result:
            result["search"]["cpu_baseline"] = cpu_baseline_search(args.dim, args.topk, args.cpu_seconds)
        if "extra" in result and "u8_frames_to_embeddings" in result["extra"]:
            result["extra"]["u8_frames_to_embeddings"]["cpu_baseline"] = cpu_baseline_preprocess(spec.image_size)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
