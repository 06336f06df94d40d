"""Two ViT-B/32 batches in flight on two streams that own DISJOINT halves of every XCD's compute units
(hipExtStreamCreateWithCUMask) against the product's two unmasked streams.

Why it could pay: the one-wave-per-SIMD GEMMs own a whole CU (512 registers per wave), so a second stream's kernels never
share a CU with them — two unmasked streams alternate full-chip kernels, and every kernel's HBM-bound epilogue (the fp32
read-modify-write burst of the residual GEMMs: ~10 of out-proj's 24 us) idles the matrix cores of all 256 CUs.  With each
stream on its own 128 CUs the two batches' kernels run side by side, out of phase: one half's store burst overlaps the
other half's MFMA loop and gets the whole HBM bandwidth to itself.

Mask bits: KFD maps bit i of the mask to XCC (i mod 8), slot (i div 8) of that XCC, so every XCC keeps CUs in both halves
(an XCC with no CU enabled would never finish its share of a dispatch).

    python tools/vit_cumask.py [steps=60]
"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


def run(eng, batches, steps, streams=None):
    # hipExtStreamCreateWithCUMask makes BLOCKING streams: anything enqueued on the legacy null stream (torch's default
    # "current stream", where forward_pipelined records the event its slot stream waits for) synchronises with them and
    # serialises the two batches.  The caller's stream is therefore a non-blocking side stream here.
    with torch.cuda.stream(CALLER):
        return _run(eng, batches, steps, streams)


def _run(eng, batches, steps, streams=None):
    if streams is not None:
        eng._slots = [{"stream": s, "ws": None} for s in streams]
        eng._next_slot = 0
    elif hasattr(eng, "_slots"):
        del eng._slots
    for i in range(8):
        h = eng.forward_pipelined(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        h = eng.forward_pipelined(batches[i % len(batches)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return h.result(), steps * batches[0].shape[0] / dt, dt / steps * 1e3


CALLER = None


def main():
    global CALLER
    CALLER = torch.cuda.Stream()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    spec = spec_for("ViT-B-32", "openai")
    eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=256)
    g = torch.Generator(device="cuda").manual_seed(1)
    batches = [torch.randn(256, 3, 224, 224, generator=g, device="cuda") for _ in range(4)]
    ref, fps, ms = run(eng, batches, steps)
    print(f"two unmasked streams          : {fps / 1e3:7.2f} k frames/s  {ms:.3f} ms/step", flush=True)
    all_bits = range(256)
    splits = {
        "slots 0-15 | 16-31 of every XCC": ([i for i in all_bits if (i // 8) < 16], [i for i in all_bits if (i // 8) >= 16]),
        "even | odd slots of every XCC  ": ([i for i in all_bits if (i // 8) % 2 == 0], [i for i in all_bits if (i // 8) % 2 == 1]),
        "160 | 96 CUs (slots 0-19 | rest)": ([i for i in all_bits if (i // 8) < 20], [i for i in all_bits if (i // 8) >= 20]),
    }
    for name, (a, b) in splits.items():
        sa, sb = masked_stream(a), masked_stream(b)
        out, fps, ms = run(eng, batches, steps, [sa, sb])
        same = torch.equal(out, ref)
        print(f"{name}: {fps / 1e3:7.2f} k frames/s  {ms:.3f} ms/step  same bits: {same}", flush=True)
    # one batch at a time on a half-chip stream (how long does a forward take on 128 CUs?)
    sa = masked_stream([i for i in all_bits if (i // 8) < 16])
    with torch.cuda.stream(sa):
        for _ in range(3):
            eng.forward(batches[0], single_stream=True)
        sa.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            eng.forward(batches[i % 4], single_stream=True)
        sa.synchronize()
        print(f"one stream on 128 CUs, one batch at a time: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward")
    for _ in range(3):
        eng.forward(batches[0], single_stream=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        eng.forward(batches[i % 4], single_stream=True)
    torch.cuda.synchronize()
    print(f"one stream on 256 CUs, one batch at a time: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward")


if __name__ == "__main__":
    main()
