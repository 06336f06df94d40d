"""ViT-B/32 (and others) without the LayerNorm fold, with it (wise_vit_config.ln_fold = 1) and — where the shape allows: up to
64 tokens, 12 heads — with attention + out-projection as one kernel on top (ln_fold = 2); same process, interleaved rounds:
two batches in flight (the bench's headline form), one batch at a time (two half batches), one stream; and small batches.

    python tools/vit_fold_ab.py [model=ViT-B-32] [steps=40]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for  # noqa: E402


def timed(fn, steps):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-32"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    spec = spec_for(model, "openai")
    sd = random_state_dict(spec, 0)
    modes = (0, 1, 2) if (spec.tokens <= 64 and spec.heads == 12) else (0, 1)
    engs = {f: VitEngine(spec, sd, max_batch=256, ln_fold=f) for f in modes}
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(256, 3, spec.image_size, spec.image_size, generator=g, device="cuda") for _ in range(4)]
    hold = {}
    for rnd in range(3):
        for f, eng in engs.items():
            p = timed(lambda i: hold.__setitem__("h", eng.forward_pipelined(xs[i % 4])), steps)
            s = timed(lambda i: hold.__setitem__("o", eng.forward(xs[i % 4])), steps)
            o = timed(lambda i: hold.__setitem__("o", eng.forward(xs[i % 4], single_stream=True)), steps)
            print(f"round {rnd} fold={int(f)}  bs=256: two in flight {p:.3f} ms ({256 / p:.1f} k frames/s)  one at a time {s:.3f} ms  "
                  f"one stream {o:.3f} ms", flush=True)
    for bs in (8, 37, 64, 128):
        row = []
        for f, eng in engs.items():
            row.append(timed(lambda i: hold.__setitem__("o", eng.forward(xs[i % 4][:bs], single_stream=True)), steps))
        print(f"bs={bs:3d} one stream: " + ", ".join(f"fold {f}: {r:.3f} ms" for f, r in zip(engs, row)))
    a = engs[0].forward(xs[0]).double()
    b = engs[1].forward(xs[0]).double()
    print("1 - cosine between fold 0 and fold 1 (max over 256 frames):", float((1 - (a * b).sum(1)).max()))
    if 2 in engs:
        print("fold 2 bit-equal to fold 1:", bool(torch.equal(engs[2].forward(xs[0]), engs[1].forward(xs[0]))))


if __name__ == "__main__":
    main()
