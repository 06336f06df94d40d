"""MS-CLAP HTSAT bs=128 x 10 s with the MLPs of stages 2 and 3 as two GEMMs each or as one kernel each (wise_mlp_stream,
wise_htsat_forward2 flags bit 1), same process, interleaved rounds:  python tools/htsat_mlp_ab.py [steps=20]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
modes = {0: dict(mlp_stream=False, attn_stream=False), 1: dict(mlp_stream=True, attn_stream=False), 2: dict(mlp_stream=True, attn_stream=True)}
engs = {f: HtsatEngine(random_htsat_state_dict(0), max_batch=128, max_samples=480000, ln_fold=False, **kw) for f, kw in modes.items()}
wav = 0.1 * torch.randn(128, 480000, generator=torch.Generator(device="cuda").manual_seed(4), device="cuda")
hold = {}


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rnd in range(3):
    for f, eng in engs.items():
        s = timed(lambda i: hold.__setitem__("o", eng.forward(wav)))
        p = timed(lambda i: hold.__setitem__("p", eng.forward_pipelined(wav)))
        print(f"round {rnd} mode {int(f)} (0 plain, 1 one-kernel MLP, 2 + one-kernel attention): one at a time {s:.3f} ms ({128 / s:.1f} k clips/s)  two in flight {p:.3f} ms ({128 / p:.1f} k clips/s)", flush=True)
a = engs[0].forward(wav).double()
for f in (1, 2):
    print(f"1 - cosine between mode 0 and mode {f} (max over 128 clips):", float((1 - (a * engs[f].forward(wav).double()).sum(1)).max()))
