"""Times wise_preproc_u8 on the GPU: python tools/preproc_bench.py [n]
Reports frames/s and GB/s against the algorithmic bytes (input rectangle the crop depends on + output)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import numpy as np
import torch

from wise_amd.feature.preprocess import ClipPreprocessor, make_plan


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for H, W in [(240, 320), (360, 640), (480, 854), (720, 1280), (1080, 1920)]:
      for mc in (False, 1, 4, "auto"):
        pre = ClipPreprocessor(224, matrix_cores=mc)
        nn = n if H * W <= 720 * 1280 else max(n // 4, 1)
        frames = torch.randint(0, 256, (nn, 3, H, W), dtype=torch.uint8, device="cuda",
                               generator=torch.Generator(device="cuda").manual_seed(H))
        out = torch.empty((nn, 3, 224, 224), dtype=torch.uint8, device="cuda")
        plan = make_plan(H, W, 224)
        # input bytes the crop depends on: the cropped column/row span of the frame
        fx, fy = W / plan.new_w, H / plan.new_h
        need_w = min(W, int(224 * fx + 4 * max(fx, 1)) + 1)
        need_h = min(H, int(224 * fy + 4 * max(fy, 1)) + 1)
        alg = nn * 3 * (need_w * need_h + 224 * 224)
        for _ in range(3):
            pre(frames, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            pre(frames, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        form = pre.chosen[(H, W)] + (" (auto)" if mc == "auto" else "")
        print(f"{H}x{W} n={nn} [{form}] tile={plan.tile} taps={plan.ndh * 4}/{plan.ndv * 4} lds={plan.lds_bytes}: "
              f"{ms * 1e3:8.1f} us  {nn / ms * 1e3:10.0f} frames/s  {alg / ms / 1e6:8.1f} GB/s algorithmic "
              f"({nn * 3 * H * W / ms / 1e6:.1f} GB/s of whole frames)", flush=True)


if __name__ == "__main__":
    main()
