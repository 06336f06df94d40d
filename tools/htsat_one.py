"""n forwards of MS-CLAP HTSAT at bs=128 x 10 s, one at a time (for rocprofv3): python tools/htsat_one.py [n] [fold 0/1]"""
import sys

import torch

sys.path.insert(0, ".")
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fold = bool(int(sys.argv[2])) if len(sys.argv) > 2 else None
eng = HtsatEngine(random_htsat_state_dict(0), max_batch=128, max_samples=480000, ln_fold=fold)
wav = 0.1 * torch.randn(128, 480000, generator=torch.Generator(device="cuda").manual_seed(4), device="cuda")
for i in range(n):
    out = eng.forward(wav)
torch.cuda.synchronize()
print("fold", eng.ln_fold, float(out.norm(dim=1).mean()))
