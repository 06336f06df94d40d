"""N whole HTSAT forwards (128 clips x 10 s), nothing else on the device: the command the HTSAT PMC passes profile"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = HtsatEngine(random_htsat_state_dict(0), max_batch=128, max_samples=480000)
w = 0.1 * torch.randn(128, 480000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
torch.cuda.synchronize()
for _ in range(n):
    eng.forward(w)
torch.cuda.synchronize()
