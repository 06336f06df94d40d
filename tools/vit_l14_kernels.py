"""a few ViT-L/14 forwards at bs=256 (one stream): the command a per-kernel profile of the attention launches is taken of"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
name = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14"
spec = spec_for(name, "openai" if name != "ViT-H-14" else "laion2b_s32b_b79k")
n_iter = 20 if name == "ViT-B-32" else 4
eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=256)
x = torch.randn(256, 3, 224, 224, device="cuda")
for _ in range(n_iter): eng.forward(x, single_stream=True)
torch.cuda.synchronize()
