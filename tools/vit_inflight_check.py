import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
spec = spec_for("ViT-B-32", "openai")
eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=256)
x = torch.randn(256, 3, 224, 224, device="cuda")
a = eng.forward(x).clone(); torch.cuda.synchronize()
bad = 0
for rep in range(10):
    hs = [eng.forward_pipelined(x) for _ in range(6)]
    outs = [h.result().clone() for h in hs]; torch.cuda.synchronize()
    bad += sum(0 if torch.equal(a, o) else 1 for o in outs)
print("ViT pipelined mismatching outputs of 60:", bad)
