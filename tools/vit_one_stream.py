"""N single-stream forwards of a CLIP image tower at bs=256 (for rocprofv3 --kernel-trace --stats): python tools/vit_one_stream.py [model] [n] [fold 0/1]"""
import sys

import torch

sys.path.insert(0, ".")
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fold = None if len(sys.argv) <= 3 else bool(int(sys.argv[3]))
spec = spec_for(model, "openai")
eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=256, ln_fold=fold)
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(256, 3, spec.image_size, spec.image_size, generator=g, device="cuda") for _ in range(4)]
for i in range(n):
    out = eng.forward(xs[i % 4], single_stream=True)
torch.cuda.synchronize()
print("ln_fold", eng.spec.ln_fold, float(out.norm(dim=1).mean()))
