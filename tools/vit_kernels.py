"""per-kernel times of the single-stream ViT forward (run under rocprofv3 --kernel-trace --stats): python tools/vit_kernels.py [model]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
import ctypes as C
from wise_amd import _lib
model = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-32"
spec = spec_for(model, "openai")
eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=256)
x = torch.randn(256, 3, 224, 224, device="cuda")
ws = eng._ws
out = torch.empty(256, spec.embed_dim, device="cuda")
for _ in range(25):
    _lib.check(eng.lib.wise_vit_forward_single(C.byref(eng.cfg), eng.wb.data_ptr(), eng.pf.data_ptr(), x.data_ptr(), 0, 256,
                                               out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "fwd")
torch.cuda.synchronize()
