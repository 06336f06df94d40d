"""every hot path run while another stream keeps the matrix cores (and LDS) busy: results must equal the solo run
bit for bit.  Victims run on the current stream; neighbours are the fused HTSAT MLP and a bare MFMA loop at several LDS
footprints (the footprint decides which victims they can share a compute unit with)."""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
from wise_amd.feature.text import TextEngine, TEXT_SPECS, random_text_state_dict
from wise_amd.feature.preprocess import ClipPreprocessor
from wise_amd.index.flat_ip import FlatIPIndex

lib = _lib.lib()
lib.wise_debug_neighbour.restype = ctypes.c_int
lib.wise_debug_neighbour.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
other = torch.cuda.Stream()
g = torch.Generator("cuda").manual_seed(1)
P = lambda t: t.data_ptr()
M = 131072
x = torch.randn(M, 96, device="cuda"); lnw = torch.ones(96, device="cuda"); lnb = torch.zeros(96, device="cuda")
W1 = (0.05 * torch.randn(384, 96, device="cuda")).bfloat16(); b1 = torch.zeros(384, device="cuda")
W2 = (0.05 * torch.randn(96, 384, device="cuda")).bfloat16(); b2 = torch.zeros(96, device="cuda")
src = torch.zeros(1024, dtype=torch.int32, device="cuda"); sink = torch.zeros(4, dtype=torch.int32, device="cuda")
neighbours = {"fused MLP": lambda: lib.wise_mlp96_fused(P(x), P(lnw), P(lnb), P(W1), P(b1), P(W2), P(b2), M, 1e-5, other.cuda_stream)}
for lds in (1024, 32768, 61440):
    neighbours[f"MFMA loop, {lds} B LDS"] = (lambda l: (lambda: lib.wise_debug_neighbour(3, 2048, l, 64, P(src), P(sink), other.cuda_stream)))(lds)

victims = {}
spec = spec_for("ViT-B-32")
vit = VitEngine(spec, random_state_dict(spec, 0), max_batch=128)
frames = torch.randint(0, 256, (128, 3, 224, 224), dtype=torch.uint8, device="cuda", generator=g)
victims["ViT-B/32 forward, 128 frames"] = lambda: vit.forward(frames).clone()
hts = HtsatEngine(random_htsat_state_dict(0), max_batch=16, max_samples=480000)
wav = 0.1 * torch.randn(16, 480000, device="cuda", generator=g)
victims["HTSAT forward, 16 clips"] = lambda: hts.forward(wav).clone()
tspec = TEXT_SPECS["ViT-B-32"]
txt = TextEngine(tspec, random_text_state_dict(tspec, 0), max_batch=32)
tok = torch.randint(1, tspec.vocab - 2, (32, tspec.context), device="cuda", generator=g); tok[:, -1] = tspec.vocab - 1
victims["CLIP text tower, 32 queries"] = lambda: txt.forward(tok).clone()
pre = ClipPreprocessor(224)
raw = torch.randint(0, 256, (32, 3, 480, 640), dtype=torch.uint8, device="cuda", generator=g)
victims["image transform, 32 frames"] = lambda: pre(raw).clone()
X = torch.nn.functional.normalize(torch.randn(1_000_000, 512, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(512).adopt(X)
q1 = X[:1] + 0.01; q32 = X[100:132] + 0.01
victims["flat top-10, nq=1"] = lambda: torch.cat([t.double().flatten() for t in idx.search_device(q1, 10)])
victims["flat top-10, nq=32"] = lambda: torch.cat([t.double().flatten() for t in idx.search_device(q32, 10)])

from wise_amd.feature.cnn14 import Cnn14Engine, random_cnn14_state_dict
cnn = Cnn14Engine(random_cnn14_state_dict(0), max_batch=16, max_samples=480000)
victims["Cnn14 forward, 16 clips"] = lambda: cnn.forward(wav).clone()
from wise_amd.feature.clap_bert import CLAP_BERT_SPEC, pack_clap_bert_weights, random_clap_bert_state_dict
from wise_amd.feature.xlmr_text import XlmrTextEngine
bert = XlmrTextEngine(CLAP_BERT_SPEC, random_clap_bert_state_dict(CLAP_BERT_SPEC, 0), max_batch=4, pack=pack_clap_bert_weights)
bert.graph_max_batch = 0          # direct launches on the current stream (the split-K + fused reduce/LayerNorm kernels)
btok = torch.zeros(1, 100, dtype=torch.int32, device="cuda"); btok[:, 0] = 101; btok[:, 1:12] = 2000; btok[:, 12] = 102
victims["CLAP 2022 BERT, 1 query"] = lambda: bert.forward(btok).clone()

bad_total = 0
for vname, vf in victims.items():
    solo = vf(); torch.cuda.synchronize()
    assert torch.equal(vf(), solo), f"{vname}: not deterministic even alone"
    for nname, nf in neighbours.items():
        wrong = 0
        for rep in range(6):
            for _ in range(4): _lib.check(nf(), nname)
            got = vf()
            for _ in range(2): _lib.check(nf(), nname)
            torch.cuda.synchronize()
            wrong += 0 if torch.equal(got, solo) else 1
        bad_total += wrong
        print(f"{vname:34s} beside {nname:24s}: {wrong} of 6 runs differ", flush=True)
print("TOTAL differing runs:", bad_total)
