"""two HTSAT forwards in flight, repeated: log-mel, residual stream and embeddings must equal the serial run bit for bit"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
B, N = 32, 480000
eng = HtsatEngine(random_htsat_state_dict(0), max_batch=B, max_samples=N)
lib = _lib.lib()
w = 0.1 * torch.randn(B, N, device="cuda")
w_copy = w.clone()
Fc = 1024
def taps(ws):
    mel = torch.empty(B * Fc, 64, device="cuda"); x = torch.empty(B * 64, 768, device="cuda")
    _lib.check(lib.wise_htsat_tap(0, ws.data_ptr(), B, N, mel.data_ptr(), mel.numel(), _lib.stream_ptr()), "tap")
    _lib.check(lib.wise_htsat_tap(1, ws.data_ptr(), B, N, x.data_ptr(), x.numel(), _lib.stream_ptr()), "tap")
    return mel, x
FL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.wise_debug_set_htsat(FL)
a = eng.forward(w).clone(); torch.cuda.synchronize()
mel0, x0 = taps(eng._ws); torch.cuda.synchronize()
stats = {"mel": 0, "x": 0, "out": 0}
for rep in range(20):
    hs = [eng.forward_pipelined(w) for _ in range(2)]
    outs = [h.result().clone() for h in hs]; torch.cuda.synchronize()
    for s, o in zip(eng._slots, outs):
        mel, x = taps(s["ws"]); torch.cuda.synchronize()
        if not torch.equal(mel, mel0):
            d = (mel != mel0)
            rows = d.any(dim=1).nonzero().flatten()
            print("mel rows differing:", rows.numel(), "first", rows[:8].tolist(), "cols of first:", d[rows[0]].nonzero().flatten()[:10].tolist(),
                  "maxdiff", float((mel - mel0).abs().max()), "clip/frame of first:", int(rows[0]) // Fc, int(rows[0]) % Fc)
            r0 = int(rows[0])
            print("  got", mel[r0, :6].tolist()); print("  ref", mel0[r0, :6].tolist())
            # does the wrong row equal the reference row of some other frame?
            eq = (mel0 == mel[r0]).all(dim=1).nonzero().flatten()
            print("  equals reference rows:", eq[:5].tolist())
            wx = (x != x0).any(dim=1).nonzero().flatten()
            print("  x rows differing:", wx.numel(), wx[:5].tolist())
        stats["mel"] += 0 if torch.equal(mel, mel0) else 1
        stats["x"] += 0 if torch.equal(x, x0) else 1
        stats["out"] += 0 if torch.equal(o, a) else 1
print(stats, 'input intact:', torch.equal(w, w_copy))
