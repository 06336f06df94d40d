"""single-query latency of the text towers under the three skinny-GEMM policies (debug library):
0 = product rule, 1 = split-K only for the residual GEMMs (the others as one ring-kernel launch), 2 = no split-K."""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib  # noqa: E402
from wise_amd.feature.clap_bert import CLAP_BERT_SPEC, pack_clap_bert_weights, random_clap_bert_state_dict  # noqa: E402
from wise_amd.feature.text import TextEngine, random_text_state_dict, text_spec_for  # noqa: E402
from wise_amd.feature.xlmr_text import XLMR_SPECS, XlmrTextEngine, random_xlmr_state_dict  # noqa: E402

lib = _lib.lib()
tspec = text_spec_for("ViT-B-32", "openai")
teng = TextEngine(tspec, random_text_state_dict(tspec, 0), max_batch=8)
toks = torch.zeros(1, tspec.context, dtype=torch.int32, device="cuda"); toks[:, 0] = 49406; toks[:, 1:6] = 1234; toks[:, 6] = 49407
xs = XLMR_SPECS["xlm-roberta-large-ViT-H-14"]
xeng = XlmrTextEngine(xs, random_xlmr_state_dict(xs, 0), max_batch=4)
xt = torch.full((1, xs.context), xs.pad_id, dtype=torch.int32, device="cuda"); xt[:, 0] = 0; xt[:, 1:9] = 4321; xt[:, 9] = 2
beng = XlmrTextEngine(CLAP_BERT_SPEC, random_clap_bert_state_dict(CLAP_BERT_SPEC, 0), max_batch=4, pack=pack_clap_bert_weights)
bt = torch.zeros(1, 100, dtype=torch.int32, device="cuda"); bt[:, 0] = 101; bt[:, 1:9] = 2345; bt[:, 9] = 102
outs = {}
for rnd in range(2):
    for pol in (0, 1, 2):
        lib.wise_debug_set_gemm_flags(pol << 10)
        for eng in (teng, xeng, beng):
            eng._graphs = {}          # captured graphs hold the previous policy's kernels
        for name, fn in (("CLIP B/32 text", lambda: teng.forward(toks)), ("XLM-R large", lambda: xeng.forward(xt)),
                         ("CLAP 2022 BERT", lambda: beng.forward(bt))):
            for _ in range(5):
                o = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                o = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50
            ref = outs.setdefault(name, o.clone())
            print(f"round {rnd} policy {pol} {name:16s} {dt * 1e3:.4f} ms   cos vs policy 0: {float((o * ref).sum()):.6f}", flush=True)
lib.wise_debug_set_gemm_flags(0)
