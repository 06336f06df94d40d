"""a few single-query forwards of the XLM-RoBERTa-large text tower (no graph): the command a per-kernel timeline is taken of"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.xlmr_text import XLMR_SPECS, XlmrTextEngine, random_xlmr_state_dict
xs = XLMR_SPECS["xlm-roberta-large-ViT-H-14"]
eng = XlmrTextEngine(xs, random_xlmr_state_dict(xs, 0), max_batch=4)
eng.graph_max_batch = 0
t = torch.full((1, xs.context), xs.pad_id, dtype=torch.int32, device="cuda"); t[:, 0] = 0; t[:, 1:9] = 4321; t[:, 9] = 2
for _ in range(5): eng.forward(t)
torch.cuda.synchronize()
