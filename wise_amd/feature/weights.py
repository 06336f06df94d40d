"""Where model weights come from.

The reference downloads checkpoints by name (open_clip / msclap).  There is no network here, so:
  * pretrained tag `seeded-<N>`  -> this repo's documented seeded initialiser (offline stand-in);
  * any other tag                -> a state dict file `<WISE_AMD_WEIGHTS_DIR>/<model>__<tag>.{safetensors,pt}`
                                    holding open_clip (or msclap) state-dict keys; missing file = error.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict

import torch


def seeded_tag(tag: str):
    if tag.startswith("seeded-"):
        try:
            return int(tag.split("-", 1)[1])
        except ValueError:
            return None
    return None


def load_state_dict_file(model: str, tag: str) -> Dict[str, torch.Tensor]:
    root = os.environ.get("WISE_AMD_WEIGHTS_DIR")
    if not root:
        raise FileNotFoundError(
            f"weights for ({model}, {tag}) requested but WISE_AMD_WEIGHTS_DIR is not set; no network is available "
            f"to fetch them.  Use the pretrained tag 'seeded-0' for seeded random weights.")
    stem = Path(root) / f"{model}__{tag}"
    st = stem.with_suffix(".safetensors")
    if st.exists():
        from safetensors.torch import load_file

        return load_file(str(st))
    pt = stem.with_suffix(".pt")
    if pt.exists():
        sd = torch.load(str(pt), map_location="cpu", weights_only=True)
        return sd.get("state_dict", sd)
    raise FileNotFoundError(f"no weight file {st} or {pt}")
