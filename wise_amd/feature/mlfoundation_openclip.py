"""MlfoundationOpenClip — drop-in for the reference's src/feature/mlfoundation_openclip.py:11-108 with
the image tower running as hand-written HIP kernels (wise_vit_forward) instead of open_clip/ATen.

Same id grammar (`mlfoundations/open_clip/<model>/<pretrained>`), attributes (`ID_PREFIX`,
`DESCRIPTION`, `DEVICE`, `input_image_size`, `output_dim`, `preprocess`) and error behaviour
(ValueError on a bad prefix / unknown model / non-tensor input).  `preprocess_image` stays a pure-CPU
picklable callable (it is shipped to DataLoader workers: extract-features.py:302-308).
"""
from __future__ import annotations

from typing import List, Union

import numpy as np
import torch
from PIL import Image

from .feature_extractor import FeatureExtractor
from .clip_tokenizer import ClipTokenizer
from .text import TextEngine, random_text_state_dict, text_spec_for
from .siglip import (SIGLIP_MEAN, SIGLIP_STD, SIGLIP_TEXT, SIGLIP_VISION, SiglipTokenizer, pack_siglip_text,
                     random_siglip_text_state_dict, random_siglip_vision_state_dict)
from .xlmr_text import XLMR_SPECS, XlmrTextEngine, XlmrTokenizer, random_xlmr_state_dict
from .vit import SPECS, VitEngine, random_state_dict, spec_for
from .weights import load_state_dict_file, seeded_tag

# open_clip OPENAI_DATASET_MEAN / STD (also /root/reference/src/dataloader/__main__.py:30-31)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

# (model, pretrained) pairs this build knows the architecture of; `seeded-<N>` is the offline stand-in
KNOWN_PRETRAINED = {
    "ViT-B-32": ("openai", "laion400m_e31", "laion400m_e32", "laion2b_e16", "laion2b_s34b_b79k"),
    "ViT-B-32-quickgelu": ("openai", "laion400m_e31", "laion400m_e32"),
    "ViT-B-16": ("openai", "laion400m_e31", "laion400m_e32", "laion2b_s34b_b88k"),
    "ViT-L-14": ("openai", "laion400m_e31", "laion400m_e32", "laion2b_s32b_b82k"),
    # head width 80; the multilingual pair is the reference's default feature id (extract-features.py:192)
    "ViT-H-14": ("laion2b_s32b_b79k",),
    "ViT-H-14-quickgelu": ("dfn5b",),
    "xlm-roberta-large-ViT-H-14": ("frozen_laion5b_s13b_b90k",),
    "xlm-roberta-base-ViT-B-32": ("laion5b_s13b_b90k",),
    # SigLIP (timm image tower, attention-pool head; tests/test-kinetics-6.sh:91 extracts with ViT-L-16-SigLIP-384)
    **{name: ("webli",) for name in SIGLIP_VISION},
}


def list_pretrained():
    return [(m, t) for m, tags in KNOWN_PRETRAINED.items() for t in tags]


class ClipImageTransform:
    """open_clip's eval transform for PIL input, restated without torchvision:
    Resize(S, bicubic, shorter side) -> CenterCrop(S) -> RGB -> ToTensor -> Normalize(mean, std);
    resize_mode 'squash' (the SigLIP models' preprocess_cfg): Resize((S, S), bicubic) without regard to aspect, no crop."""

    def __init__(self, size: int, mean=CLIP_MEAN, std=CLIP_STD, resize_mode: str = "shortest"):
        self.size = int(size)
        self.mean, self.std, self.resize_mode = tuple(mean), tuple(std), resize_mode

    def __call__(self, img: Image.Image) -> torch.Tensor:
        S = self.size
        w, h = img.size
        if self.resize_mode == "squash":
            if (w, h) != (S, S):
                img = img.resize((S, S), Image.BICUBIC)
            img = img.convert("RGB")
            x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
            return (x - torch.tensor(self.mean, dtype=torch.float32).view(3, 1, 1)) / \
                torch.tensor(self.std, dtype=torch.float32).view(3, 1, 1)
        if w <= h:
            nw, nh = S, int(S * h / w)
        else:
            nw, nh = int(S * w / h), S
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BICUBIC)
        left = int(round((nw - S) / 2.0))
        top = int(round((nh - S) / 2.0))
        img = img.crop((left, top, left + S, top + S)).convert("RGB")
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
        mean = torch.tensor(self.mean, dtype=torch.float32).view(3, 1, 1)
        std = torch.tensor(self.std, dtype=torch.float32).view(3, 1, 1)
        return (x - mean) / std


def to_pil_image(pic: torch.Tensor) -> Image.Image:
    """torchvision.transforms.functional.to_pil_image for a [C,H,W] tensor (uint8, or float in [0,1])."""
    if pic.dim() != 3:
        raise ValueError(f"pic should be 3 dimensional. Got {pic.dim()} dimensions.")
    if pic.is_floating_point():
        pic = pic.mul(255).byte()
    arr = pic.permute(1, 2, 0).cpu().numpy()
    if arr.shape[2] == 1:
        return Image.fromarray(arr[:, :, 0], mode="L")
    return Image.fromarray(arr, mode="RGB")


class _AsyncFeatures:
    """Embeddings on their way to the host: copy queued behind the forward on a side stream, `.result()` waits for it."""

    def __init__(self, pending, engine=None):
        dev = pending.device
        holder = engine if engine is not None else _AsyncFeatures
        cs = getattr(holder, "_d2h_stream", None)
        if cs is None:
            # a stream SEEN to run beside the engine's own (on the hardware queue of one of them the copy of batch i would
            # sit behind batch i + 1's forward)
            from .._streams import concurrent_streams
            cs = concurrent_streams(1, dev, beside=[sl["stream"] for sl in getattr(engine, "_slots", [])])[0]
            holder._d2h_stream = cs
        with torch.cuda.stream(cs):
            out = pending.result()                     # orders the copy stream after the forward
            self._host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            self._host.copy_(out, non_blocking=True)
            out.record_stream(cs)
            self._done = torch.cuda.Event()
            self._done.record(cs)

    def result(self) -> np.ndarray:
        self._done.synchronize()
        return self._host.numpy()


class MlfoundationOpenClip(FeatureExtractor):
    ID_PREFIX = 'mlfoundations/open_clip/'
    DESCRIPTION = 'OpenCLIP image tower as MI355X HIP kernels; see https://github.com/mlfoundations/open_clip'

    def __init__(self, id):
        if not id.startswith(self.ID_PREFIX):
            raise ValueError(f'feature id cannot start with {id} and must start with {self.ID_PREFIX}')
        id_tokens = id.split('/')
        assert len(id_tokens) == 4
        model, tag = id_tokens[2], id_tokens[3]
        seed = seeded_tag(tag)
        self._siglip = model in SIGLIP_VISION
        if (model.replace("-quickgelu", "") not in SPECS and not self._siglip) or \
                (seed is None and (model, tag) not in list_pretrained()):
            raise ValueError(f'Model ({model}, {tag}) not available in {self.ID_PREFIX}')
        self.pretrained_model_name = model
        self.pretraining_dataset = tag
        self.DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
        if self._siglip:
            # timm image tower with the attention-pool head; squash resize, mean = std = 0.5 (open_clip preprocess_cfg)
            self.spec = SIGLIP_VISION[model]
            self._state_dict = random_siglip_vision_state_dict(self.spec, seed) if seed is not None \
                else load_state_dict_file(model, tag)
            self.preprocess = ClipImageTransform(self.spec.image_size, SIGLIP_MEAN, SIGLIP_STD, "squash")
        else:
            self.spec = spec_for(model, "openai" if seed is not None else tag)
            self._state_dict = random_state_dict(self.spec, seed) if seed is not None else load_state_dict_file(model, tag)
            self.preprocess = ClipImageTransform(self.spec.image_size)
        self._engine = None
        self._gpu_preprocess = None
        # text tower (query side): built on first use; seeded models may run on the merge-less
        # byte tokenizer because open_clip's merge file does not exist offline
        # xlm-roberta-*: the text tower is open_clip's HFTextEncoder over XLM-RoBERTa (xlmr_text.py), tokenised with the
        # model's own sentencepiece vocabulary; every other model has the CLIP text transformer and its BPE tokenizer
        if self._siglip:
            self.text_spec = SIGLIP_TEXT[model]
        else:
            self.text_spec = XLMR_SPECS[model] if model in XLMR_SPECS else text_spec_for(model, "openai" if seed is not None else tag)
        self._seed = seed
        self._text_engine = None
        self._tokenizer = None
        self.input_image_size = (self.spec.image_size, self.spec.image_size)
        self.output_dim = self.spec.embed_dim
        if self.DEVICE == "cuda":
            self._find_output_dim()

    # the engine (device memory, library handle) is created on first use so that the object — and its
    # preprocess_image bound method — stays picklable for DataLoader workers
    def _get_engine(self) -> VitEngine:
        if self._engine is None:
            self._engine = VitEngine(self.spec, self._state_dict, device="cuda")
        return self._engine

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None
        st["_gpu_preprocess"] = None
        st["_text_engine"] = None
        return st

    def _find_output_dim(self):
        """Warm-up forward, as the reference does (mlfoundation_openclip.py:61-73): one random image through the
        tower; the text tower's embedding width is checked against the same dimension (:67-73)."""
        random_image = torch.rand((1, 3,) + self.input_image_size)
        feats = self.extract_image_features(self.preprocess_image(random_image))
        assert feats.shape[1] == self.output_dim
        assert self.text_spec.embed_dim == self.output_dim  # both towers embed into the same space (:67-73)

    def get_output_dim(self):
        return self.output_dim

    def get_input_image_size(self):
        return self.input_image_size

    def preprocess_image(self, images: Union[torch.Tensor, List[Image.Image]]) -> torch.Tensor:
        if isinstance(images, list) and all(isinstance(img, Image.Image) for img in images):
            return torch.stack([self.preprocess(im) for im in images], dim=0).to(device=self.DEVICE)
        elif isinstance(images, torch.Tensor) and len(images.shape) == 4:
            return torch.stack([self.preprocess(to_pil_image(im)) for im in images], dim=0).to(device=self.DEVICE)
        else:
            raise ValueError('all input to preprocess_image() must be an instance of torch.Tensor or PIL.Image')

    def preprocess_image_device(self, images: torch.Tensor) -> torch.Tensor:
        """GPU form of preprocess_image for decoded uint8 frames [n,3,H,W]: the same
        Resize -> CenterCrop as the PIL loop above, bit for bit, as one kernel; returns uint8 [n,3,S,S] on the
        device.  ToTensor + Normalize are applied by the tower when extract_image_features receives uint8."""
        if not isinstance(images, torch.Tensor) or len(images.shape) != 4 or images.dtype != torch.uint8:
            raise ValueError('input to preprocess_image_device() must be a uint8 torch.Tensor [n,3,H,W]')
        if self._gpu_preprocess is None:
            from .preprocess import ClipPreprocessor
            # the SigLIP models squash the frame to S x S (open_clip resize_mode 'squash'); the others resize the shorter
            # side and crop the centre — same kernel, different tap tables
            self._gpu_preprocess = ClipPreprocessor(self.spec.image_size, device="cuda", squash=self._siglip)
        return self._gpu_preprocess(images)

    def extract_image_features(self, images: torch.Tensor) -> np.ndarray:
        if not isinstance(images, torch.Tensor):
            raise ValueError('input to extract_features() must be an instance of torch.Tensor')
        out = self._get_engine().forward(images.to(torch.float32) if images.dtype != torch.uint8 else images)
        return out.cpu().numpy()

    def extract_image_features_async(self, images: torch.Tensor):
        """Enqueue the batch and return a handle whose `.result()` is what `extract_image_features` returns.  Two
        batches are kept in flight on the GPU (VitEngine.forward_pipelined) and the device-to-host copy goes into
        pinned memory behind the forward, so a caller that collects a batch's result after submitting the next one
        never waits on an idle GPU (wise_amd/extract.py does)."""
        if not isinstance(images, torch.Tensor):
            raise ValueError('input to extract_features() must be an instance of torch.Tensor')
        eng = self._get_engine()
        pending = eng.forward_pipelined(images.to(torch.float32) if images.dtype != torch.uint8 else images)
        return _AsyncFeatures(pending, eng)

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            if self.pretrained_model_name in XLMR_SPECS:
                self._tokenizer = XlmrTokenizer.default(self.text_spec.context)     # needs sentencepiece.bpe.model
            elif self._siglip:
                self._tokenizer = SiglipTokenizer.default(self.text_spec.context)   # needs the model's spiece.model
            else:
                self._tokenizer = ClipTokenizer.default(self.text_spec.context, allow_merge_less=self._seed is not None)
            if self._tokenizer.vocab_size > self.text_spec.vocab:
                raise RuntimeError("tokenizer vocabulary larger than the model's token embedding")
        return self._tokenizer

    def _get_text_engine(self):
        if self._text_engine is None:
            if self.pretrained_model_name in XLMR_SPECS:
                sd = random_xlmr_state_dict(self.text_spec, self._seed) if self._seed is not None else self._state_dict
                self._text_engine = XlmrTextEngine(self.text_spec, sd, device="cuda")
            elif self._siglip:
                sd = random_siglip_text_state_dict(self.text_spec, self._seed) if self._seed is not None else self._state_dict
                self._text_engine = TextEngine(self.text_spec, sd, device="cuda", pack=pack_siglip_text)
            else:
                sd = random_text_state_dict(self.text_spec, self._seed) if self._seed is not None else self._state_dict
                self._text_engine = TextEngine(self.text_spec, sd, device="cuda")
        return self._text_engine

    def preprocess_text(self, text: Union[str, List[str]]) -> torch.Tensor:
        """Token ids [n, 77] (what `self.tokenizer(text_query)` is at mlfoundation_openclip.py:106)."""
        return self.tokenizer(text)

    def extract_text_features(self, text_query: List[str]) -> np.ndarray:
        """encode_text + L2 normalise (mlfoundation_openclip.py:103-108) on the HIP text tower."""
        tokens = self.tokenizer(text_query)
        return self._get_text_engine().forward(tokens).cpu().numpy()
