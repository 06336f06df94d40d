"""MicrosoftClap — drop-in surface for the reference's src/feature/microsoft_clap.py:9-58.

`preprocess_audio` reproduces the reference exactly (transpose when shape[0] > 2, mono mix,
default_collate([audio])).  Version '2023': `extract_audio_features` drives the HTSAT HIP kernels
(wise_amd/feature/htsat.py), `preprocess_text` / `extract_text_features` the caption encoder GPT-2 base + msclap
Projection (wise_amd/feature/clap_text.py).  Version '2022': PANNs Cnn14 (wise_amd/feature/cnn14.py) and
bert-base-uncased + msclap Projection (wise_amd/feature/clap_bert.py on the XLM-RoBERTa tower's kernels).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .feature_extractor import FeatureExtractor
from .weights import seeded_tag

CLAP_MODEL_NAMES = ('2022', '2023', 'clapcap')  # msclap 1.3.3 CLAP.model_name keys


class MicrosoftClap(FeatureExtractor):
    ID_PREFIX = 'microsoft/clap/'
    DESCRIPTION = 'MS-CLAP audio / caption encoders (2023: HTSAT + GPT-2, 2022: Cnn14 + BERT) as MI355X HIP kernels; see https://github.com/microsoft/CLAP'

    def __init__(self, id):
        if not id.startswith(self.ID_PREFIX):
            raise ValueError(f'feature id cannot start with {id} and must start with {self.ID_PREFIX}')
        id_tokens = id.split('/')
        assert len(id_tokens) == 4
        if id_tokens[2] not in CLAP_MODEL_NAMES:
            raise ValueError(f'Model version {id_tokens[2]} is not available. Available models are {CLAP_MODEL_NAMES}')
        if id_tokens[2] == 'clapcap':
            # msclap's 'clapcap' wrapper holds its model under `.clapcap`, not `.clap`: the reference's own
            # extract_audio_features (`self.model.clap.audio_encoder`, microsoft_clap.py:49) raises AttributeError for
            # it, so there is no embedding behaviour to reproduce; an embedding from another architecture must never be
            # returned under this id
            raise NotImplementedError("MS-CLAP version 'clapcap' is a captioning model: the reference's extractor cannot "
                                      "produce embeddings with it either (microsoft_clap.py:49 reads `.clap`)")
        self.version = id_tokens[2]
        self.weights_tag = id_tokens[3]
        self.DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
        self.output_dim = 1024
        self._engine = None
        self._text_engine = None
        self._tokenizer = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None
        st["_text_engine"] = None
        return st

    def get_output_dim(self):
        return self.output_dim

    def preprocess_audio(self, audio: torch.Tensor) -> torch.Tensor:
        # CLAP accepts (1xN_samples)                      (microsoft_clap.py:33-40)
        if audio.shape[0] > 2:
            audio = torch.transpose(audio, 0, 1)
        # the CLAP model only accepts single channel audio
        if audio.shape[0] != 1:
            audio = torch.mean(audio, 0, keepdim=True)
        return torch.stack([audio], dim=0)  # msclap default_collate([audio]) -> [1,1,N]

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            seeded = seeded_tag(self.weights_tag) is not None
            if self.version == '2022':      # msclap config_2022: text_model bert-base-uncased, text_len 100
                from .bert_tokenizer import BertTokenizer
                self._tokenizer = BertTokenizer.default(100, allow_synthetic=seeded)
            else:                           # config_2023: gpt2, text_len 77
                from .gpt2_tokenizer import Gpt2Tokenizer
                self._tokenizer = Gpt2Tokenizer.default(77, allow_merge_less=seeded)
        return self._tokenizer

    def preprocess_text(self, text):
        """msclap `preprocess_text`: 2023 — GPT-2 ids of `text + ' <|endoftext|>'`, padded with id 0 to 77; 2022 — BERT
        WordPiece ids `[CLS] … [SEP]` padded with [PAD] to 100 (the reference returns msclap's dict of tensors; here the
        ids alone, which is all the encoders consume: the attention mask is `ids != pad`, token types are zero)."""
        return self.tokenizer(text)

    def _get_text_engine(self):
        if self._text_engine is None:
            seed = seeded_tag(self.weights_tag)
            sd = None
            if seed is None:
                from .weights import load_state_dict_file
                full = load_state_dict_file(f"clap-{self.version}", self.weights_tag)
                sd = {k[len("caption_encoder."):]: v for k, v in full.items() if k.startswith("caption_encoder.")}
            if self.version == '2022':
                from .clap_bert import CLAP_BERT_SPEC, pack_clap_bert_weights, random_clap_bert_state_dict
                from .xlmr_text import XlmrTextEngine
                self._text_engine = XlmrTextEngine(CLAP_BERT_SPEC, sd if sd is not None else
                                                   random_clap_bert_state_dict(CLAP_BERT_SPEC, seed), device="cuda",
                                                   pack=pack_clap_bert_weights)
            else:
                from .clap_text import CAPTION_SPEC, pack_caption_weights, random_caption_state_dict
                from .text import TextEngine
                self._text_engine = TextEngine(CAPTION_SPEC, sd if sd is not None else
                                               random_caption_state_dict(CAPTION_SPEC, seed), device="cuda",
                                               pack=pack_caption_weights)
        return self._text_engine

    def _get_engine(self):
        """the audio encoder msclap builds for this version: HTSAT for '2023', PANNs Cnn14 for '2022'"""
        if self._engine is None:
            seed = seeded_tag(self.weights_tag)
            sd = None
            if seed is None:
                from .weights import load_state_dict_file
                sd = load_state_dict_file(f"clap-{self.version}", self.weights_tag)
                if any(k.startswith("audio_encoder.") for k in sd):
                    sd = {k[len("audio_encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.")}
            if self.version == '2022':
                from .cnn14 import Cnn14Engine, random_cnn14_state_dict
                self._engine = Cnn14Engine(sd if sd is not None else random_cnn14_state_dict(seed), device="cuda")
            else:
                from .htsat import HtsatEngine, random_htsat_state_dict
                self._engine = HtsatEngine(sd if sd is not None else random_htsat_state_dict(seed), device="cuda")
        return self._engine

    def extract_audio_features(self, preprocessed_audio: torch.Tensor) -> np.ndarray:
        x = preprocessed_audio.reshape(preprocessed_audio.shape[0], preprocessed_audio.shape[2])
        out = self._get_engine().forward(x)
        return out.cpu().numpy()

    def extract_audio_features_async(self, preprocessed_audio: torch.Tensor):
        """Handle whose `.result()` is what `extract_audio_features` returns; two batches in flight on the GPU."""
        from .mlfoundation_openclip import _AsyncFeatures
        x = preprocessed_audio.reshape(preprocessed_audio.shape[0], preprocessed_audio.shape[2])
        eng = self._get_engine()
        return _AsyncFeatures(eng.forward_pipelined(x), eng)

    def extract_text_features(self, text: List[str]) -> np.ndarray:
        """caption_encoder + L2 normalise (microsoft_clap.py:53-58) on the HIP text-tower kernels."""
        tokens = self.preprocess_text(text)
        if int(tokens.max()) >= self._get_text_engine().spec.vocab:
            raise RuntimeError("tokenizer vocabulary larger than the model's token embedding")
        return self._get_text_engine().forward(tokens).cpu().numpy()
