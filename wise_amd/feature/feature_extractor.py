"""Abstract plugin surface — same methods and behaviour as the reference's
src/feature/feature_extractor.py:6-59 (every method, __init__ included, raises NotImplementedError)."""
from typing import List, Union

import numpy as np
import torch


class FeatureExtractor:
    def __init__(self):
        raise NotImplementedError

    def preprocess_image(self, images: Union[torch.Tensor, list]) -> torch.Tensor:
        raise NotImplementedError

    def extract_image_features(self, images: torch.Tensor) -> np.ndarray:
        raise NotImplementedError

    def preprocess_text(self, text: str) -> str:
        raise NotImplementedError

    def extract_text_features(self, text_query: List[str]) -> np.ndarray:
        raise NotImplementedError

    def preprocess_audio(self, audio: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def extract_audio_features(self, preprocessed_audio: torch.Tensor) -> np.ndarray:
        raise NotImplementedError
