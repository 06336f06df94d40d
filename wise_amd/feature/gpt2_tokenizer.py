"""GPT-2 byte-level BPE tokenizer for the MS-CLAP caption encoder (host side).

msclap 1.3.3 tokenises captions with `AutoTokenizer.from_pretrained('gpt2')` after adding '!' (id 0) as the pad
token, on `text + ' <|endoftext|>'`, padded to 77 ids (`CLAPWrapper.preprocess_text`).  The published GPT-2
algorithm is restated here:
  vocabulary = 256 byte symbols (the byte-to-unicode table, printable bytes first), one entry per merge rule of
               `merges.txt` in file order, then `<|endoftext|>` -> 50257 ids;
  split      = regex  's|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+  (case-sensitive, no
               cleaning; a leading space belongs to the word that follows it); `<|endoftext|>` is cut out first
               as a special token;
  per piece  = UTF-8 bytes -> byte symbols, then greedily apply the best-ranked merge until none applies.
The merge file is data: `$WISE_AMD_WEIGHTS_DIR/gpt2_merges.txt` (GPT-2's `merges.txt`).  Without it only a merge-less
tokenizer (bytes only) can be built, which the seeded offline model uses.  tests/test_gpt2_tokenizer.py checks the
algorithm against transformers' GPT2Tokenizer over synthetic merges.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple, Union

import regex
import torch

from .clip_tokenizer import byte_symbols, vocabulary_order

MERGES_FILE_NAME = "gpt2_merges.txt"
EOT_TEXT = "<|endoftext|>"


def read_gpt2_merges(path: Union[str, Path]) -> List[Tuple[str, str]]:
    rules = []
    with open(path, "rt", encoding="utf-8") as f:
        for line in f.read().split("\n"):
            if line.startswith("#version") or not line.strip():
                continue
            parts = line.split()
            if len(parts) == 2:
                rules.append((parts[0], parts[1]))
    return rules


class Gpt2Tokenizer:
    def __init__(self, merges: Iterable[Tuple[str, str]] = (), context_length: int = 77):
        self.context_length = int(context_length)
        self._sym = byte_symbols()
        merges = list(merges)
        vocab = vocabulary_order(self._sym) + [a + b for a, b in merges] + [EOT_TEXT]
        self.encoder: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.merges = merges
        self.rank = {pair: i for i, pair in enumerate(merges)}
        self.eot_token = self.encoder[EOT_TEXT]
        self.pad_token = 0   # '!' — msclap's add_special_tokens({'pad_token': '!'})
        self.vocab_size = len(vocab)
        self._cache: Dict[str, List[int]] = {}
        self._split = regex.compile(
            r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+")

    @classmethod
    def default(cls, context_length: int = 77, allow_merge_less: bool = False) -> "Gpt2Tokenizer":
        root = os.environ.get("WISE_AMD_WEIGHTS_DIR")
        path = Path(root) / MERGES_FILE_NAME if root else None
        if path is not None and path.exists():
            return cls(read_gpt2_merges(path), context_length)
        if allow_merge_less:
            return cls((), context_length)
        raise FileNotFoundError(f"GPT-2 merge file {MERGES_FILE_NAME} not found (set WISE_AMD_WEIGHTS_DIR to the "
                                f"directory that holds GPT-2's merges.txt under that name)")

    def _merge_piece(self, symbols: List[str]) -> List[str]:
        while len(symbols) > 1:
            best, best_rank = None, None
            for pair in zip(symbols[:-1], symbols[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            fused, i = [], 0
            while i < len(symbols):
                if i + 1 < len(symbols) and symbols[i] == best[0] and symbols[i + 1] == best[1]:
                    fused.append(best[0] + best[1])
                    i += 2
                else:
                    fused.append(symbols[i])
                    i += 1
            symbols = fused
        return symbols

    def encode(self, text: str) -> List[int]:
        """The special token is cut out of the text first (as the Hugging Face tokenizer does), so a space in front
        of it stays with the text before it and becomes a token of its own."""
        ids: List[int] = []
        segments = text.split(EOT_TEXT)
        for n, segment in enumerate(segments):
            for piece in self._split.findall(segment):
                hit = self._cache.get(piece)
                if hit is None:
                    hit = [self.encoder[s] for s in self._merge_piece([self._sym[b] for b in piece.encode("utf-8")])]
                    self._cache[piece] = hit
                ids.extend(hit)
            if n + 1 < len(segments):
                ids.append(self.eot_token)
        return ids

    def __call__(self, texts: Union[str, List[str]], context_length: Optional[int] = None) -> torch.Tensor:
        """msclap `preprocess_text`: ids of `text + ' <|endoftext|>'`, right-padded with id 0 to the context.
        (Captions longer than the context are cut to it; msclap itself does not truncate and would fail there.)"""
        if isinstance(texts, str):
            texts = [texts]
        T = context_length or self.context_length
        out = torch.zeros(len(texts), T, dtype=torch.long)
        for i, text in enumerate(texts):
            ids = self.encode(text + " " + EOT_TEXT)[:T]
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out
