"""WordPiece tokenizer of bert-base-uncased, restated for MS-CLAP 2022's `preprocess_text`
(msclap CLAPWrapper.preprocess_text: `tokenizer.encode_plus(text, add_special_tokens=True, max_length=text_len,
padding='max_length')`, reached from src/feature/microsoft_clap.py:42-43,54).

Algorithm (transformers' BertTokenizer, do_lower_case=True): clean (drop NUL / U+FFFD / control characters, every
whitespace -> ' '), put spaces around CJK ideographs, split on whitespace, lower-case, NFD and drop combining marks,
split every punctuation character into its own token; then greedy longest-match-first WordPiece per token ('##' marks a
continuation, a token of more than 100 characters or with an unmatchable rest becomes [UNK]); `[CLS] ids [SEP]`, padded
with [PAD] (id 0) to `context`.  Pinned against transformers' BertTokenizer on a synthetic vocabulary
(tests/test_bert_tokenizer.py).  Needs the model's vocab.txt; msclap passes no `truncation=`, so a longer text is not cut
there (the batch then fails to collate) — here it raises.
"""
from __future__ import annotations

import os
import unicodedata
from pathlib import Path
from typing import Dict, List, Sequence, Union

import torch

VOCAB_FILE_NAME = "bert-base-uncased/vocab.txt"


def _is_whitespace(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def basic_tokens(text: str) -> List[str]:
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or _is_control(ch):
            continue
        if _is_whitespace(ch):
            out.append(" ")
        elif _is_cjk(cp):
            out += [" ", ch, " "]
        else:
            out.append(ch)
    words = []
    for tok in "".join(out).split():
        tok = unicodedata.normalize("NFD", tok.lower())
        tok = "".join(c for c in tok if unicodedata.category(c) != "Mn")
        cur = ""
        for ch in tok:
            if _is_punctuation(ch):
                if cur:
                    words.append(cur)
                    cur = ""
                words.append(ch)
            else:
                cur += ch
        if cur:
            words.append(cur)
    return words


def synthetic_vocab(size: int = 30522) -> List[str]:
    """A stand-in vocabulary for seeded weights (no vocab.txt offline): bert-base-uncased's special ids ([PAD] 0, [UNK] 100,
    [CLS] 101, [SEP] 102, [MASK] 103), then every printable ASCII character as a word start and as a continuation."""
    vocab = [f"[unused{i}]" for i in range(size)]
    vocab[0], vocab[100], vocab[101], vocab[102], vocab[103] = "[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"
    at = 999
    for cp in range(33, 127):
        ch = chr(cp)
        if ch.lower() != ch:
            continue
        vocab[at] = ch
        vocab[at + 200] = "##" + ch
        at += 1
    return vocab


class BertTokenizer:
    def __init__(self, vocab: Sequence[str], context: int = 100):
        self.vocab: Dict[str, int] = {}
        for i, tok in enumerate(vocab):
            self.vocab.setdefault(tok, i)
        self.context = int(context)
        self.pad, self.unk, self.cls, self.sep = (self.vocab[t] for t in ("[PAD]", "[UNK]", "[CLS]", "[SEP]"))
        self.vocab_size = len(vocab)

    @classmethod
    def from_file(cls, path: Union[str, Path], context: int = 100) -> "BertTokenizer":
        with open(path, encoding="utf-8") as f:
            return cls([line.rstrip("\n") for line in f], context)

    @classmethod
    def default(cls, context: int = 100, allow_synthetic: bool = False) -> "BertTokenizer":
        root = os.environ.get("WISE_AMD_WEIGHTS_DIR")
        path = Path(root) / VOCAB_FILE_NAME if root else None
        if path is not None and path.exists():
            return cls.from_file(path, context)
        if allow_synthetic:
            return cls(synthetic_vocab(), context)
        raise FileNotFoundError(f"BERT vocabulary {VOCAB_FILE_NAME} not found (set WISE_AMD_WEIGHTS_DIR to the directory "
                                f"that holds bert-base-uncased's vocab.txt under that name)")

    def wordpiece(self, word: str) -> List[int]:
        if len(word) > 100:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, hit = len(word), None
            while start < end:
                piece = ("##" if start else "") + word[start:end]
                hit = self.vocab.get(piece)
                if hit is not None:
                    break
                end -= 1
            if hit is None:
                return [self.unk]
            ids.append(hit)
            start = end
        return ids

    def encode(self, text: str) -> List[int]:
        ids = [self.cls]
        for w in basic_tokens(text):
            ids += self.wordpiece(w)
        ids.append(self.sep)
        if len(ids) > self.context:
            raise ValueError(f"text of {len(ids)} tokens exceeds the model's text_len {self.context} (msclap does not "
                             f"truncate: such a text cannot be collated into its batch there either)")
        return ids

    def __call__(self, texts: Union[str, List[str]]) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), self.context), self.pad, dtype=torch.int64)
        for r, t in enumerate(texts):
            ids = self.encode(t)
            out[r, : len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out
