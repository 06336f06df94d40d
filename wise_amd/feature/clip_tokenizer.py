"""CLIP byte-pair tokenizer for the text tower (host side).

The reference tokenises with `open_clip.get_tokenizer(model)` (src/feature/mlfoundation_openclip.py:42,106): the
OpenAI CLIP `SimpleTokenizer` of open_clip_torch==2.24.0, an un-vendored dependency.  Its published algorithm is
restated here:
  vocabulary  = 256 byte symbols, the same 256 with the end-of-word mark `</w>`, one entry per merge rule of
                `bpe_simple_vocab_16e6.txt.gz` (lines 1 .. 48894 of the file), `<start_of_text>`, `<end_of_text>`
                -> 49408 ids, the last two being the special tokens;
  clean       = html.unescape twice, strip, collapse whitespace, lower-case (upstream also runs ftfy.fix_text first;
                ftfy is not installed here — it only matters for mojibake input);
  split       = regex  <specials>|'s|'t|'re|'ve|'m|'ll|'d|[\\p{L}]+|[\\p{N}]|[^\\s\\p{L}\\p{N}]+  (case-insensitive);
  per piece   = UTF-8 bytes -> byte symbols, last one marked `</w>`, then greedily apply the best-ranked merge
                until none applies;
  batch       = [sot] + ids + [eot], truncated to the context with the last slot forced to eot, zero padded.
The merge file is data, not code: it is looked up as `$WISE_AMD_WEIGHTS_DIR/bpe_simple_vocab_16e6.txt.gz` (the file
open_clip ships).  Without it only a merge-less tokenizer (bytes only) can be built, which is what the seeded
offline models use.  tests/test_clip_tokenizer.py checks the algorithm against transformers' CLIPTokenizer
(the `tokenizers` backend) on a vocabulary with synthetic merges.
"""
from __future__ import annotations

import gzip
import html
import os
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex
import torch

BPE_FILE_NAME = "bpe_simple_vocab_16e6.txt.gz"
FULL_MERGES = 49152 - 256 - 2  # merge rules open_clip reads from the file


def byte_symbols() -> List[str]:
    """One printable unicode character per byte value, in byte order (GPT-2 / CLIP byte-to-unicode table):
    bytes that are already printable map to themselves, the others to 256, 257, ... in increasing byte order."""
    printable = set(range(ord("!"), ord("~") + 1)) | set(range(ord("¡"), ord("¬") + 1)) | set(range(ord("®"), ord("ÿ") + 1))
    table, extra = [], 0
    for b in range(256):
        if b in printable:
            table.append(chr(b))
        else:
            table.append(chr(256 + extra))
            extra += 1
    return table


def vocabulary_order(symbols: Sequence[str]) -> List[str]:
    """The order in which open_clip numbers the 512 byte entries: printable bytes first (in the three ranges),
    then the remapped ones — i.e. sorted by the table's construction order, not by byte value."""
    printable = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    seen = set(printable)
    order = printable + [b for b in range(256) if b not in seen]
    return [symbols[b] for b in order]


def read_merges(path: Union[str, Path]) -> List[Tuple[str, str]]:
    """Merge rules of a CLIP BPE file: first line is a header, one `left right` rule per following line."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    rules = []
    for line in lines[1:FULL_MERGES + 1]:
        parts = line.split()
        if len(parts) == 2:
            rules.append((parts[0], parts[1]))
    return rules


class ClipTokenizer:
    def __init__(self, merges: Iterable[Tuple[str, str]] = (), context_length: int = 77):
        self.context_length = int(context_length)
        self._sym = byte_symbols()
        base = vocabulary_order(self._sym)
        merges = list(merges)
        vocab = base + [s + "</w>" for s in base] + [a + b for a, b in merges]
        self.sot_text, self.eot_text = "<start_of_text>", "<end_of_text>"
        vocab += [self.sot_text, self.eot_text]
        self.encoder: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.merges = merges
        self.rank: Dict[Tuple[str, str], int] = {pair: i for i, pair in enumerate(merges)}
        self.sot_token = self.encoder[self.sot_text]
        self.eot_token = self.encoder[self.eot_text]
        self.vocab_size = len(vocab)
        self._cache: Dict[str, List[int]] = {}
        self._split = regex.compile(
            r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def from_file(cls, path: Union[str, Path], context_length: int = 77) -> "ClipTokenizer":
        return cls(read_merges(path), context_length)

    @classmethod
    def default(cls, context_length: int = 77, allow_merge_less: bool = False) -> "ClipTokenizer":
        """The tokenizer open_clip would build: needs the merge file next to the weights."""
        root = os.environ.get("WISE_AMD_WEIGHTS_DIR")
        path = Path(root) / BPE_FILE_NAME if root else None
        if path is not None and path.exists():
            return cls.from_file(path, context_length)
        if allow_merge_less:
            return cls((), context_length)
        raise FileNotFoundError(
            f"CLIP merge file {BPE_FILE_NAME} not found (set WISE_AMD_WEIGHTS_DIR to the directory that holds it; "
            f"it ships with open_clip)")

    # ---- the algorithm -----------------------------------------------------------------------
    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text)).strip()
        return regex.sub(r"\s+", " ", text).strip().lower()

    def _merge_piece(self, symbols: List[str]) -> List[str]:
        """Greedy BPE: repeatedly fuse every occurrence of the adjacent pair with the best (lowest) rank."""
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

    def _piece_ids(self, piece: str) -> List[int]:
        hit = self._cache.get(piece)
        if hit is not None:
            return hit
        if piece == self.sot_text or piece == self.eot_text:
            ids = [self.encoder[piece]]
        else:
            symbols = [self._sym[b] for b in piece.encode("utf-8")]
            symbols[-1] += "</w>"
            ids = [self.encoder[s] for s in self._merge_piece(symbols)]
        self._cache[piece] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in self._split.findall(self.clean(text)):
            ids.extend(self._piece_ids(piece))
        return ids

    def __call__(self, texts: Union[str, List[str]], context_length: Optional[int] = None) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        T = context_length or self.context_length
        out = torch.zeros(len(texts), T, dtype=torch.long)
        for i, text in enumerate(texts):
            ids = [self.sot_token] + self.encode(text) + [self.eot_token]
            if len(ids) > T:
                ids = ids[:T]
                ids[-1] = self.eot_token
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out
