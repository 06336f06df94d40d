"""CLIP tokenizer restatement (wise_amd/feature/clip_tokenizer.py) against an independent implementation:
transformers' CLIPTokenizer (the `tokenizers` BPE backend) built over the same vocabulary and merge rules.
The real merge file is not available offline, so the rules are learnt here from a small corpus."""
import collections
import gzip

import pytest
import torch

from wise_amd.feature.clip_tokenizer import ClipTokenizer, byte_symbols, read_merges

CORPUS = """a photo of a dog running on the beach. this is a photo of a cat sleeping on the sofa!
the sound of rain falling on a tin roof; people cheering at a football match, a person's hands chopping onions.
children's laughter and dogs barking in the park - 3 cars and 12 bikes passed by at 10:45. it's what we've seen,
they're cooking, i'm sure she'll know, he'd go. naïve café façade über straße 東京 タワー привет мир"""


def learn_merges(text, n_rules):
    """Plain BPE training over the tokenizer's own pre-tokenisation (pieces as byte symbols + </w>)."""
    tok = ClipTokenizer(())
    sym = byte_symbols()
    words = collections.Counter()
    for piece in tok._split.findall(tok.clean(text)):
        s = [sym[b] for b in piece.encode("utf-8")]
        s[-1] += "</w>"
        words[tuple(s)] += 1
    rules = []
    for _ in range(n_rules):
        pairs = collections.Counter()
        for w, c in words.items():
            for p in zip(w[:-1], w[1:]):
                pairs[p] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        rules.append(best)
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    return rules


@pytest.fixture(scope="module")
def pair():
    rules = learn_merges(CORPUS, 300)
    ours = ClipTokenizer(rules)
    transformers = pytest.importorskip("transformers")
    vocab = dict(ours.encoder)
    vocab["<|startoftext|>"] = vocab.pop("<start_of_text>")
    vocab["<|endoftext|>"] = vocab.pop("<end_of_text>")
    hf = transformers.CLIPTokenizer(vocab=vocab, merges=[(a, b) for a, b in rules])
    return ours, hf


SAMPLES = [
    "a photo of a dog", "This is a photo of a CAT sleeping!", "the sound of rain falling on a tin roof",
    "people's cheering... at 10:45?!", "it's what we've seen; they're cooking, I'm sure she'll know, he'd go",
    "  multiple   spaces\tand\nnewlines  ", "naïve café façade über straße", "東京 タワー", "привет мир",
    "3 cars and 12 bikes", "a&amp;b &lt;tag&gt;", "", "x", "!!!", "unseenwordzzz qqq",
]


@pytest.mark.parametrize("text", SAMPLES)
def test_ids_match_transformers_clip_tokenizer(pair, text):
    ours, hf = pair
    want = hf(text, add_special_tokens=True)["input_ids"]
    got = [ours.sot_token] + ours.encode(text) + [ours.eot_token]
    if "&" in text:  # html.unescape is open_clip's cleaning step, not part of the HF tokenizer
        import html
        want = hf(html.unescape(html.unescape(text)), add_special_tokens=True)["input_ids"]
    assert got == want


def test_batch_layout_truncation_and_padding(pair):
    ours, _ = pair
    long_text = "dog " * 200
    t = ours(["a photo of a dog", long_text, ""])
    assert t.shape == (3, 77) and t.dtype == torch.long
    n = len(ours.encode("a photo of a dog"))
    assert t[0, 0] == ours.sot_token and t[0, n + 1] == ours.eot_token and int(t[0, n + 2:].abs().sum()) == 0
    assert t[1, 0] == ours.sot_token and t[1, -1] == ours.eot_token          # truncated: last slot forced to eot
    assert t[2, 0] == ours.sot_token and t[2, 1] == ours.eot_token
    assert (t.argmax(dim=-1) == torch.tensor([n + 1, 76, 1])).all()          # eot is the largest id: argmax pooling
    assert ours("x", context_length=8).shape == (1, 8)


def test_vocabulary_numbering_follows_open_clip():
    full = ClipTokenizer(())
    assert full.vocab_size == 512 + 2 and full.encoder["!"] == 0 and full.encoder["!</w>"] == 256
    sym = byte_symbols()
    assert sym[ord("a")] == "a" and sym[0] == chr(256) and sym[32] == chr(256 + 32) and len(set(sym)) == 256
    # with all 48894 merges the specials land on 49406 / 49407
    assert 512 + (49152 - 256 - 2) == 49406


def test_merge_file_round_trip(tmp_path, pair):
    ours, _ = pair
    p = tmp_path / "bpe.txt.gz"
    with gzip.open(p, "wt", encoding="utf-8") as f:
        f.write('"bpe_simple_vocab_16e6.txt#version: 0.2\n' + "\n".join(f"{a} {b}" for a, b in ours.merges) + "\n")
    assert read_merges(p) == ours.merges
    again = ClipTokenizer.from_file(p)
    assert again.encode("a photo of a dog running") == ours.encode("a photo of a dog running")


def test_default_needs_the_merge_file(monkeypatch, tmp_path):
    monkeypatch.setenv("WISE_AMD_WEIGHTS_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        ClipTokenizer.default()
    assert ClipTokenizer.default(allow_merge_less=True).vocab_size == 514
