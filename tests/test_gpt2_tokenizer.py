"""GPT-2 tokenizer restatement (wise_amd/feature/gpt2_tokenizer.py) against transformers' GPT2Tokenizer built over
the same vocabulary and merge rules (learnt here from a small corpus: the real merges.txt is not available offline)."""
import collections

import pytest
import torch

from wise_amd.feature.clip_tokenizer import byte_symbols
from wise_amd.feature.gpt2_tokenizer import EOT_TEXT, Gpt2Tokenizer, read_gpt2_merges

CORPUS = """the sound of rain falling on a tin roof; People cheering at a football match, a person's hands chopping onions.
A dog barking in the distance while children's laughter echoes - 3 cars and 12 bikes passed by at 10:45. It's what
we've heard, they're singing, I'm sure she'll know, he'd go.  Thunder and heavy rain, birds chirping at dawn,
naïve café über straße 東京 привет"""


def learn_merges(text, n_rules):
    tok = Gpt2Tokenizer(())
    sym = byte_symbols()
    words = collections.Counter(tuple(sym[b] for b in piece.encode("utf-8")) for piece in tok._split.findall(text))
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
    ours = Gpt2Tokenizer(rules)
    transformers = pytest.importorskip("transformers")
    hf = transformers.GPT2Tokenizer(vocab=dict(ours.encoder), merges=[(a, b) for a, b in rules])
    return ours, hf


SAMPLES = ["the sound of rain", "A Dog barking in the distance!", "people's cheering... at 10:45?!",
           "It's what we've heard; they're singing, I'm sure she'll know, he'd go", "  leading and   multiple spaces ",
           "naïve café über straße", "東京 привет", "3 cars and 12 bikes", "", "x", "unseenwordzzz QQQ",
           "thunder <|endoftext|>"]


@pytest.mark.parametrize("text", SAMPLES)
def test_ids_match_transformers_gpt2_tokenizer(pair, text):
    ours, hf = pair
    assert ours.encode(text) == hf(text)["input_ids"]


def test_msclap_batch_layout(pair):
    ours, _ = pair
    t = ours(["the sound of rain", ""])
    assert t.shape == (2, 77) and t.dtype == torch.long
    n = len(ours.encode("the sound of rain " + EOT_TEXT))
    assert t[0, n - 1] == ours.eot_token and int(t[0, n:].abs().sum()) == 0
    assert ((t != 0).sum(-1) - 1).tolist() == [n - 1, len(ours.encode(" " + EOT_TEXT)) - 1]  # msclap's pooled index
    long = ours(["rain " * 200])
    assert long.shape == (1, 77) and int((long != 0).sum()) == 77
    assert ours.encoder["!"] == 0 and ours.pad_token == 0 and ours.eot_token == ours.vocab_size - 1
    assert 256 + 50000 + 1 == 50257   # with GPT-2's 50000 merges the ids are GPT-2's


def test_merge_file(tmp_path, pair, monkeypatch):
    ours, _ = pair
    p = tmp_path / "gpt2_merges.txt"
    p.write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in ours.merges) + "\n", encoding="utf-8")
    assert read_gpt2_merges(p) == ours.merges
    monkeypatch.setenv("WISE_AMD_WEIGHTS_DIR", str(tmp_path))
    assert Gpt2Tokenizer.default().encode("the sound of rain") == ours.encode("the sound of rain")
    monkeypatch.setenv("WISE_AMD_WEIGHTS_DIR", str(tmp_path / "nowhere"))
    with pytest.raises(FileNotFoundError):
        Gpt2Tokenizer.default()
    assert Gpt2Tokenizer.default(allow_merge_less=True).vocab_size == 257
