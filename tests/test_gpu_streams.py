"""-m gpu: wise_amd._streams.concurrent_streams — the streams the engines keep two batches in flight on are SEEN to run side by
side (two streams on one hardware queue execute in turn: the gain of extract-features.py-style overlap silently disappears)."""
import time

import pytest
import torch

from wise_amd._streams import _runs_beside, concurrent_streams

pytestmark = pytest.mark.gpu


def test_chosen_streams_overlap_and_a_stream_does_not_overlap_itself():
    a, b, c = concurrent_streams(3, "cuda")
    dev = torch.device("cuda")
    for x, y in ((a, b), (b, a), (a, c), (c, b)):
        assert _runs_beside(x, y, dev)
    assert not _runs_beside(a, a, dev)              # the probe does tell "same queue" from "beside"
    extra = concurrent_streams(1, "cuda", beside=[a, b, c])[0]
    assert all(_runs_beside(extra, s, dev) for s in (a, b, c))


def test_engines_take_their_slots_from_it_whatever_was_taken_before():
    from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
    spec = spec_for("ViT-B-32", "openai")
    x = torch.randn(64, 3, 224, 224, device="cuda")
    dev = torch.device("cuda")
    held = []
    for before in (1, 3):
        while len(held) < before:
            held.append(torch.cuda.Stream())
        eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=64)
        ref = eng.forward(x)
        got = eng.forward_pipelined(x).result()
        torch.cuda.synchronize()
        assert torch.equal(ref, got)
        s0, s1 = (sl["stream"] for sl in eng._slots)
        assert _runs_beside(s0, s1, dev) and _runs_beside(s1, s0, dev)
