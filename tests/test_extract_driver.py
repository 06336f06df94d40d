"""The batched HP-1 driver reproduces the reference loop's outputs (ids, timestamps, store contents and order).
CPU part: fake extractors (any deterministic per-row function).  GPU part: the real HIP extractors."""
from dataclasses import dataclass

import numpy as np
import pytest
import torch

from oracle.extract_loop_ref import reference_loop
from wise_amd.extract import BatchedExtractionDriver
from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType


@dataclass
class Chunk:
    tensor: torch.Tensor
    pts: float


class FakeExtractor:
    """rows are embedded independently: mean/std features, L2-normalised"""
    def extract_image_features(self, x):
        f = torch.stack([x.float().mean(dim=(1, 2, 3)), x.float().std(dim=(1, 2, 3)), x.float()[:, 0, 0, 0],
                         torch.ones(x.shape[0])], dim=1)
        return (f / f.norm(dim=1, keepdim=True)).numpy()

    def extract_audio_features(self, x):
        x = x.reshape(x.shape[0], x.shape[2])
        f = torch.stack([x.mean(dim=1), x.std(dim=1), x[:, 0], torch.ones(x.shape[0])], dim=1)
        return (f / f.norm(dim=1, keepdim=True)).numpy()


def synthetic_loader(n_files=5, seed=0, size=8):
    """what dataset.py yields: per 4-s chunk up to 8 video frames + one audio segment (last ones ragged)."""
    g = torch.Generator().manual_seed(seed)
    for mid in range(1, n_files + 1):
        n_chunks = 2 + mid % 3
        for c in range(n_chunks):
            last = c == n_chunks - 1
            nf = 8 if not last else 3 + mid % 5
            chunks = {"video": Chunk(torch.randn(nf, 3, size, size, generator=g), c * 4.0)}
            ns = 192000 if not last else (192000 if mid % 2 else 100000)  # a short last segment is dropped
            chunks["audio"] = Chunk(torch.randn(1, 1, ns, generator=g), c * 4.0) if mid != 3 else None
            yield mid, chunks


class Recorder:
    def __init__(self):
        self.rows = []

    def __call__(self, modality, mid, ts, end_ts):
        self.rows.append((modality, mid, ts, end_ts))
        return len(self.rows)  # autoincrement from 1, shared by modalities


def run(kind, tmp, extractors, batched, video_batch=20, audio_batch=3):
    stores = {}
    for m in extractors:
        d = tmp / f"{kind}-{m}"
        d.mkdir()
        st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, m, str(d))
        st.enable_write(7, 20 * 1024 * 1024)
        stores[m] = st
    rec = Recorder()
    if batched:
        drv = BatchedExtractionDriver(extractors, stores, rec, video_batch=video_batch, audio_batch=audio_batch)
        for mid, chunks in synthetic_loader():
            drv.feed(mid, chunks)
        drv.close()
    else:
        reference_loop(synthetic_loader(), extractors, stores, rec)
    out = {}
    for m in extractors:
        rd = FeatureStoreFactory.load_store(m, tmp / f"{kind}-{m}")
        rd.enable_read()
        out[m] = [(i, v.copy()) for i, v in rd]
        out[m + "_shards"] = sorted(p.name for p in (tmp / f"{kind}-{m}").glob("*.tar"))
    return rec.rows, out


def compare(tmp_path, extractors, exact=True):
    rows_ref, ref = run("ref", tmp_path, extractors, batched=False)
    rows_new, new = run("new", tmp_path, extractors, batched=True)
    assert rows_new == rows_ref  # same vectors rows: modality, media id, timestamps, in the same id order
    for m in extractors:
        assert new[m + "_shards"] == ref[m + "_shards"]
        assert [i for i, _ in new[m]] == [i for i, _ in ref[m]]
        for (_, a), (_, b) in zip(new[m], ref[m]):
            assert a.shape == b.shape == (1, a.shape[1])
            assert np.array_equal(a, b) if exact else np.allclose(a, b, atol=1e-6)
    return rows_ref, ref


def test_batched_driver_reproduces_reference_loop_cpu(tmp_path):
    fx = FakeExtractor()
    rows, ref = compare(tmp_path, {"video": fx, "audio": fx}, exact=False)
    assert rows[0] == ("video", 1, 0.0, None) and rows[1] == ("video", 1, 0.5, None)
    assert rows[8] == ("audio", 1, 0.0, 4.0)                     # 8 frame rows, then the segment's row
    assert not any(m == "audio" and mid == 3 for m, mid, _, _ in rows)
    n_audio = sum(1 for r in rows if r[0] == "audio")
    assert len(ref["audio"]) == n_audio and len(ref["video"]) == len(rows) - n_audio


@pytest.mark.gpu
def test_batched_driver_reproduces_reference_loop_gpu(tmp_path):
    """Real extractors: batching must not change a single bit of any stored vector."""
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory

    class V:  # a small ViT so the per-chunk reference loop stays fast
        def __init__(self):
            from wise_amd.feature.vit import VitEngine, VitSpec, random_state_dict
            spec = VitSpec("t", 64, 32, 128, 2, 2, 512, 64)
            self.e = VitEngine(spec, random_state_dict(spec, 1), max_batch=64)

        def extract_image_features(self, x):
            return self.e.forward(x).cpu().numpy()

        def extract_image_features_async(self, x):   # the driver's overlapped path (two batches in flight)
            from wise_amd.feature.mlfoundation_openclip import _AsyncFeatures
            return _AsyncFeatures(self.e.forward_pipelined(x))

    clap = FeatureExtractorFactory("microsoft/clap/2023/seeded-0")
    global synthetic_loader
    orig = synthetic_loader
    synthetic_loader = lambda: orig(n_files=3, seed=1, size=64)  # noqa: E731
    try:
        compare(tmp_path, {"video": V(), "audio": clap}, exact=True)
    finally:
        synthetic_loader = orig
