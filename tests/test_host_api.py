"""CPU tests (-m "not gpu"): the plugin API mirrors the reference's surface and error behaviour, the
feature stores round-trip (the reference's own store tests, src/feature/store/test_feature_store.py),
the .faiss file IO round-trips, and the C-ABI library loads and exports every declared symbol."""
import pickle
import re
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent


# ---------------------------------------------------------------- C ABI
def test_library_exports_exactly_the_declared_symbols():
    """libwise_hip.so exports what include/wise_hip.h declares — nothing less (every declaration resolves and has a
    ctypes prototype) and nothing more (no wise_debug_* switches or C++ helpers leak out of the product library)."""
    import subprocess

    from wise_amd import _lib

    lib = _lib.load()  # raises if the .so is missing or a symbol is absent
    header = (ROOT / "include" / "wise_hip.h").read_text()
    declared = set(re.findall(r"\b(wise_[a-z0-9_]+)\s*\(", header))
    declared -= {"wise_vit_config"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in wise_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    nm = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True)
    exported = {line.split()[-1] for line in nm.stdout.splitlines() if line.strip()}
    assert exported == declared, f"exported but not declared: {sorted(exported - declared)}; missing: {sorted(declared - exported)}"
    assert lib.wise_abi_version() == 5
    # the flags of the correctness fix (no packed f32 VALU math) are the ones this very library was compiled with
    flags = lib.wise_build_flags().decode()
    assert "-fno-slp-vectorize" in flags and "-packed-fp32-ops" in flags, flags


def test_debug_switches_live_only_in_the_debug_library():
    from wise_amd import _lib, build

    assert _lib.DEBUG_LIB_PATH.exists(), "python -m wise_amd.build builds libwise_hip_debug.so beside the product"
    dbg = _lib.load_debug()
    for name in ("wise_debug_neighbour", "wise_debug_set_gemm_variant", "wise_debug_pk_overlap_probe"):
        assert hasattr(dbg, name)
        assert not hasattr(_lib.load(), name)
    assert "debug_probe.hip" not in build.HIP_SOURCES


def test_no_cpu_fallback_without_gpu():
    from wise_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        _lib.lib()
    from wise_amd.index.flat_ip import FlatIPIndex

    with pytest.raises(RuntimeError):  # rows live in HBM: without a device even building the index fails loudly
        idx = FlatIPIndex(8)
        idx.add_with_ids(np.zeros((2, 8), np.float32), np.array([1, 2]))
        idx.search(np.zeros((1, 8), np.float32), 1)


def test_product_path_does_not_import_oracle():
    for py in (ROOT / "wise_amd").rglob("*.py"):
        txt = py.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, f"{py} imports the oracle"


def test_argument_validation_without_gpu():
    """host-side checks run before any launch: bad shapes come back as WISE_E_INVALID with a message."""
    from wise_amd import _lib

    lib = _lib.load()
    assert lib.wise_ip_topk_workspace_bytes(1000, 510, 1, 10) > 0
    assert lib.wise_ip_topk_workspace_bytes(1000, 512, 1, 5000) == 0
    rc = lib.wise_ip_topk_f32(0, 10, 511, 0, 1, 10, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"multiple of 4" in lib.wise_last_error()
    # the two-stage search: shapes it does not serve are refused before anything is launched
    assert lib.wise_ip_topk_shadow_workspace_bytes(1000, 512, 1, 10) > 0
    assert lib.wise_ip_topk_shadow_workspace_bytes(1000, 516, 1, 10) == 0        # d % 8
    assert lib.wise_ip_topk_shadow_workspace_bytes(1000, 512, 1, 1000) > 0       # any k <= 1024 (REST end = 20, --topk 1000)
    assert lib.wise_ip_topk_shadow_workspace_bytes(1000, 512, 1, 1025) == 0      # k > 1024
    rc = lib.wise_ip_topk_shadow_f32(0, 0, 0, 1000, 512, 0, 1, 1025, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"k=1025" in lib.wise_last_error()
    rc = lib.wise_ip_topk_shadow_f32(0, 0, 0, 1000, 512, 0, 1, 10, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"null pointer" in lib.wise_last_error()
    rc = lib.wise_ip_shadow_bf16(0, 10, 12, 0, 0, 0)
    assert rc == -1 and b"ip_shadow_bf16" in lib.wise_last_error()
    cfg = _lib.VitConfig(224, 32, 768, 12, 12, 3072, 512, 0)
    import ctypes as C

    nb, nf = C.c_int64(), C.c_int64()
    assert lib.wise_vit_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0
    from wise_amd.feature.vit import pack_weights, random_state_dict, spec_for

    spec = spec_for("ViT-B-32")
    assert nb.value == 768 * 3072 + 12 * (3 * 768 * 768 + 768 * 768 + 2 * 3072 * 768) + 512 * 768
    bad = _lib.VitConfig(224, 32, 700, 12, 12, 3072, 512, 0)
    assert lib.wise_vit_layout(C.byref(bad), C.byref(nb), C.byref(nf)) == -1
    h14 = spec_for("ViT-H-14", "laion2b_s32b_b79k").c_config()            # head width 80
    assert lib.wise_vit_layout(C.byref(h14), C.byref(nb), C.byref(nf)) == 0
    assert nb.value == 1280 * 640 + 32 * (4 * 1280 * 1280 + 2 * 5120 * 1280) + 1024 * 1280
    bad = _lib.VitConfig(224, 14, 1280, 32, 10, 5120, 1024, 1)           # head width 128: not served
    assert lib.wise_vit_layout(C.byref(bad), C.byref(nb), C.byref(nf)) == -1


def test_weight_packing_layout():
    from wise_amd.feature.vit import VitSpec, pack_weights, random_state_dict

    spec = VitSpec("t", 28, 14, 128, 2, 2, 256, 32)
    sd = random_state_dict(spec, 3)
    wb, pf = pack_weights(spec, sd)
    assert spec.kdim == 588 and spec.kpad == 640
    conv = wb[: 128 * 640].float().reshape(128, 640)
    assert torch.equal(conv[:, :588], sd["visual.conv1.weight"].reshape(128, 588).to(torch.bfloat16).float())
    assert torch.count_nonzero(conv[:, 588:]) == 0
    # proj is stored transposed at the very end
    assert torch.equal(wb[-32 * 128:].float().reshape(32, 128), sd["visual.proj"].t().to(torch.bfloat16).float())
    assert torch.equal(pf[:128], sd["visual.class_embedding"])
    assert torch.equal(pf[-128:], sd["visual.ln_post.bias"])
    # same seed, same weights (the GPU box regenerates them from the seed)
    assert all(torch.equal(a, b) for a, b in zip(random_state_dict(spec, 3).values(), sd.values()))


# ---------------------------------------------------------------- FeatureExtractor surface
def test_feature_extractor_base_and_factory_errors():
    from wise_amd.feature.feature_extractor import FeatureExtractor
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory

    with pytest.raises(NotImplementedError):
        FeatureExtractor()
    with pytest.raises(ValueError, match="must be formatted"):
        FeatureExtractorFactory("mlfoundations/open_clip/ViT-B-32")
    with pytest.raises(ValueError, match="Unknown feature extractor id"):
        FeatureExtractorFactory("a/b/c/d")
    with pytest.raises(ValueError, match="not available"):
        FeatureExtractorFactory("mlfoundations/open_clip/ViT-Z-99/openai")
    with pytest.raises(ValueError, match="not available"):
        FeatureExtractorFactory("microsoft/clap/1999/Not-Applicable")
    # msclap's other two model keys (microsoft_clap.py:20-31): '2022' (Cnn14 + BERT) is its own pair of engines;
    # 'clapcap' cannot produce embeddings in the reference either (microsoft_clap.py:49 reads `.clap`, which msclap's
    # clapcap wrapper does not have) and is refused at construction, never served by another model's kernels
    assert FeatureExtractorFactory("microsoft/clap/2022/seeded-0").version == "2022"
    with pytest.raises(NotImplementedError, match="captioning model"):
        FeatureExtractorFactory("microsoft/clap/clapcap/seeded-0")


def test_openclip_preprocess_matches_reference_transform():
    """preprocess_image: list of PIL or 4-D tensor -> [n,3,224,224] fp32, OpenAI mean/std; else ValueError
    (mlfoundation_openclip.py:81-90).  Shapes asserted by the reference's own test
    (src/feature/test_feature_extractor.py:14-16,33)."""
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory
    from wise_amd.feature.mlfoundation_openclip import CLIP_MEAN, CLIP_STD

    fx = FeatureExtractorFactory("mlfoundations/open_clip/ViT-L-14/seeded-0")
    assert fx.get_input_image_size() == (224, 224) and fx.get_output_dim() == 768
    imgs = [Image.new("RGB", (224, 224)) for _ in range(8)]
    x = fx.preprocess_image(imgs)
    assert x.shape == (8, 3, 224, 224) and x.dtype == torch.float32
    black = torch.tensor([-m / s for m, s in zip(CLIP_MEAN, CLIP_STD)])
    assert torch.allclose(x[0, :, 0, 0].cpu(), black, atol=1e-6)   # x sits on fx.DEVICE, as in the reference
    # tensor input (decoder output: uint8 [n,3,H,W]), shorter side resized to 224 then centre crop
    frames = torch.randint(0, 256, (2, 3, 240, 320), dtype=torch.uint8)
    y = fx.preprocess_image(frames)
    assert y.shape == (2, 3, 224, 224)
    ref = Image.fromarray(frames[0].permute(1, 2, 0).numpy()).resize((298, 224), Image.BICUBIC).crop((37, 0, 261, 224))
    ref = (torch.from_numpy(np.asarray(ref)).permute(2, 0, 1).float() / 255 -
           torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)
    assert torch.allclose(y[0].cpu(), ref, atol=1e-6)
    with pytest.raises(ValueError):
        fx.preprocess_image("not an image")
    with pytest.raises(ValueError):
        fx.extract_image_features([1, 2, 3])
    # picklable for DataLoader workers (extract-features.py:302-308)
    fx2 = pickle.loads(pickle.dumps(fx))
    assert torch.equal(fx2.preprocess_image(imgs[:1]).cpu(), x[:1].cpu())


def test_clap_preprocess_audio_quirks():
    """microsoft_clap.py:33-40: transpose when shape[0] > 2, mono mix, collate -> [1,1,N]."""
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory

    fx = FeatureExtractorFactory("microsoft/clap/2023/seeded-0")
    stereo = torch.randn(2, 192000)
    out = fx.preprocess_audio(stereo)
    assert out.shape == (1, 1, 192000) and torch.allclose(out[0, 0], stereo.mean(0))
    assert fx.preprocess_audio(torch.randn(192000, 2)).shape == (1, 1, 192000)  # [N,C] is transposed
    assert fx.preprocess_audio(torch.randn(1, 1000)).shape == (1, 1, 1000)
    assert fx.get_output_dim() == 1024


# ---------------------------------------------------------------- FeatureStore (reference: test_feature_store.py)
def test_numpy_save_store_roundtrip(tmp_path):
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType

    st = FeatureStoreFactory.create_store(FeatureStoreType.NUMPY, "video", tmp_path)
    st.enable_write(3, 10 ** 6)
    feats = {i: np.random.default_rng(i).standard_normal((1, 4)).astype(np.float32) for i in range(7)}
    for i, f in feats.items():
        st.add(i, f)
    st.close()
    assert sorted(p.name for p in tmp_path.glob("*.npz")) == ["video-000000.npz", "video-000001.npz",
                                                               "video-000002.npz"]
    rd = FeatureStoreFactory.load_store("video", tmp_path)
    rd.enable_read()
    assert rd.feature_count == 7 and rd.feature_dim == 4
    got = {int(i): f for i, f in rd}
    assert all(np.array_equal(got[i], feats[i]) and got[i].shape == (1, 4) for i in feats)
    ids, vecs = zip(*rd.iter_batch(4))
    assert [len(b) for b in ids] == [3, 3, 1] and np.array_equal(np.concatenate(vecs)[0], feats[0][0])
    with pytest.raises(ValueError):
        st2 = FeatureStoreFactory.create_store(FeatureStoreType.NUMPY, "x", tmp_path)
        st2.enable_write(3, 10)
        st2.add(0, np.zeros((2, 4), np.float32))


def test_numpy_save_store_against_shards_written_by_the_reference(tmp_path):
    """f1 pinned to the reference itself: tests/golden/store_ref holds shard files written, and a read-back summary
    produced, by the reference's own NumpySaveStore (oracle/make_golden_store.py; cases: its own test of 7 adds in shards
    of 3, a roll-over case, a case whose last shard is full).  (i) this repo's reader on the reference's shards returns
    what the reference's reader returned; (ii) this repo's writer, fed the same adds, produces array-identical files."""
    import json
    import shutil

    from wise_amd.feature.store.numpy_save_store import NumpySaveStore

    ref_dir = ROOT / "tests" / "golden" / "store_ref"
    summary = json.loads((ref_dir / "summary.json").read_text())
    assert set(summary) == {"t7", "r10", "x8"}
    for name, info in summary.items():
        # (i) reader
        rdir = tmp_path / f"read_{name}"
        rdir.mkdir()
        for f in info["files"]:
            shutil.copy(ref_dir / f, rdir / f)
        rd = NumpySaveStore(name, rdir)
        rd.enable_read()
        assert rd.feature_count == info["feature_count"] and rd.feature_dim == info["feature_dim"]
        got = [(int(i), v) for i, v in rd]
        assert [i for i, _ in got] == info["read_ids"]
        assert [list(v.shape) for _, v in got] == info["read_shapes"]
        by_id = {a["id"]: np.asarray(a["row"], dtype=np.float32) for a in info["adds"]}
        assert all(v.dtype == np.float32 and np.array_equal(v[0], by_id[i]) for i, v in got)
        batches = list(rd.iter_batch(3))
        assert [i for ids, _ in batches for i in ids] == info["read_ids"]
        assert np.array_equal(np.concatenate([v for _, v in batches]), np.stack([by_id[i] for i in info["read_ids"]]))
        # (ii) writer: same adds (same input dtypes) -> same file names, same arrays, same dtypes and shapes
        wdir = tmp_path / f"write_{name}"
        wdir.mkdir()
        wr = NumpySaveStore(name, wdir)
        wr.enable_write(info["shard_maxcount"], -1)
        for a in info["adds"]:
            wr.add(a["id"], np.asarray([a["row"]], dtype=np.dtype(a["dtype"])))
        wr.close()
        assert sorted(p.name for p in wdir.glob("*.npz")) == info["files"]
        for f in info["files"]:
            ours, ref = np.load(wdir / f), np.load(ref_dir / f)
            assert sorted(ours.files) == sorted(ref.files) == ["feature_id", "features"]
            for key in ("feature_id", "features"):
                assert ours[key].dtype == ref[key].dtype and ours[key].shape == ref[key].shape
                assert np.array_equal(ours[key], ref[key])


def test_webdataset_store_format_and_order(tmp_path):
    """tar shards with '%010d.features.pyd' members holding pickled [1,D] arrays; read back in key order."""
    import tarfile

    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType

    st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, "audio", str(tmp_path))
    st.enable_write(2, 20 * 1024 * 1024)
    data = {i: np.full((1, 5), i, np.float32) for i in (0, 3, 6, 7, 8)}
    for i, f in data.items():
        st.add(i, f)
    st.close()
    tars = sorted(tmp_path.glob("audio-*.tar"))
    assert [t.name for t in tars] == ["audio-000000.tar", "audio-000001.tar", "audio-000002.tar"]
    with tarfile.open(tars[0]) as t:
        names = t.getnames()
        assert names == ["0000000000.features.pyd", "0000000003.features.pyd"]
        arr = pickle.loads(t.extractfile(names[1]).read())
        assert arr.shape == (1, 5) and arr.dtype == np.float32 and arr[0, 0] == 3
    rd = FeatureStoreFactory.load_store("audio", tmp_path)
    rd.enable_read(shard_shuffle=False)
    assert rd.feature_count == 5 and rd.feature_dim == 5
    assert [i for i, _ in rd] == [0, 3, 6, 7, 8]
    batches = list(rd.iter_batch(512))
    assert batches[0][0] == [0, 3, 6, 7, 8] and batches[0][1].shape == (5, 5)
    with pytest.raises(ValueError, match="failed to infer"):
        FeatureStoreFactory.load_store("video", tmp_path)
    # the factory's other error paths (feature_store_factory.py:21,32-38): unknown type, mixed shard kinds, foreign extension
    with pytest.raises(ValueError, match="unknown feature_store_type"):
        FeatureStoreFactory.create_store("hdf5", "audio", tmp_path)
    (tmp_path / "audio-000000.npz").write_bytes(b"")
    with pytest.raises(ValueError, match="failed to infer"):
        FeatureStoreFactory.load_store("audio", tmp_path)
    (tmp_path / "image-000000.h5").write_bytes(b"")
    with pytest.raises(ValueError, match="unknown store containing shard filenames with extension .h5"):
        FeatureStoreFactory.load_store("image", tmp_path)


def test_webdataset_store_cases_of_the_reference_test_file(tmp_path):
    """src/feature/store/test_feature_store.py:48-102 restated literally: shard_maxcount 3, shard_maxsize 256 bytes, MULTI-ROW adds
    ([3,4] arrays under ids 0 and 3 — one tar member each, read back whole), then single rows 6, 7, 8; read order [0, 3, 6, 7, 8]."""
    from wise_amd.feature.store.webdataset_store import WebdatasetStore

    featureA, featureB, featureC = np.array([[1, 2, 3, 4]]), np.array([[5, 6, 7, 8]]), np.array([[9, 10, 11, 12]])
    feature0 = np.concatenate((featureA, featureB, featureC), axis=0)
    feature3 = np.concatenate((featureC, featureB, featureA), axis=0)
    # test_webdataset_store_batch_write (:48-72)
    d1 = tmp_path / "batch"
    d1.mkdir()
    w = WebdatasetStore("wise-store", str(d1))
    w.enable_write(3, 256)
    w.add(0, feature0)
    w.add(3, feature3)
    w.close()
    del w
    r = WebdatasetStore("wise-store", str(d1))
    r.enable_read(shard_shuffle=False, shuffle_values=False)
    seen = {}
    for feature_id, feature_vector in r:
        seen[int(feature_id)] = feature_vector
    assert sorted(seen) == [0, 3]
    assert seen[0].shape == (3, 4) and np.all(np.equal(seen[0], feature0)) and np.all(np.equal(seen[3], feature3))
    assert r.feature_count == 2                                  # members, not rows (webdataset_store.py:85-91)
    # test_webdataset_store_read_order (:74-102)
    d2 = tmp_path / "order"
    d2.mkdir()
    w = WebdatasetStore("wise-store", str(d2))
    w.enable_write(3, 256, verbose=1)
    w.add(0, feature0)
    w.add(3, feature3)
    w.add(6, featureA)
    w.add(7, featureB)
    w.add(8, featureC)
    w.close()
    del w
    r = WebdatasetStore("wise-store", str(d2))
    r.enable_read(shard_shuffle=False, shuffle_values=False)
    assert [int(i) for i, _ in r] == [0, 3, 6, 7, 8]
    # a multi-row member cannot go through iter_batch: the reference squeezes axis 0 of every member (:134-137) -> ValueError
    with pytest.raises(ValueError):
        list(r.iter_batch(512))
    # add() before enable_write() (:93-95)
    with pytest.raises(ValueError, match="enable_write"):
        WebdatasetStore("wise-store", str(d2)).add(1, featureA)


# ---------------------------------------------------------------- SearchIndex surface + .faiss IO
def test_search_index_factory_and_file_io(tmp_path):
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType
    from wise_amd.index import faiss_io
    from wise_amd.index.search_index import SearchIndex
    from wise_amd.index.search_index_factory import SearchIndexFactory

    with pytest.raises(NotImplementedError):
        SearchIndex("video", "x", {})
    with pytest.raises(ValueError, match="Unknown media_type"):
        SearchIndexFactory("text", "id", {})
    with pytest.raises(AssertionError, match="features_dir missing"):
        SearchIndexFactory("video", "id", {"index_dir": tmp_path})
    fdir, idir = tmp_path / "features", tmp_path / "index"
    fdir.mkdir()
    st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, "video", str(fdir))
    st.enable_write(2048, 20 * 1024 * 1024)
    X = np.random.default_rng(0).standard_normal((1000, 16)).astype(np.float32)
    for i in range(1000):
        st.add(i + 1, X[i:i + 1])
    st.close()
    si = SearchIndexFactory("video", "mlfoundations/open_clip/ViT-B-32/seeded-0", {"features_dir": fdir,
                                                                                  "index_dir": idir})
    assert si.get_index_filename("IndexFlatIP") == idir / "video-IndexFlatIP.faiss"
    assert not si.is_index_loaded()
    si.create_index("IndexFlatIP")
    fn = si.get_index_filename("IndexFlatIP")
    assert fn.stat().st_size == 4 + 33 + 4 + 33 + 8 + 1000 * 16 * 4 + 8 + 1000 * 8
    mtime = fn.stat().st_mtime_ns
    si.create_index("IndexFlatIP")  # skip-if-exists
    assert fn.stat().st_mtime_ns == mtime
    Xr, ids = faiss_io.read_idmap_flat_ip(fn)
    assert np.array_equal(np.asarray(Xr), X) and np.array_equal(ids, np.arange(1000) + 1)
    with pytest.raises(NotImplementedError):
        si.create_index("IndexHNSWFlat", overwrite=True)       # not an index type WISE offers
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):                      # k-means and list scan are GPU-only: no CPU fallback
            si.create_index("IndexIVFFlat", overwrite=True)
    with pytest.raises(RuntimeError):
        faiss_io.read_idmap_flat_ip(idir / "missing.faiss")
    with pytest.raises(ValueError):
        si.search("video", "cat", query_type="image")


def test_shard_range_partition():
    from wise_amd.index.sharded import shard_range

    for n, w in [(10_000_000, 8), (7, 8), (0, 4), (1001, 3)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def _gfx950_code_objects(shared_object: Path):
    """The gfx950 code objects embedded in a hipcc-built shared object (one clang offload bundle per translation
    unit, concatenated in .hip_fatbin)."""
    import struct
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", str(shared_object)],
                       check=True, capture_output=True)
        blob = fat.read_bytes()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while (i := blob.find(magic, pos)) >= 0:
        (n,) = struct.unpack_from("<Q", blob, i + 24)
        off = i + 32
        for _ in range(n):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tlen].decode()
            off += tlen
            if "gfx950" in triple and size:
                out.append(blob[i + o:i + o + size])
        pos = i + 1
    return out


def test_no_packed_f32_math_in_the_product_kernels(tmp_path):
    """wise_amd/build.py compiles every product file without packed f32 VALU math (kernels using v_pk_{fma,mul,add}_f32
    returned wrong values beside MFMA-issuing kernels of another stream: DESIGN.md).  Disassemble the device code
    inside the libwise_hip.so that wise_amd._lib loads — not the intermediate objects — and hold it to that."""
    import subprocess

    from wise_amd import _lib

    llvm = Path("/opt/rocm/lib/llvm/bin")
    assert (llvm / "llvm-objdump").exists() and (llvm / "llvm-objcopy").exists(), "ROCm LLVM tools are part of the image"
    _lib.load()
    objs = _gfx950_code_objects(_lib.LIB_PATH)
    assert len(objs) >= 8, f"{len(objs)} gfx950 code objects in {_lib.LIB_PATH}"
    kernels = 0
    for n, co in enumerate(objs):
        f = tmp_path / f"dev{n}.co"
        f.write_bytes(co)
        asm = subprocess.run([str(llvm / "llvm-objdump"), "-d", str(f)], check=True, capture_output=True, text=True).stdout
        packed = [l for l in asm.splitlines() if "v_pk_fma_f32" in l or "v_pk_mul_f32" in l or "v_pk_add_f32" in l]
        assert not packed, f"code object {n}: {len(packed)} packed f32 instructions, e.g. {packed[0].strip()}"
        kernels += asm.count("v_mfma_") > 0
    assert kernels >= 4   # the disassembly really is the matrix-core kernels' code
    # and the debug twin's probe object is the one place that has them on purpose
    dbg = _gfx950_code_objects(_lib.DEBUG_LIB_PATH)
    assert len(dbg) == len(objs) + 1
