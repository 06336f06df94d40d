"""Reference-generated fixtures for the npz feature store (test infrastructure; run in the BUILD container only).

The reference's own `NumpySaveStore` (src/feature/store/numpy_save_store.py — importable here: it needs numpy alone,
SURVEY.md section 8c) writes two stores and reads them back; the shard files it produced and what its reader returned
are committed under tests/golden/store_ref/ as DATA.  tests/test_host_api.py then checks (i) this repo's reader on the
reference's shards and (ii) that this repo's writer produces array-identical shards from the same adds.

  case "t7": the reference's own test, src/feature/store/test_feature_store.py:15-46 — seven [1,4] integer rows,
             shard_maxcount 3 -> shards of 3, 3, 1 rows
  case "r10": ten float32 [1,8] rows, ids 100.., shard_maxcount 4 -> shards of 4, 4, 2 rows (roll-over inside add();
             the last shard trimmed by close())
  case "x8": eight rows, shard_maxcount 4 -> two FULL shards (close() with a full buffer)

    python oracle/make_golden_store.py        (needs /root/reference; never runs on the GPU box)
"""
import importlib.util
import json
import shutil
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference/src/feature/store")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "store_ref"


def load_reference_store():
    """Import the reference's two store modules under a throw-away package name (they use a relative import)."""
    pkg = types.ModuleType("refstore")
    pkg.__path__ = [str(REF)]
    sys.modules["refstore"] = pkg
    for name in ("feature_store", "numpy_save_store"):
        spec = importlib.util.spec_from_file_location(f"refstore.{name}", REF / f"{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refstore.{name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["refstore.numpy_save_store"].NumpySaveStore


def cases():
    a, b, c = np.array([[1, 2, 3, 4]]), np.array([[5, 6, 7, 8]]), np.array([[9, 10, 11, 12]])
    rows = np.random.default_rng(5).standard_normal((10, 1, 8)).astype(np.float32)
    return {
        "t7": (3, [(0, a), (1, b), (2, c), (3, c), (4, b), (5, a), (6, b)]),
        "r10": (4, [(100 + i, rows[i]) for i in range(10)]),
        "x8": (4, [(7 * i + 1, rows[i]) for i in range(8)]),
    }


def main():
    Store = load_reference_store()
    if OUT.exists():
        shutil.rmtree(OUT)
    OUT.mkdir(parents=True)
    summary = {}
    for name, (maxcount, adds) in cases().items():
        with tempfile.TemporaryDirectory() as tmp:
            w = Store(name, tmp)
            w.enable_write(maxcount, -1, verbose=0)
            for fid, vec in adds:
                w.add(fid, vec)
            w.close()
            del w
            r = Store(name, tmp)
            r.enable_read()
            read = [(int(fid), np.asarray(vec)) for fid, vec in r]
            files = sorted(Path(tmp).glob("*.npz"))
            for f in files:
                shutil.copy(f, OUT / f.name)
            summary[name] = {
                "shard_maxcount": maxcount,
                "files": [f.name for f in files],
                "feature_count": int(r.feature_count), "feature_dim": int(r.feature_dim),
                "read_ids": [fid for fid, _ in read],
                "read_shapes": [list(v.shape) for _, v in read],
                "adds": [{"id": int(fid), "row": np.asarray(vec, dtype=np.float64).ravel().tolist(),
                          "dtype": str(np.asarray(vec).dtype)} for fid, vec in adds],
                "shards": [{"file": f.name, "feature_id_dtype": str(np.load(f)["feature_id"].dtype),
                            "feature_id_shape": list(np.load(f)["feature_id"].shape),
                            "features_dtype": str(np.load(f)["features"].dtype),
                            "features_shape": list(np.load(f)["features"].shape)} for f in files],
            }
    (OUT / "summary.json").write_text(json.dumps(summary, indent=1))
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
