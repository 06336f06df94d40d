"""In-tree build of libwise_hip.so (hipcc, gfx950 only) and of the C oracle.

`python -m wise_amd.build` or `wise_amd.build.build_all()`.  The .so files are git-ignored but
travel to the GPU box with the working tree, so nothing is compiled there unless a source is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libwise_hip.so"
HIP_SOURCES = ["common.hip", "ip_topk.hip", "ip_topk_mfma.hip", "gemm_bf16.hip", "vit.hip", "htsat.hip", "htsat_frontend.hip", "preprocess.hip", "text.hip", "debug_probe.hip"]
ARCH = "gfx950"
# per-file flags; htsat_frontend.hip: see the note at the top of that file
FILE_FLAGS = {"htsat_frontend.hip": ["-fno-slp-vectorize"]}


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libwise_hip.so cannot be built")


def build_hip(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    srcs = [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()]
    deps = srcs + list(CSRC.glob("*.h")) + [ROOT / "include" / "wise_hip.h"]
    if not force and not _newer(LIB, deps):
        return LIB
    LIBDIR.mkdir(parents=True, exist_ok=True)
    objs = []
    procs = []
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    for s in srcs:
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _newer(o, [s] + list(CSRC.glob("*.h")) + [ROOT / "include" / "wise_hip.h"]):
            cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(s), "-o", str(o),
                   *FILE_FLAGS.get(s.name, ()), *extra_flags]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


def build_oracle(force: bool = False) -> Path | None:
    """gcc build of oracle/ C restatement (test infrastructure, never loaded by the product path)."""
    odir = ROOT / "oracle"
    src = odir / "ip_topk_ref.c"
    if not src.exists():
        return None
    out = odir / "_build" / "libwise_oracle.so"
    if not force and not _newer(out, [src]):
        return out
    out.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", str(out), str(src), "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{r.stdout}")
    return out


def build_all(force: bool = False, verbose: bool = False):
    lib = build_hip(force=force, verbose=verbose)
    orc = build_oracle(force=force)
    return lib, orc


if __name__ == "__main__":
    lib, orc = build_all(force="--force" in sys.argv, verbose=True)
    print("built", lib, orc)
