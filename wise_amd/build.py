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
# No packed f32 VALU math in the product kernels.  Kernels that use v_pk_{fma,mul,add}_f32 have twice been caught
# returning wrong values — only while another stream's MFMA-issuing kernel shared the GPU, never alone: the HTSAT front
# end (see csrc/htsat_frontend.hip) and the candidate re-scoring kernel of the two-stage search (exact dot products off
# by up to 5e-2 in ~1 call of 4 beside a bare MFMA loop; tools/concurrency_sweep.py).  Two switches, both on every
# product file: the SLP vectoriser off (it is what forms the packed operations from adjacent scalar ones), and the
# target feature itself off for the device compilation, which also scalarises explicit float4 arithmetic (the host
# pass prints "not a recognized feature ... ignoring", harmlessly).  Without packed f32 math both kernels are bit-stable
# beside every neighbour tried, and the library is no slower (GEMM +-1 %, ViT step +-0.5 %, HTSAT -2 %).
# Root cause (tools/pk_probe.py): v_pk_*_f32 with a cross-half op_sel read right behind the VALU instruction that produced
# the operand needs a wait state the compiler does not insert; another wave's MFMAs shift the issue cadence enough to
# expose it (low half = src2, product dropped).  debug_probe.hip keeps the feature: its probes emit those instructions
# on purpose.
COMMON_FLAGS = ["-fno-slp-vectorize"]
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FILE_FLAGS: dict = {name: NO_PACKED_F32 for name in HIP_SOURCES if name != "debug_probe.hip"}


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
                   *COMMON_FLAGS, *FILE_FLAGS.get(s.name, ()), *extra_flags]
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
