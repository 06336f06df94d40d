"""In-tree build of libwise_hip.so (hipcc, gfx950 only) and of its debug twin.

`python -m wise_amd.build` or `wise_amd.build.build_all()`.  The .so files are git-ignored but
travel to the GPU box with the working tree, so nothing is compiled there unless a source is newer.

Two libraries come out of the same sources:
  lib/libwise_hip.so        the product: exports exactly the symbols include/wise_hip.h declares (a linker version
                            script generated from the header), tuning / ablation switches compiled out.
  lib/libwise_hip_debug.so  the same code built with -DWISE_DEBUG_KNOBS plus csrc/debug_probe.hip: the
                            wise_debug_* entry points tools/ and the neighbour tests use.  Never loaded by wise_amd.
An object is rebuilt when its source, a header, THIS FILE or its command line changed (the command line is kept
beside the object): the flags below are a correctness fix, so a stale object built with other flags must not survive.
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libwise_hip.so"
LIB_DEBUG = LIBDIR / "libwise_hip_debug.so"
HEADER = ROOT / "include" / "wise_hip.h"
HIP_SOURCES = ["common.hip", "ip_topk.hip", "ip_topk_mfma.hip", "gemm_bf16.hip", "vit.hip", "htsat.hip",
               "htsat_frontend.hip", "preprocess.hip", "text.hip", "xlmr_text.hip", "cnn14.hip", "mlp_stream.hip", "swin_stream.hip", "ivf_build.hip"]
DEBUG_ONLY_SOURCES = ["debug_probe.hip"]
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
FILE_FLAGS: dict = {name: NO_PACKED_F32 for name in HIP_SOURCES}


def flags_string() -> str:
    """What wise_build_flags() of the product library returns (include/wise_hip.h)."""
    return " ".join(["-O3", f"--offload-arch={ARCH}", *COMMON_FLAGS, *NO_PACKED_F32])


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


def declared_symbols() -> list:
    """Every function include/wise_hip.h declares (the product library's whole export list)."""
    names = set(re.findall(r"\b(wise_[a-z0-9_]+)\s*\(", HEADER.read_text()))
    return sorted(names)


def _compile(srcs, objdir: Path, defines, force: bool, verbose: bool, extra_flags=()):
    """Compile each source to objdir; returns (objects, whether anything was rebuilt)."""
    objdir.mkdir(parents=True, exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [HEADER]
    objs, procs = [], []
    for s in srcs:
        o = objdir / (s.stem + ".o")
        objs.append(o)
        cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
               *defines, "-c", str(s), "-o", str(o), *COMMON_FLAGS, *FILE_FLAGS.get(s.name, ()), *extra_flags]
        stamp = o.with_suffix(".cmd")
        cmdline = " ".join(cmd) + "\n# build.py " + hashlib.sha256(Path(__file__).read_bytes()).hexdigest()[:16] + "\n"
        stale = force or _newer(o, [s, Path(__file__)] + headers) or not stamp.exists() or stamp.read_text() != cmdline
        if stale:
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, stamp, cmdline,
                          subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, stamp, cmdline, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n{out}")
        stamp.write_text(cmdline)
        if verbose and out.strip():
            print(out)
    return objs, bool(procs)


def _link(objs, target: Path, version_script: Path | None):
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(target), *map(str, objs)]
    if version_script is not None:
        cmd.append(f"-Wl,--version-script={version_script}")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")


def _api_define():
    return ['-DWISE_BUILD_FLAGS="' + flags_string() + '"']


def build_hip(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    """The product library: exports == include/wise_hip.h, no debug switches."""
    srcs = [CSRC / s for s in HIP_SOURCES]
    LIBDIR.mkdir(parents=True, exist_ok=True)
    objs, rebuilt = _compile(srcs, LIBDIR / "obj", _api_define(), force, verbose, extra_flags)
    vs = LIBDIR / "obj" / "exports.map"
    text = "{\n  global:\n" + "".join(f"    {n};\n" for n in declared_symbols()) + "  local: *;\n};\n"
    if not vs.exists() or vs.read_text() != text:
        vs.write_text(text)
        rebuilt = True
    if rebuilt or not LIB.exists():
        _link(objs, LIB, vs)
    return LIB


def build_debug(force: bool = False, verbose: bool = False) -> Path:
    """The debug twin: the same kernels with their tuning / ablation switches live, plus the hardware probes."""
    srcs = [CSRC / s for s in HIP_SOURCES + DEBUG_ONLY_SOURCES if (CSRC / s).exists()]
    objs, rebuilt = _compile(srcs, LIBDIR / "obj_debug", _api_define() + ["-DWISE_DEBUG_KNOBS"], force, verbose)
    vs = LIBDIR / "obj_debug" / "exports.map"
    text = "{\n  global:\n    wise_*;\n  local: *;\n};\n"
    if not vs.exists() or vs.read_text() != text:
        vs.write_text(text)
        rebuilt = True
    if rebuilt or not LIB_DEBUG.exists():
        _link(objs, LIB_DEBUG, vs)
    return LIB_DEBUG


LIB_ASAN = LIBDIR / "libwise_hip_asan.so"


def asan_runtime() -> Path | None:
    """clang's shared AddressSanitizer runtime (must be LD_PRELOADed into the python that loads LIB_ASAN)."""
    hits = sorted(Path(hipcc_path()).resolve().parent.parent.glob("lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    hits += sorted(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[0] if hits else None


def build_asan(force: bool = False, verbose: bool = False) -> Path:
    """HOST side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the
    CPU build only): argument validation, layout / workspace planners, tap builders, the error buffer — everything
    that runs before a launch.  Host pass only (--cuda-host-only: no device code, nothing can be launched); loaded by
    tests/test_host_asan.py through WISE_AMD_LIB_PATH with the sanitizer runtime preloaded.  Never used by wise_amd."""
    srcs = [CSRC / s for s in HIP_SOURCES]
    flags = ["--cuda-host-only", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined"]
    objs, rebuilt = _compile(srcs, LIBDIR / "obj_asan", _api_define(), force, verbose, flags)
    if rebuilt or not LIB_ASAN.exists():
        # a host-only object still registers its (absent) device code when the library is loaded: give it empty fat
        # binaries and registration entry points of its own (-Bsymbolic: bound here, not to the HIP runtime)
        und = subprocess.run(["nm", "-u", *map(str, objs)], stdout=subprocess.PIPE, text=True).stdout
        fat = sorted(set(re.findall(r"\b(__hip_fatbin_[0-9a-f]+)\b", und)))
        stub = LIBDIR / "obj_asan" / "hip_registration_stub.c"
        stub.write_text("".join(f"const char {n}[16] = {{0}};\n" for n in fat) +
                        "static void* handle_;\n"
                        "void** __hipRegisterFatBinary(const void* d) { (void)d; return &handle_; }\n"
                        "void __hipUnregisterFatBinary(void** h) { (void)h; }\n"
                        "void __hipRegisterFunction(void** h, const char* f, char* df, const char* dn, unsigned tl, void* a, "
                        "void* b, void* c, void* d, int* w) { (void)h; (void)f; (void)df; (void)dn; (void)tl; (void)a; (void)b; "
                        "(void)c; (void)d; (void)w; }\n"
                        "void __hipRegisterVar(void** h, char* v, char* da, const char* dn, int e, unsigned long s, int c, int g) "
                        "{ (void)h; (void)v; (void)da; (void)dn; (void)e; (void)s; (void)c; (void)g; }\n")
        stub_o = stub.with_suffix(".o")
        subprocess.run(["gcc", "-O1", "-fPIC", "-c", str(stub), "-o", str(stub_o)], check=True)
        cmd = [hipcc_path(), "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-Wl,-Bsymbolic", "-o",
               str(LIB_ASAN), *map(str, objs), str(stub_o)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB_ASAN


def build_all(force: bool = False, verbose: bool = False):
    """Both libraries.  (The oracle's C restatement has its own recipe, oracle/build.py: test infrastructure.)"""
    lib = build_hip(force=force, verbose=verbose)
    build_debug(force=force, verbose=verbose)
    return lib


if __name__ == "__main__":
    lib = build_all(force="--force" in sys.argv, verbose=True)
    print("built", lib, LIB_DEBUG)
