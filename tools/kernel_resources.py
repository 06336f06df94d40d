"""Registers, scratch and spills of the gfx950 kernels inside libwise_hip.so (the library that is loaded, not the objects):
    python tools/kernel_resources.py [substring of the demangled kernel name ...]
Reads the code objects' metadata notes.  A kernel of the one-wave-per-SIMD GEMM family with scratch or spills is a bug."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")


def code_objects(so: Path):
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(so)], check=True, capture_output=True)
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


def kernel_rows(so: Path = ROOT / "wise_amd" / "lib" / "libwise_hip.so"):
    """[(demangled name, agprs, registers, sgprs, scratch bytes, vgpr spills, sgpr spills)] of every gfx950 kernel in the library"""
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for n, co in enumerate(code_objects(so)):
            f = Path(td) / f"{n}.co"
            f.write_bytes(co)
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(f)], capture_output=True, text=True).stdout
            for ent in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", ent) or [None, "?"])[1]
                rows.append((get("name"), ent.split()[0], get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"),
                             get("vgpr_spill_count"), get("sgpr_spill_count")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    out = []
    for (mangled, agpr, vgpr, sgpr, scratch, vsp, ssp), name in zip(rows, names):
        short = re.sub(r"\(.*", "", name.replace("void wise::", "").replace("void ", ""))
        out.append((short, agpr, vgpr, sgpr, scratch, vsp, ssp))
    return out


def main():
    want = sys.argv[1:]
    for short, agpr, vgpr, sgpr, scratch, vsp, ssp in kernel_rows():
        if want and not any(w in short for w in want):
            continue
        flag = "  <-- SCRATCH / SPILLS" if scratch not in ("0", "?") or vsp not in ("0", "?") else ""
        print(f"{short:70s} {'a' + agpr:>5s} vgpr {vgpr:>3s} sgpr {sgpr:>3s} scratch {scratch:>5s} vspill {vsp:>3s} sspill {ssp:>3s}{flag}")


if __name__ == "__main__":
    main()
