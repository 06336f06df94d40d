"""Every kernel of the one-wave-per-SIMD GEMM family issues its MFMAs as volatile asm (gemm_w4.h): the compiler pads no wait
states behind them, so ANY accumulator-file copy (v_accvgpr_mov / read / write) it places between the first and the last
MFMA of a kernel reads or moves a tile that may not have landed.  Disassemble libwise_hip.so and list such kernels.
    python tools/check_mfma_loop.py [substring ...]        exit code 1 when a kernel has copies inside its MFMA range"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import kernel_resources as kr  # noqa: E402


def main():
    want = sys.argv[1:] or ["gemm_w4"]
    so = kr.ROOT / "wise_amd" / "lib" / "libwise_hip.so"
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for n, co in enumerate(kr.code_objects(so)):
            f = Path(td) / f"{n}.co"
            f.write_bytes(co)
            asm = subprocess.run([str(kr.LLVM / "llvm-objdump"), "-d", str(f)], capture_output=True, text=True).stdout
            for m in re.finditer(r"\n[0-9a-f]+ <(\S+)>:\n", asm):
                name = m.group(1)
                end = asm.find("\n\n", m.end())
                body = asm[m.end(): end if end > 0 else len(asm)].split("\n")
                mf = [i for i, l in enumerate(body) if "v_mfma_" in l]
                if len(mf) < 64:
                    continue
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                short = re.sub(r"\(.*", "", dem.replace("void wise::", ""))
                if not any(w in short for w in want):
                    continue
                inside = [l.strip() for l in body[mf[0]:mf[-1]] if "v_accvgpr_" in l]
                flag = f"  <-- {len(inside)} accumulator-file copies inside the MFMA range, e.g. {inside[0].split('//')[0].strip()}" if inside else ""
                bad += bool(inside)
                print(f"{short:60s} {len(mf):5d} MFMAs{flag}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
