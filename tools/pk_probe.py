"""packed-fp32 VALU forms run twice and compared, alone and beside an MFMA-only kernel on another stream"""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd import _lib
lib = _lib.lib()
lib.wise_debug_pk_probe.restype = ctypes.c_int
lib.wise_debug_pk_probe.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
lib.wise_debug_neighbour.restype = ctypes.c_int
lib.wise_debug_neighbour.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
sts = [torch.cuda.Stream() for _ in range(2)]
src = torch.randint(0, 2**31 - 1, (4096 * 1024 + 1024,), dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
names = ["v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_add_f32 neg", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
         "v_pk_mov_b32 op_sel", "v_fma_f32 (control)", "v_pk_mul_f32 sgpr",
         "pk->v_add_u32", "pk->ds_write/read", "pk->v_fma_f32", "pk->v_mul_lo_u32", "pk->v_pk_mov", "pk->v_rcp_f32", "ds_write_b64 then pk overwrite", "ds_write2_b64 then pk overwrite"]
for beside in ("nothing", "MFMA", "global loads"):
    rep = torch.zeros(16, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for what in range(16):
        if beside != "nothing":
            for _ in range(4):
                lib.wise_debug_neighbour(3 if beside == "MFMA" else 5, 2048, 61440, 64, src.data_ptr(), sink.data_ptr(), sts[1].cuda_stream)
        _lib.check(lib.wise_debug_pk_probe(what, 1024, 34816, 400, rep.data_ptr(), sts[0].cuda_stream), "probe")
        torch.cuda.synchronize()
    r = rep.tolist()
    print(f"beside {beside}: " + "; ".join(f"{n}: {r[i]}" for i, n in enumerate(names)), flush=True)
