"""packed-fp32 VALU forms run twice and compared, alone and beside an MFMA-only kernel on another stream"""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
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

# ---- the shape that failed in the product: rows of X dotted with a query, operands from global loads into packed fmas
lib.wise_debug_pk_dot_probe.restype = ctypes.c_int
lib.wise_debug_pk_dot_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
g = torch.Generator("cuda").manual_seed(5)
N, d, blocks = 1_000_000, 512, 64
X = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=g), dim=1)
Qd = torch.nn.functional.normalize(torch.randn(blocks, d, device="cuda", generator=g), dim=1)
rows = torch.randint(0, N, (blocks * 64,), device="cuda", generator=g, dtype=torch.int64)
rows[::7] = -1          # some "no candidate" slots, as the product kernel sees them
form = 0
def dots():
    out = torch.empty(blocks * 64, device="cuda")
    lib.wise_debug_pk_dot_probe(X.data_ptr(), d, Qd.data_ptr(), rows.data_ptr(), blocks, out.data_ptr(), sts[0].cuda_stream, form)
    return out
for form in (0, 1):
  torch.cuda.synchronize()
  ref = dots(); torch.cuda.synchronize()
  for beside in ("nothing", "MFMA"):
      wrong_calls, worst = 0, 0.0
      for rep in range(40):
          if beside == "MFMA":
              for _ in range(4): lib.wise_debug_neighbour(3, 2048, 1024, 64, src.data_ptr(), sink.data_ptr(), sts[1].cuda_stream)
          got = dots()
          if beside == "MFMA":
              for _ in range(2): lib.wise_debug_neighbour(3, 2048, 1024, 64, src.data_ptr(), sink.data_ptr(), sts[1].cuda_stream)
          torch.cuda.synchronize()
          if not torch.equal(got, ref):
              wrong_calls += 1
              worst = max(worst, float((got - ref).abs().max()))
      print(f"packed-fma dot products fed by global loads (form {form}), beside {beside}: {wrong_calls} of 40 calls differ (max |diff| {worst:.3e})", flush=True)

# ---- narrowed: v_pk_fma_f32 op_sel:[0,1,0] with the destination over src1 vs a destination of its own
lib.wise_debug_pk_overlap_probe.restype = ctypes.c_int
lib.wise_debug_pk_overlap_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for beside in ("nothing", "MFMA"):
    rep = torch.zeros(20, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        if beside == "MFMA":
            for _ in range(4): lib.wise_debug_neighbour(3, 2048, 1024, 64, src.data_ptr(), sink.data_ptr(), sts[1].cuda_stream)
        lib.wise_debug_pk_overlap_probe(1024, 200, rep.data_ptr(), sts[0].cuda_stream)
        torch.cuda.synchronize()
    r = rep.tolist()
    print(f"v_pk_fma_f32 op_sel:[0,1,0] beside {beside}: own destination {r[0]} wrong halves, destination over src1 {r[1]} wrong halves", flush=True)
    print(f"   with s_nop 0 in front: {r[16]} wrong halves; with s_nop 3 in front: {r[17]} wrong halves")
    if r[0]:
        import struct
        f = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
        print(f"   wrong low halves {r[4]} (of which = a.lo*b.LO+c.lo: {r[2]}), wrong high halves {r[5]} (of which = a.hi*b.LO+c.hi: {r[3]})")
        print("   first: a", f(r[8]), f(r[9]), "b", f(r[10]), f(r[11]), "c", f(r[12]), f(r[13]), "-> got", f(r[14]), f(r[15]),
              "expected", f(r[8]) * f(r[11]) + f(r[12]), f(r[9]) * f(r[11]) + f(r[13]))
