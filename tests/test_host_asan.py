"""The HOST side of libwise_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, no GPU): argument
validation, layout and workspace planners and the error buffer are driven through the C ABI in a child process that
preloads the sanitizer runtime; any report fails the test.  (SURVEY section 5 "race detection / sanitizers"; GPU
sanitizers are not available on this pool, so this build carries no device code.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %(root)r)
from wise_amd import _lib
lib = _lib.load(%(lib)r)
assert lib.wise_abi_version() == 5
assert b"-fno-slp-vectorize" in lib.wise_build_flags()
# flat search: planners and refusals
assert lib.wise_ip_topk_workspace_bytes(1000, 510, 1, 10) > 0
assert lib.wise_ip_topk_workspace_bytes(1000, 512, 1, 5000) == 0
assert lib.wise_ip_topk_f32(0, 10, 511, 0, 1, 10, 0, 0, 0, 0, 0, 0, 0) == -1 and b"multiple of 4" in lib.wise_last_error()
for n in (1000, 1 << 18, 10_000_000):
    for nq in (1, 2, 64, 256):
        for k in (1, 10, 16, 20, 100, 1000, 2048):
            lib.wise_ip_topk_workspace_bytes(n, 512, nq, k)
            lib.wise_ip_topk_shadow_workspace_bytes(n, 512, nq, k)
assert lib.wise_ip_topk_shadow_f32(0, 0, 0, 1000, 512, 0, 1, 10, 0, 0, 0, 0, 0, 0, 0, 0) == -1
assert lib.wise_ip_shadow_bf16(0, 10, 12, 0, 0, 0) == -1 and b"ip_shadow_bf16" in lib.wise_last_error()
lib.wise_ivf_scan_workspace_bytes(512, 256, 10)
# towers: layouts and workspaces over every model family the extractors build
nb, nf = C.c_int64(), C.c_int64()
from wise_amd.feature.vit import spec_for
for name, tag in (("ViT-B-32", "openai"), ("ViT-B-16", "openai"), ("ViT-L-14", "openai"), ("ViT-H-14", "laion2b_s32b_b79k")):
    cfg = spec_for(name, tag).c_config()
    assert lib.wise_vit_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0 and nb.value > 0
    for b in (1, 37, 256):
        assert lib.wise_vit_workspace_bytes(C.byref(cfg), b) > 0
    # forward with null buffers / short workspace: refused before any launch
    assert lib.wise_vit_forward(C.byref(cfg), 0, 0, 0, 0, 4, 0, 0, 0, 0) != 0
bad = _lib.VitConfig(224, 32, 700, 12, 12, 3072, 512, 0)
assert lib.wise_vit_layout(C.byref(bad), C.byref(nb), C.byref(nf)) == -1
tc = _lib.TextConfig(77, 49408, 512, 12, 8, 2048, 512, 0, 0, 0, 0, 0)
assert lib.wise_text_layout(C.byref(tc), C.byref(nb), C.byref(nf)) == 0
for b in (1, 2, 256):
    assert lib.wise_text_workspace_bytes(C.byref(tc), b) > 0
assert lib.wise_text_forward(C.byref(tc), 0, 0, 0, 1, 0, 0, 0, 0) != 0
xc = _lib.XlmrConfig(77, 250002, 514, 1024, 24, 16, 4096, 1024, 1024, 1, 0, 0, 0, 0)
assert lib.wise_xlmr_layout(C.byref(xc), C.byref(nb), C.byref(nf)) == 0
for b in (1, 3, 256):
    assert lib.wise_xlmr_workspace_bytes(C.byref(xc), b) > 0
assert lib.wise_htsat_layout(C.byref(nb), C.byref(nf)) == 0
assert lib.wise_htsat_workspace_bytes(128, 480000) > 0 and lib.wise_htsat_workspace_bytes(0, 480000) == 0
assert lib.wise_cnn14_layout(C.byref(nb), C.byref(nf)) == 0
assert lib.wise_cnn14_workspace_bytes(64, 480000) > 0
assert lib.wise_htsat_forward(0, 0, 0, 1, 480000, 0, 0, 0, 0) != 0
assert lib.wise_gemm_bf16(0, 0, 0, 100, 8, 8, 0, 0, 0) != 0 and b"gemm_bf16" in lib.wise_last_error()
print("asan-child-ok")
"""


@pytest.mark.timeout(900)
def test_host_side_is_clean_under_asan_and_ubsan():
    from wise_amd import build

    rt = build.asan_runtime()
    if rt is None:
        pytest.skip("clang's shared asan runtime not found")
    lib = build.build_asan()
    env = dict(os.environ, LD_PRELOAD=str(rt), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": str(ROOT), "lib": str(lib)}], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=800)
    assert r.returncode == 0 and "asan-child-ok" in r.stdout, r.stdout[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stdout and "runtime error:" not in r.stdout, r.stdout[-4000:]
