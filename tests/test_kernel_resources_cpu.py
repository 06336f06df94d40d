"""The built library's own metadata (no GPU): the kernels whose MFMAs are inline-assembly statements must live inside their
register budget WITHOUT compiler spills — a value the compiler parks in the other half of the register file is copied back right
next to a statement whose hazards it does not know (csrc/mlp_stream.hip: a build of one form for two workgroups per CU returned
wrong rows).  swin_stream.hip's statements carry an s_nop guard for that reason; mlp_stream.hip's do not and must show exactly
their accumulators in the accumulator file."""
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "wise_amd" / "lib" / "libwise_hip.so"


@pytest.fixture(scope="module")
def rows():
    if not LIB.exists():
        pytest.skip("library not built")
    spec = importlib.util.spec_from_file_location("kernel_resources", ROOT / "tools" / "kernel_resources.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.kernel_rows(LIB)
    if not r:
        pytest.skip("no code-object metadata readable here")
    return r


def test_streaming_kernels_have_no_scratch_and_their_accumulators_only(rows):
    seen = 0
    for name, agpr, vgpr, sgpr, scratch, vsp, ssp in rows:
        if "mlp_stream_kernel" in name:
            seen += 1
            assert scratch == "0" and vsp == "0", (name, scratch, vsp)
            if "true>" in name and "384" in name:
                assert int(agpr) <= 99, (name, agpr)          # 96 accumulators (+ 3 prologue values: the LayerNorm variant)
            else:
                assert int(agpr) == 96, (name, agpr)
        if "swin_qkv_attn_kernel" in name or "attn_oproj_fold_kernel" in name:
            seen += 1
            assert scratch == "0" and vsp == "0", (name, scratch, vsp)
    assert seen >= 8
