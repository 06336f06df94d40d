"""-m gpu: SURVEY.md 8(e) on the one GPU there is — the row-sharded flat index behind the SearchIndex plugin surface with
torch.distributed initialised on the `nccl` backend (RCCL) at world size 1 and the one-rank short-cut switched off, so the
collective (all_gather_into_tensor) and wise_topk_merge execute for real.  The worker is a child process with its own
time limit (tests/sharded_nccl_worker.py): a stuck collective cannot take the test session with it."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_sharded_index_over_rccl_world1(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "sharded_nccl_worker.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["ok"], res
    assert res["exchange_bytes"] == 2 * 3 * 128 * 8          # the last search: nq = 3, k = 128, (score, id) planes of int64
