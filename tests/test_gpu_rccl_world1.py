"""The RCCL code path on one GPU: a world-size-1 `nccl` process group executes init, all_reduce,
all_to_all_single + all_gather_into_tensor (the 'direct' exchange), all_gather, broadcast and barrier of nafae_amd.parallel, a
replicated DP step in both exchange modes and a frame-sharded exact step -- so the driver's 8-GPU run is not the first time
any of these calls happen.  The reference has no distributed path (model.py:91-99: --mGPUs is parsed and never read)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_world1_collectives_and_steps():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res.pt")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py"), out], env=_env(),
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        res = torch.load(out)
    print(res)
    assert res["backend"] == "nccl" and res["world"] == 1
    for k in ("reduce_allreduce_equal", "reduce_direct_equal", "all_gather_rows_equal", "all_gather_rows_i64_equal",
              "broadcast_rows_equal", "step_allreduce_loss_equal", "step_allreduce_params_equal", "step_direct_loss_equal",
              "step_direct_params_equal"):
        assert res[k], k
    assert abs(res["exact_loss"] - res["ref_loss"]) <= 1e-6 * abs(res["ref_loss"])
    assert res["exact_params_maxdiff"] <= 1e-6


def test_bench_force_dist_world1_reports_nccl():
    """`bench.py --gpus 1 --force-dist`: the timed steps all-reduce through RCCL, the line says so."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                        "--workload", "c1", "--no-cpu-baseline", "--no-other-precisions"], env=_env(), capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]        # RCCL's version banner and everything else go to stderr: stdout is the JSON line only
    out = json.loads(lines[0])
    assert out["config"]["collective_backend"] == "nccl" and out["config"]["rccl_world_size"] == 1
    assert out["config"]["grad_allreduce_bytes"] == 2201600 * 4
    assert out["value"] > 0
