"""RCCL under the test runner (-m gpu): the multi-GPU step of bench.py -- one all_gather_into_tensor of the ranks' shards of B (backend "nccl" = RCCL) followed by
sparta_vbs_spmm_gathered on the rank's part -- run as a CHILD process (never an exec of a process that holds the GPU), with one rank on a one-GPU box and with two
ranks where two GPUs are visible.  The reference has no counterpart (single process, single GPU: batch/VBR_batch_a5:6-7); the point is that the 8-GPU scaling run is
not this code base's first RCCL call."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-3000:]
    return json.loads(lines[-1])


def test_one_rank_under_nccl_all_gather_plus_gathered_product():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-path", "--backend", "nccl", "--workload", "rmat-part", "--rmat-scale", "16",
                        "--rmat-density", "1e-3", "--dtype", "bf16", "--ncols", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["steps"] == 3
    cfg = out["config"]
    assert cfg["allgather"]["mode"] == "all_gather" and 0.0 < cfg["allgather"]["ms_alone"] < 50.0, cfg["allgather"]
    assert cfg["parity_spot_check"]["max_err_over_sum_abs"] <= 1e-5
    assert out["value"] > 0.0 and out["ms_per_step"] > 0.0
    assert "1 all-gather of B per step" in cfg["workload"]


def test_two_ranks_under_nccl_when_two_gpus_are_visible():
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rmat-scale", "16", "--rmat-density", "1e-3", "--dtype", "bf16", "--ncols", "256",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["allgather"]["mode"] in ("all_gather", "peer_copies") and 0.0 < cfg["allgather"]["ms_alone"] < 100.0
    assert len(cfg["parts_detail"]) == 2 and cfg["parity_spot_check"]["max_err_over_sum_abs"] <= 1e-5
