"""RCCL on the one GPU a test box has: a world-1 `nccl` process group drives the live-group path of the sharded entry
points (scripts/rccl_world1.py: status all-gather, in-place framebuffer all-gather, two frames in flight on the lane
streams, both exchange forms, the view-sharded batch), every frame bit-identical to the single-GPU frame.  Rounds 1-4
never executed a RCCL collective: every multi-rank run used gloo (RCCL refuses two ranks per device), whose work.wait()
blocks the host where RCCL's only orders the stream."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_world1_rccl_group_runs_the_live_sharded_path():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "scripts", "rccl_world1.py")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    print("RCCL", line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_world1.json"), "w") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert rec.get("ok") is True and rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["frames_compared"] >= 40 and "view-sharded batch" in rec["modes"]
