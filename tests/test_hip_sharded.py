"""The multi-GPU entry point with a REAL process group on the HIP path: two ranks share the one GPU of
the test box over gloo (RCCL refuses two ranks on one device; gloo moves CUDA tensors and supports
the in-place, asynchronous all_gather_into_tensor the product issues).  Each rank renders its band
with the library (lane streams, split-phase frames, band-only scan, lazily sorted lists) and the
gathered frame must equal the single-GPU frame bit for bit -- blocking and with frames in flight."""
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
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_path_with_a_process_group(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "scripts", "sharded_gloo_gpu.py")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for rank in range(world):
        assert f"rank {rank}/{world}: sharded frames (blocking + pipelined) equal the single-GPU frame" in out, out[-3000:]
