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
@pytest.mark.parametrize("world,gather", [(2, "allgather"), (3, "allgather"), (3, "direct")])
def test_sharded_hip_path_with_a_process_group(world, gather):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "scripts", "sharded_gloo_gpu.py")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MOJOSPLAT_GATHER=gather)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for rank in range(world):
        assert f"rank {rank}/{world}: sharded frames (blocking + pipelined) equal the single-GPU frame" in out, out[-3000:]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver calls it (no launcher, no WORLD_SIZE): bench.py starts
    the ranks itself as a child process, relays rank 0's JSON line and the return code.  On this one-GPU box
    the two ranks share the device over gloo (MOJOSPLAT_BENCH_REHEARSE=1; RCCL refuses two ranks per device)."""
    import json
    env = dict(os.environ, MOJOSPLAT_BENCH_REHEARSE="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--workload", "cfg2", "--no-extras"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl"]["world"] == 2 and out["rccl"]["backend"] == "gloo"
    assert out["verified"] is True and out["max_abs_vs_stagewise"] == 0.0 and out["value"] > 0
    assert "REHEARSAL" in out["data"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_training_step_with_a_process_group(world):
    """Round 5: render_gaussians_trainable_sharded on the HIP path -- each rank's band forward with its alphas kept, the band's
    backward rasteriser, ONE all-reduce of the per-Gaussian gradient rows, the backward projection on the sum (SURVEY.md
    section 8(e)); image bit for bit and gradients within the float atomics' order of the single-GPU step."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "scripts", "sharded_train_gloo_gpu.py")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for rank in range(world):
        assert f"rank {rank}/{world}: sharded training steps equal the single-GPU step" in out, out[-3000:]
