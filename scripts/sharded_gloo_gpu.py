"""Two ranks on ONE GPU with the gloo backend: exercises render_gaussians_sharded's HIP path (lanes,
split-phase, band render) together with a real process group -- RCCL refuses two ranks on one
device, so this is as close as a single-GPU box gets.  python -m torch.distributed.run --nproc-per-node 2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import mojosplat_amd as ms
from mojosplat_amd.distributed import render_gaussians_sharded
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sc, cam = randscene_v1(50_000, 640, 360, ell=-3.0, seed=5, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
ref = ms.render_gaussians(*g, cam, background_color=bg)
try:
    img = render_gaussians_sharded(*g, cam, background_color=bg)
    assert torch.equal(img, ref), "blocking sharded frame differs"
    cur = None
    for k in range(6):
        nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
        if cur is not None:
            assert torch.equal(cur.wait(), ref), f"pipelined frame {k - 1} differs"
        cur = nxt
    assert torch.equal(cur.wait(), ref)
    # inputs the frame has to marshal (float64 / strided / fp16): their copies are made on the current stream
    # and read on a lane stream -- the ordering the pipelined path must get right
    g2 = (g[0].double(), torch.stack([g[1], g[1]], 1)[:, 0], g[2].double(), g[3], g[4].half())
    ref2 = ms.render_gaussians(*g2, cam, background_color=bg)
    cur = None
    for k in range(6):
        nxt = render_gaussians_sharded(*g2, cam, background_color=bg, async_op=True)
        if cur is not None:
            assert torch.equal(cur.wait(), ref2), f"pipelined marshalled frame {k - 1} differs"
        cur = nxt
    assert torch.equal(cur.wait(), ref2)
    print(f"rank {rank}/{world}: sharded frames (blocking + pipelined) equal the single-GPU frame", flush=True)
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
    raise
finally:
    dist.destroy_process_group()
