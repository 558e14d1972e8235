"""First RCCL contact on a one-GPU box: a world-1 `nccl` (= RCCL) process group on cuda:0 driving the LIVE-group path of the
sharded entry points (MOJOSPLAT_FORCE_EXCHANGE=1: distributed.py runs the exchange for a group of one): the 32-byte status
all-gather, the in-place all_gather_into_tensor of the framebuffer (input slab aliasing its slot of the output), blocking
and with two frames in flight on the lane streams, both MOJOSPLAT_GATHER modes, the view-sharded batch -- every frame
compared bit for bit with the single-GPU frame.  gloo's work.wait() blocks the host; RCCL's only orders the stream: this is
the run that exercises the product's stream ordering around the collectives.
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P scripts/rccl_world1.py
Prints one JSON line (rccl version, what ran, frames compared)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ["MOJOSPLAT_FORCE_EXCHANGE"] = "1"
import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
import mojosplat_amd as ms
import mojosplat_amd.distributed as D
from mojosplat_amd.distributed import render_gaussians_batch_sharded, render_gaussians_sharded
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rccl_version": list(torch.cuda.nccl.version()),
       "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "frames_compared": 0, "modes": []}
try:
    assert out["backend"] == "nccl" and out["world"] == 1
    for (N, W, H, ell) in ((50_000, 640, 360, -3.0), (300_000, 1920, 1080, -4.0)):
        sc, cam = randscene_v1(N, W, H, ell=ell, seed=5, device=dev)
        bg = torch.tensor(BACKGROUND_V1, device=dev)
        g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
        ref = ms.render_gaussians(*g, cam, background_color=bg)
        for gather in ("allgather", "direct"):
            os.environ["MOJOSPLAT_GATHER"] = gather
            D._plans.clear()
            D._CHECK_EVERY = 2          # (re-plan -- and read the gathered status records on the host -- every second frame)
            img = render_gaussians_sharded(*g, cam, background_color=bg)
            assert torch.equal(img, ref), f"{gather}: blocking sharded frame differs"
            out["frames_compared"] += 1
            cur = None
            for k in range(8):           # two frames in flight: frame k + 1's band is enqueued before frame k's gather
                nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
                if cur is not None:
                    assert torch.equal(cur.wait(), ref), f"{gather}: pipelined frame {k - 1} differs"
                    out["frames_compared"] += 1
                cur = nxt
            assert torch.equal(cur.wait(), ref)
            out["frames_compared"] += 1
            # a different stream as the caller's current one: the collectives must order themselves behind IT
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                a = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
                b = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
                ia, ib = a.wait(), b.wait()
                ok = torch.equal(ia, ref) and torch.equal(ib, ref)
            torch.cuda.current_stream(dev).wait_stream(side)
            assert ok, f"{gather}: frames on a side stream differ"
            out["frames_compared"] += 2
            out["modes"].append(f"{N}@{W}x{H}:{gather}")
        # a 16-bit exchange over RCCL (half-precision buffers through the same in-place all-gather)
        os.environ["MOJOSPLAT_GATHER"] = "allgather"
        for dt in (torch.float16, torch.bfloat16):
            a = render_gaussians_sharded(*g, cam, background_color=bg, exchange_dtype=dt)
            b = render_gaussians_sharded(*g, cam, background_color=bg, exchange_dtype=dt, async_op=True).wait()
            assert a.dtype == dt and torch.equal(a, ref.to(dt)) and torch.equal(b, ref.to(dt)), f"{dt} exchange differs"
            out["frames_compared"] += 2
        out["modes"].append(f"{N}@{W}x{H}:f16+bf16 exchange")
        # an empty frame (nothing on the grid): the zeros rule read from the gathered records
        far = (g[0] + torch.tensor([0.0, 0.0, 500.0], device=dev),) + g[1:]
        z = render_gaussians_sharded(*far, cam, background_color=bg)
        assert z.shape == ref.shape and (z == 0).all(), "empty frame is not the zeros image"
        out["frames_compared"] += 1
    # the other sharding axis: whole views per rank, one in-place all-gather of the batch
    sc, cam = randscene_v1(50_000, 640, 360, ell=-3.0, seed=5, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    from mojosplat_amd.utils import Camera
    cams = []
    for k in range(3):
        th = 0.2 * k
        R = torch.tensor([[torch.cos(torch.tensor(th)), 0.0, torch.sin(torch.tensor(th))], [0.0, 1.0, 0.0],
                          [-torch.sin(torch.tensor(th)), 0.0, torch.cos(torch.tensor(th))]], device=dev) @ cam.R
        cams.append(Camera(R=R.contiguous(), T=cam.T, H=cam.H, W=cam.W, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, near=cam.near, far=cam.far))
    singles = [ms.render_gaussians(*g, c, background_color=bg) for c in cams]
    views = render_gaussians_batch_sharded(*g, cams, background_color=bg)
    pend = render_gaussians_batch_sharded(*g, cams, background_color=bg, async_op=True)
    views2 = pend.wait()
    for k in range(3):
        assert torch.equal(views[k], singles[k]) and torch.equal(views2[k], singles[k]), f"view {k} differs"
        out["frames_compared"] += 2
    out["modes"].append("view-sharded batch")
    torch.cuda.synchronize()
    out["ok"] = True
finally:
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()
