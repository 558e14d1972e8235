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
    # an explicit tile size other than 16: the band plan, the slabs and the band handed to the library are all in rows
    # of THAT size (the library's 16-px-row mode is for bins that follow the rule under 16-px tiles only)
    for ts in (32, 64):
        ref_ts = ms.render_gaussians(*g, cam, background_color=bg, tile_size=ts)
        assert torch.equal(ref_ts, ref), f"tile_size={ts}: the single-GPU frame depends on the tile size"
        assert torch.equal(render_gaussians_sharded(*g, cam, background_color=bg, tile_size=ts), ref), \
            f"tile_size={ts}: blocking sharded frame differs"
        cur = None
        for k in range(3):
            nxt = render_gaussians_sharded(*g, cam, background_color=bg, tile_size=ts, async_op=True)
            if cur is not None:
                assert torch.equal(cur.wait(), ref), f"tile_size={ts}: pipelined frame {k - 1} differs"
            cur = nxt
        assert torch.equal(cur.wait(), ref)
    # ragged bands (explicit, then as the ranks' pair counts move them: the plan is re-made on every second frame
    # here) under the exchange form MOJOSPLAT_GATHER selects -- the padded in-place all-gather + compaction copy, or
    # grouped point-to-point sends / receives into the image's rows
    import mojosplat_amd.distributed as D
    D._CHECK_EVERY = 2
    th = -(-cam.H // 16)
    ragged = [0, 3, th] if world == 2 else [0, 2, 9, th]
    assert torch.equal(render_gaussians_sharded(*g, cam, background_color=bg, bounds=ragged), ref), "ragged blocking frame differs"
    assert torch.equal(render_gaussians_sharded(*g, cam, background_color=bg, bounds=ragged, async_op=True).wait(), ref)
    plans = []
    cur = None
    for k in range(10):
        plans.append(tuple(D.band_bounds(g[0], cam, 16, world)))
        nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
        if cur is not None:
            assert torch.equal(cur.wait(), ref), f"balanced pipelined frame {k - 1} differs"
        cur = nxt
    assert torch.equal(cur.wait(), ref)
    # round 5: a 16-bit exchange (every rank rounds its band once, the image comes back in that type) -- equal bands and
    # ragged ones, blocking and two frames in flight
    for dt in (torch.float16, torch.bfloat16):
        want = ref.to(dt)
        got = render_gaussians_sharded(*g, cam, background_color=bg, exchange_dtype=dt, bounds=ragged)
        assert got.dtype == dt and torch.equal(got, want), f"{dt}: ragged blocking frame is not the rounded float32 frame"
        cur = None
        for k in range(4):
            nxt = render_gaussians_sharded(*g, cam, background_color=bg, exchange_dtype=dt, async_op=True)
            if cur is not None:
                assert torch.equal(cur.wait(), want), f"{dt}: pipelined frame {k - 1} differs"
            cur = nxt
        assert torch.equal(cur.wait(), want)
    gathered = [None] * world
    dist.all_gather_object(gathered, plans)
    assert all(p == gathered[0] for p in gathered), "the ranks' plans diverged"
    print(f"rank {rank}/{world}: plans {sorted(set(plans))} gather={os.environ.get('MOJOSPLAT_GATHER', 'allgather')}", flush=True)
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
    # depth cut-offs on the ranks' bands (round 4), forced whatever the scene's size: a denser scene, pipelined, through a
    # swap to the same Gaussians with the near half nearly transparent (stale cut-offs: bins regenerated by position in the
    # band's candidate list) and back -- every gathered frame equals the single-GPU frame of its scene
    from mojosplat_amd import _fused, _hip
    _hip.config_depth_cut(2)
    sc3, cam3 = randscene_v1(300_000, 1280, 720, ell=-3.5, seed=42, device=dev)
    faint = dict(sc3)
    depth = (sc3["means3d"] @ cam3.R.T + cam3.T)[:, 2]
    faint["opacities"] = torch.where(depth < depth.median(), sc3["opacities"] * 0.02, sc3["opacities"])
    tup = lambda s_: (s_["means3d"], s_["scales"], s_["quats"], s_["opacities"], s_["features"])
    _hip.config_depth_cut(0)
    refs3 = {id(sc3): ms.render_gaussians(*tup(sc3), cam3, background_color=bg), id(faint): ms.render_gaussians(*tup(faint), cam3, background_color=bg)}
    _hip.config_depth_cut(2)
    _fused.FRAME_STATS = stats = {}
    seq = [sc3] * 5 + [faint] * 3 + [sc3] * 3
    cur = None
    for k, s_ in enumerate(seq):
        nxt = (s_, render_gaussians_sharded(*tup(s_), cam3, background_color=bg, async_op=True))
        if cur is not None:
            assert torch.equal(cur[1].wait(), refs3[id(cur[0])]), f"depth-cut band frame {k - 1} differs"
        cur = nxt
    assert torch.equal(cur[1].wait(), refs3[id(cur[0])])
    _fused.FRAME_STATS = None
    _hip.config_depth_cut(1)
    print(f"rank {rank}/{world}: band frames with depth cuts: {stats.get('depth_cut', 0)} cut, {stats.get('cut_redo_tiles', 0)} bins regenerated", flush=True)
    # the other sharding axis: C views split over the ranks by view, one all-gather (render_gaussians_batch_sharded);
    # C = 5 is not a multiple of 2 or 3 ranks, and one view looks away from the scene (the zeros-image rule is per view)
    from mojosplat_amd.distributed import render_gaussians_batch_sharded
    from mojosplat_amd.utils import Camera, look_at
    eyes = [(0.0, 1.5, 5.0), (2.0, 1.0, 4.5), (-2.5, 2.0, 4.0), (0.5, -1.0, 6.0), (0.0, 1.5, 40.0)]
    targets = [(0, 0, 0)] * 4 + [(0.0, 1.5, 80.0)]
    cams = []
    for e, t in zip(eyes, targets):
        vm = look_at(torch.tensor(e), torch.tensor(t, dtype=torch.float32), torch.tensor([0.0, 1.0, 0.0]))
        cams.append(Camera(R=vm[:3, :3].contiguous().to(dev), T=vm[:3, 3].contiguous().to(dev), H=cam.H, W=cam.W,
                           fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, near=cam.near, far=cam.far))
    views = render_gaussians_batch_sharded(*g, cams, background_color=bg)
    assert views.shape == (5, cam.H, cam.W, 3)
    for v, c in enumerate(cams):
        assert torch.equal(views[v], ms.render_gaussians(*g, c, background_color=bg)), f"view {v} differs"
    assert (views[4] == 0).all(), "a view that sees nothing is the zeros image"
    v16 = render_gaussians_batch_sharded(*g, cams, background_color=bg, exchange_dtype=torch.float16)
    assert v16.dtype == torch.float16 and torch.equal(v16, views.half()), "the view batch's 16-bit exchange is not the rounded batch"
    pend = None
    for k in range(3):   # one call ahead
        nxt = render_gaussians_batch_sharded(*g, cams, background_color=bg, async_op=True)
        if pend is not None:
            assert torch.equal(pend.wait(), views), f"pipelined view batch {k - 1} differs"
        pend = nxt
    assert torch.equal(pend.wait(), views)
    print(f"rank {rank}/{world}: sharded frames (blocking + pipelined) equal the single-GPU frame", flush=True)
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
    raise
finally:
    dist.destroy_process_group()
