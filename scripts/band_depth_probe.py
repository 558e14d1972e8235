"""How many band frames in flight does a rank's GPU want?  The band pair (ms_render_band_begin / _finish) driven directly
with D lanes -- D - 1 bands begun before the oldest is finished -- on one rank's band of a prepared scene, no exchange:
    python scripts/band_depth_probe.py cfg5 8 3 [frames]
Prints the sustained period per band for D = 1 .. 4 (D = 2 is what render_gaussians_sharded(async_op=True) does)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from mojosplat_amd import _band, _fused, _hip  # noqa: E402
from mojosplat_amd import distributed as D  # noqa: E402
from mojosplat_amd.scene_order import prepare_scene  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402

name, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 300
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda", 0)
sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = prepare_scene(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"]).arrays
cur = None
for _ in range(40):   # the library's own asynchronous path first: it settles the band's bin size
    nxt = D.render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(rank, world), async_op=True)
    if cur is not None:
        cur.wait()
    cur = nxt
cur.wait()
torch.cuda.synchronize()
th = -(-H // 16)
rows, bands = D.band_plan(th, world)
band = D._band_of(bands[rank], th)
_, mode = D._band_bin(g[0], cam, band, 16)
streams = list(_fused._lane_streams(dev)) + [torch.cuda.Stream(device=dev) for _ in range(2)]
raw = torch.cuda.current_stream(dev).cuda_stream
bufs = [torch.empty((max(world * rows * 16, H), W, 3), dtype=torch.float32, device=dev) for _ in range(4)]
out = {"workload": name, "world": world, "rank": rank, "bin_px": mode, "frames": frames, "period_us": {}}
for depth in (1, 2, 3, 4):
    pending = [None] * depth

    def run(n):
        for f in range(n):
            s = f % depth
            if pending[s] is not None:
                _band.band_finish(pending[s], raw)
            pending[s] = _band.band_begin(*g, cam, bg, mode, band, bufs[s], None, True, 1 + s, streams[s].cuda_stream, raw)
        for s in range(depth):
            if pending[s] is not None:
                _band.band_finish(pending[s], raw)
                pending[s] = None
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(frames)
    torch.cuda.synchronize()
    out["period_us"][str(depth)] = round((time.perf_counter() - t0) / frames * 1e6, 1)
print(json.dumps(out))
