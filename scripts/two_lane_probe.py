"""Does a second frame in flight on its own stream buy throughput at config 3?  Frames alternate between two lanes (own
scratch, own stream), each begun before the other lane's previous frame is finished.
python scripts/two_lane_probe.py [cfg3] [frames]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.scenes import randscene_v1, BACKGROUND_V1
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
bg = torch.tensor(BACKGROUND_V1, device=dev).to(sc["features"].dtype)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
px = int(os.environ.get("PROBE_PX", "16"))
out = {}
# one lane, the plain call
for _ in range(150):
    ms.render_gaussians(*g, cam, background_color=bg, backend="hip", bin_size=px)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    ms.render_gaussians(*g, cam, background_color=bg, backend="hip", bin_size=px)
torch.cuda.synchronize()
out["one_lane_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
ref = ms.render_gaussians(*g, cam, background_color=bg, backend="hip", bin_size=px)
# two lanes, two streams
S = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
imgs = [torch.empty((H, W, 3), device=dev), torch.empty((H, W, 3), device=dev)]
pend = [None, None]
torch.cuda.synchronize()
def run(n):
    for k in range(n):
        l = k & 1
        if pend[l] is not None:
            pend[l].finish()
        with torch.cuda.stream(S[l]):
            pend[l] = _fused.render_begin_hip(*g, cam, bg, px, out=imgs[l], lane=1 + l)
    for l in (0, 1):
        if pend[l] is not None:
            pend[l].finish(); pend[l] = None
run(150)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(frames)
torch.cuda.synchronize()
out["two_lanes_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
out["two_lane_images_equal"] = bool(torch.equal(imgs[0], ref) and torch.equal(imgs[1], ref))
# variants of the loop, to see what the product's form (below) pays for
def run_b(n, fresh):   # begin frame k + 1, THEN finish frame k (the other lane), as a caller of async_op does
    prev = None
    for k in range(n):
        l = k & 1
        with torch.cuda.stream(S[l]):
            f = _fused.render_begin_hip(*g, cam, bg, px, out=None if fresh else imgs[l], lane=1 + l)
        if prev is not None:
            prev.finish()
        prev = f
    prev.finish()
for fresh in (False, True):
    run_b(150, fresh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_b(frames, fresh)
    torch.cuda.synchronize()
    out["begin_then_finish_other_lane" + ("_fresh_image" if fresh else "") + "_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
def run_c(n):
    cur = ms.render_gaussians(*g, cam, background_color=bg, backend="hip", bin_size=px, async_op=True)
    for _ in range(n - 1):
        nxt = ms.render_gaussians(*g, cam, background_color=bg, backend="hip", bin_size=px, async_op=True)
        cur.wait()
        cur = nxt
    return cur.wait()
run_c(150)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_c(frames)
torch.cuda.synchronize()
out["render_gaussians_async_op_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
out["lane_calibration"] = {str(k_): v_ for k_, v_ in _fused.LANE_CALIBRATION.items()}
# the product's own form of it: render_gaussians_sharded(async_op=True) without a process group is a world of one --
# the whole frame on alternating lane streams, .wait() one frame later
from mojosplat_amd.distributed import render_gaussians_sharded
def run_api(n):
    cur = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
    img = None
    for _ in range(n - 1):
        nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True)
        img = cur.wait()
        cur = nxt
    return cur.wait()
run_api(150)
torch.cuda.synchronize()
t0 = time.perf_counter()
img = run_api(frames)
torch.cuda.synchronize()
out["sharded_async_world1_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
out["sharded_async_image_equal"] = bool(torch.equal(img, ref))
out.update(config=name, bin_px=px, frames=frames)
print(json.dumps(out))
