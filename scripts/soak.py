"""Usage scenarios beyond the benchmark loop, each checked against the per-stage frame and timed:
a moving camera, two image sizes in turn, a scene whose N creeps from frame to frame."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd import render as R
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
from mojosplat_amd.utils import Camera

dev = torch.device("cuda", 0)
bg = torch.tensor(BACKGROUND_V1, device=dev)


def stagewise(g, cam):
    m2, con, dep, rad = ms.project_gaussians(*g[:4], cam, backend="hip")
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, cam.H, cam.W, 16, backend="hip")
    return ms.rasterize_gaussians(m2, con, g[4], g[3], bg, ranges, ids, cam, tile_size=16, backend="hip")


def races():
    return {k[1:]: (t.choice, len(t.queue)) for k, t in R._BIN_CHOICE.items()}


sc, cam0 = randscene_v1(500_000, 1280, 720, ell=-3.6, seed=5, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])

# 1. a camera on an orbit: a new Camera (new view matrix) every frame
frames, t0, checked = 300, None, 0
for i in range(frames):
    a = 0.4 * math.sin(2 * math.pi * i / frames)
    Rm = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], device=dev, dtype=torch.float32)
    cam = Camera(R=Rm @ cam0.R, T=cam0.T, H=cam0.H, W=cam0.W, fx=cam0.fx, fy=cam0.fy, cx=cam0.cx, cy=cam0.cy)
    img = ms.render_gaussians(*g, cam, background_color=bg)
    if i % 50 == 7:
        assert torch.equal(img, stagewise(g, cam)), i
        checked += 1
    if i == 30:
        torch.cuda.synchronize(); t0 = time.perf_counter()
torch.cuda.synchronize()
print(f"orbit: {(time.perf_counter() - t0) / (frames - 31) * 1e3:.3f} ms/frame, {checked} frames checked, races {races()}", flush=True)

# 2. two image sizes in turn
cam_b = Camera(R=cam0.R, T=cam0.T, H=540, W=960, fx=cam0.fx * 0.75, fy=cam0.fy * 0.75, cx=480.0, cy=270.0)
refs = {id(cam0): stagewise(g, cam0), id(cam_b): stagewise(g, cam_b)}
for i in range(80):
    cam = cam0 if i % 2 == 0 else cam_b
    assert torch.equal(ms.render_gaussians(*g, cam, background_color=bg), refs[id(cam)]), i
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    ms.render_gaussians(*g, cam0 if i % 2 == 0 else cam_b, background_color=bg)
torch.cuda.synchronize()
print(f"two sizes in turn: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/frame, races {races()}", flush=True)

# 3. N creeps by 0.2 % per frame
R._BIN_CHOICE.clear()
n0 = 400_000
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    n = int(n0 * (1 + 0.002 * i))
    gi = tuple(t[:n] for t in g)
    img = ms.render_gaussians(*gi, cam0, background_color=bg)
    if i % 33 == 5:
        assert torch.equal(img, stagewise(gi, cam0)), i
torch.cuda.synchronize()
print(f"creeping N: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/frame, {len(R._BIN_CHOICE)} races", flush=True)

# 4. the same with a scene whose lazily sorted fronts fail (the lane must keep its fall-back to full sorts
#    although N changes from frame to frame)
R._BIN_CHOICE.clear()
sc4, cam4 = randscene_v1(240_000, 1280, 720, ell=-4.0, seed=3, device=dev)
g4 = (sc4["means3d"], sc4["scales"], sc4["quats"], sc4["opacities"], sc4["features"])
for i in range(30):
    ms.render_gaussians(*tuple(t[:200_000] for t in g4), cam4, background_color=bg)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    n = int(200_000 * (1 + 0.0005 * i))
    gi = tuple(t[:n] for t in g4)
    img = ms.render_gaussians(*gi, cam4, background_color=bg)
    if i % 33 == 5:
        assert torch.equal(img, stagewise(gi, cam4)), i
torch.cuda.synchronize()
print(f"creeping N, failing fronts: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/frame", flush=True)
