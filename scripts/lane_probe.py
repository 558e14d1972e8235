"""Per pair of torch streams: (a) a small spin kernel launched behind a WIDE spin kernel on the other stream -- does it run
under it (time ~ the wide kernel's) or after it (~ the sum)? -- (b) two single-thread spin kernels (what
_fused._lane_streams calibrates with), (c) the renderer's two-lane loop on that pair (ms a frame at config 3).
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/ubench/lane_probe.hip -o /tmp/liblaneprobe.so && python scripts/lane_probe.py"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mojosplat_amd import _fused
from mojosplat_amd.scenes import randscene_v1, BACKGROUND_V1
from bench import WORKLOADS

lib = ctypes.CDLL(os.environ.get("LANE_PROBE_LIB", "/tmp/liblaneprobe.so"))
lib.lane_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
pool = [torch.cuda.Stream(dev) for _ in range(8)]
N, W, H, ell, fp16 = WORKLOADS["cfg3"]
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
imgs = [torch.empty((H, W, 3), device=dev), torch.empty((H, W, 3), device=dev)]

def launch(s, blocks, spins, lds):
    rc = lib.lane_probe_launch(ctypes.c_void_p(s.cuda_stream), blocks, spins, lds, ctypes.c_void_p(sink.data_ptr()))
    assert rc == 0, rc

def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) * 1e6
        best = t if best is None else min(best, t)
    return best

WIDE = (16384, 6, 40 * 1024)    # blocks, spins, LDS bytes: ~4 workgroups a CU
SMALL = (256, 60, 0)
for s in pool:                    # first use, in index order
    launch(s, 64, 1, 0)
torch.cuda.synchronize()
wide_us = timed(lambda: launch(pool[0], *WIDE))
small_us = timed(lambda: launch(pool[0], *SMALL))

def run_b(n, S):
    prev = None
    for k in range(n):
        l = k & 1
        with torch.cuda.stream(S[l]):
            f = _fused.render_begin_hip(*g, cam, bg, 32, out=imgs[l], lane=1 + l)
        if prev is not None:
            prev.finish()
        prev = f
    prev.finish()

out = {"wide_alone_us": round(wide_us, 1), "small_alone_us": round(small_us, 1), "pairs": {}}
pairs = [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (0, 4), (3, 6), (2, 5), (1, 6)]
for a, b in pairs:
    under = timed(lambda: (launch(pool[a], *WIDE), launch(pool[b], *SMALL)))
    def two_sleeps():
        with torch.cuda.stream(pool[a]): torch.cuda._sleep(150000)
        with torch.cuda.stream(pool[b]): torch.cuda._sleep(150000)
    def one_sleep():
        with torch.cuda.stream(pool[a]): torch.cuda._sleep(150000)
    sleeps = timed(two_sleeps) / timed(one_sleep)
    run_b(60, [pool[a], pool[b]])
    torch.cuda.synchronize(); t0 = time.perf_counter(); run_b(200, [pool[a], pool[b]]); torch.cuda.synchronize()
    out["pairs"][f"{a},{b}"] = {"small_behind_wide_us": round(under, 1), "sleep_pair_over_solo": round(sleeps, 2),
                                "renderer_ms": round((time.perf_counter() - t0) / 200 * 1e3, 4)}
print(json.dumps(out))
