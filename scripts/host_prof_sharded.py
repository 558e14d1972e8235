"""cProfile of the host side of the asynchronous sharded entry point (one rank's band, rehearsed on one GPU):
    python scripts/host_prof_sharded.py [cfg3] [world] [rank]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from mojosplat_amd.distributed import render_gaussians_sharded
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
N, W, H, ell, fp16 = WORKLOADS[name]
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])


def pipelined(n):
    cur = None
    for _ in range(n):
        nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True, rehearse=(rank, world))
        if cur is not None:
            cur.wait()
        cur = nxt
    cur.wait()


pipelined(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipelined(1000)
torch.cuda.synchronize()
print("pipelined us/frame", round((time.perf_counter() - t0) / 1000 * 1e6, 1))
pr = cProfile.Profile()
pr.enable()
pipelined(1000)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
