import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.scenes import randscene_v1
dev = torch.device("cuda:0")
for (N, W, H, ell, px) in [(1_000_000, 1920, 1080, -4.0, 32), (1_000_000, 1920, 1080, -3.0, 64), (500_000, 1280, 720, -3.5, 32), (500_000, 1280, 720, -3.0, 64)]:
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=int(os.environ.get('SEED', 42)), device=dev)
    for mode in ("0", "2"):
        _hip.config_depth_cut(int(mode))
        _fused._state.clear(); _fused.FRAME_STATS = {}
        for _ in range(5):
            ms.render_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"], cam, bin_size=px)
        torch.cuda.synchronize()
        st = _fused._dev_state(dev, 0)
        print(json.dumps({"scene": [N, W, H, ell, px], "mode": mode, "stats": _fused.FRAME_STATS, "pairs": int(st["host_np"][0]), "heavy": int(st["host_np"][2] + st["host_np"][3] + st["host_np"][4])}))
        _fused.FRAME_STATS = None
