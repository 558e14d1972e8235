"""The sharded TRAINING step on the HIP path with a real process group: ranks share the one GPU over gloo (RCCL refuses two
ranks per device).  Every rank renders and differentiates its band (ms_render_fwd with render_alphas over the band's rows,
ms_render_bwd_rows), the per-Gaussian gradient rows are all-reduced, the backward projection runs on the sum: image and
gradients must equal the single-GPU step's (render_gaussians_trainable) -- image bit for bit, gradients within the order of
the float atomics.    python -m torch.distributed.run --nproc-per-node 2 scripts/sharded_train_gloo_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.distributed as dist
from helpers import grad_stats
from mojosplat_amd.autograd import render_gaussians_trainable
from mojosplat_amd.distributed import render_gaussians_trainable_sharded
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
names = ("means3d", "scales", "quats", "opacities", "features")
try:
    for (N, W, H, ell, ts) in ((60_000, 640, 360, -3.2, 16), (60_000, 640, 360, -3.2, 32), (4_000, 320, 200, -2.0, 16)):
        sc, cam = randscene_v1(N, W, H, ell=ell, seed=21, device=dev)
        bg = torch.tensor(BACKGROUND_V1, device=dev).requires_grad_(True)
        v_img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)
        ref_leaves = [sc[k].clone().requires_grad_(True) for k in names]
        ref = render_gaussians_trainable(*ref_leaves, cam, background_color=bg, tile_size=ts)
        ref.backward(v_img)
        ref_bg = bg.grad.clone(); bg.grad = None
        for rep in range(2):    # (the second step runs sync-free on the buffers the first one sized)
            leaves = [sc[k].clone().requires_grad_(True) for k in names]
            img = render_gaussians_trainable_sharded(*leaves, cam, background_color=bg, tile_size=ts)
            img.backward(v_img)
            assert torch.equal(img.detach(), ref.detach()), f"N={N} ts={ts} step {rep}: the sharded image differs"
            for name, a, b in zip(names, leaves, ref_leaves):
                st = grad_stats(a.grad, b.grad)
                assert st["max_norm_err"] <= 1e-4 and st["elem_rel_p999"] <= 1e-3, f"N={N} ts={ts} step {rep} {name}: {st}"
            st = grad_stats(bg.grad, ref_bg)
            assert st["max_norm_err"] <= 1e-4, f"background: {st}"
            bg.grad = None
    # an empty frame: zeros image, zero gradients, on every rank
    sc, cam = randscene_v1(500, 320, 200, ell=-3.0, seed=2, device=dev)
    leaves = [sc[k].clone().requires_grad_(True) for k in names]
    leaves[0] = (sc["means3d"] + torch.tensor([0.0, 0.0, 500.0], device=dev)).requires_grad_(True)
    img = render_gaussians_trainable_sharded(*leaves, cam, background_color=torch.tensor(BACKGROUND_V1, device=dev))
    assert (img == 0).all()
    img.sum().backward()
    assert all((l.grad == 0).all() for l in leaves)
    torch.cuda.synchronize()
    print(f"rank {rank}/{world}: sharded training steps equal the single-GPU step", flush=True)
finally:
    dist.destroy_process_group()
