"""Round 6, review item 7: where a SYNCHRONISED training step's host time goes (config 3, forward + backward, one
torch.cuda.synchronize() per step).  Wall-clock stamps around the three library calls of a step and the two ends of the
step; medians over the steps.

    python scripts/train_host_timeline.py [steps]
"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd import _hip, autograd as ag  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


class Tap:
    """A ctypes function with wall-clock stamps around every call."""

    def __init__(self, fn, log, name):
        self.fn, self.log, self.name = fn, log, name

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        self.log.append((self.name, t0, time.perf_counter()))
        return r


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    sc, cam = randscene_v1(1_000_000, 1920, 1080, ell=-4.0, seed=42, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    leaves = [t.float().clone().requires_grad_(True) for t in g]
    v_img = torch.rand(1080, 1920, 3, generator=torch.Generator().manual_seed(43)).to(dev)
    L = _hip.lib()
    log = []

    class Lib:   # the library with three of its entry points tapped
        def __getattr__(self, k):
            return getattr(L, k)
    lib = Lib()
    for name in ("ms_render_fwd", "ms_render_bwd_rows", "ms_render_bwd_finish"):
        setattr(lib, name, Tap(getattr(L, name), log, name))
    orig = _hip.lib
    _hip.lib = lambda: lib

    def step():
        for l in leaves:
            l.grad = None
        img = ag.render_gaussians_trainable(*leaves, cam, background_color=bg)
        t_fwd = time.perf_counter()
        img.backward(v_img)
        return t_fwd

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    rows = []
    for _ in range(steps):
        del log[:]
        t0 = time.perf_counter()
        t_fwd = step()
        t_enq = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        d = {n: (a, b) for n, a, b in log}
        if len(d) < 3:
            continue
        rows.append({
            "step_start_to_fwd_call": d["ms_render_fwd"][0] - t0,
            "fwd_call (enqueue + wait for the size record)": d["ms_render_fwd"][1] - d["ms_render_fwd"][0],
            "fwd_call_return_to_apply_return": t_fwd - d["ms_render_fwd"][1],
            "backward()_to_bwd_rows_call (engine hop, allocations)": d["ms_render_bwd_rows"][0] - t_fwd,
            "bwd_rows_call": d["ms_render_bwd_rows"][1] - d["ms_render_bwd_rows"][0],
            "between_bwd_calls": d["ms_render_bwd_finish"][0] - d["ms_render_bwd_rows"][1],
            "bwd_finish_call": d["ms_render_bwd_finish"][1] - d["ms_render_bwd_finish"][0],
            "bwd_finish_return_to_backward()_return": t_enq - d["ms_render_bwd_finish"][1],
            "synchronize (GPU still busy)": t1 - t_enq,
            "step": t1 - t0,
        })
    _hip.lib = orig
    out = {k: round(statistics.median(r[k] for r in rows) * 1e6, 1) for k in rows[0]}
    # streamed, for the difference
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    out["streamed_step"] = round((time.perf_counter() - t0) / steps * 1e6, 1)
    print(json.dumps({"what": "config 3 training step, host wall clock, medians in us", "steps": len(rows), **out}))


if __name__ == "__main__":
    main()
