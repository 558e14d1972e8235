"""Build libmojosplat_hip.so for gfx950 with hipcc (in-tree, no torch, no cmake).

    python -m mojosplat_amd.csrc.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "project.hip", "binning.hip", "rasterize.hip", "rasterize_bwd.hip", "rasterize_bwdq.hip",
           "project_bwd.hip", "pipeline.hip", "sh.hip"]
# per-source flags.  rasterize.hip: fp32 denormals flushed -- its blend loop selects by underflow
# (ms::kFlushK in ms_common.hpp; scripts/ubench/flush_select.hip)
SOURCE_FLAGS = {"rasterize.hip": ["-fgpu-flush-denormals-to-zero"], "rasterize_bwdq.hip": ["-fgpu-flush-denormals-to-zero"]}
HEADERS = ["ms_common.hpp", os.path.join("..", "..", "include", "mojosplat_hip.h")]
LIB = os.path.join(HERE, "libmojosplat_hip.so")
ARCH = "gfx950"


def _sources():
    return [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = _sources() + [os.path.join(HERE, h) for h in HEADERS] + \
        [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag=True builds libmojosplat_hip_diag.so with -DMS_DIAG (per-wave stamps in the rasteriser;
    scripts/raster_waves.py) next to the product library -- never loaded unless MOJOSPLAT_HIP_LIB names it."""
    # MS_VARIANT=name MS_EXTRA_FLAGS="-D..." builds libmojosplat_hip_<name>.so (kernel-tuning experiments)
    variant, extra = os.environ.get("MS_VARIANT"), os.environ.get("MS_EXTRA_FLAGS", "").split()
    lib = LIB.replace(".so", "_diag.so") if diag else LIB
    if variant:
        lib = LIB.replace(".so", f"_{variant}.so")
    if not diag and not variant and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.splitext(src)[0] + (f".{variant}.o" if variant else ".diag.o" if diag else ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-fno-slp-vectorize"] + SOURCE_FLAGS.get(os.path.basename(src), []) + \
            (["-DMS_DIAG"] if diag else []) + extra
        if verbose:
            cmd += ["-Rpass-analysis=kernel-resource-usage"]
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib + ".tmp"] + objs
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, diag="--diag" in sys.argv))
