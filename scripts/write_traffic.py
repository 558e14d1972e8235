"""The ONLY writer of profiles/traffic.json's measured fields.

On the GPU box (scripts/collect_profiles.sh):  python scripts/write_traffic.py measure <fetch.txt> <write.txt> <out.json>
    parses the pmc_summary outputs of the FETCH_SIZE / WRITE_SIZE passes and writes the per-kernel figures together with
    bench.raster_sha16() of the sources the measured library was built from.
In the build container:  python scripts/write_traffic.py install <measured.json> [source-file-name-under-profiles]
    refuses unless the hash of the CURRENT rasterize.hip + headers equals the measured one, then records the figures, the
    hash and the current commit under "cfg3" of profiles/traffic.json (history entries are kept)."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _median(path, kernel, counter):
    for line in open(path):
        if line.startswith(kernel) and f" {counter} " in line:
            return float(re.search(r"median\s+([0-9.]+)", line).group(1))
    raise SystemExit(f"{kernel} / {counter} not in {path}")


def measure(fetch, write, out):
    import bench
    k = {"rasterize_fwd": "k_rasterize_fwd", "isect_scatter": "k_isect_scatter", "project": "k_project_hist"}
    rec = {"raster_sha16": bench.raster_sha16()}
    for name, kern in k.items():
        rec[f"{name}_fetch_kb_raw"] = int(_median(fetch, kern, "FETCH_SIZE"))
        rec[f"{name}_write_kb"] = int(_median(write, kern, "WRITE_SIZE"))
    # the rasteriser's 64-byte gather requests are counted at face value (profiles/r02_fetch_calibration.md)
    rec["rasterize_fwd_bytes"] = int((rec["rasterize_fwd_fetch_kb_raw"] + rec["rasterize_fwd_write_kb"]) * 1024)
    json.dump(rec, open(out, "w"), indent=1)
    print(rec)


def install(measured, source):
    import bench
    rec = json.load(open(measured))
    have = bench.raster_sha16()
    if rec["raster_sha16"] != have:
        raise SystemExit(f"measured on rasterize.hip + headers {rec['raster_sha16']}, the tree holds {have}: measure again")
    path = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(path))
    old = {k: v for k, v in doc.get("cfg3", {}).items() if not isinstance(v, dict) and k not in ("how",)}
    hist = {k: v for k, v in doc.get("cfg3", {}).items() if isinstance(v, dict)}
    hist["before_" + rec["raster_sha16"]] = old
    rec["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
    rec["source"] = source
    rec["how"] = ("scripts/collect_profiles.sh: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --steps 5 --warmup 2 "
                  "--no-cpu-baseline --no-verify --no-extras, median over the launches; written by scripts/write_traffic.py only")
    doc["cfg3"] = {**rec, **hist}
    json.dump(doc, open(path, "w"), indent=2)
    print("installed", rec)


if __name__ == "__main__":
    if sys.argv[1] == "measure":
        measure(*sys.argv[2:5])
    else:
        install(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "traffic.json")
