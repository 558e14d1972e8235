"""Attribute a kernel's instructions to source lines (static count) from a `hipcc -S -gline-tables-only` listing.

    python scripts/isa_by_line.py <file.s> <kernel-substring> [--top N] [--src path] [--ranges a-b:name,c-d:name,...]

Groups by the (file, line) of the innermost `.loc` (inlined callees are attributed to THEIR lines) and, with --ranges,
into named line ranges of the main source file.  Classes: valu (v_*), salu (s_* except s_waitcnt / s_nop / branches),
lds (ds_*), vmem (global_/buffer_/scratch_/flat_), branch."""
import re
import sys
from collections import Counter, defaultdict


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    ranges = []
    if "--ranges" in sys.argv:
        for item in sys.argv[sys.argv.index("--ranges") + 1].split(","):
            r, name = item.split(":")
            a, b = r.split("-")
            ranges.append((int(a), int(b), name))
    files = {}
    per_line = defaultdict(Counter)
    inside = False
    cur = (None, 0)
    main_file = None
    for line in open(path):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
            continue
        if not inside:
            if re.match(r"^[_A-Za-z0-9$.]+:", line) and kern in line.split(":")[0] and not line.startswith(".L"):
                inside = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)\b", line)
        if m and not line.strip().startswith((".", ";")):
            per_line[cur][classify(m.group(1))] += 1
    tot = Counter()
    for c in per_line.values():
        tot.update(c)
    print("kernel:", kern, " totals:", dict(tot))
    by_file = defaultdict(Counter)
    for (f, l), c in per_line.items():
        by_file[files.get(f, str(f))].update(c)
    for f, c in sorted(by_file.items(), key=lambda kv: -kv[1]["valu"]):
        print(f"  {f.split('/')[-1]:28s} valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}")
    if ranges:
        main_file = max(by_file.items(), key=lambda kv: kv[1]["valu"])[0] if "--src" not in sys.argv else sys.argv[sys.argv.index("--src") + 1]
        grp = defaultdict(Counter)
        for (f, l), c in per_line.items():
            fn = files.get(f, str(f))
            name = fn.split("/")[-1]
            if fn.endswith(main_file.split("/")[-1]):
                name = next((n for a, b, n in ranges if a <= l <= b), f"{name}:other")
            grp[name].update(c)
        print("  -- by range")
        for n, c in sorted(grp.items(), key=lambda kv: -kv[1]["valu"]):
            print(f"  {n:40s} valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}")
    print("  -- top lines by valu")
    for (f, l), c in sorted(per_line.items(), key=lambda kv: -kv[1]["valu"])[:top]:
        print(f"  {files.get(f, str(f)).split('/')[-1]}:{l:<6d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}")


if __name__ == "__main__":
    main()
