"""Post-process `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- ./gather_calib`:
known bytes / (FETCH_SIZE counter in KB * 1024) per kernel = the factor to multiply FETCH_SIZE by for that pattern.
python scripts/ubench/gather_calib.py <dir with *counter_collection.csv> '<json line the program printed>'"""
import csv, glob, json, sys
known = json.loads(sys.argv[2])
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
out = {}
for name in ("k_stream", "k_rec48", "k_aos36"):
    v = sorted(float(r["Counter_Value"]) for r in rows if name in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE")
    if v:
        med = v[len(v) // 2]
        out[name] = {"known_MB": round(known[name] / 1e6, 1), "FETCH_SIZE_KB": med, "factor": round(known[name] / (med * 1024), 3)}
print(json.dumps(out))
