"""Which pairs of torch streams run concurrently on this device?  Two spin kernels (torch.cuda._sleep) on two streams take
one kernel's time if the streams sit on different hardware queues, two if they share one.
python scripts/stream_overlap_probe.py [n_streams]"""
import json, sys, time
import torch
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.Stream(dev) for _ in range(n)] + [torch.cuda.Stream(dev, priority=-1) for _ in range(2)]
cycles = 400000
def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
def one(s):
    with torch.cuda.stream(s):
        torch.cuda._sleep(cycles)
for s in streams:
    one(s)
solo = min(timed(lambda: one(streams[0])) for _ in range(5))
table = []
for i in range(len(streams)):
    row = []
    for j in range(len(streams)):
        if j <= i:
            row.append(None); continue
        t = min(timed(lambda: (one(streams[i]), one(streams[j]))) for _ in range(3))
        row.append(round(t / solo, 2))
    table.append(row)
print(json.dumps({"solo_us": round(solo, 1), "handles": [hex(s.cuda_stream) for s in streams],
                  "pair_time_over_solo (1 = concurrent, 2 = serialised)": table}))
