"""Does a producer -> consumer working set that fits the 256 MB Infinity Cache (memory-side, MALL) move faster than HBM?
Streams y = x (torch copy, 8 B / lane) over buffers of growing size, many repetitions: the second and later passes of a small
buffer hit in the cache.  usage (GPU box): python tools/mall_probe.py"""
import json
import torch

res = {}
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 8
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    reps = max(4, 8192 // mb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res[mb] = {"buffer_MB_each": mb, "ms": ms, "TBps_read_plus_write": 2 * mb * 1.048576e6 / ms / 1e9}
    print("x,y = %5d MB each: %.4f ms  %.2f TB/s (read + write)" % (mb, ms, res[mb]["TBps_read_plus_write"]))
    del x, y
print(json.dumps(res))
