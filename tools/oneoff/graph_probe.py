"""Does a hipGraph of one dycore step beat the plain launches?  (one-off probe; python tools/oneoff/graph_probe.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    c, d, m = modules.make_supercell(400, 400, 100, 1, 2e5, 2e5, 2e4)
    dt = d.compute_time_step(c)
    for _ in range(5):
        d.time_step(c, dt)
    s.synchronize()
    def timed(fn, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n):
            fn()
        e1.record(s); s.synchronize()
        return e0.elapsed_time(e1) / n
    print("plain  %.4f ms/step" % timed(lambda: d.time_step(c, dt)), flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            d.time_step(c, dt)
        for _ in range(3):
            g.replay()
        s.synchronize()
        print("graph  %.4f ms/step" % timed(lambda: g.replay()), flush=True)
        print("plain  %.4f ms/step" % timed(lambda: d.time_step(c, dt)), flush=True)
        print("graph  %.4f ms/step" % timed(lambda: g.replay()), flush=True)
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e)[:300])
