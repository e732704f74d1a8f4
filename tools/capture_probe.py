"""probe: does hipGraph stream capture survive two captured streams waiting on each other's events?"""
import sys
import torch
dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(1 << 20, device=dev)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
mode = sys.argv[1]
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s0):
    e0 = torch.cuda.Event(); e0.record(s0); s1.wait_event(e0)          # fork
    with torch.cuda.stream(s0):
        a.add_(1)
    with torch.cuda.stream(s1):
        b.add_(2)
    if mode == "mutual":
        m0, m1 = torch.cuda.Event(), torch.cuda.Event()
        m0.record(s0); m1.record(s1)
        s1.wait_event(m0); s0.wait_event(m1)                               # each lane waits for the other's first kernel
        with torch.cuda.stream(s0):
            a.add_(b)
        with torch.cuda.stream(s1):
            b.mul_(1.0)
    ej = torch.cuda.Event(); ej.record(s1); s0.wait_event(ej)          # join
g.replay(); g.replay()
torch.cuda.synchronize()
print(mode, float(a[0]), float(b[0]))
