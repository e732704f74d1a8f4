"""Is the run-to-run spread of the B = 256 forward (10.3 vs 10.9 ms) a property of the process, of the graph instantiation, or drift?
One process: N fresh networks (each captures its own hipGraph), each timed 3 x 40 steps."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    wl = bench.make_workload("lmo_ape", 512, B, "bf16", dev, 0, 1)
    step = wl["step"]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print("instantiation %d: %s ms" % (it, " ".join("%.3f" % t for t in ts)), flush=True)
    del wl, step
    torch.cuda.empty_cache()
