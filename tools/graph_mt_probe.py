"""Small-batch forward: (1) the GPU-only span of one hipGraph replay (host enqueue hidden behind a sleep kernel), (2) whether two host
threads enqueue two graphs on two streams in parallel (hipGraphLaunch host cost is ~3 us per node and bounds the B <= 8 forward)."""
import sys
import threading
import time
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
from checkerpose_amd.synthetic import build_net, det_image

dev = torch.device("cuda:0")
lib = _abi.load()


def warm(net, B):
    img = det_image(B, seed=3).to(dev)
    with torch.no_grad():
        for _ in range(3):
            net(img, None)
    buf = net.input_buffer(B)
    buf.copy_(img)
    with torch.no_grad():
        for _ in range(5):
            net(buf, None)
    torch.cuda.synchronize()
    return buf


# calibrate the sleep kernel
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10000000 / e0.elapsed_time(e1)
print("sleep: %.0f cycles per ms" % cyc_per_ms, flush=True)

nets = [build_net(npoint=512, seed=1).to(dev).eval() for _ in range(2)]
for n in nets:
    n.set_compute_dtype("bf16")
for B in (1, 8):
    bufs = [warm(n, B) for n in nets]
    # (1) GPU-only span
    spans = []
    with torch.no_grad():
        for _ in range(10):
            torch.cuda.synchronize()
            torch.cuda._sleep(int(3 * cyc_per_ms))
            e0.record()
            nets[0](bufs[0], None)
            e1.record()
            torch.cuda.synchronize()
            spans.append(e0.elapsed_time(e1))
    print("B=%d: GPU-only span of one replay (enqueued behind a 3 ms sleep): min %.3f med %.3f ms" % (B, min(spans), sorted(spans)[5]), flush=True)
    # (2) two threads, two graphs, two streams
    streams = [torch.cuda.Stream(dev) for _ in range(2)]

    def loop(k, n, out):
        with torch.no_grad(), torch.cuda.stream(streams[k]):
            t0 = time.perf_counter()
            for _ in range(n):
                nets[k](bufs[k], None)
            out[k] = time.perf_counter() - t0

    for mode in ("one thread, two streams", "two threads, two streams"):
        out = [0, 0]
        n = 200
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode.startswith("one"):
            with torch.no_grad():
                for _ in range(n):
                    for k in range(2):
                        with torch.cuda.stream(streams[k]):
                            nets[k](bufs[k], None)
            host = time.perf_counter() - t0
        else:
            th = [threading.Thread(target=loop, args=(k, n, out)) for k in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            host = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        print("B=%d %s: %d forwards each: host %.3f ms per PAIR, done %.3f ms per pair (%.0f crops/s)" % (B, mode, n, host / n * 1e3, tot / n * 1e3,
                                                                                                        2 * B * n / tot), flush=True)
    # (3) raw hipGraphLaunch host cost (no Python around it) and the Python share of net(img)
    pr = [p for k, p in nets[0]._programs.items() if k[0] == B][0]
    g, cur = pr["graph"][0], torch.cuda.current_stream(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        lib.cp_graph_launch(g, cur.cuda_stream)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("B=%d raw cp_graph_launch x200: host %.3f ms each, done %.3f ms each; nodes %d" % (B, host / 200 * 1e3, tot / 200 * 1e3, len(pr["prog"].calls)), flush=True)
    prs = [[p for k, p in n._programs.items() if k[0] == B][0] for n in nets]

    def rawloop(k, n):
        for _ in range(n):
            lib.cp_graph_launch(prs[k]["graph"][0], streams[k].cuda_stream)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=rawloop, args=(k, 200)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("B=%d raw launches from two threads: host %.3f ms per pair, done %.3f ms per pair" % (B, host / 200 * 1e3, tot / 200 * 1e3), flush=True)
