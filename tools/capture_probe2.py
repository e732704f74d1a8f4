"""probe 2: the engine's own capture calls (thread-local capture mode), 4 lanes, one event waited on by several lanes"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
lib = _abi.load()
dev = torch.device("cuda:0")
mode = sys.argv[1]
NL = int(sys.argv[2]); REPS = int(sys.argv[3])
bufs = [torch.zeros(1 << 18, device=dev) for _ in range(NL)]
streams = [torch.cuda.Stream(dev) for _ in range(NL)]
cur = torch.cuda.current_stream(dev)
streams[0].wait_stream(cur)
keep = []
_abi.check(lib.cp_graph_begin_capture(streams[0].cuda_stream))
def ev():
    keep.append(torch.cuda.Event()); return keep[-1]
e = ev(); e.record(streams[0])
for k in range(1, NL):
    streams[k].wait_event(e)
for rep in range(REPS):
    for k in range(NL):
        with torch.cuda.stream(streams[k]):
            bufs[k].add_(1.0)
    marks = []
    for k in range(NL):
        m = ev(); m.record(streams[k]); marks.append(m)
    for i in range(NL):
        for j in range(NL):
            if i == j:
                continue
            if mode == "multi":
                streams[i].wait_event(marks[j])            # one event, three waiters
            else:
                m = ev(); m.record(streams[j]); streams[i].wait_event(m)
            with torch.cuda.stream(streams[i]):
                bufs[i].add_(bufs[j], alpha=0.001)
for k in range(1, NL):
    e = ev(); e.record(streams[k]); streams[0].wait_event(e)
gx = C.c_void_p()
rc = lib.cp_graph_end_capture(streams[0].cuda_stream, C.byref(gx))
print("end capture rc", rc)
_abi.check(lib.cp_graph_launch(gx, cur.cuda_stream)); torch.cuda.synchronize()
print(mode, [float(b[0]) for b in bufs])
