"""Soak of the launched forward: many hipGraph replays per batch size (and a stretch of training steps between them), every
replay's outputs compared BITWISE with the first of its batch size, the device status word (cp_device_status: the pipelined
64 x 64 chain's hand-over / staging time-outs) read at the end of every stretch.  Prints one JSON object; exits 1 on any
difference or status bit.

    python tools/soak.py [seconds per batch size, default 40] [batch sizes, default 256,1,8,64,40]
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from checkerpose_amd import _abi                                   # noqa: E402
from checkerpose_amd.synthetic import build_net, det_image         # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
BATCHES = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 1, 8, 64, 40]
dev = torch.device("cuda:0")
net = build_net(512, seed=1).to(dev).set_compute_dtype("bf16")
res = {"seconds_per_batch": SECONDS, "stretches": []}
bad = 0
for B in BATCHES:
    img = det_image(B, seed=B).to(dev)
    first = [t.clone() for t in net(img, None)]
    torch.cuda.synchronize()
    n, diff, t0, last = 0, 0, time.time(), time.time()
    while time.time() - t0 < SECONDS:
        for _ in range(50):
            out = net(img, None)
        n += 50
        diff += int(any(not torch.equal(a, b) for a, b in zip(out, first)))       # synchronises: the 50th replay of the stretch
        if time.time() - last > 30:
            print("B=%d: %d replays, %d differing" % (B, n, diff), file=sys.stderr, flush=True)
            last = time.time()
    torch.cuda.synchronize()
    word = _abi.device_status(clear=True)
    res["stretches"].append({"batch": B, "replays": n, "compared": n // 50, "differing": diff, "status_word": word,
                             "ms_per_replay": round((time.time() - t0) / n * 1e3, 3)})
    bad += diff + int(word != 0)
# a stretch of training steps on the same device (other programs, the shared weight-gradient arena), then the first batch again
B = BATCHES[0]
from checkerpose_amd.trained_like import train_net                 # noqa: E402
t0 = time.time()
tnet, _pos, _gen, losses = train_net(npoint=512, steps=30, batch=32, seed=3, device=dev)
torch.cuda.synchronize()
res["train_stretch"] = {"steps": 30, "batch": 32, "seconds": round(time.time() - t0, 2), "status_word": _abi.device_status(clear=True),
                        "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)]}
img = det_image(B, seed=B).to(dev)
ref = [t.clone() for t in net(img, None)]
out = net(img, None)
again = {"batch": B, "equal_after_training_stretch": all(torch.equal(a, b) for a, b in zip(out, ref)), "status_word": _abi.device_status(clear=True)}
res["after_training"] = again
bad += int(not again["equal_after_training_stretch"]) + int(again["status_word"] != 0) + int(res["train_stretch"]["status_word"] != 0)
res["ok"] = bad == 0
print(json.dumps(res))
sys.exit(0 if bad == 0 else 1)
