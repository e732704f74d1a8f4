#!/usr/bin/env python3
"""bf16 accuracy contract on trained-like weights (GPU box): N steps of the HIP training program on the synthetic translation task of
checkerpose_amd/trained_like.py, then bf16 vs fp32 agreement on held-out crops.  Prints one JSON object.
  python tools/trained_like.py [--steps 300] [--batch 32] [--lr 5e-4] [--npoint 512]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--npoint", type=int, default=512)
    ap.add_argument("--held-out", type=int, default=8)
    a = ap.parse_args()
    from checkerpose_amd.trained_like import train_then_measure
    r = train_then_measure(npoint=a.npoint, steps=a.steps, batch=a.batch, lr=a.lr, held_out=a.held_out, log=lambda s: print(s, file=sys.stderr, flush=True))
    print(json.dumps(r))
