#!/usr/bin/env python3
"""Where the bf16 path's logit error comes from on trained-like weights (GPU box): trains with the HIP training program
(deterministic mode), then runs ONE block group (backbone | decoder | gnn) in bf16 and the rest in fp32, teacher-forced
(checkerpose_amd/agreement.py: attribute_groups).  Prints one JSON object.
  python tools/bf16_attribution.py [--steps 300] [--held-out 8] [--seed 1] [--no-det]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--held-out", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-det", action="store_true")
    a = ap.parse_args()
    import checkerpose_amd
    from checkerpose_amd.trained_like import train_then_measure
    checkerpose_amd.set_deterministic(not a.no_det)
    r = train_then_measure(npoint=512, steps=a.steps, batch=a.batch, seed=a.seed, held_out=a.held_out, attribution=True,
                           log=lambda s: print(s, file=sys.stderr, flush=True))
    print(json.dumps(r))
