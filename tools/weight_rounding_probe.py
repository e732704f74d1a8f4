#!/usr/bin/env python3
"""How much of the bf16 path's logit error is WEIGHT rounding (a fixed perturbation of the network) and how much activation storage:
the trained-like network (deterministic training) runs in fp32 with the conv / linear weights of one group rounded to bf16 (or to
IEEE half) on the host, teacher-forced, against the unrounded fp32 run.  Groups: the backbone as a whole and by segment, the decoder,
the keypoint side.  Prints one JSON object (mean / rms / max |dlogit| per group and format).
  python tools/weight_rounding_probe.py [--steps 300] [--seed 1] [--held-out 8]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BB = "init_net.img_backbone."
GROUPS = {
    "backbone": lambda k: k.startswith(BB),
    "backbone.stem+layer1+transition1": lambda k: k.startswith((BB + "conv1", BB + "conv2", BB + "layer1", BB + "transition1")),
    "backbone.stage2": lambda k: k.startswith((BB + "stage2", BB + "transition2")),
    "backbone.stage3": lambda k: k.startswith((BB + "stage3", BB + "transition3")),
    "backbone.stage4": lambda k: k.startswith(BB + "stage4"),
    "backbone.incre_modules": lambda k: k.startswith(BB + "incre_modules"),
    "decoder": lambda k: k.startswith(("up_net", "seg_block")),
    "gnn": lambda k: not k.startswith((BB, "up_net", "seg_block")),
    "all": lambda k: True,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--held-out", type=int, default=8)
    a = ap.parse_args()
    import checkerpose_amd
    from checkerpose_amd.trained_like import make_batch, train_net
    checkerpose_amd.set_deterministic(True)
    dev = torch.device("cuda")
    net, pos, gen, losses = train_net(512, a.steps, 32, 5e-4, a.seed, dev, None)
    checkerpose_amd.set_deterministic(False)
    res = {"steps": a.steps, "seed": a.seed, "groups": {}}
    with torch.no_grad():
        net.eval()
        net.set_kernel_selection("per_crop")
        net.set_compute_dtype("fp32")
        net.clone_outputs = True
        img = make_batch(a.held_out, pos, gen, dev)[0]
        ref = [t.clone() for t in net(img, None)]
        zr = torch.cat(ref[:3], 1)
        t = torch.zeros(a.held_out, 13, 512, device=dev)
        t[:, 0:1], t[:, 1:7], t[:, 7:13] = ref[0], ref[1], ref[2]
        res["logit_rms"] = round(float(zr.pow(2).mean().sqrt()), 4)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        for fmt, tdt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            for name, sel in GROUPS.items():
                if fmt == "f16" and name not in ("backbone", "all"):
                    continue
                sd = {}
                n = 0
                for k, v in sd0.items():
                    if sel(k) and k.endswith("weight") and v.dim() >= 2:      # conv / linear weights (BatchNorm vectors stay fp32 in every path)
                        sd[k] = v.to(tdt).to(v.dtype)
                        n += v.numel()
                    else:
                        sd[k] = v
                net.load_state_dict(sd)
                out = net.forward_teacher_forced(img, t)
                d = (torch.cat([x.float() for x in out[:3]], 1) - zr).abs()
                res["groups"]["%s:%s" % (fmt, name)] = {"weights_rounded": n, "mean_abs_dlogit": round(float(d.mean()), 6),
                                                        "rms_dlogit": round(float(d.pow(2).mean().sqrt()), 6), "max_abs_dlogit": round(float(d.max()), 5)}
                print(fmt, name, res["groups"]["%s:%s" % (fmt, name)], file=sys.stderr, flush=True)
        net.load_state_dict(sd0)
    print(json.dumps(res))
