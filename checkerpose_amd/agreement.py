"""Accuracy contract of the reduced-precision (bf16) path, as numbers: how far its outputs are from a reference run of the
same network (the fp32 HIP path, which tests/test_gpu_parity.py pins to the CPU oracle at 1e-4; or the oracle itself).

The 1e-4 criterion of north_star cannot apply to bf16 storage (8 significant bits), and ADD(-S) is not measurable offline
(no datasets / checkpoints / PnP solver), so the bf16 path is judged on what the downstream PnP stage consumes
(test.py:294-329): the thresholded bits of all 13 logit rows, the final pixel ids, the two segmentation masks, plus a
bound on the logit error itself.  Used by tests (thresholds) and by bench.py (reported in the JSON line).
"""
import torch

ROWS = ("roi", "x5", "x4", "x3", "x2", "x1", "x0", "y5", "y4", "y3", "y2", "y1", "y0")   # MSB first (pipeline.py:72-82)
_Z0 = torch.tensor([0x33C00000], dtype=torch.int32).view(torch.float32)                  # include/checkerpose_hip.h


def _bit(z):
    return z > _Z0.to(z.device)


def logit_agreement(out, ref):
    """out / ref: 6-tuples (roi (B,1,N), x_bits (B,nx,N), y_bits (B,ny,N), seg (B,2,h,w), x_id, y_id) of the same forward.
    Returns a dict of plain floats (JSON-able)."""
    o = [t.detach().float().cpu() if t.dtype != torch.int64 else t.detach().cpu() for t in out]
    r = [t.detach().float().cpu() if t.dtype != torch.int64 else t.detach().cpu() for t in ref]
    zo, zr = torch.cat(o[:3], 1), torch.cat(r[:3], 1)                      # (B, 1+nx+ny, N)
    nx = o[1].shape[1]
    names = ["roi"] + ["x%d" % (nx - 1 - i) for i in range(nx)] + ["y%d" % (o[2].shape[1] - 1 - i) for i in range(o[2].shape[1])]
    agree = (_bit(zo) == _bit(zr)).float().mean(dim=(0, 2))
    d = (zo - zr).abs()
    rms = float(zr.pow(2).mean().sqrt())
    res = {
        "bit_agreement_per_row": {n: round(float(a), 5) for n, a in zip(names, agree)},
        "bit_agreement_min_row": round(float(agree.min()), 5),
        "bit_agreement_all_rows": round(float((_bit(zo) == _bit(zr)).float().mean()), 5),
        "x_id_equal": round(float((o[4] == r[4]).float().mean()), 5),
        "y_id_equal": round(float((o[5] == r[5]).float().mean()), 5),
        "xy_id_equal": round(float(((o[4] == r[4]) & (o[5] == r[5])).float().mean()), 5),
        "id_abs_err_mean_px": round(float(((o[4] - r[4]).abs() + (o[5] - r[5]).abs()).float().mean()), 4),
        "seg_agreement": round(float((_bit(o[3]) == _bit(r[3])).float().mean()), 5),
        "max_abs_dlogit": round(float(d.max()), 5),
        "mean_abs_dlogit": round(float(d.mean()), 6),
        "logit_rms": round(rms, 4),
        "mean_abs_dlogit_over_rms": round(float(d.mean()) / max(rms, 1e-30), 6),
        "max_abs_dseg": round(float((o[3] - r[3]).abs().max()), 5),
    }
    return res
