"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (numpy, explicit formulas -- no autograd) of the training-side ops of SURVEY.md 8f row N1:
the loss head of train.py:307-320 and the backward of the two fused graph ops.  Only tests/ may import this
file.  PINNED: tests/golden/make_golden_train.py ran the REFERENCE's own modules (losses/code_loss.py,
losses/mask_loss.py, model/init.py StaticGraph_module, model/pipeline.py Index2Feat_module) with torch autograd
in the build container; tests/test_oracle_train.py re-checks every function below against those vectors.
Line numbers are relative to /root/reference/checkerpose.
"""
import numpy as np


def _sigmoid(z):
    z = z.astype(np.float64)
    return 1.0 / (1.0 + np.exp(-z))


def code_loss(pred, gt, mask=None, loss_type="BCE"):
    """UnmaskedCodeLoss.forward (losses/code_loss.py:17-27; mask None) / MaskedCodeLoss.forward (:43-62).
    pred, gt (B,nb,N); mask (B,1,N).  Returns (loss, d loss / d pred) in float64."""
    z = pred.astype(np.float64)
    y = gt.astype(np.float64)
    s = _sigmoid(pred)
    if loss_type == "BCE":        # nn.BCEWithLogitsLoss(reduction="none"), code_loss.py:11,35
        raw = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
        d = s - y
    elif loss_type == "L1":       # nn.L1Loss on sigmoid(pred), code_loss.py:23-24,53-54
        raw = np.abs(s - y)
        d = np.sign(s - y) * s * (1 - s)
    else:
        raise ValueError(loss_type)
    if mask is None:              # reduction="mean"
        denom = float(z.size)
        return raw.sum() / denom, d / denom
    m = mask.astype(np.float64)
    denom = max(m.sum(), 1.0) * z.shape[1]     # code_loss.py:59-60
    return (raw * m).sum() / denom, d * m / denom


def masked_ce_loss(pred, gt_class, mask):
    """MaskedCodeLoss(loss_type="CE").forward (losses/code_loss.py:36-37,47-61): nn.CrossEntropyLoss(reduction="none") over the
    class axis of pred (B,C,N) against class ids gt_class (B,1,N), times mask (B,1,N), / clamp(mask.sum(), 1) (num_bits = 1).
    Returns (loss, d loss / d pred) in float64."""
    z = pred.astype(np.float64)
    y = gt_class[:, 0, :].astype(np.int64)
    m = mask[:, 0, :].astype(np.float64)
    zs = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(zs).sum(axis=1))
    sm = np.exp(zs - lse[:, None, :])
    picked = np.take_along_axis(zs, y[:, None, :], axis=1)[:, 0, :]
    denom = max(m.sum(), 1.0)
    onehot = np.zeros_like(z)
    np.put_along_axis(onehot, y[:, None, :], 1.0, axis=1)
    return ((lse - picked) * m).sum() / denom, (sm - onehot) * m[:, None, :] / denom


def nearest_index(out_size, in_size):
    """F.interpolate(mode='nearest') source indices (mask_loss.py:14): min(floor(dst * fp32(in/out)), in-1)."""
    scale = np.float32(in_size) / np.float32(out_size)
    return np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64), in_size - 1)


def mask_loss_interpolate(pred, gt):
    """MaskLoss_interpolate.forward (losses/mask_loss.py:11-17).  pred (B,C,h,w) logits (channel 0 used), gt
    (B,Hm,Wm).  Returns (loss, d loss / d pred) -- gradient zero on channels != 0."""
    B, C, h, w = pred.shape
    s = _sigmoid(pred[:, 0])
    r = gt[:, nearest_index(h, gt.shape[1])][:, :, nearest_index(w, gt.shape[2])].astype(np.float64)
    denom = float(B * h * w)
    d = np.zeros(pred.shape, np.float64)
    d[:, 0] = np.sign(s - r) * s * (1 - s) / denom
    return np.abs(s - r).sum() / denom, d


def edgeconv_gather_max(pq, idx, slope):
    """Factored StaticGraph_module aggregation (init.py:64-68 with the BN folded, see include/checkerpose_hip.h):
    pq (B,N,2C) = [P'|Q'], idx (N,K).  Returns (out (B,N,C), kstar (B,N,C) first arg-max)."""
    C = pq.shape[2] // 2
    P, Q = pq[:, :, :C].astype(np.float64), pq[:, :, C:].astype(np.float64)
    nb = P[:, idx, :]                               # (B,N,K,C)
    kstar = nb.argmax(axis=2)                       # first maximum, like torch.max on the CPU
    y = nb.max(axis=2) + Q
    return np.where(y > 0, y, y * slope), kstar


def edgeconv_gather_max_bwd(pq, idx, gout, slope):
    """autograd of the above: gq = gout * leaky'(y); dQ' = gq; dP'[b, idx[i,k*], c] += gq[b,i,c].
    Returns dpq (B,N,2C) float64."""
    B, N, C2 = pq.shape
    C = C2 // 2
    P, Q = pq[:, :, :C].astype(np.float64), pq[:, :, C:].astype(np.float64)
    nb = P[:, idx, :]
    kstar = nb.argmax(axis=2)
    y = nb.max(axis=2) + Q
    gq = gout.astype(np.float64) * np.where(y > 0, 1.0, slope)
    dP = np.zeros((B, N, C), np.float64)
    j = idx[np.arange(N)[None, :, None], kstar]     # (B,N,C) winning neighbour of every (i,c)
    bb = np.arange(B)[:, None, None]
    cc = np.arange(C)[None, None, :]
    np.add.at(dP, (np.broadcast_to(bb, j.shape), j, np.broadcast_to(cc, j.shape)), gq)
    return np.concatenate([dP, gq], axis=2)


def index2feat_gather_bwd(gout, x_id, y_id, mask, Hp, Wp, E, k=2):
    """autograd of Index2Feat_module.forward's gathers (pipeline.py:158-162) times the RoI mask (:280):
    gout (B,N,4E) -> dpatches (B,Hp,Wp,E) float64 (channels-last)."""
    B, N, _ = gout.shape
    d = np.zeros((B, Hp, Wp, E), np.float64)
    g = gout.astype(np.float64) * mask.astype(np.float64)[:, :, None]
    for tap in range(4):
        yy = 2 * y_id + (k if tap & 1 else 0)       # sf2, sf4: row + k
        xx = 2 * x_id + (k if tap & 2 else 0)       # sf3, sf4: col + k
        np.add.at(d, (np.arange(B)[:, None], yy, xx), g[:, :, tap * E:(tap + 1) * E])
    return d
