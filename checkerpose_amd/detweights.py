"""Closed-form deterministic parameter / input fill.

No checkpoint of the reference is available offline, so every
parity test and the benchmark fill the state dict with a closed-form function
of (key name, flat element index): a splitmix64 integer hash -> 24-bit uniform,
which is exactly representable in fp32 and therefore regenerates bit-identically
on any machine (no libm involved).  The same fill is applied to the reference's
own modules when the golden vectors are made (tests/golden/make_golden.py).
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x ^= x >> np.uint64(30)
    x = (x * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x ^= x >> np.uint64(27)
    x = (x * np.uint64(0x94D049BB133111EB)) & _M64
    x ^= x >> np.uint64(31)
    return x


def uniform_pm1(name, numel, seed=0):
    """numel values in [-1, 1), function of (name, seed, index) only; float32."""
    key = np.uint64(zlib.crc32(name.encode()) + (int(seed) << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(numel, dtype=np.uint64)
        h = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + _splitmix64(key))
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1) on a 2^-24 grid
    return (2.0 * u - 1.0).astype(np.float32)


def det_tensor(name, shape, scale=1.0, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(uniform_pm1(name, n, seed) * np.float32(scale)).reshape(shape)


def _residual_damp(k, sd):
    """Backbone residual / fuse sums would otherwise double the variance ~40 times in a row:
    damp the BatchNorm that closes a residual branch (bn3 of a Bottleneck, bn2 of a BasicBlock),
    the shortcut BN and the HRNet fuse-layer BNs so features stay O(1)."""
    stem = k[: -len("weight")]
    if "img_backbone" not in k and not k.startswith(("layer", "stage", "transition", "incre", "conv", "bn")):
        return 1.0
    if stem.endswith("bn3.") or (stem.endswith("bn2.") and (stem[:-4] + "bn3.weight") not in sd):
        return 0.15 if ("layer" in stem or "branches" in stem or "incre" in stem) else 1.0
    if "fuse_layers" in stem:
        return 0.22
    return 1.0


def _head_gain(k):
    """Keep graph features and logits O(1..5) like a trained net's, so that the absolute 1e-4 logit tolerance
    is a ~1e-5 relative one: post-ReLU inputs have a non-zero mean (plain He gain over-amplifies the 1024-wide
    conv1x1) and the max-over-neighbours of EdgeConv adds a positive drift per layer."""
    if k.endswith("conv1x1.weight"):
        return 0.2
    if "pre_query_block" in k and k.endswith("conv.0.weight"):
        return 0.6
    if k.endswith("pre_graph_module.0.weight"):
        return 0.7
    if k.endswith("query_block.mlps.4.weight") or k.endswith("mlp.weight"):
        return 1.0
    if k.endswith("seg_block.weight"):
        return 0.3
    return 1.0


@torch.no_grad()
def fill_state_dict_(sd, seed=0, bias_scale=0.1):
    """In-place deterministic fill of a state dict (reference or build modules alike).

    conv / linear weights: uniform with He variance (2 / fan_in), so activations stay
    O(1) through the ~70 layers; BatchNorm gamma has MIXED SIGNS (exercises the
    max-vs-min selection of the factored EdgeConv), running_var in [0.5, 1.5],
    running_mean and beta small.  Keys are classified by suffix and tensor rank only.
    """
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            v.zero_()
            continue
        if k.endswith("running_var"):
            v.copy_(1.0 + 0.5 * det_tensor(k, v.shape, 1.0, seed))
        elif k.endswith("running_mean"):
            v.copy_(det_tensor(k, v.shape, 0.1, seed))
        elif v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            if "up_net.0.0." in k:  # ConvTranspose2d weight is (in, out, kh, kw); stride 2 -> ~kh*kw/4 taps hit
                fan_in = v.shape[0] * v.shape[2] * v.shape[3] / 4.0
            std = (2.0 / fan_in) ** 0.5
            std *= _head_gain(k)
            v.copy_(det_tensor(k, v.shape, std * 3.0 ** 0.5, seed))
        elif k.endswith("weight"):  # BatchNorm gamma: |gamma| in [0.6, 1.4], ~25 % negative
            u = det_tensor(k, v.shape, 1.0, seed)
            s = det_tensor(k + "#sign", v.shape, 1.0, seed)
            v.copy_((1.0 + 0.4 * u) * torch.where(s < -0.5, -1.0, 1.0) * _residual_damp(k, sd))
        elif k.endswith("bias"):
            v.copy_(det_tensor(k, v.shape, bias_scale, seed))
        else:
            raise KeyError("unclassified state-dict key " + k)
    return sd


def det_image(batch, size=256, seed=0):
    """Synthetic crop batch (B,3,size,size) ~ unit variance (ImageNet-normalised crops are;
    reference: checkerpose/bop_dataset_pytorch.py:385-398)."""
    return det_tensor("img", (batch, 3, size, size), 3.0 ** 0.5, seed)
