"""Backbone feature extractors: parameter containers in timm's naming.

Mirrors reference checkerpose/model/backbone.py:39-50 (`get_timm_backbone`), which obtains
`timm.create_model(name, features_only=True, out_indices=(1,2,3,4))`.  timm is a third-party
dependency absent from the reference tree, so the layouts below restate timm's published
`HighResolutionNetFeatures(hrnet_w18 | hrnet_w18_small | hrnet_w30, feature_location="incre")` and ResNet-34
(SURVEY.md Appendix A); the state-dict keys follow timm's so released checkpoints load unchanged.

These classes only HOLD parameters (nn.Conv2d / nn.BatchNorm2d children are never called);
the arithmetic runs in the HIP engine (checkerpose_amd/engine.py).  Calling forward() directly
is not supported -- the owning InitNet_GNN drives the backbone as part of its fused program.
"""
import torch
import torch.nn as nn


def _conv(cin, cout, k, s=1, p=0):
    return nn.Conv2d(cin, cout, k, s, p, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(cin, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, downsample=None):
        super().__init__()
        self.conv1 = _conv(cin, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample


def _down(cin, cout, stride=1):
    return nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))


class HighResolutionModule(nn.Module):
    def __init__(self, chans, nblocks=4):
        super().__init__()
        self.chans = tuple(chans)
        self.branches = nn.ModuleList(
            nn.Sequential(*[BasicBlock(c, c) for _ in range(nblocks)]) for c in chans)
        fuse = []
        for i, ci in enumerate(chans):
            row = []
            for j, cj in enumerate(chans):
                if j > i:
                    row.append(nn.Sequential(_conv(cj, ci, 1), nn.BatchNorm2d(ci),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(nn.Identity())
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = ci if last else cj
                        mods = [_conv(cj, cout, 3, 2, 1), nn.BatchNorm2d(cout)]
                        if not last:
                            mods.append(nn.ReLU(False))
                        steps.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)


# timm `hrnet.cfg_cls` (published layouts) of the HRNet names the reference accepts (backbone.py:43, init.py:15-24):
#   layer1 = (Bottleneck blocks, planes) ; stages 2..4 = (modules, BasicBlocks per branch, branch channels).  stem_width 64 and the
#   "incre" head channels (32, 64, 128, 256) x Bottleneck expansion 4 are the same for all of them.
HRNET_CFGS = {
    "hrnet_w18": dict(layer1=(4, 64), stages=((1, 4, (18, 36)), (4, 4, (18, 36, 72)), (3, 4, (18, 36, 72, 144)))),
    "hrnet_w30": dict(layer1=(4, 64), stages=((1, 4, (30, 60)), (4, 4, (30, 60, 120)), (3, 4, (30, 60, 120, 240)))),
    "hrnet_w18_small": dict(layer1=(1, 32), stages=((1, 2, (16, 32)), (1, 2, (16, 32, 64)), (1, 2, (16, 32, 64, 128)))),
}


class HRNetFeatures(nn.Module):
    """timm HighResolutionNetFeatures(<name>, feature_location="incre"), out_indices (1, 2, 3, 4):
    outputs [128@64^2, 256@32^2, 512@16^2, 1024@8^2] for a 256^2 crop, whatever the body's widths."""
    out_channels = (128, 256, 512, 1024)

    def __init__(self, name="hrnet_w18"):
        super().__init__()
        cfg = HRNET_CFGS[name]
        self.name = name
        self.cfg = cfg
        n1, p1 = cfg["layer1"]
        self.conv1 = _conv(3, 64, 3, 2, 1)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = _conv(64, 64, 3, 2, 1)
        self.bn2 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(Bottleneck(64, p1, _down(64, 4 * p1)), *[Bottleneck(4 * p1, p1) for _ in range(n1 - 1)])
        pre = (4 * p1,)
        for si, (nmod, nblk, chans) in enumerate(cfg["stages"]):
            # timm _make_transition_layer: a kept branch whose width changes gets conv3x3 + BN + ReLU (only transition1's branch 0
            # here), an unchanged one nn.Identity; every NEW branch is a stride-2 conv3x3 + BN + ReLU from the previous LAST branch
            tr = []
            for i, c in enumerate(chans):
                if i < len(pre):
                    tr.append(nn.Identity() if pre[i] == c else nn.Sequential(_conv(pre[i], c, 3, 1, 1), nn.BatchNorm2d(c), nn.ReLU(False)))
                else:
                    assert i == len(pre), "one new branch per stage"
                    tr.append(nn.Sequential(nn.Sequential(_conv(pre[-1], c, 3, 2, 1), nn.BatchNorm2d(c), nn.ReLU(False))))
            setattr(self, "transition%d" % (si + 1), nn.ModuleList(tr))
            setattr(self, "stage%d" % (si + 2), nn.Sequential(*[HighResolutionModule(chans, nblk) for _ in range(nmod)]))
            pre = chans
        self.incre_modules = nn.ModuleList(
            nn.Sequential(Bottleneck(c, p, _down(c, p * 4))) for c, p in zip(pre, (32, 64, 128, 256)))

    def forward(self, x):
        raise RuntimeError("checkerpose_amd backbones are parameter containers; run them through InitNet_GNN "
                           "(HIP engine). There is no PyTorch fallback path.")


class HRNetW18Features(HRNetFeatures):
    """hrnet_w18 (the shipped configs' backbone)"""
    STAGES = (("stage2", 1, (18, 36)), ("stage3", 4, (18, 36, 72)), ("stage4", 3, (18, 36, 72, 144)))

    def __init__(self):
        super().__init__("hrnet_w18")


class ResNet34Features(nn.Module):
    """resnet34 features_only out_indices (1,2,3,4): [64@64^2, 128@32^2, 256@16^2, 512@8^2]."""
    name = "resnet34"
    out_channels = (64, 128, 256, 512)
    LAYERS = (3, 4, 6, 3)

    def __init__(self):
        super().__init__()
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (n, planes) in enumerate(zip(self.LAYERS, (64, 128, 256, 512))):
            blocks = []
            for k in range(n):
                stride = 2 if (k == 0 and li > 0) else 1
                ds = _down(cin, planes, stride) if (stride != 1 or cin != planes) else None
                blocks.append(BasicBlock(cin, planes, stride, ds))
                cin = planes
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))

    def forward(self, x):
        raise RuntimeError("checkerpose_amd backbones are parameter containers; run them through InitNet_GNN "
                           "(HIP engine). There is no PyTorch fallback path.")


def _timm_style_init_(m):
    """The from-scratch initialisation timm applies in `HighResolutionNet.init_weights` / `ResNet.init_weights`
    (restated from timm's published source, not verifiable offline): Conv2d kaiming_normal_(fan_out, relu); BatchNorm
    weight 1 / bias 0; ResNet additionally zero-initialises the last BatchNorm gamma of every residual block
    (`zero_init_last=True`, timm's default)."""
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(mod, nn.BatchNorm2d):
            nn.init.ones_(mod.weight)
            nn.init.zeros_(mod.bias)
    if isinstance(m, ResNet34Features):
        for mod in m.modules():
            if isinstance(mod, BasicBlock):
                nn.init.zeros_(mod.bn2.weight)


def _load_pretrained_(m, model_name):
    """`pretrained=True` (reference backbone.py:48-49 -> timm downloads ImageNet weights; pretrain.py:180-183 relies on it).
    There is no network here, so the weights come from a LOCAL timm checkpoint:
      CHECKERPOSE_AMD_TIMM_CKPT      = path of the <model_name> checkpoint file, or
      CHECKERPOSE_AMD_TIMM_CKPT_DIR  = directory holding <model_name>.pth
    (timm's classification checkpoint: the keys this features_only layout has are loaded, classifier / downsamp_modules /
    final_layer keys are dropped exactly as timm's FeatureInfo path does).  Without one, warn loudly -- or raise when
    CHECKERPOSE_AMD_REQUIRE_PRETRAINED=1 -- and fall back to timm's from-scratch init: training the init net from
    PyTorch-default weights silently is not what the reference's pretrain.py does."""
    import os
    import warnings
    path = os.environ.get("CHECKERPOSE_AMD_TIMM_CKPT")
    if not path and os.environ.get("CHECKERPOSE_AMD_TIMM_CKPT_DIR"):
        path = os.path.join(os.environ["CHECKERPOSE_AMD_TIMM_CKPT_DIR"], model_name + ".pth")
    if path and os.path.exists(path):
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd
        own = m.state_dict()
        missing = [k for k in own if k not in sd and not k.endswith("num_batches_tracked")]
        if missing:
            raise RuntimeError("timm checkpoint %s lacks %d keys of %s (first: %s)" % (path, len(missing), model_name, missing[0]))
        bad = [k for k in own if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
        if bad:
            raise RuntimeError("timm checkpoint %s: shape mismatch at %s" % (path, bad[0]))
        m.load_state_dict({k: sd[k] for k in own if k in sd}, strict=False)
        return True
    msg = ("checkerpose_amd.get_timm_backbone(%r, pretrained=True): no local timm checkpoint (set CHECKERPOSE_AMD_TIMM_CKPT or "
           "CHECKERPOSE_AMD_TIMM_CKPT_DIR; there is no network to download ImageNet weights as the reference does) -- the backbone "
           "is initialised FROM SCRATCH with timm's init scheme" % model_name)
    if os.environ.get("CHECKERPOSE_AMD_REQUIRE_PRETRAINED", "0") == "1":
        raise RuntimeError(msg)
    warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return False


def get_timm_backbone(model_name="resnet34", concat_decoder=True, pretrained=True):
    """Same name/arguments as reference backbone.py:39.  pretrained=True loads a local timm checkpoint (see
    _load_pretrained_) or warns; either way a from-scratch backbone gets timm's init, not PyTorch's default."""
    if model_name in HRNET_CFGS:                  # hrnet_w18 | hrnet_w18_small | hrnet_w30 (backbone.py:43)
        m = HRNetFeatures(model_name)
    elif model_name == "resnet34":
        m = ResNet34Features()
    else:
        raise ValueError("timm_backbone {} not supported yet".format(model_name))  # backbone.py:47
    _timm_style_init_(m)
    if pretrained:
        _load_pretrained_(m, model_name)
    return m
