"""Drop-in for reference checkerpose/model/pipeline.py (full progressive network): same class names,
constructor arguments, forward() signature, 6-tuple of outputs and state-dict keys (pipeline.py:130-384).
Children are parameter containers only; the forward is one HIP launch program (see ../netbuilder.py).
"""
import torch
import torch.nn as nn

from ._runtime import HipForwardMixin
from .init import StaticGraph_module, knn  # noqa: F401  (re-exported like the reference module does)

IMG_FEATS_DIMS = {"resnet34": [64, 128, 256, 512], "hrnet_w18": [128, 256, 512, 1024], "hrnet_w18_small": [128, 256, 512, 1024],
                  "hrnet_w30": [128, 256, 512, 1024]}   # pipeline.py:6-15


def get_MLP_leakyReLU_layers(dims, doLastAct, negative_slope=0.1):
    """pipeline.py:61-69 (container with the reference's Sequential indices)."""
    layers = []
    for i in range(1, len(dims)):
        layers.append(nn.Linear(dims[i - 1], dims[i]))
        if i == len(dims) - 1 and not doLastAct:
            continue
        layers.append(nn.LeakyReLU(negative_slope=negative_slope))
    return nn.Sequential(*layers)


class Index2Feat_module(nn.Module):
    """pipeline.py:130-147 container."""

    def __init__(self, feat_dim, embed_dim=None, kernel_size=2):
        super().__init__()
        self.kernel_size = kernel_size
        self.embed_dim = embed_dim if embed_dim is not None else (feat_dim * kernel_size * kernel_size)
        self.patch_generator = nn.Conv2d(feat_dim, self.embed_dim, kernel_size=kernel_size, stride=1, padding=kernel_size - 1)


class MLP_QueryNet(nn.Module):
    """pipeline.py:168-172 container (`pts` never enters the arithmetic: pipeline.py:174-180)."""

    def __init__(self, feat_dims=(256, 256, 64), pt_dim=3, out_dim=4, leaky_slope=0.01):
        super().__init__()
        self.mlps = get_MLP_leakyReLU_layers(dims=tuple(feat_dims) + (out_dim,), doLastAct=False, negative_slope=leaky_slope)


def get_gdrn_upsample_module(is_convtrans=False, in_channels=512, num_filters=256, kernel_size=3, padding=1, output_padding=1):
    """pipeline.py:183-211 container (same Sequential indices -> same state-dict keys)."""
    layers = []
    if is_convtrans:
        layers.append(nn.ConvTranspose2d(in_channels, num_filters, kernel_size=kernel_size, stride=2, padding=padding,
                                         output_padding=output_padding, bias=False))
        layers.append(nn.BatchNorm2d(num_filters))
        layers.append(nn.ReLU(inplace=True))
        layers.append(nn.Conv2d(num_filters, num_filters, kernel_size=3, stride=1, padding=1, bias=False))
    else:
        layers.append(nn.UpsamplingBilinear2d(scale_factor=2))
        layers.append(nn.Conv2d(in_channels, num_filters, kernel_size=3, stride=1, padding=1, bias=False))
    layers.append(nn.BatchNorm2d(num_filters))
    layers.append(nn.ReLU(inplace=True))
    layers.append(nn.Conv2d(num_filters, num_filters, kernel_size=3, stride=1, padding=1, bias=False))
    layers.append(nn.BatchNorm2d(num_filters))
    layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class Refine_moduleGNN(nn.Module):
    """pipeline.py:214-260 container."""

    def __init__(self, npoint, p3d_normed, num_filters=256, max_batch_size=64, query_dims=None, local_k=4,
                 leaky_slope=0.01, num_graph_module=2, graph_k=20, graph_leaky_slope=0.2, query_type="mlp",
                 graph_feat_dim=64, knn_idx=None):
        super().__init__()
        self.npoint = npoint
        if query_type == "mlp":
            self.query_dims = (num_filters, 256, 64) if query_dims is None else tuple(query_dims)
        else:
            raise ValueError("query type {} not supported in Refine_module".format(query_type))   # pipeline.py:232
        self.local_feat_ext_block = Index2Feat_module(num_filters, self.query_dims[0] // 4, local_k)
        self.pre_graph_module = get_MLP_leakyReLU_layers(
            (self.query_dims[0] + graph_feat_dim, self.query_dims[0], self.query_dims[0]), True, leaky_slope)
        self.pre_query_block = nn.ModuleList()
        if knn_idx is None:
            knn_idx = knn(p3d_normed, graph_k)
        for _ in range(num_graph_module):
            self.pre_query_block.append(StaticGraph_module(self.query_dims[0], self.query_dims[0], knn_idx, graph_leaky_slope))
        self.query_block = MLP_QueryNet(self.query_dims, 3, 2, leaky_slope)


class PoseNet_GNNskip(HipForwardMixin, nn.Module):
    LM = False

    def __init__(self, init_net, npoint, p3d_normed, res_log2=6, num_filters=256, max_batch_size=64, query_dims=None,
                 seg_output_dim=2, local_k=4, leaky_slope=0.01, num_graph_module=2, graph_k=20, graph_leaky_slope=0.2,
                 query_type="mlp"):
        super().__init__()
        self.npoint = npoint
        self.init_net = init_net
        self.num_refine_steps = res_log2 - 3
        if not 0 <= self.num_refine_steps <= 3:
            raise ValueError("PoseNet_GNNskip: res_log2 must be in 3..6")
        if getattr(init_net, "res_log2", 3) != 3:
            raise ValueError("PoseNet_GNNskip: the init net must predict 1 + 3 + 3 bits (pipeline.py:363-365 splits them so)")
        if graph_k != init_net.graph_k:
            raise ValueError("PoseNet_GNNskip: graph_k must equal the init net's (one shared kNN table)")
        self.cfg = dict(res_log2=res_log2, num_filters=num_filters, query_dims=tuple(query_dims) if query_dims else None,
                        seg_output_dim=seg_output_dim, local_k=local_k, leaky_slope=leaky_slope,
                        num_graph_module=num_graph_module, graph_slope=graph_leaky_slope)
        feats = IMG_FEATS_DIMS[init_net.backbone_name]
        self.up_net = nn.ModuleList()
        for i in range(self.num_refine_steps):
            if i == 0:
                self.up_net.append(get_gdrn_upsample_module(True, feats[-1], num_filters))
            else:
                self.up_net.append(get_gdrn_upsample_module(False, num_filters + feats[-i - 1], num_filters))
        self.refine_net = nn.ModuleList()
        knn_idx = init_net.knn_idx     # identical table (same points, same k): pipeline.py:248 recomputes it per module
        for i in range(self.num_refine_steps):
            ng = num_graph_module if isinstance(num_graph_module, int) else num_graph_module[i]
            gdim = 64 if i == 0 else (num_filters if query_dims is None else query_dims[0])
            self.refine_net.append(Refine_moduleGNN(npoint, p3d_normed, num_filters, max_batch_size, query_dims, local_k,
                                                    leaky_slope, ng, graph_k, graph_leaky_slope, query_type, gdim,
                                                    knn_idx=knn_idx))
        self.seg_block = nn.Conv2d(num_filters, seg_output_dim, kernel_size=1, padding=0, bias=True)
        self._init_runtime()
        import weakref
        object.__setattr__(init_net, "_owner", weakref.ref(self))     # init_net.load_state_dict() must drop OUR folded weights too

    def _net_cfg(self):
        c = dict(self.cfg)
        c.update(kind="pose", npoint=self.npoint, backbone=self.init_net.backbone_name,
                 init_num_graph_module=len(self.init_net.pre_query_block),
                 init_graph_slope=self.init_net.graph_leaky_slope, num_conv1x1=getattr(self.init_net, "num_conv1x1", 1))
        return c

    def _knn_table(self):
        return self.init_net.knn_idx

    def _keypoints(self):
        return self.init_net._p3d

    def set_compute_dtype(self, name):
        self.init_net.set_compute_dtype(name)
        return super().set_compute_dtype(name)

    def _outputs(self, res, active):
        bits = res["bits"]
        return (bits[:, 0:1], bits[:, 1:4 + active], bits[:, 7:10 + active], res["seg"], res["x64"], res["y64"])

    def forward_teacher_forced(self, img, teacher_bits, stage=None, obj_ids=None):
        """Test hook: same forward, but every discrete decision (RoI mask, pixel ids that pick
        the next stage's gather locations) is decoded from `teacher_bits` (B,13,N) -- e.g. the oracle's logits --
        so one flipped bit cannot mask or fake agreement downstream.  Returns the 6-tuple; ids are the teacher's."""
        active = stage if stage is not None else self.num_refine_steps
        return self._outputs(self._run(img, obj_ids, stage=stage, teacher_bits=teacher_bits), active)

    def forward_injected_feats(self, img, feats, stage=None, obj_ids=None, teacher_bits=None):
        """Test hook: the forward with the backbone's four features GIVEN (NCHW fp32 list, as `img_backbone` returns them in the
        reference, init.py:111) -- exactly how the reference-made `*_injected` goldens were produced (a timm stub returning preset
        features), so the whole head is compared with the reference's own outputs directly.  `img` only supplies B and the size."""
        active = stage if stage is not None else self.num_refine_steps
        return self._outputs(self._run(img, obj_ids, stage=stage, teacher_bits=teacher_bits, inject_feats=feats), active)

    def forward_hooks(self, img, teacher_bits=None, feats=None, dec=None, want_feats=False, want_dec=False, stage=None, obj_ids=None):
        """Attribution hook (agreement.attribute_groups): one forward with any of the three block groups' inputs GIVEN -- `feats`
        (the backbone's four features, NCHW fp32) replaces the backbone, `dec` (the decoder stages' output maps, NCHW fp32) replaces
        what the refinement stages gather from -- and / or their outputs returned (`want_feats`, `want_dec`).  Two models of the
        same weights at different compute dtypes exchange these tensors to run ONE group in bf16 and the rest in fp32.
        Returns (6-tuple, {"img_feats": [...], "dec_feats": [...]})."""
        active = stage if stage is not None else self.num_refine_steps
        res = self._run(img, obj_ids, stage=stage, teacher_bits=teacher_bits, inject_feats=feats, inject_dec=dec,
                        want_feats=want_feats, want_dec=want_dec)
        return self._outputs(res, active), {k: res[k] for k in ("img_feats", "dec_feats") if k in res}

    def forward(self, img, p3d_normed, stage=None):
        """pipeline.py:351-384.  `p3d_normed` is accepted for signature parity; it has no numeric effect in the
        reference either (only forwarded to MLP_QueryNet, which ignores it: pipeline.py:174-180,295)."""
        active = stage if stage is not None else self.num_refine_steps
        res = self._run(img, None, stage=stage)
        return self._outputs(res, active)
