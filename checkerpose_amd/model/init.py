"""Drop-in for reference checkerpose/model/init.py: same class names, constructor arguments, forward()
signature/returns and state-dict keys (init.py:54-128); the forward runs as HIP kernels on an MI355X.

The nn.Conv2d / nn.BatchNorm2d / nn.Linear children exist only to own parameters under the reference's key
names (so `load_state_dict` of a reference checkpoint works, test.py:226-227); they are never called.
"""
import torch
import torch.nn as nn

from .backbone import get_timm_backbone
from ._runtime import HipForwardMixin

CONV1X1_IN_CHANS = {"resnet34": 512, "hrnet_w18": 1024, "hrnet_w18_small": 1024, "hrnet_w30": 1024}   # init.py:15-24 (supported backbones)


def knn(x, k):
    """init.py:27-32, same op sequence (fp32, on CPU for a machine-independent tie order).
    x (G,3,N) -> (G,N,k) int64; construction-time only."""
    x = x.detach().float().cpu()
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise_distance = -xx - inner - xx.transpose(2, 1)
    return pairwise_distance.topk(k=k, dim=-1)[1]


class StaticGraph_module(nn.Module):
    """Parameter container of init.py:54-62; the EdgeConv arithmetic is cp_conv2d_igemm + cp_edgeconv_gather_max."""

    def __init__(self, input_dim, output_dim, knn_idx, leaky_slope=0.2):
        super().__init__()
        self.knn_idx = knn_idx          # plain attribute, not a buffer -- as in the reference (init.py:57)
        self.leaky_slope = leaky_slope
        self.conv = nn.Sequential(
            nn.Conv2d(input_dim * 2, output_dim, kernel_size=1, bias=False),
            nn.BatchNorm2d(output_dim),
            nn.LeakyReLU(negative_slope=leaky_slope))


class InitNet_GNN(HipForwardMixin, nn.Module):
    LM = False

    def __init__(self, npoint, p3d_normed, res_log2=3, backbone_name="resnet34", pretrain_backbone=True,
                 num_conv1x1=1, max_batch_size=64, num_graph_module=2, graph_k=20, graph_leaky_slope=0.2):
        super().__init__()
        if not 1 <= res_log2 <= 6:
            raise ValueError("InitNet_GNN: res_log2 must be in 1..6 (1 + 2*res_log2 <= 13 output rows)")
        if num_conv1x1 < 1:
            raise ValueError("InitNet_GNN: num_conv1x1 must be >= 1")
        self.num_out_bits = 1 + 2 * res_log2
        self.npoint = npoint
        self.backbone_name = backbone_name
        self.img_backbone = get_timm_backbone(model_name=backbone_name, concat_decoder=True, pretrained=pretrain_backbone)
        if num_conv1x1 == 1:
            self.conv1x1 = nn.Conv2d(CONV1X1_IN_CHANS[backbone_name], npoint, kernel_size=1, stride=1, padding=0)
        else:                                               # init.py:87-95: same Sequential layout -> keys conv1x1.{0,2,..}
            layers = [nn.Conv2d(CONV1X1_IN_CHANS[backbone_name], npoint, kernel_size=1, stride=1, padding=0)]
            for _ in range(num_conv1x1 - 1):
                layers += [nn.LeakyReLU(negative_slope=0.01), nn.Conv2d(npoint, npoint, kernel_size=1, stride=1, padding=0)]
            self.conv1x1 = nn.Sequential(*layers)
        self.num_conv1x1 = num_conv1x1
        self.res_log2 = res_log2
        self.pre_query_block = nn.ModuleList()
        self.knn_idx = knn(p3d_normed, graph_k)             # (G, N, K) int64, G = 1 or #objects (LM)
        self._p3d = p3d_normed.detach().float().cpu()        # plain attribute: the tiled EdgeConv's patch schedule (N > 512) needs it
        self.graph_k = graph_k
        self.graph_leaky_slope = graph_leaky_slope
        self.max_batch_size = max_batch_size               # kept for signature parity; no limit is imposed
        for _ in range(num_graph_module):
            self.pre_query_block.append(StaticGraph_module(64, 64, self.knn_idx, graph_leaky_slope))
        self.mlp = nn.Linear(64, self.num_out_bits)
        self._init_runtime()

    # ---- HipForwardMixin hooks
    def _net_cfg(self):
        return dict(kind="init", npoint=self.npoint, backbone=self.backbone_name, img_size=None,
                    init_num_graph_module=len(self.pre_query_block), init_graph_slope=self.graph_leaky_slope,
                    graph_k=self.graph_k, init_res_log2=self.res_log2, num_conv1x1=self.num_conv1x1)

    def _knn_table(self):
        return self.knn_idx

    def _keypoints(self):
        return self._p3d

    def _out_rows(self, bits):
        """the (B, 1 + 2 res_log2, N) output of init.py:120-122 out of the launch program's logit block; shared with the LM twin"""
        r = self.res_log2          # rows of the (B,13,N) logit block: [roi | x bits at 1.. | y bits at 7..]; res_log2 > 3: packed
        return torch.cat([bits[:, 0:4], bits[:, 7:10]], dim=1) if r == 3 else bits[:, :1 + 2 * r]

    def forward(self, img, return_img_feats=False, return_graph_feats=False):
        """init.py:109-128: returns out (B,7,N) | (out, img_feats) | (out, img_feats, graph_feats)."""
        res = self._run(img, None, want_feats=return_img_feats or return_graph_feats, want_graph=return_graph_feats)
        out = self._out_rows(res["bits"])
        if return_img_feats:
            return out, res["img_feats"]
        if return_graph_feats:
            return out, res["img_feats"], res["graph_feats"]
        return out
