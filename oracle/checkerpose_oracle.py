"""ORACLE -- test infrastructure, NOT product code.

CPU fp32 restatement (plain PyTorch tensor ops, functional over a flat state dict)
of CheckerPose's forward hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file; the product path
(checkerpose_amd/) never does and fails loudly without its HIP library.

Pinning status
--------------
* Head (everything after the backbone: conv1x1, EdgeConv stacks, decoder, local
  feature gather, MLPs, bit decode, seg head, LM per-object graphs): PINNED.  The
  reference's own modules (/root/reference/checkerpose/model/{init,pipeline,init_lm,
  pipeline_lm}.py) were imported in the build container with a `timm` stub and this
  restatement was checked against them; the golden vectors under tests/golden/ were
  produced by the REFERENCE modules (tests/golden/make_golden.py) and
  tests/test_oracle.py re-checks this file against them on every run.
* Train mode (bn_train(): batch-statistics BatchNorm + torch autograd through this file): PINNED for the head.
  tests/golden/make_golden_trainstep.py ran ONE TRAINING STEP of the reference's own modules in .train() mode with the
  reference's loss classes combined as train.py:307-318 does; tests/test_oracle_train.py re-checks losses, logits, ids,
  every parameter gradient and BatchNorm running statistics of this restatement against it (2e-5).
* Backbone (a1: timm `hrnet_w18` / `resnet34` features_only): PARITY UNPINNED.  The
  arithmetic lives in the third-party package `timm` (version not pinned anywhere in
  the reference; backbone.py:5,35,48-49), which is absent from /root/reference and
  not installable offline.  `hrnet_features` / `resnet34_features` below restate timm's
  published HighResolutionNetFeatures(hrnet_w18, feature_location='incre') and
  ResNet-34 layouts; they are anchored on the reference's
  channel/stride contract (pipeline.py:6-15, init.py:15-24,111) and CROSS-CHECKED (not pinned:
  it is not the reference's dependency) against the one independent implementation this image
  holds, HF transformers' ResNet: `resnet34_features` == `ResNetModel` configured as ResNet-34
  with the same weights, `_bottleneck` / `_basic_block` == `ResNetBottleNeckLayer` /
  `ResNetBasicLayer` on HRNet's own blocks (tests/test_oracle.py, 1e-5).  The HRNet ASSEMBLY
  (`_hr_module`, `hrnet_features`: transitions, branches, fuse layers) has no second
  implementation offline.

Every function cites the reference file:line (relative to /root/reference/checkerpose)
it follows.
"""
import torch
import torch.nn.functional as F

# pipeline.py:6-15 / init.py:15-24
IMG_FEATS_DIMS = {"resnet34": [64, 128, 256, 512], "hrnet_w18": [128, 256, 512, 1024], "hrnet_w18_small": [128, 256, 512, 1024],
                  "hrnet_w30": [128, 256, 512, 1024]}
CONV1X1_IN_CHANS = {"resnet34": 512, "hrnet_w18": 1024, "hrnet_w18_small": 1024, "hrnet_w30": 1024}


# --------------------------------------------------------------------------- generic pieces
BN_TRAIN = {"on": False}     # tests flip this for the train-mode (batch statistics) restatement, see bn_train()


class bn_train:
    """context manager: BatchNorm layers run like nn.BatchNorm2d in .train() mode (batch statistics, running stats of
    `sd` updated in place with momentum 0.1) -- the state every module is in during reference train.py:300-320."""

    def __enter__(self):
        BN_TRAIN["on"] = True

    def __exit__(self, *a):
        BN_TRAIN["on"] = False


def _bn(sd, p, x, eps=1e-5):
    """BatchNorm (nn.BatchNorm2d defaults eps=1e-5, momentum 0.1), any rank, channel dim 1; eval mode unless bn_train()."""
    if BN_TRAIN["on"]:
        if (p + ".num_batches_tracked") in sd:
            sd[p + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)
    shape = [1, -1] + [1] * (x.dim() - 2)
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + eps)
    return x * s.view(shape) + (sd[p + ".bias"] - sd[p + ".running_mean"] * s).view(shape)


FORCE_MASK = {}     # tests only: {activation key: bool tensor, True where the device's pre-activation was > 0}


def _act(x, key, slope=0.0):
    """ReLU (slope 0) / LeakyReLU.  With a FORCE_MASK entry for `key` the branch of every element is the device's choice: the
    kink at 0 is a discontinuity of the derivative, and among ~1e8 activations a few sit within the ~1e-6 forward difference
    of 0 -- gradient-parity tests pin them (as they pin the arg-max of the EdgeConv max and the bit decisions)."""
    m = FORCE_MASK.get(key)
    if m is None:
        return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)
    return torch.where(m, x, x * slope)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _conv_bn(sd, pc, pb, x, stride=1, padding=0, relu=True):
    x = _bn(sd, pb, _conv(sd, pc, x, stride, padding))
    return _act(x, pb) if relu else x


# --------------------------------------------------------------------------- graph ops
def knn(x, k):
    """init.py:27-32 == pipeline.py:18-23.  x (G,3,N) -> (G,N,k) int64.  Same op sequence."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise_distance = -xx - inner - xx.transpose(2, 1)
    return pairwise_distance.topk(k=k, dim=-1)[1]


FORCE_KSTAR = {}    # tests only: {module prefix: (B,C',N) int64 neighbour slot} -- teacher-forces the arg-max of the max over K


def static_graph_module(sd, p, x, knn_idx, slope=0.2):
    """StaticGraph_module.forward, init.py:54-68 == pipeline.py:45-59, with
    get_graph_feature init.py:36-49.  x (B,C,N), knn_idx (1|B,N,K) -> (B,C',N).
    Written exactly as the reference does (per-edge 1x1 conv), not factored.
    If FORCE_KSTAR holds an entry for `p`, the max over K is replaced by a gather at those slots: the max routes its
    gradient to ONE neighbour, a discontinuous choice, so gradient-parity tests pin it to the device's choice (the two
    forwards agree only to ~1e-5 relative and near-ties among the K=20 candidates would otherwise be routed differently)."""
    B, C, N = x.shape
    idx = knn_idx.expand(B, -1, -1) if knn_idx.shape[0] == 1 else knn_idx
    K = idx.shape[2]
    nb = torch.gather(x.unsqueeze(3).expand(B, C, N, K), 2,
                      idx.unsqueeze(1).expand(B, C, N, K))          # x_j   (B,C,N,K)
    ctr = x.unsqueeze(3).expand(B, C, N, K)                        # x_i
    e = torch.cat([nb - ctr, ctr], dim=1)                          # (B,2C,N,K)
    e = F.conv2d(e, sd[p + ".conv.0.weight"])
    e = _bn(sd, p + ".conv.1", e)
    if p in FORCE_KSTAR and p in FORCE_MASK:      # device: leaky AFTER the (forced) selection, sign taken from its output
        return _act(e.gather(-1, FORCE_KSTAR[p].unsqueeze(-1)).squeeze(-1), p, slope)
    e = F.leaky_relu(e, slope)
    if p in FORCE_KSTAR:
        return e.gather(-1, FORCE_KSTAR[p].unsqueeze(-1)).squeeze(-1)
    return e.max(dim=-1)[0]


def mlp_leaky(sd, p, x, idxs, slope, last_act):
    """get_MLP_leakyReLU_layers pipeline.py:61-69; x (B,N,C); idxs = Sequential indices of the Linears."""
    for n, i in enumerate(idxs):
        x = F.linear(x, sd["%s.%d.weight" % (p, i)], sd["%s.%d.bias" % (p, i)])
        if n < len(idxs) - 1 or last_act:
            x = _act(x, "%s.%d" % (p, i), slope)
    return x


# --------------------------------------------------------------------------- bit decode (pipeline.py:72-127)
def mask_from_prob(z):
    """from_mask_prob_to_mask pipeline.py:120-127: sigmoid(z) > 0.5 -> 1.0/0.0"""
    return torch.where(torch.sigmoid(z) > 0.5, 1.0, 0.0)


def id_from_code_prob(z):
    """from_code_prob_to_id pipeline.py:84-92 + from_code_to_id :72-82 (MSB first). z (B,bits,N) -> (B,N) int64"""
    code = torch.where(torch.sigmoid(z) > 0.5, 1, 0)
    L = code.shape[1]
    ids = code[:, 0, :] * (2 ** (L - 1))
    for i in range(1, L):
        ids = ids + code[:, i, :] * (2 ** (L - 1 - i))
    return ids


def id_from_bit_prob(z):
    """from_bit_prob_to_id pipeline.py:103-110. z (B,1,N) -> (B,N) int64"""
    return torch.where(torch.sigmoid(z[:, 0, :]) > 0.5, 1, 0)


# --------------------------------------------------------------------------- head blocks
def index2feat(sd, p, feat, x_id, y_id, k=2):
    """Index2Feat_module.forward pipeline.py:149-164. feat (B,C,H,W); ids (B,N) int64 -> (B,4*E,N)"""
    patches = _conv(sd, p + ".patch_generator", feat, 1, k - 1)
    B = feat.shape[0]
    bi = torch.arange(B).view(B, 1).expand(-1, x_id.shape[1])
    sf1 = patches[bi, :, 2 * y_id, 2 * x_id]
    sf2 = patches[bi, :, 2 * y_id + k, 2 * x_id]
    sf3 = patches[bi, :, 2 * y_id, 2 * x_id + k]
    sf4 = patches[bi, :, 2 * y_id + k, 2 * x_id + k]
    return torch.cat([sf1, sf2, sf3, sf4], dim=2).permute(0, 2, 1)


def upsample_module(sd, p, x, is_convtrans):
    """get_gdrn_upsample_module pipeline.py:183-211 (Sequential indices as built there)."""
    if is_convtrans:
        x = F.conv_transpose2d(x, sd[p + ".0.weight"], None, stride=2, padding=1, output_padding=1)
        x = _act(_bn(sd, p + ".1", x), p + ".1")
        x = _conv_bn(sd, p + ".3", p + ".4", x, 1, 1)
        x = _conv_bn(sd, p + ".6", p + ".7", x, 1, 1)
    else:
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)  # nn.UpsamplingBilinear2d
        x = _conv_bn(sd, p + ".1", p + ".2", x, 1, 1)
        x = _conv_bn(sd, p + ".4", p + ".5", x, 1, 1)
    return x


def refine_module(sd, p, img_feat, graph_feat, roi_mask_bit, x_id, y_id, knn_idx, n_graph,
                  local_k=2, slope=0.01, graph_slope=0.2):
    """Refine_moduleGNN.forward pipeline.py:262-298 (p3d_normed has no numeric effect: :174-180)."""
    local = index2feat(sd, p + ".local_feat_ext_block", img_feat, x_id, y_id, local_k)
    local = local * roi_mask_bit
    local = torch.cat([local, graph_feat], dim=1).permute(0, 2, 1)
    local = mlp_leaky(sd, p + ".pre_graph_module", local, (0, 2), slope, True).permute(0, 2, 1)
    for i in range(n_graph):
        local = static_graph_module(sd, "%s.pre_query_block.%d" % (p, i), local, knn_idx, graph_slope)
    bits = mlp_leaky(sd, p + ".query_block.mlps", local.permute(0, 2, 1), (0, 2, 4), slope, False)
    return bits.permute(0, 2, 1), local


# --------------------------------------------------------------------------- backbones (UNPINNED, see header)
def _basic_block(sd, p, x, stride=1):
    """timm resnet.BasicBlock: conv3x3-bn-relu-conv3x3-bn (+shortcut) relu"""
    sc = x
    y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, stride, 1)
    y = _conv_bn(sd, p + ".conv2", p + ".bn2", y, 1, 1, relu=False)
    if (p + ".downsample.0.weight") in sd:
        sc = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, stride, 0, relu=False)
    return _act(y + sc, p + ".bn2")


def _bottleneck(sd, p, x):
    """timm resnet.Bottleneck (stride 1): 1x1-bn-relu, 3x3-bn-relu, 1x1-bn (+shortcut) relu"""
    sc = x
    y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, 1, 0)
    y = _conv_bn(sd, p + ".conv2", p + ".bn2", y, 1, 1)
    y = _conv_bn(sd, p + ".conv3", p + ".bn3", y, 1, 0, relu=False)
    if (p + ".downsample.0.weight") in sd:
        sc = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, 1, 0, relu=False)
    return _act(y + sc, p + ".bn3")


HRNET_W18 = dict(stage2=(1, (18, 36)), stage3=(4, (18, 36, 72)), stage4=(3, (18, 36, 72, 144)), blocks=4)


def _count(sd, fmt):
    """number of consecutive indices k = 0, 1, .. for which the key fmt % k exists in the state dict"""
    k = 0
    while (fmt % k) in sd:
        k += 1
    return k


def _hr_module(sd, p, xs, nblocks=4):
    """timm HighResolutionModule.forward: per-branch BasicBlocks, then fuse (sum over j, ReLU)."""
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        for k in range(nblocks):
            xs[b] = _basic_block(sd, "%s.branches.%d.%d" % (p, b, k), xs[b])
    out = []
    for i in range(nb):
        y = None
        for j in range(nb):
            q = "%s.fuse_layers.%d.%d" % (p, i, j)
            if j == i:
                t = xs[j]
            elif j > i:   # conv1x1 + bn + nearest upsample 2^(j-i)
                t = _conv_bn(sd, q + ".0", q + ".1", xs[j], 1, 0, relu=False)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:         # (i-j) stride-2 3x3 convs; all but the last keep channels and have ReLU
                t = xs[j]
                for k in range(i - j):
                    t = _conv_bn(sd, "%s.%d.0" % (q, k), "%s.%d.1" % (q, k), t, 2, 1, relu=(k != i - j - 1))
            y = t if y is None else y + t
        out.append(_act(y, "%s.fuse%d" % (p, i)))
    return out


def hrnet_features(sd, p, x):
    """timm HighResolutionNetFeatures(hrnet_w18 | hrnet_w18_small | hrnet_w30, features_only, out_indices=(1,2,3,4)) as called from
    backbone.py:48-49: returns [128@/4, 256@/8, 512@/16, 1024@/32].  The variants differ in widths and COUNTS only (timm cfg_cls:
    layer1 blocks, modules per stage, BasicBlocks per branch); the counts are read off the state dict's keys, the widths off its
    shapes (hrnet_w18: HRNET_W18 above)."""
    x = _conv_bn(sd, p + "conv1", p + "bn1", x, 2, 1)
    x = _conv_bn(sd, p + "conv2", p + "bn2", x, 2, 1)
    for k in range(_count(sd, p + "layer1.%d.conv1.weight")):
        x = _bottleneck(sd, "%slayer1.%d" % (p, k), x)
    xs = [_conv_bn(sd, p + "transition1.0.0", p + "transition1.0.1", x, 1, 1),
          _conv_bn(sd, p + "transition1.1.0.0", p + "transition1.1.0.1", x, 2, 1)]
    for si, stage in enumerate(("stage2", "stage3", "stage4")):
        nmod = _count(sd, p + stage + ".%d.branches.0.0.conv1.weight")
        nblk = _count(sd, p + stage + ".0.branches.0.%d.conv1.weight")
        if si > 0:  # transition{2,3}: new branch from the LAST branch of the previous stage
            t = "%stransition%d.%d.0" % (p, si + 1, si + 1)
            xs = xs + [_conv_bn(sd, t + ".0", t + ".1", xs[-1], 2, 1)]
        for m in range(nmod):
            xs = _hr_module(sd, "%s%s.%d" % (p, stage, m), xs, nblk)
    return [_bottleneck(sd, "%sincre_modules.%d.0" % (p, i), f) for i, f in enumerate(xs)]


def resnet34_features(sd, p, x):
    """timm resnet34 features_only out_indices=(1,2,3,4): [64@/4,128@/8,256@/16,512@/32]."""
    x = _conv_bn(sd, p + "conv1", p + "bn1", x, 2, 3)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nblk in enumerate((3, 4, 6, 3)):
        for k in range(nblk):
            x = _basic_block(sd, "%slayer%d.%d" % (p, li + 1, k), x, stride=2 if (k == 0 and li > 0) else 1)
        feats.append(x)
    return feats


BACKBONES = {"hrnet_w18": hrnet_features, "hrnet_w18_small": hrnet_features, "hrnet_w30": hrnet_features, "resnet34": resnet34_features}


# --------------------------------------------------------------------------- the two nets
def init_net_forward(sd, p, img, knn_idx, npoint, backbone="hrnet_w18", n_graph=2, graph_slope=0.2,
                     img_feats=None):
    """InitNet_GNN.forward init.py:109-128.  `img_feats` may be injected to test the head independently of the (unpinned)
    backbone.  num_conv1x1 > 1 (init.py:87-95: Conv2d, then LeakyReLU(0.01) + Conv2d(N -> N) pairs) is recognised by its
    state-dict keys `conv1x1.{0,2,..}`; res_log2 by the rows of `mlp.weight`.
    Returns (out (B,1+2*res_log2,N), img_feats, graph_feats (B,64,N))."""
    if img_feats is None:
        img_feats = BACKBONES[backbone](sd, p + "img_backbone.", img)
    if (p + "conv1x1.weight") in sd:
        out = _conv(sd, p + "conv1x1", img_feats[-1])
    else:
        out, j = _conv(sd, p + "conv1x1.0", img_feats[-1]), 2
        while (p + "conv1x1.%d.weight" % j) in sd:
            out = _conv(sd, p + "conv1x1.%d" % j, F.leaky_relu(out, 0.01))
            j += 2
    g = out.reshape(-1, npoint, 64).permute(0, 2, 1)
    for i in range(n_graph):
        g = static_graph_module(sd, "%spre_query_block.%d" % (p, i), g, knn_idx, graph_slope)
    out = F.linear(g.permute(0, 2, 1), sd[p + "mlp.weight"], sd[p + "mlp.bias"]).permute(0, 2, 1)
    return out, img_feats, g


def posenet_forward(sd, img, knn_idx, npoint, backbone="hrnet_w18", res_log2=6, init_n_graph=2, n_graph=3,
                    local_k=2, slope=0.01, graph_slope=0.2, init_graph_slope=0.2, stage=None, img_feats=None,
                    forced=None):
    """PoseNet_GNNskip.forward pipeline.py:351-384 (and the LM twin pipeline_lm.py:392-425 when knn_idx
    is the per-sample (B,N,K) table `self.knn_idx[obj_ids-1]`, pipeline_lm.py:57).

    forced: optional dict {"roi": (B,1,N) f32, "x": [ids per stage], "y": [...]} -- teacher forcing of the
    discrete feedback for per-stage parity tests.
    Returns the reference's 6-tuple plus a dict of intermediates."""
    nref = res_log2 - 3
    active = stage if stage is not None else nref
    ng = (n_graph,) * nref if isinstance(n_graph, int) else tuple(n_graph)
    bits, feats, g = init_net_forward(sd, "init_net.", img, knn_idx, npoint, backbone, init_n_graph,
                                      init_graph_slope, img_feats)
    roi, xb, yb = bits[:, 0:1], bits[:, 1:4], bits[:, 4:]
    mask = mask_from_prob(roi)
    x_id, y_id = id_from_code_prob(xb), id_from_code_prob(yb)
    inter = {"img_feats": feats, "graph0": g, "mask": mask, "x_ids": [x_id], "y_ids": [y_id], "up": [], "graph": []}
    if forced is not None:
        mask, x_id, y_id = forced["roi"], forced["x"][0], forced["y"][0]
    f = feats[-1]
    for i in range(active):
        if i > 0:
            f = torch.cat([f, feats[-i - 1]], dim=1)
        f = upsample_module(sd, "up_net.%d" % i, f, is_convtrans=(i == 0))
        nb, g = refine_module(sd, "refine_net.%d" % i, f, g, mask, x_id, y_id, knn_idx, ng[i], local_k, slope,
                              graph_slope)
        xb = torch.cat([xb, nb[:, 0:1]], dim=1)
        yb = torch.cat([yb, nb[:, 1:2]], dim=1)
        x_id = x_id * 2 + id_from_bit_prob(nb[:, 0:1])
        y_id = y_id * 2 + id_from_bit_prob(nb[:, 1:2])
        inter["up"].append(f); inter["graph"].append(g)
        inter["x_ids"].append(x_id); inter["y_ids"].append(y_id)
        if forced is not None and i + 1 < len(forced["x"]):
            x_id, y_id = forced["x"][i + 1], forced["y"][i + 1]
    seg = _conv(sd, "seg_block", f)
    return (roi, xb, yb, seg, x_id, y_id), inter


# --------------------------------------------------------------------------- post-forward decode (next-row N2)
def correspondences(roi, seg, x_id, y_id, roi_xy_ori, discard_bd_pixel=0):
    """Host-side correspondence extraction of the reference, per image: test.py:294-329 (sigmoid > 0.5 thresholds,
    seg channel 0 = visible, 1 = full) + test_network_with_test_data.py:50-59 (`disc_p2d = roi_xy_ori[y_id, x_id]`,
    `valid = roi_bit > 0.5 [and seg_mask[y_id, x_id] > 0.5]`).  Tensors: roi (B,1,N), seg (B,2,H,W), ids (B,N),
    roi_xy_ori (B,2,H,W).  `discard_bd_pixel` d > 0: `bd_mask[d:H-d, d:W-d] = 1`, keep only keypoints with
    `bd_mask[y_id, x_id] > 0.5` (:60-63).  Returns p2d (B,N,2), valid (B,N,3) uint8 [all, full, visible], count (B,3).
    Pinned by tests/golden/n2_from_id_to_pose.npz (the lists the reference's from_id_to_pose hands to its solver)."""
    B, _, N = roi.shape
    roi_bit = torch.where(torch.sigmoid(roi) > 0.5, 1.0, 0.0)[:, 0]                 # (B,N)
    segb = torch.where(torch.sigmoid(seg) > 0.5, 1.0, 0.0)
    grid = roi_xy_ori.permute(0, 2, 3, 1)                                            # (B,H,W,2)  test.py:327
    bi = torch.arange(B).view(B, 1).expand(B, N)
    p2d = grid[bi, y_id, x_id]                                                       # (B,N,2)
    v0 = roi_bit > 0.5
    if discard_bd_pixel > 0:
        d, (H, W) = discard_bd_pixel, seg.shape[2:]
        bd_mask = torch.zeros(H, W)
        bd_mask[d:H - d, d:W - d] = 1.0
        v0 = v0 & (bd_mask[y_id, x_id] > 0.5)
    v1 = v0 & (segb[:, 1][bi, y_id, x_id] > 0.5)
    v2 = v0 & (segb[:, 0][bi, y_id, x_id] > 0.5)
    valid = torch.stack([v0, v1, v2], dim=2).to(torch.uint8)
    return p2d, valid, valid.sum(dim=1).to(torch.int32)


# --------------------------------------------------------------------------- input side (next-row N3)
def preprocess_uint8(img_u8):
    """bop_dataset_pytorch.py:385-391: transforms.ToTensor() (uint8 HWC -> float CHW / 255) then
    transforms.Normalize((0.485,0.456,0.406),(0.229,0.224,0.225)) (sub mean, div std).  (B,H,W,3) u8 -> (B,3,H,W) f32."""
    x = img_u8.permute(0, 3, 1, 2).to(torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return x.sub(mean).div(std)
