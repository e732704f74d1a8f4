#!/usr/bin/env python3
"""Round-2 golden vectors, made by running the REFERENCE's own code in the build container (needs /root/reference;
nothing of the reference travels, only the small outputs of this script are committed):

  python tests/golden/make_golden_r2.py [lm4096] [n2] [sigmoid] [keys] [ycbv]

  lm4096   BASELINE config #5: LM 13-object shared estimator at npt=4096.  Keypoint data (15 x 4096 x 3, from the
           reference's fps pickles) -> checkerpose_amd/data/; `knn` (init.py:27) tables -> knn_lm4096.npz (3 objects in
           full + a checksum per object for all 15); PoseNet_GNNskip of pipeline_lm.py:392-425 end to end with injected
           backbone features -> e2e_lm4096_injected.npz.  Also the 21 YCB-V objects' first 512 keypoints (config #4).
  n2       the reference's own correspondence extraction: test.py:294-314 thresholds + from_id_to_pose
           (test_network_with_test_data.py:32-66) with cv2.solvePnPRansac stubbed to RECORD (valid_p3d, valid_disc_p2d),
           check_seg in {False, full, visib} x discard_bd_pixel in {0, 2} -> n2_from_id_to_pose.npz
  sigmoid  from_mask_prob_to_mask / from_code_prob_to_id / from_bit_prob_to_id (pipeline.py:84-127) swept over the
           logits around the largest fp32 z with sigmoid(z) == 0.5 -> sigmoid_threshold.npz
  keys     state-dict key/shape list of the reference module tree (head; backbone = stub) -> head_state_dict_keys.json
  ycbv     BASELINE config #4: reference `knn` tables of the 21 YCB-V objects at 512 keypoints -> knn_ycbv512.npz
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden as MG  # noqa: E402  (installs the timm stub, puts the reference on sys.path)
from make_golden import R_init, R_pipe, R_pipe_lm, _STUB, build_ref, decision_margin, inject_feats, load_fps, p3d, save  # noqa: E402

DATA = os.path.join(ROOT, "checkerpose_amd", "data")
LM_IDS = [1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15]          # test_network_with_test_data.py:533


def knn_checksum(idx):
    """order-sensitive 64-bit checksum of one (N,K) index table (known-answer for the objects not stored in full)"""
    a = idx.astype(np.uint64).reshape(-1)
    w = (np.arange(a.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(1000003)
    return int((a * w).sum() % np.uint64(1 << 61))


def do_lm4096():
    os.makedirs(DATA, exist_ok=True)
    ape = load_fps("lmo", 1)
    np.save(os.path.join(DATA, "fps_lmo_obj01.npy"), ape)                                    # (4096,3) f64
    lm = np.stack([load_fps("lm", o) for o in range(1, 16)]).astype(np.float32)              # (15,4096,3)
    np.save(os.path.join(DATA, "fps_lm_15x4096.npy"), lm)
    ycbv = np.stack([load_fps("ycbv", o)[:512] for o in range(1, 22)]).astype(np.float32)    # (21,512,3)
    np.save(os.path.join(DATA, "fps_ycbv_21x512.npy"), ycbv)
    lm_p3d = torch.cat([p3d(lm[o].astype(np.float64), 4096) for o in range(15)], 0)          # (15,3,4096)
    idx = R_init.knn(lm_p3d, 20).numpy()                                                     # (15,4096,20)
    save("knn_lm4096", objs=np.array([1, 9, 15]), idx=idx[[0, 8, 14]].astype(np.int16),
         checksum=np.array([knn_checksum(idx[o]) for o in range(15)], dtype=np.int64))
    # end to end, LM twin with per-sample graphs at N=4096 (config #5), features injected through the timm stub
    obj_ids = torch.tensor([5, 12])
    _STUB["feats"] = inject_feats(2, seed=2)
    net = build_ref(4096, lm_p3d, "inject", seed=0, lm=True)
    MG.save_e2e("e2e_lm4096_injected", net, lambda: net(torch.zeros(2, 3, 256, 256), lm_p3d[obj_ids - 1], obj_ids), seed=0,
                extra=lambda: {"obj_ids": obj_ids.numpy()})


def do_n2():
    """Import the reference's from_id_to_pose with the absent third-party modules stubbed; the stubbed solver records
    the correspondence list the reference hands to it."""
    rec = {}

    def solve(p3, p2, K, distCoeffs=None, reprojectionError=2, iterationsCount=150, flags=0):
        rec["p3d"], rec["p2d"] = np.array(p3), np.array(p2)
        return True, np.zeros((3, 1)), np.zeros((3, 1)), np.arange(p3.shape[0]).reshape(-1, 1)   # "all inliers"

    cv2 = types.ModuleType("cv2")
    cv2.solvePnPRansac, cv2.SOLVEPNP_EPNP = solve, 1
    cv2.Rodrigues = lambda r, jacobian=None: (np.eye(3), None)
    for name in ("sklearn", "sklearn.metrics", "tools_for_BOP", "tools_for_BOP.common_dataset_info", "metric", "tqdm",
                 "binary_code_helper", "binary_code_helper.class_id_encoder_decoder", "bop_toolkit_lib"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["sklearn.metrics"].f1_score = None
    sys.modules["tools_for_BOP.common_dataset_info"].get_obj_info = None
    sys.modules["metric"].Calculate_ADD_Error_BOP = sys.modules["metric"].Calculate_ADI_Error_BOP = None
    sys.modules["tqdm"].tqdm = None
    sys.modules["binary_code_helper.class_id_encoder_decoder"].class_code_vecs_to_class_id_vec = None
    sys.modules["bop_toolkit_lib"].pose_error = None
    sys.modules["cv2"] = cv2
    import test_network_with_test_data as T                                       # the reference's own module

    g = np.load(os.path.join(HERE, "e2e_injected.npz"))                             # reference-made forward outputs (B=2)
    B, N = 2, 512
    act = torch.nn.Sigmoid()                                                        # test.py:278 activation_function
    roi = torch.where(act(torch.from_numpy(g["roi"])) > 0.5, 1.0, 0.0).numpy().transpose(0, 2, 1)      # test.py:294-303
    # the random-init seg logits are ~95 % negative: re-centre each map on its median so that both masks are mixed
    seg_shift = np.median(g["seg"], axis=(2, 3), keepdims=True).astype(np.float32)
    seg = torch.where(act(torch.from_numpy(g["seg"] - seg_shift)) > 0.5, 1.0, 0.0)                     # test.py:313-314
    seg_visib, seg_full = seg[:, 0].numpy(), seg[:, 1].numpy()                                          # test.py:316-317
    xid, yid = g["xid"].astype(np.int64), g["yid"].astype(np.int64)
    from checkerpose_amd.detweights import det_tensor
    grid = (det_tensor("roi_xy", (B, 2, 64, 64), 300.0) + 320.0)
    roi_xy = grid.numpy().transpose(0, 2, 3, 1)                                     # test.py:327 (B,H,W,2)
    p3d_xyz = load_fps("lmo", 1)[:N]
    out = {"grid_name": "roi_xy", "grid_scale": 300.0, "grid_shift": 320.0, "seg_shift": seg_shift}
    for b in range(B):
        for cs, sm in (("all", None), ("full", seg_full), ("visib", seg_visib)):
            for bd in (0, 2):
                rec.clear()
                R, t, inl = T.from_id_to_pose(p3d_xyz=p3d_xyz, roi_xy_ori=roi_xy[b], cam_K=np.eye(3), roi_mask_bit=roi[b],
                                              pixel_x_id=xid[b], pixel_y_id=yid[b], check_seg=sm is not None,
                                              seg_mask=None if sm is None else sm[b], discard_bd_pixel=bd,
                                              return_inliers=True)
                key = "b%d_%s_bd%d" % (b, cs, bd)
                assert inl is not None and len(inl) >= 4, "fewer than 4 valid points: the solver stub was not called"
                assert np.array_equal(rec["p3d"], p3d_xyz[inl])
                out[key + "_idx"] = inl.astype(np.int16); out[key + "_p2d"] = rec["p2d"].astype(np.float32)
                print(key, len(out[key + "_idx"]))
    save("n2_from_id_to_pose", **out)


def do_sigmoid():
    lo, hi = 0, 0x3F800000
    one = lambda u: torch.tensor([u], dtype=torch.int32).view(torch.float32)   # noqa: E731
    while hi - lo > 1:                                                              # through the reference's own function
        mid = (lo + hi) // 2
        if bool(R_pipe.from_mask_prob_to_mask(one(mid).view(1, 1, 1).repeat(1, 1, 64)).all()):
            hi = mid
        else:
            lo = mid
    print("largest fp32 z with sigmoid(z) == 0.5: bits 0x%08x = %.9e" % (lo, float(one(lo))))
    u = np.concatenate([np.arange(lo - 300, lo + 300), np.arange(0, 8), np.arange(0x00800000 - 4, 0x00800000 + 4),
                        np.linspace(1, 0x34600000, 424).astype(np.int64)]).astype(np.int32)      # 0 .. 2e-7, dense at the edge
    z = torch.from_numpy(u).view(torch.float32)
    z = torch.cat([z, -z])                                                                        # 2048 logits incl. -0.0
    n = z.numel()
    mask = R_pipe.from_mask_prob_to_mask(z.view(1, 1, n))                                          # (1,1,n) f32
    ids3 = R_pipe.from_code_prob_to_id(torch.stack([z, z.flip(0), z.roll(7)]).view(1, 3, n))       # (1,n) MSB first
    bit = R_pipe.from_bit_prob_to_id(z.view(1, 1, n))
    assert bool((mask.view(-1) == (z > one(lo)).float()).all())
    save("sigmoid_threshold", z0_bits=np.int64(lo), z_bits=z.view(torch.int32).numpy(), mask=mask.numpy().astype(np.uint8),
         ids3=ids3.numpy().astype(np.int16), bit=bit.numpy().astype(np.uint8))


def do_ycbv():
    """BASELINE config #4: the reference's `knn` (init.py:27) on the first 512 FPS keypoints of each of the 21 YCB-V objects
    (datasets/BOP_DATASETS/ycbv/fps_202212/obj_0000NN.pkl): 3 objects in full + an order-sensitive checksum for all 21."""
    tabs = []
    for o in range(1, 22):
        idx = R_init.knn(p3d(load_fps("ycbv", o), 512), 20)[0].numpy()
        assert (idx[:, 0] == np.arange(512)).all()
        tabs.append(idx)
    save("knn_ycbv512", objs=np.array([1, 11, 21]), idx=np.stack([tabs[0], tabs[10], tabs[20]]).astype(np.int16),
         checksum=np.array([knn_checksum(t) for t in tabs], dtype=np.int64))


def do_keys():
    P512 = p3d(load_fps("lmo", 1), 512)
    _STUB["feats"] = inject_feats(1)
    net = build_ref(512, P512, "inject", seed=0)
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}                 # head only: the stub backbone has no parameters
    with open(os.path.join(HERE, "head_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("head_state_dict_keys.json: %d keys" % len(keys))


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    what = sys.argv[1:] or ["lm4096", "n2", "sigmoid", "keys", "ycbv"]
    for w in what:
        {"lm4096": do_lm4096, "n2": do_n2, "sigmoid": do_sigmoid, "keys": do_keys, "ycbv": do_ycbv}[w]()
