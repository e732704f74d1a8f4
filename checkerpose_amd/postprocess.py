"""On-device post-forward decode: the 2D-3D correspondence list the reference builds on the
host in test.py:294-329 / test_network_with_test_data.py:from_id_to_pose :50-66, computed by one HIP kernel so that
only (B,N,2) floats + (B,N,3) validity bytes leave the GPU.  The reference's PnP (Progressive-X / cv2) stays untouched and can
consume these; `solve_pnp_ransac` is the opt-in on-device twin of its cv2 branch (row N4), after which 12 doubles per crop leave."""
import numpy as np
import torch

from . import _abi


def correspondences(outputs, roi_xy_ori=None, discard_bd_pixel=0, Bboxes=None):
    """outputs: the 6-tuple of PoseNet_GNNskip.forward (full 6+6 bits); roi_xy_ori: (B,2,H,W) fp32 CUDA tensor
    (the dataset's original-image coordinate grid of the crop, bop_dataset_pytorch.py) -- or, instead of it, Bboxes: the (B,4)
    final boxes (x, y, w, h) of the crops (`get_final_Bbox`, :188-222; host array or int32 CUDA tensor), from which the kernel
    builds the very same grid entries on the fly (16 bytes per crop travel instead of 32 KB).  Returns
    (p2d (B,N,2) f32, valid (B,N,3) uint8 [all | in full mask | in visible mask], count (B,3) int32).
    discard_bd_pixel: from_id_to_pose's border filter (test_network_with_test_data.py:60-63), 0 = off."""
    roi, xb, yb, seg, xid, yid = outputs
    if not roi.is_cuda or (roi_xy_ori is not None and not roi_xy_ori.is_cuda):
        raise RuntimeError("checkerpose_amd.postprocess: CUDA/HIP tensors required (no CPU fallback)")
    if (roi_xy_ori is None) == (Bboxes is None):
        raise ValueError("give either roi_xy_ori or Bboxes")
    B, _, N = roi.shape
    H, W = seg.shape[2], seg.shape[3]
    if roi_xy_ori is not None and tuple(roi_xy_ori.shape) != (B, 2, H, W):
        raise ValueError("roi_xy_ori must be (B,2,%d,%d)" % (H, W))
    lib = _abi.load()
    # the three logit views are slices of one (B,13,N) block when they come from the module; rebuild it otherwise
    base = roi._base if roi._base is not None and tuple(roi._base.shape) == (B, 13, N) else None
    bits = base if base is not None else torch.cat([roi, xb, yb], 1).contiguous()
    if bits.shape[1] != 13:
        raise ValueError("need the full 13-row logit block (all refinement stages active)")
    seg = seg.contiguous(); xid = xid.contiguous(); yid = yid.contiguous()
    p2d = torch.empty(B, N, 2, dtype=torch.float32, device=roi.device)
    valid = torch.empty(B, N, 3, dtype=torch.uint8, device=roi.device)
    count = torch.empty(B, 3, dtype=torch.int32, device=roi.device)
    st = torch.cuda.current_stream(roi.device).cuda_stream
    if Bboxes is not None:
        bb = Bboxes if torch.is_tensor(Bboxes) else torch.as_tensor(np.asarray(Bboxes), dtype=torch.int32)
        bb = bb.to(device=roi.device, dtype=torch.int32).contiguous()
        if tuple(bb.shape) != (B, 4):
            raise ValueError("Bboxes must be (B,4): x, y, w, h of every crop")
        _abi.check(lib.cp_correspondences_bbox(st, bits.data_ptr(), seg.data_ptr(), xid.data_ptr(), yid.data_ptr(), bb.data_ptr(),
                                               p2d.data_ptr(), valid.data_ptr(), count.data_ptr(), B, N, H, W, int(discard_bd_pixel)),
                   "cp_correspondences_bbox")
        return p2d, valid, count
    rxy = roi_xy_ori.contiguous().float()
    _abi.check(lib.cp_correspondences(st, bits.data_ptr(), seg.data_ptr(), xid.data_ptr(), yid.data_ptr(), rxy.data_ptr(),
                                      p2d.data_ptr(), valid.data_ptr(), count.data_ptr(), B, N, H, W, int(discard_bd_pixel)),
               "cp_correspondences")
    return p2d, valid, count


PNP_MAX_ITERS = 256        # csrc/pnp.hip: 4 rounds of 64 hypotheses


def solve_pnp_ransac(p3d_xyz, p2d, valid, cam_K, column=0, reproj_threshold=2.0, iterations=150, seed=0):
    """On-device twin of from_id_to_pose's cv2 branch (test_network_with_test_data.py:100-114; defaults reprojErr_thresh=2,
    cv_max_iters=150): EPnP + RANSAC over the correspondences of `correspondences()`.
      p3d_xyz (N,3) or (B,N,3) model keypoints in original units; p2d (B,N,2), valid (B,N,3) from correspondences();
      column 0 = all RoI keypoints | 1 = also inside the full mask | 2 = inside the visible mask (check_seg variants);
      cam_K (3,3) or (B,3,3).
    Returns (R (B,3,3) f64, t (B,3,1) f64, inliers (B,N) bool, status (B,) int32: 0 = the reference's identity fallback)."""
    if not (p2d.is_cuda and valid.is_cuda):
        raise RuntimeError("checkerpose_amd.postprocess: CUDA/HIP tensors required (no CPU fallback)")
    if not 0 < int(iterations) <= PNP_MAX_ITERS:        # no silent clamp: cv2 would run them all
        raise ValueError("iterations (cv_max_iters) must be in 1..%d for cp_pnp_ransac, got %r" % (PNP_MAX_ITERS, iterations))
    lib = _abi.load()
    dev = p2d.device
    B, N, _ = p2d.shape
    if tuple(valid.shape) != (B, N, 3) or valid.dtype != torch.uint8 or not 0 <= column < 3:
        raise ValueError("valid must be the (B,N,3) uint8 tensor of correspondences(), column in 0..2")
    p3 = torch.as_tensor(p3d_xyz, dtype=torch.float32, device=dev).contiguous()
    K = torch.as_tensor(cam_K, dtype=torch.float32, device=dev).contiguous()
    if p3.shape[-2:] != (N, 3) or K.shape[-2:] != (3, 3):
        raise ValueError("p3d_xyz must be (N,3) / (B,N,3) and cam_K (3,3) / (B,3,3)")
    p2 = p2d.contiguous().float()
    va = valid.contiguous()
    pose = torch.empty(B, 12, dtype=torch.float64, device=dev)
    inl = torch.empty(B, N, dtype=torch.uint8, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib.cp_pnp_ransac_scratch_bytes(B, N), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _abi.check(lib.cp_pnp_ransac(st, p3.data_ptr(), 3 * N if p3.dim() == 3 else 0, p2.data_ptr(), va.data_ptr() + column, 3, K.data_ptr(),
                                 9 if K.dim() == 3 else 0, B, N, float(reproj_threshold), int(iterations), int(seed) & 0xFFFFFFFF,
                                 pose.data_ptr(), inl.data_ptr(), status.data_ptr(), scratch.data_ptr()), "cp_pnp_ransac")
    return pose[:, :9].view(B, 3, 3), pose[:, 9:].view(B, 3, 1), inl.bool(), status


def estimate_poses(net, frames, Bboxes, p3d_xyz, cam_K, img_index=None, obj_ids=None, padding_ratio=1.5, crop_size=256,
                   resize_method="crop_square_resize", check_seg=False, discard_bd_pixel=0, reproj_threshold=2.0, iterations=150, seed=0):
    """The inner loop of the reference's test.py (:198-330) for a whole batch without leaving the GPU: detection boxes on full uint8
    frames -> padded RoI crops (`padding_Bbox` + `get_roi`, bop_dataset_pytorch.py:344-354; preprocess.get_roi_batch) -> network forward
    (uint8 input, normalised on the device) -> correspondences from the crops' final boxes (`get_final_Bbox`; N2) -> EPnP + RANSAC (N4).
      frames: uint8 CUDA tensor (n_img, H, W, 3) or (H, W, 3); Bboxes: (B, 4) detection boxes (x, y, w, h), None = no detection;
      p3d_xyz (N,3) / (B,N,3) model keypoints in original units; cam_K (3,3) / (B,3,3); obj_ids for the LM shared estimator.
    -> (R (B,3,3) f64, t (B,3,1) f64, inliers (B,N) bool, status (B,) int32 (0: identity fallback), final boxes (B,4) int array)"""
    from . import preprocess as PP
    if frames.dim() == 3:
        frames = frames.unsqueeze(0)
    H, W = int(frames.shape[1]), int(frames.shape[2])
    padded = [None if b is None else PP.padding_Bbox(b, padding_ratio) for b in Bboxes]
    crops = PP.get_roi_batch(frames, padded, crop_size, PP.INTER_LINEAR, resize_method, img_index=img_index)
    final = np.array([[0, 0, 0, 0] if b is None else PP.get_final_Bbox(b, resize_method, W, H) for b in padded], dtype=np.int32)
    with torch.no_grad():
        out = net(crops, None) if obj_ids is None else net(crops, None, obj_ids)
    p2d, valid, _ = correspondences(out, discard_bd_pixel=discard_bd_pixel, Bboxes=final)
    R, t, inl, status = solve_pnp_ransac(p3d_xyz, p2d, valid, cam_K, column=1 if check_seg else 0, reproj_threshold=reproj_threshold,
                                         iterations=iterations, seed=seed)
    return R, t, inl, status, final


def from_id_to_pose(p3d_xyz, roi_xy_ori, cam_K, roi_mask_bit, pixel_x_id, pixel_y_id, check_seg=False, seg_mask=None,
                    use_progressivex=False, neighborhood_ball_radius=20, spatial_coherence_weight=0.1, prog_max_iters=400,
                    discard_bd_pixel=0, return_inliers=False, reprojErr_thresh=2, cv_max_iters=150, device="cuda:0", seed=0):
    """Same name, arguments (numpy arrays of ONE image) and returns as the reference's `from_id_to_pose`
    (test_network_with_test_data.py:32-115), with its cv2 branch running on the device (`cp_pnp_ransac`):
      the validity mask is built exactly as :50-66 (RoI bit > 0.5, optional seg mask at the predicted pixel, optional border
      discard), then EPnP + RANSAC (reprojErr_thresh, cv_max_iters) -> R (3,3), t (3,1) [, inlier indices into ALL keypoints];
      fewer than 4 valid correspondences -> R = I, t = 0, inliers None (:111-114).
    `use_progressivex=True` is the third-party pyprogressivex solver of the reference and is not rebuilt: ValueError.
    For whole batches straight from the network's outputs use correspondences() + solve_pnp_ransac() instead."""
    import numpy as np
    if use_progressivex:
        raise ValueError("use_progressivex=True needs the third-party pyprogressivex solver; only the cv2 (EPnP + RANSAC) branch is built")
    num_all_pt = p3d_xyz.shape[0]
    roi_h, roi_w, _ = roi_xy_ori.shape
    disc_p2d = roi_xy_ori[pixel_y_id, pixel_x_id]
    valid_mask = (roi_mask_bit[:, 0] > 0.5)
    if check_seg:
        valid_mask = np.logical_and(valid_mask, seg_mask[pixel_y_id, pixel_x_id] > 0.5)
    if discard_bd_pixel > 0:
        bd_mask = np.zeros((roi_h, roi_w))
        bd_mask[discard_bd_pixel:(roi_h - discard_bd_pixel), discard_bd_pixel:(roi_w - discard_bd_pixel)] = 1.0
        valid_mask = np.logical_and(valid_mask, bd_mask[pixel_y_id, pixel_x_id] > 0.5)
    if int(valid_mask.sum()) < 4:
        R_predict, t_predict, inliers = np.eye(3), np.zeros((3, 1)), None
    else:
        dev = torch.device(device)
        valid = torch.zeros(1, num_all_pt, 3, dtype=torch.uint8, device=dev)
        valid[0, :, 0] = torch.from_numpy(np.ascontiguousarray(valid_mask)).to(dev)
        p2d = torch.from_numpy(np.ascontiguousarray(disc_p2d, dtype=np.float32)).to(dev)[None]
        R, t, inl, status = solve_pnp_ransac(torch.from_numpy(np.ascontiguousarray(p3d_xyz, dtype=np.float32)).to(dev), p2d, valid,
                                             torch.from_numpy(np.ascontiguousarray(cam_K, dtype=np.float32)).to(dev), column=0,
                                             reproj_threshold=float(reprojErr_thresh), iterations=int(cv_max_iters), seed=seed)
        if int(status[0]) == 1:
            R_predict, t_predict = R[0].cpu().numpy(), t[0].cpu().numpy()
            inliers = np.nonzero(inl[0].cpu().numpy())[0]
        else:                                     # no hypothesis with a full sample of inliers: cv2 reports failure; identity like :111-114
            R_predict, t_predict, inliers = np.eye(3), np.zeros((3, 1)), None
    if return_inliers:
        return R_predict, t_predict, inliers
    return R_predict, t_predict
