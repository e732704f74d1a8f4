"""On-device post-forward decode: the 2D-3D correspondence list the reference builds on the
host in test.py:294-329 / test_network_with_test_data.py:from_id_to_pose :50-66, computed by one HIP kernel so that
only (B,N,2) floats + (B,N,3) validity bytes leave the GPU.  PnP itself (Progressive-X / cv2) stays untouched."""
import torch

from . import _abi


def correspondences(outputs, roi_xy_ori, discard_bd_pixel=0):
    """outputs: the 6-tuple of PoseNet_GNNskip.forward (full 6+6 bits); roi_xy_ori: (B,2,H,W) fp32 CUDA tensor
    (the dataset's original-image coordinate grid of the crop, bop_dataset_pytorch.py).  Returns
    (p2d (B,N,2) f32, valid (B,N,3) uint8 [all | in full mask | in visible mask], count (B,3) int32).
    discard_bd_pixel: from_id_to_pose's border filter (test_network_with_test_data.py:60-63), 0 = off."""
    roi, xb, yb, seg, xid, yid = outputs
    if not (roi.is_cuda and roi_xy_ori.is_cuda):
        raise RuntimeError("checkerpose_amd.postprocess: CUDA/HIP tensors required (no CPU fallback)")
    B, _, N = roi.shape
    H, W = seg.shape[2], seg.shape[3]
    if tuple(roi_xy_ori.shape) != (B, 2, H, W):
        raise ValueError("roi_xy_ori must be (B,2,%d,%d)" % (H, W))
    lib = _abi.load()
    # the three logit views are slices of one (B,13,N) block when they come from the module; rebuild it otherwise
    base = roi._base if roi._base is not None and tuple(roi._base.shape) == (B, 13, N) else None
    bits = base if base is not None else torch.cat([roi, xb, yb], 1).contiguous()
    if bits.shape[1] != 13:
        raise ValueError("need the full 13-row logit block (all refinement stages active)")
    seg = seg.contiguous(); xid = xid.contiguous(); yid = yid.contiguous()
    rxy = roi_xy_ori.contiguous().float()
    p2d = torch.empty(B, N, 2, dtype=torch.float32, device=roi.device)
    valid = torch.empty(B, N, 3, dtype=torch.uint8, device=roi.device)
    count = torch.empty(B, 3, dtype=torch.int32, device=roi.device)
    st = torch.cuda.current_stream(roi.device).cuda_stream
    _abi.check(lib.cp_correspondences(st, bits.data_ptr(), seg.data_ptr(), xid.data_ptr(), yid.data_ptr(), rxy.data_ptr(),
                                      p2d.data_ptr(), valid.data_ptr(), count.data_ptr(), B, N, H, W, int(discard_bd_pixel)),
               "cp_correspondences")
    return p2d, valid, count
