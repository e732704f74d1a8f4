"""Input side on the device: the RoI crops of the reference's data loader as ONE HIP launch per batch.

The reference's `bop_dataset_pytorch.py` crops every detection on the host: `padding_Bbox` (:147-163) -> `get_roi(x, Bbox,
crop_size, interpolation, resize_method)` (:132-145: `crop_square_resize` :55-91 in every config, or `crop_resize` :94-108, then
`cv2.resize`) -> `get_final_Bbox` (:188-222) -> `transform_pre` (:385-398).  Here the full uint8 images go to the GPU once and
`get_roi_batch` cuts all crops of a batch there (`cp_crop_resize_u8`, csrc/preprocess.hip); the uint8 crops feed the model's uint8
forward (`net(crops_u8, ...)`: ToTensor + Normalize on the device, `cp_u8hwc_to_nhwc_norm`).  The window arithmetic (a handful of
integer operations per box, on the host) is the reference's; the resize is cv2's 8-bit arithmetic (see the kernel's header).
No CPU fallback: CPU tensors raise."""
import numpy as np
import torch

from . import _abi

INTER_NEAREST, INTER_LINEAR = 0, 1          # cv2.INTER_NEAREST / cv2.INTER_LINEAR


def padding_Bbox(Bbox, padding_ratio):
    """the detection box (x, y, w, h) grown about its centre (bop_dataset_pytorch.py:147-163) -> int array (x, y, w, h)"""
    x, y, w, h = (float(v) for v in Bbox)
    pw, ph = int(w * padding_ratio), int(h * padding_ratio)
    return np.array([int(x + 0.5 * w - pw / 2), int(y + 0.5 * h - ph / 2), pw, ph])


def roi_window(Bbox, resize_method, img_w, img_h):
    """(x1, y1, x2, y2, roi_w, roi_h) of cp_crop_resize_u8 for one box: the square about the box centre, cut with int() as the
    reference does (crop_square_resize :55-77), or the box clamped to the image (crop_resize :94-106)"""
    x1, y1, bw, bh = (v for v in Bbox)
    x2, y2 = x1 + bw, y1 + bh
    if resize_method == "crop_square_resize":
        if bh > bw:
            c = 0.5 * (x1 + x2)
            x1, x2 = c - bh / 2, c + bh / 2
        else:
            c = 0.5 * (y1 + y2)
            y1, y2 = c - bw / 2, c + bw / 2
        side = int(max(bh, bw))
        return int(x1), int(y1), int(x2), int(y2), side, side
    if resize_method == "crop_resize":
        x1, y1, x2, y2 = _clamped_box(x1, y1, x2, y2, img_w, img_h)
        # the reference cuts `img[y1:y2, x1:x2]` (:106): a NEGATIVE end (box wholly left of / above the frame) is Python's
        # "counted from the far edge", so such a box yields the frame minus a margin, not an empty crop (n3_windows.npz pins it)
        x2 = x2 if x2 >= 0 else max(img_w + x2, 0)
        y2 = y2 if y2 >= 0 else max(img_h + y2, 0)
        return x1, y1, x2, y2, max(x2 - x1, 0), max(y2 - y1, 0)
    raise NotImplementedError("unknown decoder type: %s" % resize_method)      # the reference's message (:145)


def _clamped_box(x1, y1, x2, y2, img_w, img_h):
    """crop_resize :94-104 / get_final_Bbox :212-220"""
    return int(max(x1, 0)), int(max(y1, 0)), int(min(x2, img_w)), int(min(y2, img_h))


def get_final_Bbox(Bbox, resize_method, max_x, max_y):
    """the box the crop actually covers (bop_dataset_pytorch.py:188-222), what the post-processing maps pixel codes back with
    (crop_resize: the clamped corners as they are, also where x2 < x1 -- the reference's own result for a box off the frame)"""
    if resize_method == "crop_resize":
        x1, y1, x2, y2 = _clamped_box(Bbox[0], Bbox[1], Bbox[0] + Bbox[2], Bbox[1] + Bbox[3], max_x, max_y)
    else:
        x1, y1, x2, y2, _, _ = roi_window(Bbox, resize_method, max_x, max_y)
    return np.array([x1, y1, x2 - x1, y2 - y1])


def get_roi_batch(images, Bboxes, crop_size, interpolation=INTER_LINEAR, resize_method="crop_square_resize", img_index=None, out=None):
    """`get_roi` (bop_dataset_pytorch.py:132-145) for a batch, on the GPU.
    images: uint8 CUDA tensor (n_img, H, W, C) or (H, W, C) (C <= 4, the loader's cv2.imread layout); Bboxes: (B, 4) boxes
    (x, y, w, h), ALREADY padded / augmented as the loader does before get_roi; None entries give a zero crop (the loader's dummy
    input for a missing detection, :325-337); img_index: for each box the image it is cut from (default: image b, or the only
    image).  -> uint8 CUDA tensor (B, crop_size, crop_size, C)."""
    if not (torch.is_tensor(images) and images.is_cuda and images.dtype == torch.uint8):
        raise RuntimeError("checkerpose_amd.preprocess: images must be a uint8 CUDA tensor (no CPU path)")
    if images.dim() == 3:
        images = images.unsqueeze(0)
    if images.dim() != 4 or images.shape[3] > 4 or not images.is_contiguous():
        raise RuntimeError("checkerpose_amd.preprocess: images (n_img, H, W, C <= 4), contiguous")
    if interpolation not in (INTER_NEAREST, INTER_LINEAR):
        raise NotImplementedError("interpolation %r: cv2.INTER_NEAREST (0) and cv2.INTER_LINEAR (1) are built" % (interpolation,))
    n_img, H, W, C_ = (int(v) for v in images.shape)
    B = len(Bboxes)
    win = np.zeros((B, 6), dtype=np.int32)
    for b, box in enumerate(Bboxes):
        if box is not None:
            win[b] = roi_window([int(v) for v in box], resize_method, W, H)
    if img_index is None:
        if n_img not in (1, B):
            raise RuntimeError("checkerpose_amd.preprocess: img_index is needed when %d boxes come from %d images" % (B, n_img))
        idx_t = None
    else:
        idx_np = np.asarray(img_index, dtype=np.int32).reshape(-1)
        if idx_np.shape[0] != B or (B and (idx_np.min() < 0 or idx_np.max() >= n_img)):
            raise RuntimeError("checkerpose_amd.preprocess: img_index must hold one valid image number per box")
        idx_t = torch.from_numpy(idx_np).to(images.device)
    if out is None:
        out = torch.empty(B, crop_size, crop_size, C_, dtype=torch.uint8, device=images.device)
    elif tuple(out.shape) != (B, crop_size, crop_size, C_) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != images.device:
        raise RuntimeError("checkerpose_amd.preprocess: out must be a contiguous uint8 (B, crop, crop, C) tensor on the images' device")
    if B == 0:
        return out
    win_t = torch.from_numpy(win).to(images.device)
    lib = _abi.load()
    st = torch.cuda.current_stream(images.device).cuda_stream
    _abi.check(lib.cp_crop_resize_u8(st, images.data_ptr(), n_img, H, W, C_, win_t.data_ptr(), idx_t.data_ptr() if idx_t is not None else None,
                                     out.data_ptr(), B, int(crop_size), int(interpolation)), "cp_crop_resize_u8")
    return out                                # (win_t / idx_t may be freed: torch's allocator reuses a block in stream order)
