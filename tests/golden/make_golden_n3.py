#!/usr/bin/env python3
"""Row N3 (input side) pinned by the REFERENCE's own code: the window arithmetic of the data loader's crop.

Runs ONLY in the build container (needs /root/reference; nothing of the reference travels, only n3_windows.npz is committed):

  python tests/golden/make_golden_n3.py

`bop_dataset_pytorch.py` is imported with the third-party modules this image lacks stubbed (`cv2`, `mmcv`, `imageio`, `torchvision`,
`imgaug` through `GDR_Net_Augmentation`; the same trick make_golden_r2.py uses for `from_id_to_pose`).  Its own functions then run
over 240 boxes (random ones in and partly outside a 640 x 480 frame, boxes wholly outside, 1-pixel boxes, the hand-worked boxes of
tests/test_preprocess.py):
  * `padding_Bbox(Bbox, 1.5)`                                         (:147-163)
  * `get_final_Bbox(padded, method, W, H)` for both crop methods      (:188-222)
  * `crop_square_resize(img, padded, 256, INTER_LINEAR)` (:55-91) and `crop_resize(..)` (:94-108) with `cv2.resize` replaced by a
    RECORDER of the array it is handed -- the pre-resize window.  The frame is a position-coded uint8 image (pixel (y, x) holds
    (x % 256, y % 256, 16 (x // 256) + y // 256 + 1)), so the recorded window says which frame pixel every window pixel came from
    and which ones are zero padding.  Stored per box: its shape, the CRC-32 of its bytes, the count of non-zero pixels.
    Boxes on which the reference itself raises (numpy broadcast error: the box lies outside the frame; the loader prints "fail to
    get_roi" there, :297-322) are flagged `raised` and carry no window.
  * `mapping_pixel_position_to_original_position_2d(roi_xy, final_box, 64)` (:223-235) on the loader's own `roi_xy` grid
    (:266-269), cast to float32 as :380 does, for 6 final boxes -> the `roi_xy_ori` grids cp_correspondences_bbox rebuilds on the fly.
"""
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "checkerpose"))

REC = {}


def _resize_recorder(src, dsize, interpolation=None, **kw):
    REC["src"] = np.array(src, copy=True)
    REC["dsize"], REC["interp"] = tuple(dsize), interpolation
    return np.zeros((dsize[1], dsize[0]) + src.shape[2:], dtype=src.dtype)


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
    cv2.resize = _resize_recorder
    sys.modules["cv2"] = cv2
    for name in ("mmcv", "imageio", "torchvision", "torchvision.transforms", "imgaug", "imgaug.augmenters", "GDR_Net_Augmentation",
                 "binary_code_helper", "binary_code_helper.class_id_encoder_decoder"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["GDR_Net_Augmentation"].get_affine_transform = None
    sys.modules["binary_code_helper.class_id_encoder_decoder"].class_id_vec_to_class_code_vecs = None


def coded_frame(H, W):
    """uint8 (H, W, 3): every pixel names its own position; never all-zero, so zero = padding"""
    y, x = np.mgrid[0:H, 0:W]
    return np.stack([x % 256, y % 256, 16 * (x // 256) + y // 256 + 1], -1).astype(np.uint8)


def boxes_under_test(W, H):
    rng = np.random.default_rng(20261004)
    bx = [[10, 12, 21, 30], [4, 0, 13, 26], [8, 20, 40, 10], [-5, 30, 30, 40], [50, 40, 30, 20], [0, 0, W, H], [20, 10, 1, 1], [30, 20, 7, 3]]
    for _ in range(150):                                    # detections inside the frame
        w, h = int(rng.integers(8, 260)), int(rng.integers(8, 260))
        bx.append([int(rng.integers(0, W - 8)), int(rng.integers(0, H - 8)), w, h])
    for _ in range(60):                                     # leaving the frame on any side (negative corners, beyond the far edges)
        w, h = int(rng.integers(4, 400)), int(rng.integers(4, 400))
        bx.append([int(rng.integers(-300, W + 100)), int(rng.integers(-300, H + 100)), w, h])
    bx += [[W + 50, 10, 40, 40], [10, H + 80, 30, 60], [-400, -400, 20, 30], [W - 1, H - 1, 1, 1], [0, 0, 1, 1], [W - 3, 5, 9, 200],
           [5, H - 2, 300, 6], [-1, -1, 2, 2], [100, 100, 0, 0], [100, 100, 1, 0], [100, 100, 0, 1], [-20, 200, 19, 19], [-20, 200, 20, 20],
           [-20, 200, 21, 21], [W - 10, 200, 9, 40], [W - 10, 200, 10, 40], [W - 10, 200, 11, 40], [300, -7, 40, 6], [300, -7, 40, 7],
           [300, -7, 40, 8], [319, 239, 2, 2], [0, 0, 641, 481]]
    return np.array(bx, dtype=np.int64)


def main():
    install_stubs()
    import bop_dataset_pytorch as D                                             # the reference's own module
    W, H, CROP = 640, 480, 256
    frame = coded_frame(H, W)
    raw = boxes_under_test(W, H)
    n = len(raw)
    padded = np.zeros((n, 4), np.int64)
    out = {"frame_hw": np.array([H, W]), "raw": raw, "crop_size": np.int64(CROP)}
    for m in ("crop_square_resize", "crop_resize"):
        final = np.zeros((n, 4), np.int64)
        shape = np.zeros((n, 2), np.int64)
        crc = np.zeros(n, np.int64)
        nnz = np.zeros(n, np.int64)
        raised = np.zeros(n, np.uint8)
        for i, b in enumerate(raw):
            padded[i] = D.padding_Bbox(np.array(b), 1.5)
            final[i] = D.get_final_Bbox(padded[i].copy(), m, W, H)
            REC.clear()
            try:
                D.get_roi(frame, padded[i].copy(), CROP, interpolation=1, resize_method=m)
                src = REC["src"]
                assert REC["dsize"] == (CROP, CROP) and REC["interp"] == 1
                if src.size == 0:                           # crop_resize of a box outside the frame: cv2.resize would assert on an empty source
                    raised[i] = 2
                    continue
                shape[i] = src.shape[:2]
                crc[i] = zlib.crc32(np.ascontiguousarray(src).tobytes())
                nnz[i] = int(src.any(-1).sum())
            except ValueError:                              # numpy broadcast error inside crop_square_resize: the window misses the frame
                raised[i] = 1
        k = "sq" if m == "crop_square_resize" else "cr"
        out.update({k + "_final": final, k + "_shape": shape, k + "_crc": crc, k + "_nnz": nnz, k + "_raised": raised})
        print("%-20s %d boxes, %d raised in the reference, %d empty sources" % (m, n, int((raised == 1).sum()), int((raised == 2).sum())))
    out["padded"] = padded
    # the loader's coordinate grid of a crop (:266-269 + :223-235 + the float32 cast of :380)
    S = 64
    pix = np.linspace(0, S - 1, S)
    roi_xy = np.asarray(np.meshgrid(pix, pix)).transpose((1, 2, 0))
    pick = [0, 1, 3, 9, 170, 200]
    grids = np.stack([D.mapping_pixel_position_to_original_position_2d(roi_xy, out["sq_final"][i], S) for i in pick])
    out["grid_boxes"] = out["sq_final"][pick]
    out["grid_xy_ori"] = grids.astype(np.float32).transpose(0, 3, 1, 2)         # (6, 2, 64, 64) as the loader returns it (:380)
    path = os.path.join(HERE, "n3_windows.npz")
    np.savez_compressed(path, **out)
    print("wrote n3_windows.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
