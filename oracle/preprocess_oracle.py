"""TEST INFRASTRUCTURE (CPU oracle) -- never imported by the product path.

Input-side crop of the reference's data loader (second half): `get_roi(x, Bbox, crop_size, interpolation,
resize_method)` of `bop_dataset_pytorch.py:132-145` with `resize_method` = `crop_square_resize` (:55-91, the configs' choice) or
`crop_resize` (:94-108), behind `padding_Bbox` (:147-163); `get_final_Bbox` (:188-222).

PARITY UNPINNED for the resize itself: the reference calls `cv2.resize`, a third-party dependency (opencv-python, unpinned in the
reference's requirements) that is absent from this image, so no golden vector can be produced here.  `resize_u8` restates OpenCV's
published 8-bit algorithm (modules/imgproc/src/resize.cpp, 4.x): INTER_LINEAR in fixed point -- coefficients
saturate_cast<short>(c * 2048) with round-half-even, source column clamped with fx reset at both borders, source rows clamped with
the coefficients kept, horizontal pass in int32, vertical pass `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2` --
and INTER_NEAREST (`min(floor(dx * scale), w - 1)`).  The window arithmetic (int() truncation towards zero, zero padding outside the
image) follows the reference's own lines and is pinned by a run of the reference's own crop functions (tests/golden/n3_windows.npz).
Cross-check of the resize (not a pin): torch's `interpolate` samples with the same half-pixel geometry in floating point;
tests/test_preprocess.py holds INTER_LINEAR to one grey level of it and INTER_NEAREST to equality.  Pure numpy, meant for small cases."""
import numpy as np

INTER_NEAREST, INTER_LINEAR = 0, 1          # cv2's values


def padding_bbox(bbox, padding_ratio):
    """bop_dataset_pytorch.py:147-163"""
    x1, y1, bw, bh = (float(v) for v in bbox)
    cx, cy = x1 + 0.5 * bw, y1 + 0.5 * bh
    pw, ph = int(bw * padding_ratio), int(bh * padding_ratio)
    return np.array([int(cx - pw / 2), int(cy - ph / 2), pw, ph])


def window(bbox, resize_method, img_w, img_h):
    """-> (x1, y1, x2, y2, roi_w, roi_h): roi pixel (ry, rx) is image pixel (y1 + ry, x1 + rx) where that lies in
    [max(x1, 0), min(x2, img_w)) x [max(y1, 0), min(y2, img_h)), zero elsewhere (crop_square_resize :55-91 / crop_resize :94-108)"""
    x1, y1, bw, bh = (v for v in bbox)
    x2, y2 = x1 + bw, y1 + bh
    if resize_method == "crop_square_resize":
        cx, cy = 0.5 * (x1 + x2), 0.5 * (y1 + y2)
        if bh > bw:
            x1, x2 = cx - bh / 2, cx + bh / 2
        else:
            y1, y2 = cy - bw / 2, cy + bw / 2
        side = int(max(bh, bw))
        return int(x1), int(y1), int(x2), int(y2), side, side
    if resize_method == "crop_resize":
        x1, y1, x2, y2 = int(max(x1, 0)), int(max(y1, 0)), int(min(x2, img_w)), int(min(y2, img_h))
        # `img[y1:y2, x1:x2]` (:106) with a negative end index counts from the far edge (Python slicing): reproduced, not "fixed"
        x2 = x2 if x2 >= 0 else max(img_w + x2, 0)
        y2 = y2 if y2 >= 0 else max(img_h + y2, 0)
        return x1, y1, x2, y2, max(x2 - x1, 0), max(y2 - y1, 0)
    raise NotImplementedError(resize_method)


def final_bbox(bbox, resize_method, max_x, max_y):
    """bop_dataset_pytorch.py:188-222"""
    if resize_method == "crop_resize":
        x1, y1 = int(max(bbox[0], 0)), int(max(bbox[1], 0))
        x2, y2 = int(min(bbox[0] + bbox[2], max_x)), int(min(bbox[1] + bbox[3], max_y))
    else:
        x1, y1, x2, y2, _, _ = window(bbox, resize_method, max_x, max_y)
    return np.array([x1, y1, x2 - x1, y2 - y1])


def roi(img, win):
    """the pre-resize window: roi pixel (ry, rx) = image pixel (y1 + ry, x1 + rx) inside the image and inside [x1, x2) x [y1, y2),
    zero elsewhere (crop_square_resize :78-89; pinned by tests/golden/n3_windows.npz, made by the reference's own function)"""
    x1, y1, x2, y2, rw, rh = win
    H, W = img.shape[:2]
    out = np.zeros((max(rh, 0), max(rw, 0)) + img.shape[2:], dtype=img.dtype)
    ya, yb = max(y1, 0), min(y2, H, y1 + rh)
    xa, xb = max(x1, 0), min(x2, W, x1 + rw)
    if yb > ya and xb > xa:
        out[ya - y1:yb - y1, xa - x1:xb - x1] = img[ya:yb, xa:xb]
    return out


def _coeffs(dst, src):
    """per destination index: (source index, short coefficient pair) of cv2's 8-bit INTER_LINEAR; columns: reset=True"""
    scale = 1.0 / (dst / src)                                  # cv2: scale_x = 1. / inv_scale_x, both doubles
    idx, a0, a1 = np.zeros(dst, np.int64), np.zeros(dst, np.int64), np.zeros(dst, np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        idx[d], a0[d], a1[d] = s, int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048.0))), int(np.rint(f * np.float32(2048.0)))
    return idx, a0, a1


def resize_u8(src, dsize_w, dsize_h, interpolation):
    src = np.asarray(src)
    assert src.dtype == np.uint8
    sh, sw = src.shape[:2]
    s3 = src.reshape(sh, sw, -1).astype(np.int64)
    out = np.zeros((dsize_h, dsize_w, s3.shape[2]), np.uint8)
    if interpolation == INTER_NEAREST:
        xs = [min(int(np.floor(d * (1.0 / (dsize_w / sw)))), sw - 1) for d in range(dsize_w)]
        ys = [min(int(np.floor(d * (1.0 / (dsize_h / sh)))), sh - 1) for d in range(dsize_h)]
        out[:] = src.reshape(sh, sw, -1)[np.array(ys)[:, None], np.array(xs)[None, :]]
        return out.reshape((dsize_h, dsize_w) + src.shape[2:])
    xi, xa0, xa1 = _coeffs(dsize_w, sw)
    yi, yb0, yb1 = _coeffs(dsize_h, sh)
    for d in range(dsize_w):                                  # columns: index clamped AND the fraction reset at both borders
        if xi[d] < 0:
            xi[d], xa0[d], xa1[d] = 0, 2048, 0
        if xi[d] >= sw - 1:
            xi[d], xa0[d], xa1[d] = sw - 1, 2048, 0
    hor = np.zeros((sh, dsize_w, s3.shape[2]), np.int64)
    for d in range(dsize_w):
        hor[:, d] = s3[:, xi[d]] * xa0[d] + s3[:, min(xi[d] + 1, sw - 1)] * xa1[d]
    for d in range(dsize_h):                                  # rows: indices clamped, coefficients kept
        r0, r1 = min(max(yi[d], 0), sh - 1), min(max(yi[d] + 1, 0), sh - 1)
        v = (((yb0[d] * (hor[r0] >> 4)) >> 16) + ((yb1[d] * (hor[r1] >> 4)) >> 16) + 2) >> 2
        out[d] = np.clip(v, 0, 255).astype(np.uint8)
    return out.reshape((dsize_h, dsize_w) + src.shape[2:])


def get_roi(img, bbox, crop_size, interpolation, resize_method):
    """bop_dataset_pytorch.py:132-145 (crop_resize_by_warp_affine is not used by any config)"""
    win = window(bbox, resize_method, img.shape[1], img.shape[0])
    return resize_u8(roi(img, win), crop_size, crop_size, interpolation)


def normalise(roi_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """transform_pre, bop_dataset_pytorch.py:385-398: ToTensor (/255) then Normalize -> (3, H, W) float32"""
    x = roi_u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
