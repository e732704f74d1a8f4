"""Row N3, second half: the data loader's RoI crop (bop_dataset_pytorch.py:132-145) on the device.  CPU: the oracle's restatement of
cv2's 8-bit resize against hand-checked values and invariants, the host-side window arithmetic against the reference's formulas
worked by hand; GPU: cp_crop_resize_u8 bit-exact against the oracle.  (cv2 is absent from the image: parity with it is unpinned.)"""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO
from checkerpose_amd import preprocess as PP


def _img(h, w, c=3, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, c), dtype=np.uint8)


def test_resize_oracle_known_values_and_invariants():
    img = _img(48, 64)
    assert np.array_equal(PO.resize_u8(img, 64, 48, PO.INTER_LINEAR), img)                      # same size: every coefficient is (2048, 0)
    assert np.array_equal(PO.resize_u8(img, 64, 48, PO.INTER_NEAREST), img)
    assert np.unique(PO.resize_u8(np.full((20, 30, 3), 137, np.uint8), 256, 256, PO.INTER_LINEAR)).tolist() == [137]
    # cv2.resize(np.uint8([[0, 100]]), (4, 1)) = [[0, 25, 75, 100]]: centres at -0.25, 0.25, 0.75, 1.25 -> clamped, 1/4, 3/4, clamped
    assert PO.resize_u8(np.array([[0, 100]], np.uint8), 4, 1, PO.INTER_LINEAR).tolist() == [[0, 25, 75, 100]]
    # 2:1 reduction samples between pixel pairs (no area averaging): (10 + 20) / 2, (30 + 40) / 2
    assert PO.resize_u8(np.array([[10, 20, 30, 40]], np.uint8), 2, 1, PO.INTER_LINEAR).tolist() == [[15, 35]]
    assert PO.resize_u8(np.array([[10, 20, 30, 40]], np.uint8), 2, 1, PO.INTER_NEAREST).tolist() == [[10, 30]]
    # rounding: 0.25 * 255 + 0.75 * 0 = 63.75 -> 64
    assert PO.resize_u8(np.array([[255, 0]], np.uint8), 4, 1, PO.INTER_LINEAR).tolist() == [[255, 191, 64, 0]]


def test_resize_oracle_against_torch_interpolate():
    """Independent cross-check of the UNPINNED resize restatement (cv2 is absent): torch's `interpolate` implements the same sampling
    geometry -- half-pixel centres, source index clamped at the borders, no antialiasing; legacy `nearest` = floor(dst * scale) -- in
    floating point.  cv2's 8-bit INTER_LINEAR differs from it only by its 11-bit coefficients and the final rounding: at most one
    grey level.  INTER_NEAREST must agree exactly.  Up- and down-scaling, non-integer ratios, the loader's 256 x 256 target."""
    import torch.nn.functional as F
    for (h, w, oh, ow, seed) in ((48, 64, 256, 256, 1), (300, 217, 256, 256, 2), (97, 131, 64, 80, 3), (31, 17, 256, 256, 4), (640, 480, 256, 256, 5)):
        img = _img(h, w, seed=seed)
        x = torch.from_numpy(img).permute(2, 0, 1)[None].float()
        lin = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        got = PO.resize_u8(img, ow, oh, PO.INTER_LINEAR).astype(np.float64)
        d = np.abs(got - lin)
        assert d.max() <= 1.0 + 1e-3, (h, w, oh, ow, float(d.max()))
        assert d.mean() <= 0.3, float(d.mean())
        near = F.interpolate(x, size=(oh, ow), mode="nearest")[0].permute(1, 2, 0).numpy().astype(np.uint8)
        assert np.array_equal(PO.resize_u8(img, ow, oh, PO.INTER_NEAREST), near), (h, w, oh, ow)


def test_window_arithmetic_follows_the_reference_formulas():
    """padding_Bbox :147-163 and the square window of crop_square_resize :55-77 / get_final_Bbox :188-207, worked by hand"""
    # box (10, 12, 21, 30), ratio 1.5: padded w = int(31.5) = 31, h = 45; centre (20.5, 27) -> (int(5.0), int(4.5), 31, 45)
    b = PP.padding_Bbox([10, 12, 21, 30], 1.5)
    assert b.tolist() == [5, 4, 31, 45] and PO.padding_bbox([10, 12, 21, 30], 1.5).tolist() == b.tolist()
    # taller than wide: x range becomes centre +- h/2 = 20.5 +- 22.5 = (-2.0, 43.0); side 45
    assert PP.roi_window(b, "crop_square_resize", 64, 48) == (-2, 4, 43, 49, 45, 45)
    assert PP.get_final_Bbox(b, "crop_square_resize", 64, 48).tolist() == [-2, 4, 45, 45]
    # int() truncates towards zero: centre 10.5, h 26 -> x1 = -2.5 -> -2, x2 = 23.5 -> 23 (25 columns for a 26-wide roi)
    assert PP.roi_window([4, 0, 13, 26], "crop_square_resize", 64, 48) == (-2, 0, 23, 26, 26, 26)
    # wider than tall
    assert PP.roi_window([8, 20, 40, 10], "crop_square_resize", 64, 48) == (8, 5, 48, 45, 40, 40)
    assert PP.roi_window([-5, 30, 30, 40], "crop_resize", 64, 48) == (0, 30, 25, 48, 25, 18)
    assert PP.get_final_Bbox([-5, 30, 30, 40], "crop_resize", 64, 48).tolist() == [0, 30, 25, 18]
    for box in ([5, 4, 31, 45], [4, 0, 13, 26], [8, 20, 40, 10], [-5, 30, 30, 40]):
        for m in ("crop_square_resize", "crop_resize"):
            assert PP.roi_window(box, m, 64, 48) == PO.window(box, m, 64, 48)
    with pytest.raises(NotImplementedError):
        PP.roi_window([0, 0, 4, 4], "crop_resize_by_warp_affine", 64, 48)


def _coded_frame(H, W):
    """the position-coded frame of tests/golden/make_golden_n3.py: pixel (y, x) = (x % 256, y % 256, 16 (x // 256) + y // 256 + 1)"""
    y, x = np.mgrid[0:H, 0:W]
    return np.stack([x % 256, y % 256, 16 * (x // 256) + y // 256 + 1], -1).astype(np.uint8)


def _n3():
    import os
    import zlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "n3_windows.npz"))
    return g, zlib.crc32


def test_window_arithmetic_matches_a_run_of_the_reference():
    """n3_windows.npz = the reference's OWN padding_Bbox / get_final_Bbox / crop_square_resize / crop_resize
    (bop_dataset_pytorch.py:55-108,147-163,188-222) run over 240 boxes (inside, across every edge, outside a 640 x 480 frame, 0- and
    1-pixel boxes) with cv2.resize replaced by a recorder of the pre-resize window on a position-coded frame.  The product's host
    arithmetic (preprocess.py) and the oracle's restatement must reproduce every padded box, every final box, and every window: its
    shape, which frame pixel each window pixel came from and which are zero padding (CRC-32 of the window's bytes)."""
    g, crc32 = _n3()
    H, W = (int(v) for v in g["frame_hw"])
    frame = _coded_frame(H, W)
    n = len(g["raw"])
    assert n >= 200
    windows = 0
    for i in range(n):
        raw, padded = g["raw"][i].tolist(), g["padded"][i].tolist()
        assert PP.padding_Bbox(raw, 1.5).tolist() == padded and PO.padding_bbox(raw, 1.5).tolist() == padded, raw
        for k, m in (("sq", "crop_square_resize"), ("cr", "crop_resize")):
            assert PP.get_final_Bbox(padded, m, W, H).tolist() == g[k + "_final"][i].tolist(), (raw, m)
            assert PO.final_bbox(padded, m, W, H).tolist() == g[k + "_final"][i].tolist(), (raw, m)
            win = PP.roi_window(padded, m, W, H)
            assert win == PO.window(padded, m, W, H)
            if g[k + "_raised"][i] == 1:
                # the reference raises inside crop_square_resize (its window misses the frame; the loader then has no crop, :297-322);
                # the kernel's contract there is a window WITHOUT any frame pixel = an all-zero crop
                assert not PO.roi(frame, win).any(), (raw, m)
                continue
            if g[k + "_raised"][i] == 2:                    # empty source handed to cv2.resize (which asserts): an empty window here too
                assert win[4] <= 0 or win[5] <= 0 or not PO.roi(frame, win).size, (raw, m)
                continue
            r = PO.roi(frame, win)
            assert list(r.shape[:2]) == g[k + "_shape"][i].tolist(), (raw, m, r.shape)
            assert int(r.any(-1).sum()) == int(g[k + "_nnz"][i]) and crc32(np.ascontiguousarray(r).tobytes()) == int(g[k + "_crc"][i]), (raw, m)
            windows += 1
    assert windows >= 400
    # the loader's coordinate grid of a crop (mapping_pixel_position_to_original_position_2d :223-235 on roi_xy :266-269, float32 :380)
    pix = np.linspace(0, 63, 64)
    gx, gy = np.meshgrid(pix, pix)
    for b, want in zip(g["grid_boxes"], g["grid_xy_ori"]):
        mine = np.stack([b[2] / 64 * gx + b[0], b[3] / 64 * gy + b[1]]).astype(np.float32)
        assert np.array_equal(mine, want)


def test_roi_zero_padding_outside_the_image():
    img = _img(48, 64)
    win = PO.window([5, 4, 31, 45], "crop_square_resize", 64, 48)          # (-2, 4, 43, 49, 45, 45): leaves the image left and below
    r = PO.roi(img, win)
    assert r.shape == (45, 45, 3) and not r[:, :2].any() and not r[44:].any()          # columns -2, -1 and row 48
    assert np.array_equal(r[:44, 2:], img[4:48, 0:43])


def test_get_roi_batch_refuses_cpu_tensors():
    with pytest.raises(RuntimeError, match="uint8 CUDA tensor"):
        PP.get_roi_batch(torch.zeros(8, 8, 3, dtype=torch.uint8), [[0, 0, 4, 4]], 4)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["crop_square_resize", "crop_resize"])
@pytest.mark.parametrize("interp", [PP.INTER_LINEAR, PP.INTER_NEAREST])
def test_device_crops_equal_the_oracle_bit_for_bit(method, interp):
    imgs = np.stack([_img(48, 64, 3, seed=s) for s in (1, 2)])
    boxes = [[5, 4, 31, 45], [4, 0, 13, 26], [8, 20, 40, 10], [-5, 30, 30, 40], [50, 40, 30, 20], None, [0, 0, 64, 48], [20, 10, 1, 1],
             [30, 20, 7, 3]]
    idx = [0, 1, 0, 1, 1, 0, 0, 1, 0]
    for crop in (32, 20):
        got = PP.get_roi_batch(torch.from_numpy(imgs).cuda(), boxes, crop, interp, method, img_index=idx).cpu().numpy()
        for b, (box, im) in enumerate(zip(boxes, idx)):
            if box is None:
                assert not got[b].any()
                continue
            win = PO.window(box, method, 64, 48)
            if win[4] <= 0 or win[5] <= 0:
                assert not got[b].any()
                continue
            want = PO.resize_u8(PO.roi(imgs[im], win), crop, crop, interp)
            assert np.array_equal(got[b], want), (method, interp, crop, box)
    mask = _img(48, 64, 1, seed=3)                                             # a visibility mask: one channel, nearest (loader :310)
    got = PP.get_roi_batch(torch.from_numpy(mask).cuda(), [[5, 4, 31, 45]], 16, PP.INTER_NEAREST, method).cpu().numpy()
    assert np.array_equal(got[0], PO.get_roi(mask, [5, 4, 31, 45], 16, PO.INTER_NEAREST, method))


@pytest.mark.gpu
def test_device_windows_equal_the_reference_recorded_windows():
    """cp_crop_resize_u8's pre-resize window against the windows the REFERENCE's crop_square_resize handed to cv2.resize
    (n3_windows.npz): with crop_size = the window's side the resize is the identity (INTER_NEAREST: source index = destination
    index; INTER_LINEAR: every coefficient pair is (2048, 0)), so the crop IS the window -- same CRC-32 as the recorded one, for every
    box the reference produced a window for; where the reference raised (window off the frame) the crop is all zero."""
    g, crc32 = _n3()
    H, W = (int(v) for v in g["frame_hw"])
    frame = torch.from_numpy(_coded_frame(H, W)).cuda()
    by_side = {}
    for i in range(len(g["raw"])):
        if g["sq_raised"][i] == 2:
            continue
        side = int(max(g["padded"][i][2], g["padded"][i][3]))
        by_side.setdefault(side, []).append(i)
    checked = 0
    for side, ids in sorted(by_side.items()):
        for interp in (PP.INTER_NEAREST, PP.INTER_LINEAR):
            got = PP.get_roi_batch(frame, [g["padded"][i].tolist() for i in ids], side, interp, "crop_square_resize").cpu().numpy()
            for j, i in enumerate(ids):
                if g["sq_raised"][i] == 1:
                    assert not got[j].any()
                    continue
                assert list(got[j].shape[:2]) == g["sq_shape"][i].tolist()
                assert crc32(np.ascontiguousarray(got[j]).tobytes()) == int(g["sq_crc"][i]), (g["raw"][i], interp)
                checked += 1
    assert checked >= 400
    # crop_resize (:94-108; rectangular windows, so no identity resize): the kernel against the oracle, whose windows the fixture pins
    # on the CPU -- incl. the boxes wholly left of / above the frame, where the reference's `img[y1:y2, x1:x2]` counts a negative end
    # from the far edge
    fr = _coded_frame(H, W)
    wrap = [i for i in range(len(g["raw"])) if g["cr_raised"][i] == 0 and (g["padded"][i][0] + g["padded"][i][2] < 0 or g["padded"][i][1] + g["padded"][i][3] < 0)]
    assert len(wrap) >= 3
    ids = wrap[:6] + [i for i in range(8, 240, 23) if g["cr_raised"][i] == 0]
    got = PP.get_roi_batch(frame, [g["padded"][i].tolist() for i in ids], 64, PP.INTER_LINEAR, "crop_resize").cpu().numpy()
    for j, i in enumerate(ids):
        assert np.array_equal(got[j], PO.get_roi(fr, g["padded"][i].tolist(), 64, PO.INTER_LINEAR, "crop_resize")), g["raw"][i]
    # ... and the coordinate grid cp_correspondences_bbox rebuilds from a final box == the loader's own roi_xy_ori
    from checkerpose_amd.postprocess import correspondences
    from checkerpose_amd.synthetic import build_net, det_image
    net = build_net(seed=1).cuda()
    nb = len(g["grid_boxes"])
    out = net(det_image(nb, seed=5).cuda(), None)
    want = correspondences(out, torch.from_numpy(g["grid_xy_ori"]).cuda())
    got = correspondences(out, Bboxes=g["grid_boxes"])
    for a, b in zip(want, got):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_device_crops_at_full_size_and_into_the_model():
    """640 x 480 frames, 64 detections, 256 x 256 crops: windows of exactly 256 x 256 pixels inside the image come out as the image
    patch itself, and the model's uint8 forward on device-made crops equals its forward on the same crops uploaded from the host"""
    from checkerpose_amd.synthetic import build_net
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (4, 480, 640, 3), dtype=np.uint8)
    boxes, idx = [], []
    for b in range(64):
        if b % 2 == 0:
            boxes.append([int(rng.integers(0, 640 - 256)), int(rng.integers(0, 480 - 256)), 256, 256])
        else:
            boxes.append(PP.padding_Bbox([int(rng.integers(-40, 600)), int(rng.integers(-40, 440)), int(rng.integers(20, 200)), int(rng.integers(20, 200))], 1.5))
        idx.append(b % 4)
    crops = PP.get_roi_batch(torch.from_numpy(frames).cuda(), boxes, 256, PP.INTER_LINEAR, "crop_square_resize", img_index=idx)
    c = crops.cpu().numpy()
    for b in range(0, 64, 2):
        x, y = boxes[b][0], boxes[b][1]
        assert np.array_equal(c[b], frames[idx[b], y:y + 256, x:x + 256])
    for b in (1, 33):                                                           # two of the padded boxes against the oracle
        assert np.array_equal(c[b], PO.get_roi(frames[idx[b]], boxes[b], 256, PO.INTER_LINEAR, "crop_square_resize"))
    net = build_net(npoint=512, seed=1).cuda().eval()
    net.set_compute_dtype("bf16")
    p3d = torch.zeros(8, 3, 512, device="cuda")
    with torch.no_grad():
        a = net(crops[:8], p3d)
        bq = net(torch.from_numpy(c[:8]).cuda(), p3d)
    for u, v in zip(a, bq):
        assert torch.equal(u, v)


@pytest.mark.gpu
def test_estimate_poses_is_the_composition_of_its_stages():
    """postprocess.estimate_poses (test.py's inner loop on the device: boxes on full frames -> crops -> forward -> correspondences from
    the final boxes -> EPnP + RANSAC) == the stages called one by one, incl. a missing detection (zero crop, degenerate box)"""
    from checkerpose_amd.synthetic import build_net
    from checkerpose_amd import postprocess as Q
    rng = np.random.default_rng(9)
    frames = torch.from_numpy(rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)).cuda()
    boxes = [[100, 80, 120, 90], [300, 200, 60, 140], None, [-10, 400, 90, 90]]
    idx = [0, 1, 0, 1]
    net = build_net(npoint=512, seed=1).cuda().eval()
    net.set_compute_dtype("bf16")
    p3d = torch.from_numpy(rng.normal(size=(512, 3)).astype(np.float32) * 50).cuda()
    K = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1]], dtype=np.float32)
    R, t, inl, status, final = Q.estimate_poses(net, frames, boxes, p3d, K, img_index=idx)
    padded = [None if b is None else PP.padding_Bbox(b, 1.5) for b in boxes]
    crops = PP.get_roi_batch(frames, padded, 256, PP.INTER_LINEAR, "crop_square_resize", img_index=idx)
    assert not crops[2].any()
    want_final = np.array([[0, 0, 0, 0] if b is None else PP.get_final_Bbox(b, "crop_square_resize", 640, 480) for b in padded])
    assert np.array_equal(final, want_final)
    with torch.no_grad():
        out = net(crops, None)
    p2d, valid, _ = Q.correspondences(out, Bboxes=want_final)
    R2, t2, inl2, st2 = Q.solve_pnp_ransac(p3d, p2d, valid, K)
    assert torch.equal(R, R2) and torch.equal(t, t2) and torch.equal(inl, inl2) and torch.equal(status, st2)
    assert tuple(R.shape) == (4, 3, 3) and tuple(t.shape) == (4, 3, 1) and status.dtype == torch.int32
