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
