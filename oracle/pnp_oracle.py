"""ORACLE -- test infrastructure, NOT product code.  Only tests/ may import this file.

CPU restatement (numpy, float64) of the pose solver behind SURVEY.md 8f row N4: `cv2.solvePnPRansac(valid_p3d, valid_disc_p2d,
cam_K, distCoeffs=None, reprojectionError=2, iterationsCount=150, flags=cv2.SOLVEPNP_EPNP)` as the reference calls it
(test_network_with_test_data.py:100-110), including the reference's own fallbacks (identity pose when fewer than 4 valid
correspondences, :111-114).

PARITY UNPINNED.  The arithmetic lives in the third-party package `opencv-python` (cv2; version not pinned anywhere in the
reference), which is absent from /root/reference and not installable offline, and the reference holds no fixture of a solver
output.  What is restated here is the PUBLISHED algorithm:
  * EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) in the form OpenCV's epnp.cpp implements it: 4 control points from the PCA of
    the model points, barycentric coordinates, the 2n x 12 system M, the null-space basis from the 4 smallest eigenvectors of
    M^T M, the three beta approximations (N = 4, 2, 3 unknown-products variants) each refined by 5 Gauss-Newton steps, camera-frame
    control points -> absolute orientation (SVD of the 3x3 cross-covariance, row-flip on a reflection), best of the three by mean
    reprojection error;
  * the RANSAC frame of OpenCV's solvePnPRansac: minimal sets of 5 correspondences, squared reprojection
    error <= threshold^2 as the inlier test, the hypothesis with the most inliers (first one on ties, at least the sample size),
    a final EPnP over its inliers; the returned inlier list is that hypothesis' set.
Deliberate differences (no way to pin them, and nothing downstream depends on them): the sample sequence comes from a counter-based
hash (`sample_indices`, shared bit for bit with the device kernel) instead of OpenCV's MWC generator; OpenCV's stopping rule
(RANSACUpdateNumIters at confidence 0.99) is applied between rounds of 64 hypotheses, not after every better model, so at least as
many hypotheses are evaluated as OpenCV would; small linear systems are
solved through the normal equations instead of an SVD.  With exactly 4 valid correspondences OpenCV runs no RANSAC but its P3P
kernel on the first three points and lets the fourth choose among the solutions: restated here (round 4) with Grunert's quartic
(OpenCV's p3p.cpp follows Gao et al.: another derivation of the same up-to-four solutions).  Known-answer anchoring instead of golden vectors: synthetic poses with
exact and outlier-contaminated correspondences must be recovered (tests/test_pnp.py).
"""
import numpy as np

PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


def _hash32(a, b, c, d):
    """counter-based 32-bit hash (all arithmetic mod 2^32), identical in pnp.hip"""
    M = 0xFFFFFFFF
    h = (a * 0x9E3779B1 + 0x7F4A7C15) & M
    for v in (b, c, d):
        h ^= (v + 0x9E3779B9 + ((h << 6) & M) + (h >> 2)) & M
        h = (h * 0x85EBCA6B) & M
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & M
        h ^= h >> 16
    return h


def sample_indices(seed, crop, hyp, n_valid, m):
    """m distinct positions in [0, n_valid): draw k takes hash(seed, crop, hyp, try) % n_valid, duplicates are redrawn"""
    out, t = [], 0
    while len(out) < m:
        r = _hash32(seed, crop, hyp, t) % n_valid
        t += 1
        if r not in out:
            out.append(r)
    return out


def _solve_normal(A, b):
    """least squares through the normal equations (Gaussian elimination with partial pivoting on A^T A)"""
    return np.linalg.solve(A.T @ A, A.T @ b)


def epnp(pw, uv, K):
    """pw (n,3) model points, uv (n,2) pixels, K (3,3).  Returns (R (3,3), t (3,), mean reprojection error)."""
    pw, uv = np.asarray(pw, np.float64), np.asarray(uv, np.float64)
    n = pw.shape[0]
    fu, fv, uc, vc = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    cw = np.zeros((4, 3))
    cw[0] = pw.mean(0)
    P0 = pw - cw[0]
    ev, evec = np.linalg.eigh(P0.T @ P0)                       # ascending
    for i in range(3):
        cw[i + 1] = cw[0] + np.sqrt(max(ev[2 - i], 0.0) / n) * evec[:, 2 - i]
    CC = (cw[1:] - cw[0]).T                                   # columns = control-point axes
    a123 = np.linalg.solve(CC, (pw - cw[0]).T).T              # (n,3)
    alphas = np.concatenate([1.0 - a123.sum(1, keepdims=True), a123], 1)      # (n,4)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j] * fu
        M[0::2, 3 * j + 2] = alphas[:, j] * (uc - uv[:, 0])
        M[1::2, 3 * j + 1] = alphas[:, j] * fv
        M[1::2, 3 * j + 2] = alphas[:, j] * (vc - uv[:, 1])
    w, V = np.linalg.eigh(M.T @ M)                            # ascending: columns 0..3 = the null-space basis v1..v4
    v = [V[:, i] for i in range(4)]
    L = np.zeros((6, 10))
    for r, (a, b) in enumerate(PAIRS):
        dv = [v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for i in range(4)]
        L[r] = [dv[0] @ dv[0], 2 * dv[0] @ dv[1], dv[1] @ dv[1], 2 * dv[0] @ dv[2], 2 * dv[1] @ dv[2], dv[2] @ dv[2],
                2 * dv[0] @ dv[3], 2 * dv[1] @ dv[3], 2 * dv[2] @ dv[3], dv[3] @ dv[3]]
    rho = np.array([np.sum((cw[a] - cw[b]) ** 2) for a, b in PAIRS])

    def approx(kind):
        be = np.zeros(4)
        if kind == 1:                                          # unknowns B11 B12 B13 B14
            b4 = _solve_normal(L[:, [0, 1, 3, 6]], rho)
            if b4[0] < 0:
                be[0] = np.sqrt(-b4[0]); be[1:] = -b4[1:] / be[0]
            else:
                be[0] = np.sqrt(b4[0]); be[1:] = b4[1:] / be[0]
        elif kind == 2:                                        # B11 B12 B22
            b3 = _solve_normal(L[:, [0, 1, 2]], rho)
            if b3[0] < 0:
                be[0] = np.sqrt(-b3[0]); be[1] = np.sqrt(-b3[2]) if b3[2] < 0 else 0.0
            else:
                be[0] = np.sqrt(b3[0]); be[1] = np.sqrt(b3[2]) if b3[2] > 0 else 0.0
            if b3[1] < 0:
                be[0] = -be[0]
        else:                                                  # B11 B12 B22 B13 B23
            b5 = _solve_normal(L[:, [0, 1, 2, 3, 4]], rho)
            if b5[0] < 0:
                be[0] = np.sqrt(-b5[0]); be[1] = np.sqrt(-b5[2]) if b5[2] < 0 else 0.0
            else:
                be[0] = np.sqrt(b5[0]); be[1] = np.sqrt(b5[2]) if b5[2] > 0 else 0.0
            if b5[1] < 0:
                be[0] = -be[0]
            be[2] = b5[3] / be[0] if be[0] != 0 else 0.0
        return be

    def gauss_newton(be):
        for _ in range(5):
            A, r = np.zeros((6, 4)), np.zeros(6)
            for i in range(6):
                l = L[i]
                A[i] = [2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3],
                        l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3],
                        l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3],
                        l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3]]
                r[i] = rho[i] - (l[0] * be[0] ** 2 + l[1] * be[0] * be[1] + l[2] * be[1] ** 2 + l[3] * be[0] * be[2] + l[4] * be[1] * be[2]
                                 + l[5] * be[2] ** 2 + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] ** 2)
            AtA = A.T @ A + 1e-18 * np.trace(A.T @ A) * np.eye(4)      # (betas of an unused basis vector stay where they are)
            be = be + np.linalg.solve(AtA, A.T @ r)
        return be

    def pose(be):
        ccs = sum(be[i] * v[i] for i in range(4)).reshape(4, 3)
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            ccs, pcs = -ccs, -pcs
        pc0, pw0 = pcs.mean(0), pw.mean(0)
        ABt = (pcs - pc0).T @ (pw - pw0)
        U, _, Vt = np.linalg.svd(ABt)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        return R, t, reprojection_error(pw, uv, K, R, t)

    best = None
    for kind in (1, 2, 3):
        be = approx(kind)
        if not np.all(np.isfinite(be)):
            continue
        R, t, err = pose(gauss_newton(be))
        if np.isfinite(err) and (best is None or err < best[2]):
            best = (R, t, err)
    if best is None:
        return np.eye(3), np.zeros(3), np.inf
    return best


def project(pw, K, R, t):
    pc = pw @ R.T + t
    return np.stack([K[0, 2] + K[0, 0] * pc[:, 0] / pc[:, 2], K[1, 2] + K[1, 1] * pc[:, 1] / pc[:, 2]], 1)


def reprojection_error(pw, uv, K, R, t):
    return float(np.sqrt(((project(pw, K, R, t) - uv) ** 2).sum(1)).mean())


def needed_iterations(best, nv, m, iterations, confidence=0.99):
    """OpenCV RANSACUpdateNumIters: iterations after which an all-inlier sample has been drawn with `confidence`"""
    if best < m:
        return iterations
    ep = min(max(1.0 - best / nv, 0.0), 1.0)
    num = np.log(1.0 - confidence)
    denom = 1.0 - (1.0 - ep) ** m
    if denom < np.finfo(np.float64).tiny:
        return 0
    denom = np.log(denom)
    if denom >= 0 or -num >= iterations * (-denom):
        return iterations
    return int(np.rint(num / denom))


def _absolute_orientation(pc, pw):
    """R, t with pc ~ R pw + t (SVD of the 3x3 cross-covariance, proper rotation also for 3 points / rank 2)"""
    c0, w0 = pc.mean(0), pw.mean(0)
    U, _, Vt = np.linalg.svd((pc - c0).T @ (pw - w0))
    R = U @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    return R, c0 - R @ w0


def p3p_distances(pw, uv, K):
    """Grunert's quartic for the three camera distances (as reviewed by Haralick et al., IJCV 1994): up to four (s1, s2, s3), each
    polished by Newton steps on the three law-of-cosines equations.  pw (3,3), uv (3,2).  Returns (bearings (3,3), [distances])."""
    f = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0], (uv[:, 1] - K[1, 2]) / K[1, 1], np.ones(3)], 1)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    a2, b2, c2 = ((pw[1] - pw[2]) ** 2).sum(), ((pw[0] - pw[2]) ** 2).sum(), ((pw[0] - pw[1]) ** 2).sum()
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    if min(a2, b2, c2) <= 0:
        return f, []
    q = (a2 - c2) / b2
    A = [(q - 1) ** 2 - 4 * c2 / b2 * ca ** 2,
         4 * (q * (1 - q) * cb - (1 - (a2 + c2) / b2) * ca * cg + 2 * c2 / b2 * ca ** 2 * cb),
         2 * (q ** 2 - 1 + 2 * q ** 2 * cb ** 2 + 2 * (b2 - c2) / b2 * ca ** 2 - 4 * (a2 + c2) / b2 * ca * cb * cg + 2 * (b2 - a2) / b2 * cg ** 2),
         4 * (-q * (1 + q) * cb + 2 * a2 / b2 * cg ** 2 * cb - (1 - (a2 + c2) / b2) * ca * cg),
         (1 + q) ** 2 - 4 * a2 / b2 * cg ** 2]
    out = []
    for v in np.roots(A):
        if abs(v.imag) > 1e-6 * (1 + abs(v.real)) or v.real <= 0:
            continue
        v = float(v.real)
        den = 2 * (cg - v * ca)
        if abs(den) < 1e-12:
            continue
        u = ((q - 1) * v * v - 2 * q * cb * v + 1 + q) / den
        if u <= 0 or 1 + v * v - 2 * v * cb <= 0:
            continue
        s1 = np.sqrt(b2 / (1 + v * v - 2 * v * cb))
        sd = np.array([s1, u * s1, v * s1])
        for _ in range(3):                                  # Newton polish
            F = np.array([sd[1] ** 2 + sd[2] ** 2 - 2 * sd[1] * sd[2] * ca - a2, sd[0] ** 2 + sd[2] ** 2 - 2 * sd[0] * sd[2] * cb - b2,
                          sd[0] ** 2 + sd[1] ** 2 - 2 * sd[0] * sd[1] * cg - c2])
            J = np.array([[0, 2 * sd[1] - 2 * sd[2] * ca, 2 * sd[2] - 2 * sd[1] * ca],
                          [2 * sd[0] - 2 * sd[2] * cb, 0, 2 * sd[2] - 2 * sd[0] * cb],
                          [2 * sd[0] - 2 * sd[1] * cg, 2 * sd[1] - 2 * sd[0] * cg, 0]])
            if abs(np.linalg.det(J)) < 1e-30:
                break
            sd = sd - np.linalg.solve(J, F)
        if (sd > 0).all():
            out.append(sd)
    return f, out


def solve_four_points(pw4, uv4, K):
    """What OpenCV's solvePnPRansac does with exactly 4 correspondences (calib3d solvepnp.cpp: model_points == npoints -> no RANSAC,
    solvePnP with the P3P kernel): P3P on the first three, the fourth picks among the up-to-four poses (smallest reprojection error).
    Returns (R, t) or None."""
    f, dists = p3p_distances(pw4[:3], uv4[:3], K)
    best = None
    for sd in dists:
        R, t = _absolute_orientation(sd[:, None] * f, pw4[:3])
        pc = R @ pw4[3] + t
        if pc[2] <= 0:
            continue
        e = np.hypot(K[0, 2] + K[0, 0] * pc[0] / pc[2] - uv4[3, 0], K[1, 2] + K[1, 1] * pc[1] / pc[2] - uv4[3, 1])
        if best is None or e < best[0]:
            best = (e, R, t)
    return None if best is None else (best[1], best[2])


def solve_pnp_ransac(p3d, p2d, valid, K, threshold=2.0, iterations=150, seed=0, crop=0):
    """p3d (N,3), p2d (N,2), valid (N,) bool, K (3,3).  Returns (R, t, inlier mask (N,) bool, status): status 0 = the
    reference's identity fallback (fewer than 4 valid points, or no hypothesis with a full sample of inliers)."""
    p3d, p2d = np.asarray(p3d, np.float64), np.asarray(p2d, np.float64)
    vid = np.nonzero(np.asarray(valid))[0]
    nv = len(vid)
    ident = (np.eye(3), np.zeros(3), np.zeros(len(p3d), bool), 0)
    if nv < 4:
        return ident
    if nv == 4:                                                # no RANSAC on exactly 4 points: P3P + the 4th point, all four "inliers"
        rt = solve_four_points(p3d[vid], p2d[vid], np.asarray(K, np.float64))
        if rt is None:
            return ident
        mask = np.zeros(len(p3d), bool)
        mask[vid] = True
        return rt[0], rt[1], mask, 1
    m = 5
    best_cnt, best_mask = m - 1, None
    counts = []
    for h in range(iterations):
        if h > 0 and h % 64 == 0 and h >= needed_iterations(max(counts), nv, m, iterations):
            break                                              # OpenCV's stopping rule, applied between rounds of 64 hypotheses
        counts.append(-1)
        s = vid[sample_indices(seed, crop, h, nv, m)]
        try:
            R, t, err = epnp(p3d[s], p2d[s], K)
        except np.linalg.LinAlgError:                          # coplanar / collinear sample: no barycentric frame
            continue
        if not np.isfinite(err):
            continue
        d2 = ((project(p3d[vid], K, R, t) - p2d[vid]) ** 2).sum(1)
        inl = d2 <= threshold * threshold
        counts[-1] = int(inl.sum())
        if inl.sum() > best_cnt:
            best_cnt, best_mask = int(inl.sum()), inl
    if best_mask is None:
        return ident
    sel = vid[best_mask]
    R, t, _ = epnp(p3d[sel], p2d[sel], K)
    mask = np.zeros(len(p3d), bool)
    mask[sel] = True
    return R, t, mask, 1
