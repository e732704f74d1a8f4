// Pose from correspondences on the device: the solver call of the reference's from_id_to_pose
//   cv2.solvePnPRansac(valid_p3d, valid_disc_p2d, cam_K, None, reprojectionError=2, iterationsCount=150, flags=cv2.SOLVEPNP_EPNP)
// (test_network_with_test_data.py:100-110; identity pose below 4 valid correspondences, :111-114) as a consumer of
// cp_correspondences' output: TWO launches for the whole batch, so the (B, N, 2) coordinates and validity
// masks never leave the GPU -- only 12 doubles per crop do.  opencv-python is not vendored by the reference (and absent here): this
// is the published algorithm -- EPnP (Lepetit, Moreno-Noguer, Fua 2009) in the structure of OpenCV's epnp.cpp inside the RANSAC
// frame of OpenCV's solvePnPRansac -- restated in oracle/pnp_oracle.py, which states the deliberate differences (sample sequence
// from a counter-based hash, no early termination).  All arithmetic in fp64.
//   launch 1 (grid: 64 hypotheses x crop, one wave each; the crop's correspondences staged in LDS, valid indices compacted by
//            ballot scans): lane h draws 5 (4 if only 4 are valid) distinct correspondences, EPnP -> pose h, counts the valid
//            correspondences with squared reprojection error <= threshold^2 -> a 14-double record in scratch;
//   launch 2 (one 256-thread workgroup per crop): best = most inliers (first on ties, at least a full sample); its inlier list
//            is compacted; EPnP over the inliers with every loop over the points shared by the 256 threads (block reductions:
//            centroid, scatter matrix, the 78 entries of M^T M, the candidates' centroids / cross-covariances / errors) and
//            the small dense algebra (12 x 12 Jacobi, betas, Gauss-Newton, 3 x 3 SVD) on single threads.
// (The first version -- everything in one workgroup, serial passes on thread 0 straight from global memory -- took 7.2 ms per
// batch whatever its size; see tools/pnp_bench.py.)
#include "common.h"

namespace {

struct PnpParams {
  const float* p3d; const float* p2d; const uint8_t* valid; const float* K;
  double* pose; uint8_t* inliers; int32_t* status; int32_t* scratch;
  long long p3d_bs, K_bs;
  int B, N, valid_stride, iters, round;
  float thr;
  uint32_t seed;
};

__device__ __forceinline__ uint32_t hash32(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t h = a * 0x9E3779B1u + 0x7F4A7C15u;
  const uint32_t v[3] = {b, c, d};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    h ^= v[i] + 0x9E3779B9u + (h << 6) + (h >> 2);
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
  }
  return h;
}

// cyclic Jacobi on a symmetric n x n matrix (row-major a, destroyed: eigenvalues end on its diagonal); eigenvectors = columns of v
template <int n>
__device__ void jacobi_eig(double* a, double* v) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) v[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += a[i * n + i] * a[i * n + i];
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    }
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {                     // A <- J^T A J
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
}

// the same on matrices that live in LDS with an element stride (launch 1: element e of lane l at [e * 64 + l], conflict-free;
// launch 2: stride 1).  In private (scratch) memory the ~76 000 dependent loads / stores of a 12 x 12 solve took ~4 ms per wave.
template <int n>
__device__ void jacobi_eig_strided(double* a, double* v, int st) {
#define JA(i, j) a[((i) * n + (j)) * st]
#define JV(i, j) v[((i) * n + (j)) * st]
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) JV(i, j) = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += JA(i, i) * JA(i, i);
      for (int j = i + 1; j < n; ++j) off += JA(i, j) * JA(i, j);
    }
    if (off <= 1e-26 * diag || off == 0.0) break;             // off-diagonal mass below 1e-13 of the diagonal's
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = JA(p, q);
        if (apq * apq <= 1e-34 * fabs(JA(p, p) * JA(q, q)) || apq == 0.0) continue;      // already negligible: skip the rotation
        const double theta = (JA(q, q) - JA(p, p)) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < n; ++k) {                     // A <- J^T A J
          const double akp = JA(k, p), akq = JA(k, q);
          JA(k, p) = c * akp - s * akq;
          JA(k, q) = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < n; ++k) {
          const double apk = JA(p, k), aqk = JA(q, k);
          JA(p, k) = c * apk - s * aqk;
          JA(q, k) = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < n; ++k) {
          const double vkp = JV(k, p), vkq = JV(k, q);
          JV(k, p) = c * vkp - s * vkq;
          JV(k, q) = s * vkp + c * vkq;
        }
      }
  }
#undef JA
#undef JV
}

// the same solve shared by the first 12 lanes of ONE wave (launch 2's single final solve): lane k owns index k of every 12-long
// loop; all 12 lanes derive the same rotation from the same three LDS words; LDS operations of a wave execute in program order, so
// the column / row / eigenvector updates need no barrier between them.  Call with the whole wave converged; lanes >= 12 idle.
__device__ void jacobi_eig12_wave(double* a, double* v, int lane) {
  const int k = lane;
  const bool on = lane < 12;
  if (on)
    for (int j = 0; j < 12; ++j) v[k * 12 + j] = k == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < 12; ++i) {
      diag += a[i * 12 + i] * a[i * 12 + i];
      for (int j = i + 1; j < 12; ++j) off += a[i * 12 + j] * a[i * 12 + j];
    }
    if (off <= 1e-26 * diag || off == 0.0) break;             // every lane reads the same words: uniform decision
    for (int p = 0; p < 11; ++p)
      for (int q = p + 1; q < 12; ++q) {
        const double apq = a[p * 12 + q], app = a[p * 12 + p], aqq = a[q * 12 + q];
        if (apq * apq <= 1e-34 * fabs(app * aqq) || apq == 0.0) continue;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        if (on) {
          const double akp = a[k * 12 + p], akq = a[k * 12 + q];
          a[k * 12 + p] = c * akp - s * akq;
          a[k * 12 + q] = s * akp + c * akq;
        }
        if (on) {
          const double apk = a[p * 12 + k], aqk = a[q * 12 + k];
          a[p * 12 + k] = c * apk - s * aqk;
          a[q * 12 + k] = s * apk + c * aqk;
          const double vkp = v[k * 12 + p], vkq = v[k * 12 + q];
          v[k * 12 + p] = c * vkp - s * vkq;
          v[k * 12 + q] = s * vkp + c * vkq;
        }
      }
  }
}

// x = argmin |A x - b| through the normal equations, A (6 x m) row-major with row pitch 4/5 given by `ld`; ridge * trace on the diagonal
template <int m>
__device__ bool solve_normal(const double* A, int ld, const double* b, double ridge, double* x) {
  double G[m][m + 1];
  double tr = 0.0;
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int r = 0; r < 6; ++r) s += A[r * ld + i] * A[r * ld + j];
      G[i][j] = s;
    }
    double s = 0.0;
    for (int r = 0; r < 6; ++r) s += A[r * ld + i] * b[r];
    G[i][m] = s;
    tr += G[i][i];
  }
  for (int i = 0; i < m; ++i) G[i][i] += ridge * tr;
  for (int c = 0; c < m; ++c) {                           // Gaussian elimination, partial pivoting
    int piv = c;
    for (int r = c + 1; r < m; ++r)
      if (fabs(G[r][c]) > fabs(G[piv][c])) piv = r;
    if (G[piv][c] == 0.0) return false;
    if (piv != c)
      for (int k = 0; k <= m; ++k) { const double tmp = G[c][k]; G[c][k] = G[piv][k]; G[piv][k] = tmp; }
    for (int r = c + 1; r < m; ++r) {
      const double f = G[r][c] / G[c][c];
      for (int k = c; k <= m; ++k) G[r][k] -= f * G[c][k];
    }
  }
  for (int i = m - 1; i >= 0; --i) {
    double s = G[i][m];
    for (int k = i + 1; k < m; ++k) s -= G[i][k] * x[k];
    x[i] = s / G[i][i];
  }
  return true;
}

// one-sided Jacobi SVD of a 3 x 3 matrix (row-major m): R = U V^T of its SVD, third row negated when det R < 0 (epnp.cpp)
__device__ void procrustes_rotation(const double* m, double* R) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) a[i] = m[i];
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += a[3 * i + p] * a[3 * i + p]; be += a[3 * i + q] * a[3 * i + q]; ga += a[3 * i + p] * a[3 * i + q]; }
        if (fabs(ga) <= 1e-16 * sqrt(al * be) || ga == 0.0) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double aip = a[3 * i + p], aiq = a[3 * i + q];
          a[3 * i + p] = c * aip - s * aiq; a[3 * i + q] = s * aip + c * aiq;
          const double vip = v[3 * i + p], viq = v[3 * i + q];
          v[3 * i + p] = c * vip - s * viq; v[3 * i + q] = s * vip + c * viq;
        }
      }
    if (!rotated) break;
  }
  // columns of a = sigma_j u_j; order them by sigma (descending) so that a vanishing third one can be completed by a cross product
  double sg[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) sg[j] = sqrt(a[j] * a[j] + a[3 + j] * a[3 + j] + a[6 + j] * a[6 + j]);
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (sg[ord[j]] > sg[ord[i]]) { const int tmp = ord[i]; ord[i] = ord[j]; ord[j] = tmp; }
  double U[9], V[9];
  for (int k = 0; k < 3; ++k) {
    const int j = ord[k];
    const double inv = sg[j] > 0.0 ? 1.0 / sg[j] : 0.0;
    for (int i = 0; i < 3; ++i) { U[3 * i + k] = a[3 * i + j] * inv; V[3 * i + k] = v[3 * i + j]; }
  }
  if (sg[ord[2]] <= 1e-12 * sg[ord[0]]) {                 // rank 2: u3 = +-(u1 x u2), sign such that U, V have the same handedness
    const double c0 = U[3] * U[7] - U[6] * U[4], c1 = U[6] * U[1] - U[0] * U[7], c2 = U[0] * U[4] - U[3] * U[1];
    const double dv = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    const double sgn = dv >= 0.0 ? 1.0 : -1.0;
    U[2] = sgn * c0; U[5] = sgn * c1; U[8] = sgn * c2;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0.0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
}

// ---- exactly four valid correspondences: OpenCV's solvePnPRansac runs no RANSAC there (calib3d solvepnp.cpp: model_points ==
// npoints) but solvePnP with its P3P kernel -- P3P on the first three points, the fourth picks among the up-to-four poses by its
// reprojection error; all four are reported as inliers.  P3P here: Grunert's quartic in v = s3 / s1 (Haralick et al., IJCV 1994),
// roots by Durand-Kerner + Newton, the three distances polished by Newton steps on the law-of-cosines system, pose by absolute
// orientation (procrustes_rotation: rank 2 with three points).  oracle/pnp_oracle.py:solve_four_points is the same arithmetic.
__device__ int quartic_real_roots(const double* A, double* roots) {
  double mx = 0.0;
  for (int i = 0; i < 5; ++i) mx = fmax(mx, fabs(A[i]));
  if (!(mx > 0.0) || fabs(A[0]) < 1e-14 * mx) return 0;       // (a vanishing leading coefficient: a measure-zero configuration)
  double c[4];                                                // monic: v^4 + c0 v^3 + c1 v^2 + c2 v + c3
  double bound = 0.0;
  for (int i = 0; i < 4; ++i) { c[i] = A[i + 1] / A[0]; bound = fmax(bound, fabs(c[i])); }
  bound += 1.0;
  double zr[4], zi[4];
  {
    double pr = 1.0, pi = 0.0;                                // (0.4 + 0.9 i)^k * bound
    for (int k = 0; k < 4; ++k) {
      zr[k] = pr * bound; zi[k] = pi * bound;
      const double nr = pr * 0.4 - pi * 0.9, ni = pr * 0.9 + pi * 0.4;
      pr = nr; pi = ni;
    }
  }
  for (int it = 0; it < 200; ++it) {
    double change = 0.0;
    for (int k = 0; k < 4; ++k) {
      double pr = 1.0, pi = 0.0;                              // p(z_k) by Horner
      for (int i = 0; i < 4; ++i) {
        const double nr = pr * zr[k] - pi * zi[k] + c[i], ni = pr * zi[k] + pi * zr[k];
        pr = nr; pi = ni;
      }
      double dr = 1.0, di = 0.0;                              // prod_{j != k} (z_k - z_j)
      for (int j = 0; j < 4; ++j) {
        if (j == k) continue;
        const double er = zr[k] - zr[j], ei = zi[k] - zi[j];
        const double nr = dr * er - di * ei, ni = dr * ei + di * er;
        dr = nr; di = ni;
      }
      const double dn = dr * dr + di * di;
      if (!(dn > 0.0)) continue;
      const double qr = (pr * dr + pi * di) / dn, qi = (pi * dr - pr * di) / dn;
      zr[k] -= qr; zi[k] -= qi;
      change = fmax(change, fabs(qr) + fabs(qi));
    }
    if (change < 1e-15 * bound) break;
  }
  int n = 0;
  for (int k = 0; k < 4; ++k) {
    if (fabs(zi[k]) > 1e-6 * (1.0 + fabs(zr[k]))) continue;
    double v = zr[k];
    for (int it = 0; it < 2; ++it) {                          // Newton on the real polynomial
      const double f = (((A[0] * v + A[1]) * v + A[2]) * v + A[3]) * v + A[4];
      const double d = ((4.0 * A[0] * v + 3.0 * A[1]) * v + 2.0 * A[2]) * v + A[3];
      if (d != 0.0) v -= f / d;
    }
    roots[n++] = v;
  }
  return n;
}

// pose from exactly four correspondences (pw (4,3) float, uv (4,2) float): true + R (row-major), t; false: no admissible solution
__device__ bool solve_four_points(const float* pw, const float* uv, double fu, double fv, double uc, double vc, double* Rout, double* tout) {
  double f[3][3], X[4][3];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 3; ++c) X[i][c] = (double)pw[3 * i + c];
  for (int i = 0; i < 3; ++i) {
    const double a = ((double)uv[2 * i] - uc) / fu, b = ((double)uv[2 * i + 1] - vc) / fv;
    const double inv = 1.0 / sqrt(a * a + b * b + 1.0);
    f[i][0] = a * inv; f[i][1] = b * inv; f[i][2] = inv;
  }
  auto d2 = [&](int i, int j) { double s = 0.0; for (int c = 0; c < 3; ++c) s += (X[i][c] - X[j][c]) * (X[i][c] - X[j][c]); return s; };
  auto dot = [&](int i, int j) { return f[i][0] * f[j][0] + f[i][1] * f[j][1] + f[i][2] * f[j][2]; };
  const double a2 = d2(1, 2), b2 = d2(0, 2), c2 = d2(0, 1);
  if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return false;
  const double ca = dot(1, 2), cb = dot(0, 2), cg = dot(0, 1);
  const double q = (a2 - c2) / b2;
  const double A[5] = {(q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca,
                       4.0 * (q * (1.0 - q) * cb - (1.0 - (a2 + c2) / b2) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb),
                       2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca - 4.0 * (a2 + c2) / b2 * ca * cb * cg +
                              2.0 * (b2 - a2) / b2 * cg * cg),
                       4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - (a2 + c2) / b2) * ca * cg),
                       (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg};
  double roots[4];
  const int nr = quartic_real_roots(A, roots);
  bool have = false;
  double best = INFINITY;
  for (int r = 0; r < nr; ++r) {
    const double v = roots[r];
    if (!(v > 0.0)) continue;
    const double den = 2.0 * (cg - v * ca), w = 1.0 + v * v - 2.0 * v * cb;
    if (fabs(den) < 1e-12 || !(w > 0.0)) continue;
    const double u = ((q - 1.0) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
    if (!(u > 0.0)) continue;
    double sd[3];
    sd[0] = sqrt(b2 / w); sd[1] = u * sd[0]; sd[2] = v * sd[0];
    for (int it = 0; it < 3; ++it) {                          // Newton polish of (s1, s2, s3)
      const double F0 = sd[1] * sd[1] + sd[2] * sd[2] - 2.0 * sd[1] * sd[2] * ca - a2;
      const double F1 = sd[0] * sd[0] + sd[2] * sd[2] - 2.0 * sd[0] * sd[2] * cb - b2;
      const double F2 = sd[0] * sd[0] + sd[1] * sd[1] - 2.0 * sd[0] * sd[1] * cg - c2;
      const double J01 = 2.0 * sd[1] - 2.0 * sd[2] * ca, J02 = 2.0 * sd[2] - 2.0 * sd[1] * ca;
      const double J10 = 2.0 * sd[0] - 2.0 * sd[2] * cb, J12 = 2.0 * sd[2] - 2.0 * sd[0] * cb;
      const double J20 = 2.0 * sd[0] - 2.0 * sd[1] * cg, J21 = 2.0 * sd[1] - 2.0 * sd[0] * cg;
      // J = [[0, J01, J02], [J10, 0, J12], [J20, J21, 0]]: Cramer
      const double det = -J01 * (0.0 - J12 * J20) + J02 * (J10 * J21);
      if (!(fabs(det) > 1e-30)) break;
      const double x0 = (F0 * (0.0 - J12 * J21) - J01 * (0.0 - J12 * F2) + J02 * (F1 * J21)) / det;
      const double x1 = (0.0 - F0 * (0.0 - J12 * J20) + J02 * (J10 * F2 - F1 * J20)) / det;
      const double x2 = (J01 * (0.0 - (J10 * F2 - F1 * J20)) + F0 * (J10 * J21)) / det;
      sd[0] -= x0; sd[1] -= x1; sd[2] -= x2;
    }
    if (!(sd[0] > 0.0) || !(sd[1] > 0.0) || !(sd[2] > 0.0)) continue;
    double pc[3][3], c0[3] = {0, 0, 0}, w0[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c) { pc[i][c] = sd[i] * f[i][c]; c0[c] += pc[i][c] / 3.0; w0[c] += X[i][c] / 3.0; }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i)
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) H[3 * a + c] += (pc[i][a] - c0[a]) * (X[i][c] - w0[c]);
    double R[9], t[3];
    procrustes_rotation(H, R);
    for (int a = 0; a < 3; ++a) t[a] = c0[a] - (R[3 * a] * w0[0] + R[3 * a + 1] * w0[1] + R[3 * a + 2] * w0[2]);
    const double Z = R[6] * X[3][0] + R[7] * X[3][1] + R[8] * X[3][2] + t[2];
    if (!(Z > 0.0)) continue;
    const double du = uc + fu * (R[0] * X[3][0] + R[1] * X[3][1] + R[2] * X[3][2] + t[0]) / Z - (double)uv[6];
    const double dv = vc + fv * (R[3] * X[3][0] + R[4] * X[3][1] + R[5] * X[3][2] + t[1]) / Z - (double)uv[7];
    const double e = sqrt(du * du + dv * dv);
    if (e < best) {
      best = e; have = true;
      for (int i = 0; i < 9; ++i) Rout[i] = R[i];
      for (int i = 0; i < 3; ++i) tout[i] = t[i];
    }
  }
  return have;
}

constexpr int PNP_THREADS = 256, PNP_MAX_ITERS = 256, PNP_NMAX = 4096, PNP_HYP = 14;      // doubles per hypothesis record: count, -, 12 pose

struct Points {            // the correspondences of one EPnP call: idx[0 .. n) into this crop's (N, 3) / (N, 2) arrays
  const float* p3d; const float* p2d; const int32_t* idx; int n;
  double fu, fv, uc, vc;
};

struct Frame { double cw[4][3]; double ci[9]; };            // control points, inverse of [cw1 - cw0 | cw2 - cw0 | cw3 - cw0]

__device__ __forceinline__ void alphas_of(const Frame& f, const float* pw, double* al) {
  const double d0 = pw[0] - f.cw[0][0], d1 = pw[1] - f.cw[0][1], d2 = pw[2] - f.cw[0][2];
  al[1] = f.ci[0] * d0 + f.ci[1] * d1 + f.ci[2] * d2;
  al[2] = f.ci[3] * d0 + f.ci[4] * d1 + f.ci[5] * d2;
  al[3] = f.ci[6] * d0 + f.ci[7] * d1 + f.ci[8] * d2;
  al[0] = 1.0 - al[1] - al[2] - al[3];
}

// control points from the model points' centroid c and scatter matrix S (row-major 3 x 3, destroyed) -- the PCA of epnp.cpp's
// choose_control_points -- and the inverse of the barycentric basis
__device__ bool epnp_frame_from(const double* c, const double* Sin, int n, Frame& f) {
  double S[9], E[9];
  for (int i = 0; i < 9; ++i) S[i] = Sin[i];
  for (int k = 0; k < 3; ++k) f.cw[0][k] = c[k];
  jacobi_eig<3>(S, E);
  int ord[3] = {0, 1, 2};                                  // descending eigenvalues
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (S[4 * ord[j]] > S[4 * ord[i]]) { const int tmp = ord[i]; ord[i] = ord[j]; ord[j] = tmp; }
  double CC[9];
  for (int j = 0; j < 3; ++j) {
    const double ev = S[4 * ord[j]];
    const double k = sqrt((ev > 0.0 ? ev : 0.0) / n);
    for (int a = 0; a < 3; ++a) {
      CC[3 * a + j] = k * E[3 * a + ord[j]];
      f.cw[j + 1][a] = f.cw[0][a] + CC[3 * a + j];
    }
  }
  const double det = CC[0] * (CC[4] * CC[8] - CC[5] * CC[7]) - CC[1] * (CC[3] * CC[8] - CC[5] * CC[6]) + CC[2] * (CC[3] * CC[7] - CC[4] * CC[6]);
  if (!(fabs(det) > 0.0)) return false;                    // coplanar / collinear sample: no barycentric frame
  const double id = 1.0 / det;
  f.ci[0] = (CC[4] * CC[8] - CC[5] * CC[7]) * id; f.ci[1] = (CC[2] * CC[7] - CC[1] * CC[8]) * id; f.ci[2] = (CC[1] * CC[5] - CC[2] * CC[4]) * id;
  f.ci[3] = (CC[5] * CC[6] - CC[3] * CC[8]) * id; f.ci[4] = (CC[0] * CC[8] - CC[2] * CC[6]) * id; f.ci[5] = (CC[2] * CC[3] - CC[0] * CC[5]) * id;
  f.ci[6] = (CC[3] * CC[7] - CC[4] * CC[6]) * id; f.ci[7] = (CC[1] * CC[6] - CC[0] * CC[7]) * id; f.ci[8] = (CC[0] * CC[4] - CC[1] * CC[3]) * id;
  return true;
}

__device__ bool epnp_frame(const Points& P, Frame& f) {
  const int n = P.n;
  double c[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) { const float* pw = P.p3d + 3 * (size_t)P.idx[i]; c[0] += pw[0]; c[1] += pw[1]; c[2] += pw[2]; }
  for (int k = 0; k < 3; ++k) c[k] /= n;
  double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float* pw = P.p3d + 3 * (size_t)P.idx[i];
    const double d[3] = {pw[0] - c[0], pw[1] - c[1], pw[2] - c[2]};
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) S[3 * a + b] += d[a] * d[b];
  }
  return epnp_frame_from(c, S, n, f);
}

// the two rows of M that correspondence i contributes: r0 = [a_j fu, 0, a_j (uc - u)]_j, r1 = [0, a_j fv, a_j (vc - v)]_j
__device__ __forceinline__ void m_rows(const Points& P, const Frame& f, int i, double* r0, double* r1) {
  const size_t k = (size_t)P.idx[i];
  double al[4];
  alphas_of(f, P.p3d + 3 * k, al);
  const double u = P.p2d[2 * k], v = P.p2d[2 * k + 1];
  for (int j = 0; j < 4; ++j) {
    r0[3 * j] = al[j] * P.fu; r0[3 * j + 1] = 0.0; r0[3 * j + 2] = al[j] * (P.uc - u);
    r1[3 * j] = 0.0; r1[3 * j + 1] = al[j] * P.fv; r1[3 * j + 2] = al[j] * (P.vc - v);
  }
}

__device__ double reproj_mean(const Points& P, const double* R, const double* t) {
  double s = 0.0;
  for (int i = 0; i < P.n; ++i) {
    const size_t k = (size_t)P.idx[i];
    const float* pw = P.p3d + 3 * k;
    const double X = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + t[0], Y = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + t[1];
    const double iz = 1.0 / (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + t[2]);
    const double du = P.uc + P.fu * X * iz - P.p2d[2 * k], dv = P.vc + P.fv * Y * iz - P.p2d[2 * k + 1];
    s += sqrt(du * du + dv * dv);
  }
  return s / P.n;
}

// behind M^T M (row-major 12 x 12 with element stride st, in LDS like its eigenvector matrix V; destroyed): null-space basis v[4][12], the three beta approximations each refined by 5
// Gauss-Newton steps, and per approximation the camera-frame control points cc[kind][4][3] (sign fixed: the first point in front of
// the camera); kok[kind] = that approximation produced finite betas
__device__ void epnp_betas(const Points& P, const Frame& f, double* MtM, double* V, int st, double (*v)[12], double (*cc)[4][3], bool* kok) {
  if (st > 0) jacobi_eig_strided<12>(MtM, V, st);            // st < 0: the caller ran jacobi_eig12_wave on (MtM, V), element stride 1
  else st = 1;
  int ord[4];                                              // the 4 smallest eigenvalues, ascending
  {
    bool used[12];
    for (int i = 0; i < 12; ++i) used[i] = false;
    for (int k = 0; k < 4; ++k) {
      int best = -1;
      for (int i = 0; i < 12; ++i)
        if (!used[i] && (best < 0 || MtM[13 * i * st] < MtM[13 * best * st])) best = i;
      used[best] = true;
      ord[k] = best;
    }
  }
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 12; ++i) v[k][i] = V[(i * 12 + ord[k]) * st];
  const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
  double L[6][10], rho[6];
  for (int r = 0; r < 6; ++r) {
    double dv[4][3];
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 3; ++c) dv[k][c] = v[k][3 * pa[r] + c] - v[k][3 * pb[r] + c];
    auto dot = [&](int a, int b) { return dv[a][0] * dv[b][0] + dv[a][1] * dv[b][1] + dv[a][2] * dv[b][2]; };
    L[r][0] = dot(0, 0); L[r][1] = 2 * dot(0, 1); L[r][2] = dot(1, 1); L[r][3] = 2 * dot(0, 2); L[r][4] = 2 * dot(1, 2);
    L[r][5] = dot(2, 2); L[r][6] = 2 * dot(0, 3); L[r][7] = 2 * dot(1, 3); L[r][8] = 2 * dot(2, 3); L[r][9] = dot(3, 3);
    double d = 0.0;
    for (int c = 0; c < 3; ++c) { const double e = f.cw[pa[r]][c] - f.cw[pb[r]][c]; d += e * e; }
    rho[r] = d;
  }
  for (int kind = 1; kind <= 3; ++kind) {
    double be[4] = {0, 0, 0, 0};
    bool ok;
    if (kind == 1) {
      double A[6][4], b4[4];
      for (int r = 0; r < 6; ++r) { A[r][0] = L[r][0]; A[r][1] = L[r][1]; A[r][2] = L[r][3]; A[r][3] = L[r][6]; }
      ok = solve_normal<4>(&A[0][0], 4, rho, 0.0, b4);
      if (ok) {
        if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
        else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
      }
    } else if (kind == 2) {
      double A[6][3], b3[3];
      for (int r = 0; r < 6; ++r) { A[r][0] = L[r][0]; A[r][1] = L[r][1]; A[r][2] = L[r][2]; }
      ok = solve_normal<3>(&A[0][0], 3, rho, 0.0, b3);
      if (ok) {
        if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = b3[2] < 0 ? sqrt(-b3[2]) : 0.0; }
        else { be[0] = sqrt(b3[0]); be[1] = b3[2] > 0 ? sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) be[0] = -be[0];
      }
    } else {
      double A[6][5], b5[5];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 5; ++c) A[r][c] = L[r][c];
      ok = solve_normal<5>(&A[0][0], 5, rho, 0.0, b5);
      if (ok) {
        if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = b5[2] < 0 ? sqrt(-b5[2]) : 0.0; }
        else { be[0] = sqrt(b5[0]); be[1] = b5[2] > 0 ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) be[0] = -be[0];
        be[2] = be[0] != 0.0 ? b5[3] / be[0] : 0.0;
      }
    }
    ok = ok && isfinite(be[0]) && isfinite(be[1]) && isfinite(be[2]) && isfinite(be[3]);
    kok[kind - 1] = ok;
    if (!ok) continue;
    for (int it = 0; it < 5; ++it) {                        // Gauss-Newton on the 6 distance constraints
      double A[6][4], res[6], dx[4];
      for (int r = 0; r < 6; ++r) {
        const double* l = L[r];
        A[r][0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
        A[r][1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
        A[r][2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
        A[r][3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
        res[r] = rho[r] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] +
                           l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
      }
      if (!solve_normal<4>(&A[0][0], 4, res, 1e-18, dx)) break;
      for (int k = 0; k < 4; ++k) be[k] += dx[k];
    }
    double (*c4)[3] = cc[kind - 1];
    for (int j = 0; j < 4; ++j)
      for (int c = 0; c < 3; ++c) c4[j][c] = be[0] * v[0][3 * j + c] + be[1] * v[1][3 * j + c] + be[2] * v[2][3 * j + c] + be[3] * v[3][3 * j + c];
    double al[4];
    alphas_of(f, P.p3d + 3 * (size_t)P.idx[0], al);
    const double z0 = al[0] * c4[0][2] + al[1] * c4[1][2] + al[2] * c4[2][2] + al[3] * c4[3][2];
    if (z0 < 0.0)
      for (int j = 0; j < 4; ++j)
        for (int c = 0; c < 3; ++c) c4[j][c] = -c4[j][c];
  }
}

// the whole solve behind M^T M on ONE thread (hypotheses of 5 correspondences): absolute orientation of the three candidates, best
// by mean reprojection error
__device__ bool epnp_finish(const Points& P, const Frame& f, double* MtM, double* V, int st, double* Rout, double* tout) {
  double v[4][12], ccs[3][4][3];
  bool kok[3];
  epnp_betas(P, f, MtM, V, st, v, ccs, kok);
  double pw0[3] = {0, 0, 0};
  for (int i = 0; i < P.n; ++i) { const float* pw = P.p3d + 3 * (size_t)P.idx[i]; pw0[0] += pw[0]; pw0[1] += pw[1]; pw0[2] += pw[2]; }
  for (int c = 0; c < 3; ++c) pw0[c] /= P.n;
  double best_err = INFINITY;
  bool found = false;
  for (int kind = 0; kind < 3; ++kind) {
    if (!kok[kind]) continue;
    double (*cc)[3] = ccs[kind];
    auto pc_of = [&](int i, double* pc) {
      double al[4];
      alphas_of(f, P.p3d + 3 * (size_t)P.idx[i], al);
      for (int c = 0; c < 3; ++c) pc[c] = al[0] * cc[0][c] + al[1] * cc[1][c] + al[2] * cc[2][c] + al[3] * cc[3][c];
    };
    double pc0[3] = {0, 0, 0};
    for (int i = 0; i < P.n; ++i) { double pc[3]; pc_of(i, pc); pc0[0] += pc[0]; pc0[1] += pc[1]; pc0[2] += pc[2]; }
    for (int c = 0; c < 3; ++c) pc0[c] /= P.n;
    double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < P.n; ++i) {
      double pc[3];
      pc_of(i, pc);
      const float* pw = P.p3d + 3 * (size_t)P.idx[i];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) ABt[3 * a + b] += (pc[a] - pc0[a]) * (pw[b] - pw0[b]);
    }
    double R[9], t[3];
    procrustes_rotation(ABt, R);
    for (int a = 0; a < 3; ++a) t[a] = pc0[a] - (R[3 * a] * pw0[0] + R[3 * a + 1] * pw0[1] + R[3 * a + 2] * pw0[2]);
    const double err = reproj_mean(P, R, t);
    if (isfinite(err) && err < best_err) {
      best_err = err;
      found = true;
      for (int i = 0; i < 9; ++i) Rout[i] = R[i];
      for (int i = 0; i < 3; ++i) tout[i] = t[i];
    }
  }
  return found;
}

__device__ __forceinline__ bool is_inlier(const Points& P, const double* R, const double* t, int k, double thr2) {
  const float* pw = P.p3d + 3 * (size_t)k;
  const double X = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + t[0], Y = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + t[1];
  const double iz = 1.0 / (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + t[2]);
  const double du = P.uc + P.fu * X * iz - P.p2d[2 * (size_t)k], dv = P.vc + P.fv * Y * iz - P.p2d[2 * (size_t)k + 1];
  return du * du + dv * dv <= thr2;
}


// ---- shared staging: this crop's correspondences in LDS (every later pass reads them dozens of times), valid indices compacted in
// ascending order by a block-wide scan.  T = threads of the calling workgroup (a multiple of 64).
struct CropLds { float* p3d; float* p2d; int32_t* vidx; int* nv; int* wsum; };

template <int T, typename IdxT, typename F>
__device__ __forceinline__ int block_compact(F flag_of_point, int N, IdxT* out, int* wsum, int tid) {
  // ascending compaction of the points whose flag is set: chunks of T points, wave ballots + a scan over the waves' counts
  int base = 0;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < N; c0 += T) {
    const int i = c0 + tid;
    const bool f = i < N && flag_of_point(i);
    const unsigned long long m = __ballot(f);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (f) out[off + __popcll(m & ((1ull << lane) - 1ull))] = (IdxT)i;
    int tot = 0;
    for (int w = 0; w < T / 64; ++w) tot += wsum[w];
    base += tot;
    __syncthreads();
  }
  return base;
}

// OpenCV's RANSACUpdateNumIters(confidence = 0.99, outlier ratio, sample size, max): iterations after which a sample of all
// inliers has been drawn with that confidence, given the best inlier count so far
__device__ __forceinline__ int needed_iters(int best, int nv, int m, int iters) {
  if (best < m) return iters;
  double ep = 1.0 - (double)best / (double)nv;
  ep = ep < 0.0 ? 0.0 : (ep > 1.0 ? 1.0 : ep);
  const double num = log(1.0 - 0.99);
  double denom = 1.0 - pow(1.0 - ep, (double)m);
  if (denom < 2.2250738585072014e-308) return 0;
  denom = log(denom);
  if (denom >= 0.0 || -num >= (double)iters * (-denom)) return iters;
  return (int)nearbyint(num / denom);
}
// Hypotheses are evaluated in rounds of 64 (one launch each); round r runs only while 64 r is below the number of iterations the
// rule asks for given the best count of rounds 0 .. r - 1.  Returns how many hypothesis records are valid (a multiple of 64, or iters).
__device__ __forceinline__ int hypotheses_run(const double* hb, int nv, int m, int iters, int upto_round) {
  int best = -1, done = iters < 64 ? iters : 64;
  for (int r = 1; 64 * r < iters && r <= upto_round; ++r) {
    for (int h = 64 * (r - 1); h < 64 * r; ++h) { const int c = (int)hb[(size_t)h * PNP_HYP]; best = c > best ? c : best; }
    if (64 * r >= needed_iters(best, nv, m, iters)) return done;
    done = iters < 64 * (r + 1) ? iters : 64 * (r + 1);
  }
  return done;
}

// ---------------------------------------------------------------------------------------------- launch 1: the hypotheses
// one launch per round of 64 hypotheses, grid (1, B), one wave per workgroup, a hypothesis per lane (M^T M and its eigenvectors of
// all 64 lanes in 147 KB of LDS).  Round r > 0 first applies OpenCV's stopping rule to the records of the earlier rounds: with
// 70 % inliers 25 iterations suffice, so rounds 1 and 2 of the default 150 iterations usually return at once
__global__ __launch_bounds__(64) void pnp_hypotheses_kernel(const PnpParams p, double* __restrict__ hyp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* const sA = (double*)smem;                          // [144][64]: M^T M of lane l at [e * 64 + l]
  double* const sV = sA + 144 * 64;                          // [144][64]: its eigenvectors
  uint16_t* const vidx = (uint16_t*)(sV + 144 * 64);         // [N] valid indices, ascending
  __shared__ int wsum[1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t* valid = p.valid + (size_t)b * p.N * p.valid_stride;
  const int vs = p.valid_stride;
  const int nv = block_compact<64>([&](int i) { return valid[(size_t)i * vs] != 0; }, p.N, vidx, wsum, tid);
  const int h = p.round * 64 + tid;
  if (nv < 5) return;                                        // < 4: identity; exactly 4: P3P in the selection launch, no hypotheses
  if (p.round > 0 && hypotheses_run(hyp + (size_t)b * p.iters * PNP_HYP, nv, 5, p.iters, p.round) <= 64 * p.round)
    return;                                                  // the rule was satisfied by the earlier rounds: whole workgroup, uniform
  if (h >= p.iters) return;
  Points P;                                                  // straight from global memory: the scoring loop reads the SAME point on
  P.p3d = p.p3d + (size_t)b * p.p3d_bs;                      // every lane (one cache line per wave), the solve only its 5 samples
  P.p2d = p.p2d + (size_t)b * p.N * 2;
  const float* K = p.K + (size_t)b * p.K_bs;
  P.fu = K[0]; P.fv = K[4]; P.uc = K[2]; P.vc = K[5];
  const double thr2 = (double)p.thr * (double)p.thr;
  const int m = 5;
  int32_t sel[5];
  int got = 0;
  uint32_t tries = 0;
  while (got < m) {
    const int r = (int)(hash32(p.seed, (uint32_t)b, (uint32_t)h, tries++) % (uint32_t)nv);
    bool dup = false;
    for (int k = 0; k < got; ++k) dup = dup || sel[k] == (int32_t)vidx[r];
    if (!dup) sel[got++] = (int32_t)vidx[r];
  }
  Points S = P;
  S.idx = sel; S.n = m;
  Frame f;
  double R[9], t[3];
  bool ok = epnp_frame(S, f);
  if (ok) {
    double* const A = sA + tid;
    for (int i = 0; i < 144; ++i) A[i * 64] = 0.0;
    for (int i = 0; i < m; ++i) {
      double r0[12], r1[12];
      m_rows(S, f, i, r0, r1);
#pragma unroll
      for (int a = 0; a < 12; ++a)
#pragma unroll
        for (int c = a; c < 12; ++c) A[(a * 12 + c) * 64] += r0[a] * r0[c] + r1[a] * r1[c];
    }
    for (int a = 0; a < 12; ++a)
      for (int c = 0; c < a; ++c) A[(a * 12 + c) * 64] = A[(c * 12 + a) * 64];
    ok = epnp_finish(S, f, A, sV + tid, 64, R, t);
  }
  double* rec = hyp + ((size_t)b * p.iters + h) * PNP_HYP;
  int cnt = -1;
  if (ok) {
    cnt = 0;
    for (int i = 0; i < nv; ++i) cnt += is_inlier(P, R, t, (int)vidx[i], thr2) ? 1 : 0;
    for (int i = 0; i < 9; ++i) rec[2 + i] = R[i];
    for (int i = 0; i < 3; ++i) rec[11 + i] = t[i];
  }
  rec[0] = (double)cnt;
}

// ---------------------------------------------------------------------------------------------- launch 2: selection + final EPnP
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* sred, int tid) {      // v[NV] summed over the 256 threads -> v on every thread
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if (lane == 0) sred[wave * NV + k] = x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = sred[k] + sred[NV + k] + sred[2 * NV + k] + sred[3 * NV + k];
  __syncthreads();
}

__global__ __launch_bounds__(PNP_THREADS) void pnp_select_refit_kernel(const PnpParams p, const double* __restrict__ hyp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const s3 = (float*)smem;
  float* const s2 = s3 + 3 * p.N;
  int32_t* const vidx = (int32_t*)(s2 + 2 * p.N);
  int32_t* const iidx = vidx + p.N;
  bool* const flag = (bool*)(iidx + p.N);
  __shared__ double sred[4 * 27];
  __shared__ double s_mtm[144], s_evec[144];
  __shared__ double s_cc[3][4][3], s_Rt[3][12];
  __shared__ Frame s_frame;
  __shared__ int wsum[PNP_THREADS / 64], s_best, s_ok, s_kok[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* g3 = p.p3d + (size_t)b * p.p3d_bs;
  const float* g2 = p.p2d + (size_t)b * p.N * 2;
  const uint8_t* valid = p.valid + (size_t)b * p.N * p.valid_stride;
  for (int i = tid; i < 3 * p.N; i += PNP_THREADS) s3[i] = g3[i];
  for (int i = tid; i < 2 * p.N; i += PNP_THREADS) s2[i] = g2[i];
  for (int i = tid; i < p.N; i += PNP_THREADS) { flag[i] = valid[(size_t)i * p.valid_stride] != 0; p.inliers[(size_t)b * p.N + i] = 0; }
  __syncthreads();
  const int nv = block_compact<PNP_THREADS>([&](int i) { return flag[i]; }, p.N, vidx, wsum, tid);
  double* pose = p.pose + (size_t)b * 12;
  auto identity = [&]() {                                    // the reference's fallback: identity pose, no inliers
    if (tid == 0) {
      for (int i = 0; i < 9; ++i) pose[i] = (i % 4 == 0) ? 1.0 : 0.0;
      pose[9] = pose[10] = pose[11] = 0.0;
      p.status[b] = 0;
    }
  };
  if (nv < 4) { identity(); return; }
  const int m = 5;
  Points P;
  P.p3d = s3; P.p2d = s2;
  const float* K = p.K + (size_t)b * p.K_bs;
  P.fu = K[0]; P.fv = K[4]; P.uc = K[2]; P.vc = K[5];
  if (nv == 4) {                                             // OpenCV: no RANSAC, P3P + the fourth point; all four are inliers
    if (tid == 0) {
      float pw4[12], uv4[8];
      for (int i = 0; i < 4; ++i) {
        const int k = vidx[i];
        for (int c = 0; c < 3; ++c) pw4[3 * i + c] = s3[3 * k + c];
        uv4[2 * i] = s2[2 * k]; uv4[2 * i + 1] = s2[2 * k + 1];
      }
      double R4[9], t4[3];
      if (solve_four_points(pw4, uv4, P.fu, P.fv, P.uc, P.vc, R4, t4)) {
        for (int i = 0; i < 9; ++i) pose[i] = R4[i];
        for (int i = 0; i < 3; ++i) pose[9 + i] = t4[i];
        for (int i = 0; i < 4; ++i) p.inliers[(size_t)b * p.N + vidx[i]] = 1;
        p.status[b] = 1;
      } else {
        for (int i = 0; i < 9; ++i) pose[i] = (i % 4 == 0) ? 1.0 : 0.0;
        pose[9] = pose[10] = pose[11] = 0.0;
        p.status[b] = 0;
      }
    }
    return;
  }
  const double thr2 = (double)p.thr * (double)p.thr;
  const double* hb = hyp + (size_t)b * p.iters * PNP_HYP;
  if (tid == 0) {                                            // most inliers, first on ties, at least a full sample
    int best = -1, bc = m - 1;
    const int nrun = hypotheses_run(hb, nv, m, p.iters, PNP_MAX_ITERS / 64);
    for (int h = 0; h < nrun; ++h) {
      const int c = (int)hb[(size_t)h * PNP_HYP];
      if (c > bc) { bc = c; best = h; }
    }
    s_best = best;
  }
  __syncthreads();
  const int best = s_best;
  if (best < 0) { identity(); return; }
  double Rb[9], tb[3];
  for (int i = 0; i < 9; ++i) Rb[i] = hb[(size_t)best * PNP_HYP + 2 + i];
  for (int i = 0; i < 3; ++i) tb[i] = hb[(size_t)best * PNP_HYP + 11 + i];
  __syncthreads();
  for (int i = tid; i < p.N; i += PNP_THREADS) flag[i] = false;
  __syncthreads();
  for (int i = tid; i < nv; i += PNP_THREADS) {
    const int k = vidx[i];
    const bool in = is_inlier(P, Rb, tb, k, thr2);
    flag[k] = in;
    if (in) p.inliers[(size_t)b * p.N + k] = 1;
  }
  __syncthreads();
  const int n = block_compact<PNP_THREADS>([&](int i) { return flag[i]; }, p.N, iidx, wsum, tid);
  Points S = P;
  S.idx = iidx; S.n = n;
  // ---- final EPnP over the n inliers, loops over the points shared by the 256 threads, the small dense algebra on thread 0
  double acc[27];
  for (int k = 0; k < 3; ++k) acc[k] = 0.0;
  for (int i = tid; i < n; i += PNP_THREADS) { const float* pw = s3 + 3 * iidx[i]; acc[0] += pw[0]; acc[1] += pw[1]; acc[2] += pw[2]; }
  block_sum<3>(acc, sred, tid);
  const double pw0[3] = {acc[0] / n, acc[1] / n, acc[2] / n};
  for (int k = 0; k < 6; ++k) acc[k] = 0.0;
  for (int i = tid; i < n; i += PNP_THREADS) {
    const float* pw = s3 + 3 * iidx[i];
    const double d0 = pw[0] - pw0[0], d1 = pw[1] - pw0[1], d2 = pw[2] - pw0[2];
    acc[0] += d0 * d0; acc[1] += d0 * d1; acc[2] += d0 * d2; acc[3] += d1 * d1; acc[4] += d1 * d2; acc[5] += d2 * d2;
  }
  block_sum<6>(acc, sred, tid);
  if (tid == 0) {
    const double Sm[9] = {acc[0], acc[1], acc[2], acc[1], acc[3], acc[4], acc[2], acc[4], acc[5]};
    s_ok = epnp_frame_from(pw0, Sm, n, s_frame) ? 1 : 0;
  }
  __syncthreads();
  bool ok = s_ok != 0;
  if (ok && tid < 78) {                                      // entry (a, c), a <= c, of M^T M per thread
    int a = 0, rem = tid;
    while (rem >= 12 - a) { rem -= 12 - a; ++a; }
    const int c = a + rem;
    double e = 0.0;
    for (int i = 0; i < n; ++i) {
      double r0[12], r1[12];
      m_rows(S, s_frame, i, r0, r1);
      e += r0[a] * r0[c] + r1[a] * r1[c];
    }
    s_mtm[a * 12 + c] = e;
    s_mtm[c * 12 + a] = e;
  }
  __syncthreads();
  if (ok && tid < 64) jacobi_eig12_wave(s_mtm, s_evec, tid);  // wave 0: the 12 x 12 eigen-solve on 12 lanes
  __syncthreads();
  if (ok && tid == 0) {                                      // null-space basis, betas of the three approximations, camera-frame control points
    double v[4][12], cc[3][4][3];
    bool kok[3];
    epnp_betas(S, s_frame, s_mtm, s_evec, -1, v, cc, kok);
    for (int k = 0; k < 3; ++k) {
      s_kok[k] = kok[k] ? 1 : 0;
      for (int j = 0; j < 4; ++j)
        for (int c = 0; c < 3; ++c) s_cc[k][j][c] = cc[k][j][c];
    }
  }
  __syncthreads();
  if (ok) {
    // pc0 of the three candidates (9 sums), then their cross-covariances with the model points (27 sums), then their errors (3)
    auto pc_of = [&](int k, int i, double* pc) {
      double al[4];
      alphas_of(s_frame, s3 + 3 * iidx[i], al);
      for (int c = 0; c < 3; ++c) pc[c] = al[0] * s_cc[k][0][c] + al[1] * s_cc[k][1][c] + al[2] * s_cc[k][2][c] + al[3] * s_cc[k][3][c];
    };
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int i = tid; i < n; i += PNP_THREADS)
      for (int k = 0; k < 3; ++k) { double pc[3]; pc_of(k, i, pc); acc[3 * k] += pc[0]; acc[3 * k + 1] += pc[1]; acc[3 * k + 2] += pc[2]; }
    block_sum<9>(acc, sred, tid);
    double pc0[3][3];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) pc0[k][c] = acc[3 * k + c] / n;
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    for (int i = tid; i < n; i += PNP_THREADS) {
      const float* pw = s3 + 3 * iidx[i];
      for (int k = 0; k < 3; ++k) {
        double pc[3];
        pc_of(k, i, pc);
        for (int a = 0; a < 3; ++a)
          for (int c = 0; c < 3; ++c) acc[9 * k + 3 * a + c] += (pc[a] - pc0[k][a]) * (pw[c] - pw0[c]);
      }
    }
    block_sum<27>(acc, sred, tid);
    if (tid < 3 && s_kok[tid]) {                             // absolute orientation of candidate `tid`
      double R[9], t[3];
      procrustes_rotation(acc + 9 * tid, R);
      for (int a = 0; a < 3; ++a) t[a] = pc0[tid][a] - (R[3 * a] * pw0[0] + R[3 * a + 1] * pw0[1] + R[3 * a + 2] * pw0[2]);
      for (int i = 0; i < 9; ++i) s_Rt[tid][i] = R[i];
      for (int i = 0; i < 3; ++i) s_Rt[tid][9 + i] = t[i];
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) acc[k] = 0.0;
    for (int i = tid; i < n; i += PNP_THREADS) {
      const int kk = iidx[i];
      const float* pw = s3 + 3 * kk;
      for (int k = 0; k < 3; ++k) {
        if (!s_kok[k]) continue;
        const double* R = s_Rt[k];
        const double X = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + R[9], Y = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + R[10];
        const double iz = 1.0 / (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + R[11]);
        const double du = P.uc + P.fu * X * iz - s2[2 * kk], dv = P.vc + P.fv * Y * iz - s2[2 * kk + 1];
        acc[k] += sqrt(du * du + dv * dv);
      }
    }
    block_sum<3>(acc, sred, tid);
    if (tid == 0) {
      int pick = -1;
      double be = INFINITY;
      for (int k = 0; k < 3; ++k)
        if (s_kok[k] && isfinite(acc[k]) && acc[k] < be) { be = acc[k]; pick = k; }
      if (pick >= 0) {
        for (int i = 0; i < 12; ++i) pose[i] = s_Rt[pick][i];
      } else ok = false;
      s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    ok = s_ok != 0;
  }
  if (tid == 0) {
    if (!ok) {                                               // degenerate inlier set: keep the winning hypothesis
      for (int i = 0; i < 9; ++i) pose[i] = Rb[i];
      for (int i = 0; i < 3; ++i) pose[9 + i] = tb[i];
    }
    p.status[b] = 1;
  }
}

}  // namespace

extern "C" size_t cp_pnp_ransac_scratch_bytes(int B, int N) { (void)N; return (size_t)B * PNP_MAX_ITERS * PNP_HYP * sizeof(double); }

extern "C" int cp_pnp_ransac(cp_stream_t stream, const float* p3d, long long p3d_bstride, const float* p2d, const uint8_t* valid,
                             int valid_stride, const float* cam_K, long long K_bstride, int B, int N, float reproj_threshold,
                             int iterations, uint32_t seed, double* pose, uint8_t* inliers, int32_t* status, void* scratch) {
  if (!p3d || !p2d || !valid || !cam_K || !pose || !inliers || !status || !scratch) return CP_ERR_INVALID;
  if (B <= 0 || N <= 0 || N > PNP_NMAX || valid_stride <= 0 || iterations <= 0 || iterations > PNP_MAX_ITERS || !(reproj_threshold > 0.f))
    return CP_ERR_INVALID;
  if (p3d_bstride != 0 && p3d_bstride < 3LL * N) return CP_ERR_INVALID;
  if (K_bstride != 0 && K_bstride < 9) return CP_ERR_INVALID;
  if (((uintptr_t)pose & 7) || ((uintptr_t)scratch & 7) || ((uintptr_t)status & 3)) return CP_ERR_ALIGN;
  PnpParams p;
  p.p3d = p3d; p.p2d = p2d; p.valid = valid; p.K = cam_K; p.pose = pose; p.inliers = inliers; p.status = status; p.scratch = nullptr;
  p.p3d_bs = p3d_bstride; p.K_bs = K_bstride; p.B = B; p.N = N; p.valid_stride = valid_stride; p.iters = iterations; p.thr = reproj_threshold;
  p.seed = seed;
  const size_t lds1 = (size_t)2 * 144 * 64 * 8 + (size_t)N * 2 + 16, lds2 = (size_t)N * (5 * 4 + 8 + 1) + 16;
  static CpDeviceOnce once;                  // both kernels are sized for PNP_NMAX keypoints, once per device
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)pnp_hypotheses_kernel, (size_t)2 * 144 * 64 * 8 + PNP_NMAX * 2 + 16) &&
                                  cp_set_max_lds((const void*)pnp_select_refit_kernel, (size_t)PNP_NMAX * 29 + 16));
  hipStream_t st = (hipStream_t)stream;
  for (int r = 0; 64 * r < iterations; ++r) {
    p.round = r;
    CP_LAUNCH(pnp_hypotheses_kernel, dim3((unsigned)B), dim3(64), lds1, st, p, (double*)scratch);
  }
  p.round = 0;
  CP_LAUNCH(pnp_select_refit_kernel, dim3((unsigned)B), dim3(PNP_THREADS), lds2, st, p, (const double*)scratch);
  return cp_check_launch();
}
