// Pose from correspondences on the device: the solver call of the reference's from_id_to_pose
//   cv2.solvePnPRansac(valid_p3d, valid_disc_p2d, cam_K, None, reprojectionError=2, iterationsCount=150, flags=cv2.SOLVEPNP_EPNP)
// (test_network_with_test_data.py:100-110; identity pose below 4 valid correspondences, :111-114) as a consumer of
// cp_correspondences' output: ONE launch for the whole batch, one workgroup per crop, so the (B, N, 2) coordinates and validity
// masks never leave the GPU -- only 12 doubles per crop do.  opencv-python is not vendored by the reference (and absent here): this
// is the published algorithm -- EPnP (Lepetit, Moreno-Noguer, Fua 2009) in the structure of OpenCV's epnp.cpp inside the RANSAC
// frame of OpenCV's solvePnPRansac -- restated in oracle/pnp_oracle.py, which states the deliberate differences (sample sequence
// from a counter-based hash, no early termination).  All arithmetic in fp64.
//   phase 1  thread 0 compacts the valid indices (ascending);
//   phase 2  thread h < iterations: draws 5 (4 if only 4 are valid) distinct correspondences, EPnP -> pose h, counts the valid
//            correspondences with squared reprojection error <= threshold^2;
//   phase 3  best = most inliers (first on ties, at least a full sample); its inlier list is compacted;
//   phase 4  EPnP over the inliers: the 78 distinct entries of M^T M by 78 threads, the rest by thread 0.
#include "common.h"

namespace {

struct PnpParams {
  const float* p3d; const float* p2d; const uint8_t* valid; const float* K;
  double* pose; uint8_t* inliers; int32_t* status; int32_t* scratch;
  long long p3d_bs, K_bs;
  int B, N, valid_stride, iters;
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

__device__ bool epnp_frame(const Points& P, Frame& f) {
  const int n = P.n;
  double c[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) { const float* pw = P.p3d + 3 * (size_t)P.idx[i]; c[0] += pw[0]; c[1] += pw[1]; c[2] += pw[2]; }
  for (int k = 0; k < 3; ++k) f.cw[0][k] = c[k] / n;
  double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, E[9];
  for (int i = 0; i < n; ++i) {
    const float* pw = P.p3d + 3 * (size_t)P.idx[i];
    const double d[3] = {pw[0] - f.cw[0][0], pw[1] - f.cw[0][1], pw[2] - f.cw[0][2]};
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) S[3 * a + b] += d[a] * d[b];
  }
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

// everything behind M^T M (row-major 12 x 12, destroyed): null-space basis, betas, Gauss-Newton, absolute orientation
__device__ bool epnp_finish(const Points& P, const Frame& f, double* MtM, double* Rout, double* tout) {
  double V[144];
  jacobi_eig<12>(MtM, V);
  int ord[4];                                              // the 4 smallest eigenvalues, ascending
  {
    bool used[12];
    for (int i = 0; i < 12; ++i) used[i] = false;
    for (int k = 0; k < 4; ++k) {
      int best = -1;
      for (int i = 0; i < 12; ++i)
        if (!used[i] && (best < 0 || MtM[13 * i] < MtM[13 * best])) best = i;
      used[best] = true;
      ord[k] = best;
    }
  }
  double v[4][12];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 12; ++i) v[k][i] = V[i * 12 + ord[k]];
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
  double pw0[3] = {0, 0, 0};
  for (int i = 0; i < P.n; ++i) { const float* pw = P.p3d + 3 * (size_t)P.idx[i]; pw0[0] += pw[0]; pw0[1] += pw[1]; pw0[2] += pw[2]; }
  for (int c = 0; c < 3; ++c) pw0[c] /= P.n;
  double best_err = INFINITY;
  bool found = false;
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
    if (!ok || !(isfinite(be[0]) && isfinite(be[1]) && isfinite(be[2]) && isfinite(be[3]))) continue;
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
    // camera-frame control points, sign, absolute orientation
    double cc[4][3];
    for (int j = 0; j < 4; ++j)
      for (int c = 0; c < 3; ++c) cc[j][c] = be[0] * v[0][3 * j + c] + be[1] * v[1][3 * j + c] + be[2] * v[2][3 * j + c] + be[3] * v[3][3 * j + c];
    auto pc_of = [&](int i, double* pc) {
      double al[4];
      alphas_of(f, P.p3d + 3 * (size_t)P.idx[i], al);
      for (int c = 0; c < 3; ++c) pc[c] = al[0] * cc[0][c] + al[1] * cc[1][c] + al[2] * cc[2][c] + al[3] * cc[3][c];
    };
    double p0[3];
    pc_of(0, p0);
    if (p0[2] < 0.0)
      for (int j = 0; j < 4; ++j)
        for (int c = 0; c < 3; ++c) cc[j][c] = -cc[j][c];
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

constexpr int PNP_THREADS = 256, PNP_MAX_ITERS = 256;

__global__ __launch_bounds__(PNP_THREADS) void pnp_ransac_kernel(const PnpParams p) {
  __shared__ double s_pose[PNP_MAX_ITERS][12];
  __shared__ int s_cnt[PNP_MAX_ITERS];
  __shared__ double s_mtm[144];
  __shared__ Frame s_frame;
  __shared__ int s_nv, s_best, s_ninl, s_ok;
  const int b = blockIdx.x, tid = threadIdx.x;
  Points P;
  P.p3d = p.p3d + (size_t)b * p.p3d_bs;
  P.p2d = p.p2d + (size_t)b * p.N * 2;
  const float* K = p.K + (size_t)b * p.K_bs;
  P.fu = K[0]; P.fv = K[4]; P.uc = K[2]; P.vc = K[5];
  const uint8_t* valid = p.valid + (size_t)b * p.N * p.valid_stride;
  int32_t* vidx = p.scratch + (size_t)b * 2 * p.N;         // valid indices, then the inlier list
  int32_t* iidx = vidx + p.N;
  const double thr2 = (double)p.thr * (double)p.thr;
  if (tid == 0) {
    int n = 0;
    for (int i = 0; i < p.N; ++i)
      if (valid[(size_t)i * p.valid_stride]) vidx[n++] = i;
    s_nv = n;
  }
  for (int i = tid; i < p.N; i += PNP_THREADS) p.inliers[(size_t)b * p.N + i] = 0;
  __syncthreads();
  const int nv = s_nv;
  double* pose = p.pose + (size_t)b * 12;
  if (nv < 4) {                                            // the reference's fallback: identity pose, no inliers
    if (tid == 0) {
      for (int i = 0; i < 9; ++i) pose[i] = (i % 4 == 0) ? 1.0 : 0.0;
      pose[9] = pose[10] = pose[11] = 0.0;
      p.status[b] = 0;
    }
    return;
  }
  const int m = nv >= 5 ? 5 : 4;
  // ---- hypotheses
  for (int h = tid; h < p.iters; h += PNP_THREADS) {
    int32_t sel[5];
    int got = 0;
    uint32_t tries = 0;
    while (got < m) {
      const int r = (int)(hash32(p.seed, (uint32_t)b, (uint32_t)h, tries++) % (uint32_t)nv);
      bool dup = false;
      for (int k = 0; k < got; ++k) dup = dup || sel[k] == vidx[r];
      if (!dup) sel[got++] = vidx[r];
    }
    Points S = P;
    S.idx = sel; S.n = m;
    Frame f;
    double R[9], t[3];
    bool ok = epnp_frame(S, f);
    if (ok) {
      double MtM[144];
      for (int i = 0; i < 144; ++i) MtM[i] = 0.0;
      for (int i = 0; i < m; ++i) {
        double r0[12], r1[12];
        m_rows(S, f, i, r0, r1);
        for (int a = 0; a < 12; ++a)
          for (int c = a; c < 12; ++c) MtM[a * 12 + c] += r0[a] * r0[c] + r1[a] * r1[c];
      }
      for (int a = 0; a < 12; ++a)
        for (int c = 0; c < a; ++c) MtM[a * 12 + c] = MtM[c * 12 + a];
      ok = epnp_finish(S, f, MtM, R, t);
    }
    int cnt = -1;
    if (ok) {
      cnt = 0;
      for (int i = 0; i < nv; ++i) cnt += is_inlier(P, R, t, vidx[i], thr2) ? 1 : 0;
      for (int i = 0; i < 9; ++i) s_pose[h][i] = R[i];
      for (int i = 0; i < 3; ++i) s_pose[h][9 + i] = t[i];
    }
    s_cnt[h] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    int best = -1, bc = m - 1;
    for (int h = 0; h < p.iters; ++h)
      if (s_cnt[h] > bc) { bc = s_cnt[h]; best = h; }
    s_best = best;
    int n = 0;
    if (best >= 0) {
      double R[9], t[3];
      for (int i = 0; i < 9; ++i) R[i] = s_pose[best][i];
      for (int i = 0; i < 3; ++i) t[i] = s_pose[best][9 + i];
      for (int i = 0; i < nv; ++i)
        if (is_inlier(P, R, t, vidx[i], thr2)) { iidx[n++] = vidx[i]; p.inliers[(size_t)b * p.N + vidx[i]] = 1; }
    }
    s_ninl = n;
    s_ok = 0;
    if (best >= 0) {
      Points S = P;
      S.idx = iidx; S.n = n;
      s_ok = epnp_frame(S, s_frame) ? 1 : 0;
    }
  }
  __syncthreads();
  if (s_best < 0) {
    if (tid == 0) {
      for (int i = 0; i < 9; ++i) pose[i] = (i % 4 == 0) ? 1.0 : 0.0;
      pose[9] = pose[10] = pose[11] = 0.0;
      p.status[b] = 0;
    }
    return;
  }
  // ---- final EPnP over the inliers: entry (a, c), a <= c, of M^T M per thread
  Points S = P;
  S.idx = iidx; S.n = s_ninl;
  if (s_ok && tid < 78) {
    int a = 0, rem = tid;
    while (rem >= 12 - a) { rem -= 12 - a; ++a; }
    const int c = a + rem;
    double acc = 0.0;
    for (int i = 0; i < S.n; ++i) {
      double r0[12], r1[12];
      m_rows(S, s_frame, i, r0, r1);
      acc += r0[a] * r0[c] + r1[a] * r1[c];
    }
    s_mtm[a * 12 + c] = acc;
    s_mtm[c * 12 + a] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    double R[9], t[3];
    bool ok = s_ok != 0;
    if (ok) {
      double MtM[144];
      for (int i = 0; i < 144; ++i) MtM[i] = s_mtm[i];
      ok = epnp_finish(S, s_frame, MtM, R, t);
    }
    if (!ok) {                                             // degenerate inlier set: keep the winning hypothesis
      for (int i = 0; i < 9; ++i) R[i] = s_pose[s_best][i];
      for (int i = 0; i < 3; ++i) t[i] = s_pose[s_best][9 + i];
    }
    for (int i = 0; i < 9; ++i) pose[i] = R[i];
    for (int i = 0; i < 3; ++i) pose[9 + i] = t[i];
    p.status[b] = 1;
  }
}

}  // namespace

extern "C" size_t cp_pnp_ransac_scratch_bytes(int B, int N) { return (size_t)B * 2 * N * sizeof(int32_t); }

extern "C" int cp_pnp_ransac(cp_stream_t stream, const float* p3d, long long p3d_bstride, const float* p2d, const uint8_t* valid,
                             int valid_stride, const float* cam_K, long long K_bstride, int B, int N, float reproj_threshold,
                             int iterations, uint32_t seed, double* pose, uint8_t* inliers, int32_t* status, void* scratch) {
  if (!p3d || !p2d || !valid || !cam_K || !pose || !inliers || !status || !scratch) return CP_ERR_INVALID;
  if (B <= 0 || N <= 0 || valid_stride <= 0 || iterations <= 0 || iterations > PNP_MAX_ITERS || !(reproj_threshold > 0.f)) return CP_ERR_INVALID;
  if (p3d_bstride != 0 && p3d_bstride < 3LL * N) return CP_ERR_INVALID;
  if (K_bstride != 0 && K_bstride < 9) return CP_ERR_INVALID;
  if (((uintptr_t)pose & 7) || ((uintptr_t)scratch & 3) || ((uintptr_t)status & 3)) return CP_ERR_ALIGN;
  PnpParams p;
  p.p3d = p3d; p.p2d = p2d; p.valid = valid; p.K = cam_K; p.pose = pose; p.inliers = inliers; p.status = status; p.scratch = (int32_t*)scratch;
  p.p3d_bs = p3d_bstride; p.K_bs = K_bstride; p.B = B; p.N = N; p.valid_stride = valid_stride; p.iters = iterations; p.thr = reproj_threshold;
  p.seed = seed;
  CP_LAUNCH(pnp_ransac_kernel, dim3((unsigned)B), dim3(PNP_THREADS), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}
