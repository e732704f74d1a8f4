// Train-mode EdgeConv (StaticGraph_module, reference checkerpose/model/init.py:54-68 == pipeline.py:45-59) in the
// FACTORED form, with BatchNorm2d in batch-statistics mode and its full backward -- without ever materialising the
// (B, C', N, K) edge tensor.  pq = [P | Q] = x [W1 ; W2-W1]^T is the raw node GEMM output (B, N, 2C'), and the edge
// pre-activation is e[b,i,k,c] = P[b, idx[i,k], c] + Q[b,i,c].
//
//   statistics  : sum_k e   = S + K Q,   sum_k e^2 = S2 + 2 Q S + K Q^2     (S, S2 = gather-sums of P, P^2)
//   forward     : out = leaky( max_k (s P_j) + s Q_i + t ),  s = gamma*rstd, t = beta - mu*s ; k* saved (uint8)
//   backward    : gq = gout * leaky'(out);  dz[b,i,k] = gq [k == k*];  m1 = mean(dz), m2 = mean(dz * ehat) over all
//                 B*N*K edges;  de = s (dz - m1 - ehat m2)  is DENSE over the edges, but its two marginals are not:
//     dQ_i = sum_k de           = s ( gq_i - K m1 - m2 rstd (S_i + K Q_i - K mu) )
//     dP_j = sum_{(i,k)->j} de  = s ( G_j - deg_j m1 - m2 rstd (deg_j P_j + R_j - deg_j mu) )
//                 G_j = sum of gq_i over the reverse edges that won the max, R_j = sum of Q_i over all reverse edges.
//   The node GEMM's backward then sees D = [dP - dQ | dQ] (so that dW1 = D1^T x, dW2 = D2^T x, dx = [W1 W2]-view).
//
// Same mapping as the eval gather kernel (graph_ops.hip): thread = one 16-byte channel group of one keypoint, the
// neighbour lists of a block's keypoints staged in LDS, all blocks of a crop on one XCD (blockIdx % 8 label) so the
// crop's tables are served by that XCD's L2; the reverse graph (rev_ptr / rev_edge, static per object) turns every
// scatter into a deterministic gather.  Per-channel sums: fp64 block partials + one finalize launch.
#include "common.h"
#include <cstring>

int cp_bn_finalize_launch(hipStream_t st, const double* partial, int nblk, int CP, int C, double count, const float* gamma,
                          const float* beta, float eps, float momentum, float* rmean, float* rvar, float* scale, float* shift,
                          float* mean, float* rstd);
int cp_bn_bwd_finalize_launch(hipStream_t st, const double* partial, int nblk, int CP, int C, double count, const float* gamma,
                              const float* mean, const float* rstd, float* coef, float* dgamma, float* dbeta);

struct EdgeTrainParams {
  const void* pq; const int32_t* idx; const int32_t* gids;
  const int32_t* rev_ptr; const int32_t* rev_edge;
  const float* scale; const float* shift;       // forward: s, t
  const float* coef; int Cvec;                   // backward: a, b, cr, mu
  void* out; int out_cs, out_coff;               // forward output / saved forward output (backward)
  const void* gout; int g_cs, g_coff;
  uint8_t* kstar;
  void* dpq;                                     // (B,N,2C) dtype: [dP - dQ | dQ]
  double* partial;
  int B, N, K, S, npb;                           // S sub-blocks per crop, npb keypoints per sub-block
  float slope;
};

template <typename Tag, int TPK>
__device__ __forceinline__ bool edge_block_setup(const EdgeTrainParams& p, int& b, int& sidx, int& g) {
  const int label = blockIdx.x & 7, j = blockIdx.x >> 3;
  b = label + 8 * (j / p.S);
  sidx = j % p.S;
  if (b >= p.B) return false;
  g = p.gids ? p.gids[b] : 0;
  return true;
}

// block-level fp64 reduction of per-thread (s1[E], s2[E]) over the KPB keypoint lanes -> partial[blk][2][C]
template <typename Tag, int TPK>
__device__ __forceinline__ void edge_block_reduce(const double* s1, const double* s2, double* red, double* partial, int blk) {
  constexpr int E = Tag::E, KPB = 256 / TPK, C = TPK * E;
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < E; ++e) { red[tid * 2 * E + e] = s1[e]; red[tid * 2 * E + E + e] = s2[e]; }
  __syncthreads();
  for (int o = tid; o < 2 * C; o += 256) {
    const int which = o / C, c = o - which * C;
    const int cg = c / E, e = c - cg * E;
    double s = 0.0;
    for (int r = 0; r < KPB; ++r) s += red[(r * TPK + cg) * 2 * E + which * E + e];
    partial[((size_t)blk * 2 + which) * C + c] = s;
  }
}

// MODE 0: batch statistics of e.  MODE 1: backward reductions (sum gq, sum gq * ehat*).
template <typename Tag, int TPK, int MODE>
__global__ __launch_bounds__(256) void edge_reduce_kernel(const EdgeTrainParams p) {
  constexpr int E = Tag::E, KPB = 256 / TPK, C = TPK * E;
  using T = typename Tag::elem;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* red = (double*)smem;                               // 256 * 2E doubles
  int32_t* s_idx = (int32_t*)(smem + 256 * 2 * E * sizeof(double));
  int b, sidx, g;
  if (!edge_block_setup<Tag, TPK>(p, b, sidx, g)) return;
  const int kp_l = threadIdx.x / TPK, cg = threadIdx.x % TPK;
  const u32x4* rows = (const u32x4*)p.pq + (size_t)b * p.N * (2 * TPK);
  double s1[E], s2[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
  float mu[E], rs[E];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < E; ++e) { mu[e] = p.coef[3 * p.Cvec + cg * E + e]; rs[e] = p.scale[cg * E + e]; }   // scale := rstd here
  }
  const int kp_end = min((sidx + 1) * p.npb, p.N);
  for (int kp0 = sidx * p.npb; kp0 < kp_end; kp0 += KPB) {
    const int nkp = min(KPB, kp_end - kp0);
    if (MODE == 0) {
      __syncthreads();
      const int32_t* gidx = p.idx + ((size_t)g * p.N + kp0) * p.K;
      for (int t = threadIdx.x; t < nkp * p.K; t += 256) s_idx[t] = gidx[t];
      __syncthreads();
    }
    if (kp_l >= nkp) continue;
    const int i = kp0 + kp_l;
    float f[E], q[E];
    Vec16<Tag>::unpack(rows[(size_t)i * (2 * TPK) + TPK + cg], q);
    if (MODE == 0) {
      float a1[E], a2[E];
#pragma unroll
      for (int e = 0; e < E; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
      const int32_t* my = s_idx + kp_l * p.K;
      int k = 0;
      for (; k + 4 <= p.K; k += 4) {        // four rows in flight
        u32x4 r4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r4[u] = rows[(size_t)my[k + u] * (2 * TPK) + cg];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          Vec16<Tag>::unpack(r4[u], f);
#pragma unroll
          for (int e = 0; e < E; ++e) { a1[e] += f[e]; a2[e] += f[e] * f[e]; }
        }
      }
      for (; k < p.K; ++k) {
        Vec16<Tag>::unpack(rows[(size_t)my[k] * (2 * TPK) + cg], f);
#pragma unroll
        for (int e = 0; e < E; ++e) { a1[e] += f[e]; a2[e] += f[e] * f[e]; }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const double S = a1[e], S2 = a2[e], Q = q[e], K = p.K;
        s1[e] += S + K * Q;
        s2[e] += S2 + 2.0 * Q * S + K * Q * Q;
      }
    } else {
      const size_t node = (size_t)b * p.N + i;
      float go[E], ov[E];
      Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.gout + node * p.g_cs + p.g_coff + cg * E), go);
      Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.out + node * p.out_cs + p.out_coff + cg * E), ov);
      const uint8_t* ks = p.kstar + node * C + cg * E;
      const int32_t* nb = p.idx + ((size_t)g * p.N + i) * p.K;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float gq = ov[e] > 0.f ? go[e] : go[e] * p.slope;
        const int j = nb[ks[e]];
        const float pj = load_elem<Tag>(p.pq, ((size_t)b * p.N + j) * (2 * C) + cg * E + e);
        const float eh = (pj + q[e] - mu[e]) * rs[e];
        s1[e] += (double)gq;
        s2[e] += (double)gq * (double)eh;
      }
    }
  }
  if (MODE == 0) __syncthreads();
  edge_block_reduce<Tag, TPK>(s1, s2, red, p.partial, b * p.S + sidx);
}

// forward: out = leaky(max_k (s P_j) + s Q_i + t), first arg-max saved
template <typename Tag, int TPK>
__global__ __launch_bounds__(256) void edge_train_fwd_kernel(const EdgeTrainParams p) {
  constexpr int E = Tag::E, KPB = 256 / TPK, C = TPK * E;
  using T = typename Tag::elem;
  extern __shared__ __attribute__((aligned(16))) int32_t s_idx2[];
  int b, sidx, g;
  if (!edge_block_setup<Tag, TPK>(p, b, sidx, g)) return;
  const int kp_l = threadIdx.x / TPK, cg = threadIdx.x % TPK;
  const u32x4* rows = (const u32x4*)p.pq + (size_t)b * p.N * (2 * TPK);
  float sc[E], sh[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { sc[e] = p.scale[cg * E + e]; sh[e] = p.shift[cg * E + e]; }
  const int kp_end = min((sidx + 1) * p.npb, p.N);
  for (int kp0 = sidx * p.npb; kp0 < kp_end; kp0 += KPB) {
    const int nkp = min(KPB, kp_end - kp0);
    __syncthreads();
    const int32_t* gidx = p.idx + ((size_t)g * p.N + kp0) * p.K;
    for (int t = threadIdx.x; t < nkp * p.K; t += 256) s_idx2[t] = gidx[t];
    __syncthreads();
    if (kp_l >= nkp) continue;
    const int i = kp0 + kp_l;
    float m[E], f[E];
    int ks[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { m[e] = -INFINITY; ks[e] = 0; }
    const int32_t* my = s_idx2 + kp_l * p.K;
    int k = 0;
    for (; k + 4 <= p.K; k += 4) {          // four rows in flight; the comparisons stay in list order (first arg-max wins)
      u32x4 r4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r4[u] = rows[(size_t)my[k + u] * (2 * TPK) + cg];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        Vec16<Tag>::unpack(r4[u], f);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float v = f[e] * sc[e];
          if (v > m[e]) { m[e] = v; ks[e] = k + u; }
        }
      }
    }
    for (; k < p.K; ++k) {
      Vec16<Tag>::unpack(rows[(size_t)my[k] * (2 * TPK) + cg], f);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float v = f[e] * sc[e];
        if (v > m[e]) { m[e] = v; ks[e] = k; }
      }
    }
    Vec16<Tag>::unpack(rows[(size_t)i * (2 * TPK) + TPK + cg], f);
    const size_t node = (size_t)b * p.N + i;
    uint8_t* kp = p.kstar + node * C + cg * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float y = m[e] + (f[e] * sc[e] + sh[e]);
      m[e] = y > 0.f ? y : y * p.slope;
      kp[e] = (uint8_t)ks[e];
    }
    *(u32x4*)((T*)p.out + node * p.out_cs + p.out_coff + cg * E) = Vec16<Tag>::pack(m);
  }
}

// backward marginals -> D = [dP - dQ | dQ]
template <typename Tag, int TPK>
__global__ __launch_bounds__(256) void edge_train_bwd_kernel(const EdgeTrainParams p) {
  constexpr int E = Tag::E, KPB = 256 / TPK, C = TPK * E;
  constexpr int RU = 6;      // reverse edges in flight: 222 VGPRs, still two waves per SIMD (8: 256+ VGPRs, one wave)
  using T = typename Tag::elem;
  extern __shared__ __attribute__((aligned(16))) int32_t s_idx3[];
  int b, sidx, g;
  if (!edge_block_setup<Tag, TPK>(p, b, sidx, g)) return;
  const int kp_l = threadIdx.x / TPK, cg = threadIdx.x % TPK;
  const u32x4* rows = (const u32x4*)p.pq + (size_t)b * p.N * (2 * TPK);
  float ca[E], cb[E], cr[E], mu[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = cg * E + e;
    ca[e] = p.coef[c]; cb[e] = p.coef[p.Cvec + c]; cr[e] = p.coef[2 * p.Cvec + c]; mu[e] = p.coef[3 * p.Cvec + c];
  }
  const int32_t* rp = p.rev_ptr + (size_t)g * (p.N + 1);
  const int32_t* re = p.rev_edge + (size_t)g * p.N * p.K;
  const float Kf = (float)p.K;
  const int kp_end = min((sidx + 1) * p.npb, p.N);
  for (int kp0 = sidx * p.npb; kp0 < kp_end; kp0 += KPB) {
    const int nkp = min(KPB, kp_end - kp0);
    __syncthreads();
    const int32_t* gidx = p.idx + ((size_t)g * p.N + kp0) * p.K;
    for (int t = threadIdx.x; t < nkp * p.K; t += 256) s_idx3[t] = gidx[t];
    __syncthreads();
    if (kp_l >= nkp) continue;
    const int j = kp0 + kp_l;
    const size_t node = (size_t)b * p.N + j;
    float f[E], S[E], pj[E], qj[E], go[E], ov[E], dq[E], dp[E];
    // ---- dQ_j: forward-graph gather-sum of P
#pragma unroll
    for (int e = 0; e < E; ++e) S[e] = 0.f;
    const int32_t* my = s_idx3 + kp_l * p.K;
    int k = 0;
    for (; k + 4 <= p.K; k += 4) {          // four rows in flight (one dependent L2 round trip per row made this loop a latency chain)
      u32x4 r4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r4[u] = rows[(size_t)my[k + u] * (2 * TPK) + cg];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        Vec16<Tag>::unpack(r4[u], f);
#pragma unroll
        for (int e = 0; e < E; ++e) S[e] += f[e];
      }
    }
    for (; k < p.K; ++k) {
      Vec16<Tag>::unpack(rows[(size_t)my[k] * (2 * TPK) + cg], f);
#pragma unroll
      for (int e = 0; e < E; ++e) S[e] += f[e];
    }
    Vec16<Tag>::unpack(rows[(size_t)j * (2 * TPK) + cg], pj);
    Vec16<Tag>::unpack(rows[(size_t)j * (2 * TPK) + TPK + cg], qj);
    Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.gout + node * p.g_cs + p.g_coff + cg * E), go);
    Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.out + node * p.out_cs + p.out_coff + cg * E), ov);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float gq = ov[e] > 0.f ? go[e] : go[e] * p.slope;
      dq[e] = ca[e] * (gq - Kf * cb[e] - cr[e] * (S[e] + Kf * qj[e] - Kf * mu[e]));
    }
    // ---- dP_j: reverse-graph gather (winners' gq, all Q_i)
    float G[E], R[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { G[e] = 0.f; R[e] = 0.f; }
    const int e0 = rp[j], e1 = rp[j + 1];
    for (int t0 = e0; t0 < e1; t0 += RU) {    // RU reverse edges in flight (edge id -> three rows + the arg-max bytes: two dependent trips each)
      int ei[RU], kk[RU];
      bool ok[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        ok[u] = t0 + u < e1;
        const int eid = re[ok[u] ? t0 + u : e1 - 1];
        ei[u] = eid / p.K;
        kk[u] = eid - ei[u] * p.K;
      }
      u32x4 rq[RU], rg[RU], ro[RU];
      uint8_t ks[RU][E];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const size_t ni = (size_t)b * p.N + ei[u];
        rq[u] = rows[(size_t)ei[u] * (2 * TPK) + TPK + cg];
        rg[u] = *(const u32x4*)((const T*)p.gout + ni * p.g_cs + p.g_coff + cg * E);
        ro[u] = *(const u32x4*)((const T*)p.out + ni * p.out_cs + p.out_coff + cg * E);
        const uint8_t* kp = p.kstar + ni * C + cg * E;
        if (E == 8) { const uint2 v = *(const uint2*)kp; memcpy(ks[u], &v, 8); }
        else { const uint32_t v = *(const uint32_t*)kp; memcpy(ks[u], &v, 4); }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (!ok[u]) continue;
        float gi[E], oi[E];
        Vec16<Tag>::unpack(rq[u], f);
        Vec16<Tag>::unpack(rg[u], gi);
        Vec16<Tag>::unpack(ro[u], oi);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          R[e] += f[e];
          if ((int)ks[u][e] == kk[u]) G[e] += oi[e] > 0.f ? gi[e] : gi[e] * p.slope;
        }
      }
    }
    const float deg = (float)(e1 - e0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      dp[e] = ca[e] * (G[e] - deg * cb[e] - cr[e] * (deg * pj[e] + R[e] - deg * mu[e]));
      dp[e] -= dq[e];
    }
    T* dst = (T*)p.dpq + node * (2 * C) + cg * E;
    *(u32x4*)dst = Vec16<Tag>::pack(dp);
    *(u32x4*)(dst + C) = Vec16<Tag>::pack(dq);
  }
}

// ------------------------------------------------------------------------------------------------ host
static void edge_plan(int B, int N, int KPB, int* S, int* npb) {
  const int chunks = (N + KPB - 1) / KPB;
  int s = 512 / (B > 0 ? B : 1);
  s = s < 1 ? 1 : (s > chunks ? chunks : s);
  const int cps = (chunks + s - 1) / s;          // chunks per sub-block
  *npb = cps * KPB;
  *S = (N + *npb - 1) / *npb;
}

extern "C" size_t cp_edge_train_workspace_bytes(int B, int C) {
  const size_t nb = (size_t)(B > 512 ? B : 512);
  const size_t Cvec = (size_t)(C + 15) / 16 * 16;
  return nb * 2 * Cvec * sizeof(double) + 4 * Cvec * sizeof(float);
}

static int edge_check(int dtype, const void* pq, const int32_t* idx, int B, int N, int K, int C, int G) {
  if (!pq || !idx || B <= 0 || N <= 0 || K <= 0 || K > 64 || C <= 0 || G <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (C % cp_chan_align(dtype) || !cp_aligned16(pq)) return CP_ERR_ALIGN;
  return CP_OK;
}

#define CP_EDGE_DISPATCH(FN)                                                                                        \
  if (dtype == CP_F32) {                                                                                             \
    switch (tpk) { case 8: FN(F32Tag, 8); break; case 16: FN(F32Tag, 16); break; case 32: FN(F32Tag, 32); break;    \
                   case 64: FN(F32Tag, 64); break; case 128: FN(F32Tag, 128); break; default: return CP_ERR_INVALID; } \
  } else {                                                                                                           \
    switch (tpk) { case 4: FN(BF16Tag, 4); break; case 8: FN(BF16Tag, 8); break; case 16: FN(BF16Tag, 16); break;   \
                   case 32: FN(BF16Tag, 32); break; case 64: FN(BF16Tag, 64); break; default: return CP_ERR_INVALID; } \
  }

extern "C" int cp_edgeconv_train_fwd(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx, const int32_t* graph_ids,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, void* out, int out_cstride, int out_coff, uint8_t* kstar,
                                     float* scale, float* shift, float* mean, float* rstd, void* workspace, int B, int N, int K,
                                     int C, int G, float slope) {
  int rc = edge_check(dtype, pq, idx, B, N, K, C, G);
  if (rc) return rc;
  if (!out || !kstar || !scale || !shift || !mean || !rstd || !workspace) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (out_cstride % E || out_coff % E || out_coff + C > out_cstride || !cp_aligned16(out)) return CP_ERR_ALIGN;
  if (C % 16) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int tpk = C / E, KPB = 256 / tpk;
  EdgeTrainParams p = {};
  p.pq = pq; p.idx = idx; p.gids = graph_ids; p.out = out; p.out_cs = out_cstride; p.out_coff = out_coff; p.kstar = kstar;
  p.scale = scale; p.shift = shift; p.partial = (double*)workspace; p.B = B; p.N = N; p.K = K; p.slope = slope;
  edge_plan(B, N, KPB, &p.S, &p.npb);
  const int grid = 8 * ((B + 7) / 8) * p.S;
  const size_t lds_r = (size_t)256 * 2 * E * sizeof(double) + (size_t)KPB * K * sizeof(int32_t);
#define FN_STATS(TAG, T) do { cp_mark_kernel("edge_reduce_kernel<%s, %d, 0>", #TAG, T); \
    hipLaunchKernelGGL((edge_reduce_kernel<TAG, T, 0>), dim3(grid), dim3(256), lds_r, st, p); } while (0)
  CP_EDGE_DISPATCH(FN_STATS)
#undef FN_STATS
  if ((rc = cp_check_launch())) return rc;
  if ((rc = cp_bn_finalize_launch(st, p.partial, B * p.S, C, C, (double)B * N * K, gamma, beta, eps, momentum, running_mean,
                                  running_var, scale, shift, mean, rstd)))
    return rc;
#define FN_FWD(TAG, T) do { cp_mark_kernel("edge_train_fwd_kernel<%s, %d>", #TAG, T); \
    hipLaunchKernelGGL((edge_train_fwd_kernel<TAG, T>), dim3(grid), dim3(256), (size_t)KPB * K * sizeof(int32_t), st, p); } while (0)
  CP_EDGE_DISPATCH(FN_FWD)
#undef FN_FWD
  return cp_check_launch();
}

extern "C" int cp_edgeconv_train_bwd(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx, const int32_t* rev_ptr,
                                     const int32_t* rev_edge, const int32_t* graph_ids, const void* out, int out_cstride,
                                     int out_coff, const uint8_t* kstar, const void* gout, int gout_cstride, int gout_coff,
                                     const float* gamma, const float* mean, const float* rstd, void* dpq, float* dgamma,
                                     float* dbeta, void* workspace, int B, int N, int K, int C, int G, float slope) {
  int rc = edge_check(dtype, pq, idx, B, N, K, C, G);
  if (rc) return rc;
  if (!rev_ptr || !rev_edge || !out || !kstar || !gout || !mean || !rstd || !dpq || !workspace) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (out_cstride % E || out_coff % E || gout_cstride % E || gout_coff % E || C % 16) return CP_ERR_ALIGN;
  if (!cp_aligned16(out) || !cp_aligned16(gout) || !cp_aligned16(dpq)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int tpk = C / E, KPB = 256 / tpk;
  const size_t nb = (size_t)(B > 512 ? B : 512);
  float* coef = (float*)((char*)workspace + nb * 2 * C * sizeof(double));
  EdgeTrainParams p = {};
  p.pq = pq; p.idx = idx; p.gids = graph_ids; p.rev_ptr = rev_ptr; p.rev_edge = rev_edge;
  p.out = const_cast<void*>(out); p.out_cs = out_cstride; p.out_coff = out_coff; p.kstar = const_cast<uint8_t*>(kstar);
  p.gout = gout; p.g_cs = gout_cstride; p.g_coff = gout_coff; p.dpq = dpq; p.partial = (double*)workspace;
  p.coef = coef; p.Cvec = C; p.scale = rstd;      // the reduce kernel reads rstd through `scale`, mu through coef[3]
  p.B = B; p.N = N; p.K = K; p.slope = slope;
  edge_plan(B, N, KPB, &p.S, &p.npb);
  const int grid = 8 * ((B + 7) / 8) * p.S;
  // coef[3] (mu) must be in place before the reduce kernel runs: copy mean -> coef[3*C] first
  if ((rc = cp_memcpy_d2d(stream, coef + 3 * (size_t)C, mean, (size_t)C * sizeof(float)))) return rc;
  const size_t lds_r = (size_t)256 * 2 * E * sizeof(double) + (size_t)KPB * K * sizeof(int32_t);
#define FN_RED(TAG, T) do { cp_mark_kernel("edge_reduce_kernel<%s, %d, 1>", #TAG, T); \
    hipLaunchKernelGGL((edge_reduce_kernel<TAG, T, 1>), dim3(grid), dim3(256), lds_r, st, p); } while (0)
  CP_EDGE_DISPATCH(FN_RED)
#undef FN_RED
  if ((rc = cp_check_launch())) return rc;
  if ((rc = cp_bn_bwd_finalize_launch(st, p.partial, B * p.S, C, C, (double)B * N * K, gamma, mean, rstd, coef, dgamma, dbeta)))
    return rc;
#define FN_BWD(TAG, T) do { cp_mark_kernel("edge_train_bwd_kernel<%s, %d>", #TAG, T); \
    hipLaunchKernelGGL((edge_train_bwd_kernel<TAG, T>), dim3(grid), dim3(256), (size_t)KPB * K * sizeof(int32_t), st, p); } while (0)
  CP_EDGE_DISPATCH(FN_BWD)
#undef FN_BWD
  return cp_check_launch();
}
