// Training-side ops: backward of the fused graph ops and the loss head.
//   * cp_edgeconv_gather_max_bwd : backward of cp_edgeconv_gather_max (gradient goes to the arg-max neighbour)
//   * cp_index2feat_gather_bwd   : backward of cp_index2feat_gather (scatter-add into the patch map)
//   * cp_code_loss               : UnmaskedCodeLoss / MaskedCodeLoss (losses/code_loss.py:6-62), value + d/dlogits
//   * cp_mask_loss               : MaskLoss_interpolate (losses/mask_loss.py:6-17), value + d/dlogits
// Gradients are fp32 whatever the forward's storage type.  All of this is HBM-bound streaming work: one pass over
// the saved forward tensors, 16 bytes per lane, no float atomics except the Index2Feat scatter (data-dependent
// collisions).  Reductions are two-level with a fixed order (bit-reproducible run to run).
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------
// EdgeConv backward.  Forward (graph_ops.hip): out[b,i,c] = leaky( max_k P'[b, idx[g,i,k], c] + Q'[b,i,c] ).
//   pass A (thread = one 16-byte channel group of one keypoint, same mapping as the forward): recompute the K-way
//          max keeping the FIRST arg-max k* (torch.max's index on the CPU), gq = gout * leaky'(y);
//          dQ'[b,i,c] = gq, kstar[b,i,c] = k*.
//   pass B (thread = 4 channels of one node j): dP'[b,j,c] = sum over the reverse edges (i,k) of j, in edge order, of
//          (kstar[b,i,c] == k ? dQ'[b,i,c] : 0)  -- a gather over the static reverse graph instead of float atomics:
//          deterministic, and the per-crop rows stay in the XCD's L2 (same blockIdx % 8 labelling as the forward).
template <typename Tag, int TPK>
__global__ __launch_bounds__(256) void edgeconv_bwd_a_kernel(
    const void* __restrict__ pq, const int32_t* __restrict__ idx, const int32_t* __restrict__ graph_ids,
    const float* __restrict__ gout, float* __restrict__ dpq, uint8_t* __restrict__ kstar, int B, int N, int K, int chunks,
    int g_cs, int g_coff, float slope) {
  constexpr int E = Tag::E;
  constexpr int KPB = 256 / TPK;
  constexpr int C = TPK * E;
  extern __shared__ __attribute__((aligned(16))) int32_t s_idx[];   // KPB * K
  const int label = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int b = label + 8 * (jb / chunks);
  const int chunk = jb % chunks;
  if (b >= B) return;                        // whole block exits together (before any barrier)
  const int g = graph_ids ? graph_ids[b] : 0;
  const int kp0 = chunk * KPB;
  const int nkp = min(KPB, N - kp0);
  const int32_t* gidx = idx + ((size_t)g * N + kp0) * K;
  for (int t = threadIdx.x; t < nkp * K; t += 256) s_idx[t] = gidx[t];
  __syncthreads();

  const int kp_l = threadIdx.x / TPK, cg = threadIdx.x % TPK;
  if (kp_l >= nkp) return;
  const int i = kp0 + kp_l;
  const u32x4* rows = (const u32x4*)pq + (size_t)b * N * (2 * TPK);
  float m[E], f[E];
  int ks[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { m[e] = -INFINITY; ks[e] = 0; }
  const int32_t* my = s_idx + kp_l * K;
  for (int k = 0; k < K; ++k) {
    Vec16<Tag>::unpack(rows[(size_t)my[k] * (2 * TPK) + cg], f);
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (f[e] > m[e]) { m[e] = f[e]; ks[e] = k; }
  }
  Vec16<Tag>::unpack(rows[(size_t)i * (2 * TPK) + TPK + cg], f);          // Q'_i
  const size_t node = (size_t)b * N + i;
  const float* gp = gout + node * g_cs + g_coff + cg * E;
  float* dq = dpq + node * (2 * C) + C + cg * E;
  uint32_t kpack[E / 4];
#pragma unroll
  for (int v = 0; v < E / 4; ++v) {
    const f32x4 gv = *(const f32x4*)(gp + 4 * v);
    f32x4 o;
    kpack[v] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = 4 * v + j;
      const float y = m[e] + f[e];
      o[j] = y > 0.f ? gv[j] : gv[j] * slope;
      kpack[v] |= (uint32_t)ks[e] << (8 * j);
    }
    *(f32x4*)(dq + 4 * v) = o;
  }
  uint32_t* kp = (uint32_t*)(kstar + node * C + cg * E);
#pragma unroll
  for (int v = 0; v < E / 4; ++v) kp[v] = kpack[v];
}

__global__ __launch_bounds__(256) void edgeconv_bwd_b_kernel(
    const int32_t* __restrict__ rev_ptr, const int32_t* __restrict__ rev_edge, const int32_t* __restrict__ graph_ids,
    float* __restrict__ dpq, const uint8_t* __restrict__ kstar, int B, int N, int K, int C, int chunks) {
  const int tpn = C / 4;                     // threads per node
  const int npb = 256 / tpn;                 // nodes per block (C <= 1024 -> tpn <= 256)
  const int label = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int b = label + 8 * (jb / chunks);
  const int chunk = jb % chunks;
  if (b >= B) return;
  const int g = graph_ids ? graph_ids[b] : 0;
  const int nl = threadIdx.x / tpn, cg = threadIdx.x % tpn;
  const int j = chunk * npb + nl;
  if (nl >= npb || j >= N) return;
  const int32_t* rp = rev_ptr + (size_t)g * (N + 1);
  const int32_t* re = rev_edge + (size_t)g * N * K;
  const int e0 = rp[j], e1 = rp[j + 1];
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = e0; t < e1; ++t) {
    const int eid = re[t];
    const int i = eid / K, k = eid - i * K;
    const size_t node = (size_t)b * N + i;
    const uint32_t kk = *(const uint32_t*)(kstar + node * C + cg * 4);
    const f32x4 gq = *(const f32x4*)(dpq + node * (2 * C) + C + cg * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if ((int)((kk >> (8 * c)) & 255u) == k) acc[c] += gq[c];
  }
  *(f32x4*)(dpq + ((size_t)b * N + j) * (2 * C) + cg * 4) = acc;
}

extern "C" size_t cp_edgeconv_bwd_workspace_bytes(int B, int N, int C) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  return ((size_t)B * N * C + 15) / 16 * 16;
}

template <typename Tag, int TPK>
static int launch_edge_bwd_a(hipStream_t st, const void* pq, const int32_t* idx, const int32_t* gids, const float* gout,
                             float* dpq, uint8_t* kstar, int B, int N, int K, int g_cs, int g_coff, float slope) {
  constexpr int KPB = 256 / TPK;
  const int chunks = (N + KPB - 1) / KPB;
  const int grid = 8 * ((B + 7) / 8) * chunks;
  cp_mark_kernel("edgeconv_bwd_a_kernel<%s, %d>", Tag::dtype == CP_BF16 ? "BF16Tag" : "F32Tag", TPK);
  hipLaunchKernelGGL((edgeconv_bwd_a_kernel<Tag, TPK>), dim3(grid), dim3(256), KPB * K * sizeof(int32_t), st, pq, idx, gids,
                     gout, dpq, kstar, B, N, K, chunks, g_cs, g_coff, slope);
  return cp_check_launch();
}

extern "C" int cp_edgeconv_gather_max_bwd(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx,
                                          const int32_t* rev_ptr, const int32_t* rev_edge, const int32_t* graph_ids,
                                          const float* gout, float* dpq, void* workspace, int B, int N, int K, int C,
                                          int G, int gout_cstride, int gout_coff, float slope) {
  if (!pq || !idx || !rev_ptr || !rev_edge || !gout || !dpq || !workspace) return CP_ERR_INVALID;
  if (B <= 0 || N <= 0 || K <= 0 || K > 64 || C <= 0 || G <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || C > 1024 || (256 % (C / 4)) || gout_cstride % 4 || gout_coff % 4 || gout_coff + C > gout_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(pq) || !cp_aligned16(gout) || !cp_aligned16(dpq) || !cp_aligned16(workspace)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  uint8_t* kstar = (uint8_t*)workspace;
  const int tpk = C / E;
  int rc = CP_ERR_INVALID;
#define CP_EDGE(TAG, T) case T: rc = launch_edge_bwd_a<TAG, T>(st, pq, idx, graph_ids, gout, dpq, kstar, B, N, K, gout_cstride, gout_coff, slope); break;
  if (dtype == CP_F32) {
    switch (tpk) { CP_EDGE(F32Tag, 8) CP_EDGE(F32Tag, 16) CP_EDGE(F32Tag, 32) CP_EDGE(F32Tag, 64) CP_EDGE(F32Tag, 128) default: return CP_ERR_INVALID; }
  } else {
    switch (tpk) { CP_EDGE(BF16Tag, 4) CP_EDGE(BF16Tag, 8) CP_EDGE(BF16Tag, 16) CP_EDGE(BF16Tag, 32) CP_EDGE(BF16Tag, 64) default: return CP_ERR_INVALID; }
  }
#undef CP_EDGE
  if (rc != CP_OK) return rc;
  const int npb = 256 / (C / 4);
  const int chunks = (N + npb - 1) / npb;
  CP_LAUNCH(edgeconv_bwd_b_kernel, dim3(8 * ((B + 7) / 8) * chunks), dim3(256), 0, st, rev_ptr, rev_edge, graph_ids, dpq,
                     kstar, B, N, K, C, chunks);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Index2Feat backward: dpatches[b, ty, tx, e] += gout[b, i, coff + tap*E + e] * mask[b,i]   (taps as the forward).
// Several keypoints may share a sub-pixel (ids are predictions), so this is a scatter with collisions: hardware
// fp32 atomics (global_atomic_add_f32), skipped entirely for keypoints outside the RoI (mask == 0).
template <typename Tag>
__global__ void index2feat_bwd_kernel(const void* __restrict__ gout, const int32_t* __restrict__ x_id,
                                      const int32_t* __restrict__ y_id, const float* __restrict__ mask,
                                      float* __restrict__ dpatches, int N, int Hp, int Wp, int EG4, int k, int g_cs, int g_coff,
                                      size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*N*4*EG4
  if (i >= total) return;
  const int g = (int)(i % EG4);
  size_t t = i / EG4;
  const int tap = (int)(t & 3);
  const size_t kp = t >> 2;
  const float mk = mask[kp];
  if (mk == 0.f) return;
  const size_t b = kp / N;
  const int y = 2 * y_id[kp] + ((tap & 1) ? k : 0);
  const int x = 2 * x_id[kp] + ((tap & 2) ? k : 0);
  if ((unsigned)y >= (unsigned)Hp || (unsigned)x >= (unsigned)Wp) return;
  const size_t ge = kp * g_cs + g_coff + (size_t)(tap * EG4 + g) * 4;
  f32x4 gv;
  if (Tag::E == 4) gv = *(const f32x4*)((const float*)gout + ge);
  else {
    const u32x2 r2 = *(const u32x2*)((const uint16_t*)gout + ge);
    gv = f32x4{__uint_as_float(r2.x << 16), __uint_as_float(r2.x & 0xffff0000u), __uint_as_float(r2.y << 16),
               __uint_as_float(r2.y & 0xffff0000u)};
  }
  float* dst = dpatches + (((b * Hp + y) * Wp + x) * EG4 + g) * 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) unsafeAtomicAdd(dst + c, gv[c] * mk);
}

// Deterministic form (cp_set_deterministic): a GATHER -- one thread per (crop, patch pixel, channel quad) walks the crop's 4 N
// (keypoint, tap) slots in slot order (their destination pixels staged in LDS, 2048 slots at a time) and adds the ones that land on
// its pixel: every sum has one fixed order, every pixel is written exactly once (no memset, no atomics).  ~0.2 ms at B = 32.
constexpr int I2F_DET_CHUNK = 2048;
template <typename Tag>
__global__ __launch_bounds__(256) void index2feat_bwd_det_kernel(const void* __restrict__ gout, const int32_t* __restrict__ x_id,
                                                                 const int32_t* __restrict__ y_id, const float* __restrict__ mask,
                                                                 float* __restrict__ dpatches, int N, int Hp, int Wp, int EG4, int k,
                                                                 int g_cs, int g_coff) {
  __shared__ int dest[I2F_DET_CHUNK];
  __shared__ float mks[I2F_DET_CHUNK / 4];
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;           // over Hp * Wp * EG4
  const int npix = Hp * Wp;
  const bool live = idx < npix * EG4;
  const int pix = idx / EG4, g = idx - pix * EG4;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < 4 * N; s0 += I2F_DET_CHUNK) {
    __syncthreads();
    for (int s = threadIdx.x; s < I2F_DET_CHUNK; s += 256) {
      const int slot = s0 + s;
      int d = -1;
      if (slot < 4 * N) {
        const size_t kp = (size_t)b * N + (slot >> 2);
        const int tap = slot & 3;
        const float mk = mask[kp];
        if (tap == 0) mks[s >> 2] = mk;
        const int y = 2 * y_id[kp] + ((tap & 1) ? k : 0);
        const int x = 2 * x_id[kp] + ((tap & 2) ? k : 0);
        if (mk != 0.f && (unsigned)y < (unsigned)Hp && (unsigned)x < (unsigned)Wp) d = y * Wp + x;
      }
      dest[s] = d;
    }
    __syncthreads();
    if (!live) continue;
    const int n_here = min(I2F_DET_CHUNK, 4 * N - s0);
    for (int s = 0; s < n_here; ++s) {
      if (dest[s] != pix) continue;
      const size_t kp = (size_t)b * N + ((s0 + s) >> 2);
      const int tap = s & 3;
      const size_t ge = kp * g_cs + g_coff + (size_t)(tap * EG4 + g) * 4;
      f32x4 gv;
      if (Tag::E == 4) gv = *(const f32x4*)((const float*)gout + ge);
      else {
        const u32x2 r2 = *(const u32x2*)((const uint16_t*)gout + ge);
        gv = f32x4{__uint_as_float(r2.x << 16), __uint_as_float(r2.x & 0xffff0000u), __uint_as_float(r2.y << 16),
                   __uint_as_float(r2.y & 0xffff0000u)};
      }
      const float mk = mks[s >> 2];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += gv[c] * mk;
    }
  }
  if (live) *(f32x4*)(dpatches + (((size_t)b * npix + pix) * EG4 + g) * 4) = acc;
}

// gout in `dtype` (the training program keeps activation gradients in the storage type); dpatches stays fp32
extern "C" int cp_index2feat_gather_bwd_t(cp_stream_t stream, int dtype, const void* gout, const int32_t* x_id,
                                          const int32_t* y_id, const float* mask, float* dpatches, int B, int N, int Hp, int Wp,
                                          int E_ch, int k, int gout_cstride, int gout_coff) {
  if (!gout || !x_id || !y_id || !mask || !dpatches || B <= 0 || N <= 0 || Hp <= 0 || Wp <= 0 || E_ch <= 0 || k <= 0)
    return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (E_ch % 4 || gout_cstride % 4 || gout_coff % 4 || gout_coff + 4 * E_ch > gout_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(gout) || !cp_aligned16(dpatches)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int EG4 = E_ch / 4;
  if (cp_deterministic()) {
    if (B > 65535) return CP_ERR_RANGE;
    const dim3 grid((unsigned)(((size_t)Hp * Wp * EG4 + 255) / 256), (unsigned)B);
    if (dtype == CP_F32)
      CP_LAUNCH(index2feat_bwd_det_kernel<F32Tag>, grid, dim3(256), 0, st, gout, x_id, y_id, mask, dpatches, N, Hp, Wp, EG4, k, gout_cstride, gout_coff);
    else
      CP_LAUNCH(index2feat_bwd_det_kernel<BF16Tag>, grid, dim3(256), 0, st, gout, x_id, y_id, mask, dpatches, N, Hp, Wp, EG4, k, gout_cstride, gout_coff);
    return cp_check_launch();
  }
  if (cp_memset_zero(stream, dpatches, (size_t)B * Hp * Wp * E_ch * sizeof(float)) != CP_OK) return CP_ERR_HIP;
  const size_t total = (size_t)B * N * 4 * EG4;
  if (dtype == CP_F32)
    CP_LAUNCH(index2feat_bwd_kernel<F32Tag>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gout, x_id, y_id, mask,
              dpatches, N, Hp, Wp, EG4, k, gout_cstride, gout_coff, total);
  else
    CP_LAUNCH(index2feat_bwd_kernel<BF16Tag>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gout, x_id, y_id, mask,
              dpatches, N, Hp, Wp, EG4, k, gout_cstride, gout_coff, total);
  return cp_check_launch();
}

extern "C" int cp_index2feat_gather_bwd(cp_stream_t stream, const float* gout, const int32_t* x_id, const int32_t* y_id,
                                        const float* mask, float* dpatches, int B, int N, int Hp, int Wp, int E_ch, int k,
                                        int gout_cstride, int gout_coff) {
  return cp_index2feat_gather_bwd_t(stream, CP_F32, gout, x_id, y_id, mask, dpatches, B, N, Hp, Wp, E_ch, k, gout_cstride,
                                    gout_coff);
}

// ------------------------------------------------------------------------------------------------------------------
// Loss head.  One reduction kernel (<= LOSS_BLOCKS blocks, per-block partial in double; the last block to finish --
// a ticket counter in the workspace -- sums the partials in block order and writes the loss and the denominator),
// then one elementwise kernel for d loss / d logits.
constexpr int LOSS_BLOCKS = 256;
struct LossWs { double part[LOSS_BLOCKS][2]; float denom; uint32_t ticket; };

extern "C" size_t cp_loss_workspace_bytes(void) { return (sizeof(LossWs) + 15) / 16 * 16; }

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }
// nn.BCEWithLogitsLoss element: (1 - y) z + softplus(-z) = max(z, 0) - z y + log1p(exp(-|z|))
__device__ __forceinline__ float bce_logits(float z, float y) { return fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoid_acc(float z) { return 1.f / (1.f + expf(-z)); }

struct CodeLossArgs {
  const float* pred; const float* gt; const float* mask; float* dpred;
  long long pred_bs, gt_bs, dpred_bs;
  int B, nb, N, type, masked;
};
struct MaskLossArgs {
  const float* pred; const float* gt; float* dpred;
  long long pred_bs, dpred_bs;
  int B, h, w, Hm, Wm;
};

__device__ __forceinline__ void loss_block_reduce(double s0, double s1, LossWs* ws, float* loss, int mode_masked, double count,
                                                  int nb) {
  __shared__ double sh[2][4];
  __shared__ bool last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o); s1 += __shfl_down(s1, o); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = s0; sh[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws->part[blockIdx.x][0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    ws->part[blockIdx.x][1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    __threadfence();
    last = atomicAdd(&ws->ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    double t0 = 0.0, t1 = 0.0;
    for (unsigned i = 0; i < gridDim.x; ++i) { t0 += ((volatile double*)ws->part[i])[0]; t1 += ((volatile double*)ws->part[i])[1]; }
    // MaskedCodeLoss: sum / (clamp(mask.sum(), 1) * #bits)  (code_loss.py:59-61); otherwise reduction="mean"
    const double denom = mode_masked ? fmax(t1, 1.0) * nb : count;
    *loss = (float)(t0 / denom);
    ws->denom = (float)denom;
    ws->ticket = 0;
  }
}

__global__ __launch_bounds__(256) void code_loss_kernel(const CodeLossArgs a, LossWs* ws, float* loss) {
  const long long per = (long long)a.nb * a.N, total = per * a.B;
  double s0 = 0.0, s1 = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / per, r = e - b * per;
    const float z = a.pred[b * a.pred_bs + r], y = a.gt[b * a.gt_bs + r];
    float raw = a.type == 0 ? bce_logits(z, y) : fabsf(sigmoid_acc(z) - y);
    if (a.masked) {
      const int n = (int)(r % a.N);
      const float m = a.mask[b * a.N + n];
      raw *= m;
      if (r < a.N) s1 += m;
    }
    s0 += raw;
  }
  loss_block_reduce(s0, s1, ws, loss, a.masked, (double)total, a.nb);
}

__global__ __launch_bounds__(256) void code_loss_grad_kernel(const CodeLossArgs a, const LossWs* ws) {
  const long long per = (long long)a.nb * a.N, total = per * a.B;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const float inv = 1.f / ws->denom;
  const long long b = e / per, r = e - b * per;
  const float z = a.pred[b * a.pred_bs + r], y = a.gt[b * a.gt_bs + r];
  const float s = sigmoid_acc(z);
  float d;
  if (a.type == 0) d = s - y;
  else { const float df = s - y; d = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * s * (1.f - s); }
  if (a.masked) d *= a.mask[b * a.N + (int)(r % a.N)];
  a.dpred[b * a.dpred_bs + r] = d * inv;
}

extern "C" int cp_code_loss(cp_stream_t stream, int loss_type, const float* pred, long long pred_bstride, const float* gt,
                            long long gt_bstride, const float* mask, int B, int nbits, int N, float* loss, float* dpred,
                            long long dpred_bstride, void* workspace) {
  if (!pred || !gt || !loss || !workspace || B <= 0 || nbits <= 0 || N <= 0) return CP_ERR_INVALID;
  if (loss_type != CP_LOSS_BCE && loss_type != CP_LOSS_L1) return CP_ERR_INVALID;
  const long long per = (long long)nbits * N;
  if (pred_bstride < per || gt_bstride < per || (dpred && dpred_bstride < per)) return CP_ERR_INVALID;
  if (!cp_aligned16(workspace)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  LossWs* ws = (LossWs*)workspace;
  if (hipMemsetAsync(&ws->ticket, 0, sizeof(uint32_t), st) != hipSuccess) return CP_ERR_HIP;
  CodeLossArgs a{pred, gt, mask, dpred, pred_bstride, gt_bstride, dpred_bstride, B, nbits, N, loss_type, mask ? 1 : 0};
  const long long total = per * B;
  const int blocks = (int)((total + 255) / 256 < LOSS_BLOCKS ? (total + 255) / 256 : LOSS_BLOCKS);
  CP_LAUNCH(code_loss_kernel, dim3(blocks), dim3(256), 0, st, a, ws, loss);
  if (dpred) CP_LAUNCH(code_loss_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, ws);
  return cp_check_launch();
}

// MaskedCodeLoss(loss_type="CE") (losses/code_loss.py:36-37,47-61): nn.CrossEntropyLoss(reduction="none") over the class axis
// of pred (B, C, N) against class ids gt (B, N) (ids handed over as fp32: exact up to 2^24), times the mask, summed and divided by
// clamp(mask.sum(), 1) (num_bits = 1 for CE).  One thread per (b, n): max-shifted log-sum-exp in fp32, value summed in fp64.
struct CeLossArgs {
  const float* pred; const float* gt; const float* mask; float* dpred;
  long long pred_bs, dpred_bs;
  int B, C, N;
};

__global__ __launch_bounds__(256) void ce_loss_kernel(const CeLossArgs a, LossWs* ws, float* loss) {
  const long long total = (long long)a.B * a.N;
  double s0 = 0.0, s1 = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / a.N;
    const int n = (int)(e - b * a.N);
    const float* p = a.pred + b * a.pred_bs + n;
    float mx = p[0];
    for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, p[(long long)c * a.N]);
    float se = 0.f;
    for (int c = 0; c < a.C; ++c) se += expf(p[(long long)c * a.N] - mx);
    const int y = (int)a.gt[e];
    const float m = a.mask[e];
    // nn.CrossEntropyLoss: ignore_index (-100) contributes 0 loss; any other id outside [0, C) raises there -- a kernel cannot, so
    // every out-of-range id is treated as ignored (never read: p[y * N] would be out of bounds even under mask = 0)
    if (y >= 0 && y < a.C) s0 += (double)((mx + logf(se) - p[(long long)y * a.N]) * m);
    s1 += m;
  }
  loss_block_reduce(s0, s1, ws, loss, 1, (double)total, 1);
}

__global__ __launch_bounds__(256) void ce_loss_grad_kernel(const CeLossArgs a, const LossWs* ws) {
  const long long total = (long long)a.B * a.N;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long long b = e / a.N;
  const int n = (int)(e - b * a.N);
  const float* p = a.pred + b * a.pred_bs + n;
  float* d = a.dpred + b * a.dpred_bs + n;
  float mx = p[0];
  for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, p[(long long)c * a.N]);
  float se = 0.f;
  for (int c = 0; c < a.C; ++c) se += expf(p[(long long)c * a.N] - mx);
  const int y = (int)a.gt[e];
  const float w = (y >= 0 && y < a.C) ? a.mask[e] / ws->denom : 0.f, inv = 1.f / se;      // ignored / out-of-range id: zero gradient
  for (int c = 0; c < a.C; ++c) d[(long long)c * a.N] = (expf(p[(long long)c * a.N] - mx) * inv - (c == y ? 1.f : 0.f)) * w;
}

extern "C" int cp_masked_ce_loss(cp_stream_t stream, const float* pred, long long pred_bstride, const float* gt_class,
                                 const float* mask, int B, int C, int N, float* loss, float* dpred, long long dpred_bstride,
                                 void* workspace) {
  if (!pred || !gt_class || !mask || !loss || !workspace || B <= 0 || C <= 0 || N <= 0) return CP_ERR_INVALID;
  const long long per = (long long)C * N;
  if (pred_bstride < per || (dpred && dpred_bstride < per)) return CP_ERR_INVALID;
  if (!cp_aligned16(workspace)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  LossWs* ws = (LossWs*)workspace;
  if (hipMemsetAsync(&ws->ticket, 0, sizeof(uint32_t), st) != hipSuccess) return CP_ERR_HIP;
  CeLossArgs a{pred, gt_class, mask, dpred, pred_bstride, dpred_bstride, B, C, N};
  const long long total = (long long)B * N;
  const int blocks = (int)((total + 255) / 256 < LOSS_BLOCKS ? (total + 255) / 256 : LOSS_BLOCKS);
  CP_LAUNCH(ce_loss_kernel, dim3(blocks), dim3(256), 0, st, a, ws, loss);
  if (dpred) CP_LAUNCH(ce_loss_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, ws);
  return cp_check_launch();
}

// F.interpolate(mode="nearest") source index (mask_loss.py:14): min(floor(dst * in/out), in - 1), scale in fp32
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  const float scale = (float)in / (float)out;
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ __launch_bounds__(256) void mask_loss_kernel(const MaskLossArgs a, LossWs* ws, float* loss) {
  const long long per = (long long)a.h * a.w, total = per * a.B;
  double s0 = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / per, r = e - b * per;
    const int y = (int)(r / a.w), x = (int)(r - (long long)y * a.w);
    const float z = a.pred[b * a.pred_bs + r];
    const float g = a.gt[((long long)b * a.Hm + nearest_src(y, a.Hm, a.h)) * a.Wm + nearest_src(x, a.Wm, a.w)];
    s0 += fabsf(sigmoid_acc(z) - g);
  }
  loss_block_reduce(s0, 0.0, ws, loss, 0, (double)total, 1);
}

__global__ __launch_bounds__(256) void mask_loss_grad_kernel(const MaskLossArgs a, const LossWs* ws) {
  const long long per = (long long)a.h * a.w, total = per * a.B;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long long b = e / per, r = e - b * per;
  const int y = (int)(r / a.w), x = (int)(r - (long long)y * a.w);
  const float s = sigmoid_acc(a.pred[b * a.pred_bs + r]);
  const float g = a.gt[((long long)b * a.Hm + nearest_src(y, a.Hm, a.h)) * a.Wm + nearest_src(x, a.Wm, a.w)];
  const float df = s - g;
  a.dpred[b * a.dpred_bs + r] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * s * (1.f - s) / ws->denom;
}

extern "C" int cp_mask_loss(cp_stream_t stream, const float* pred, long long pred_bstride, const float* gt, int B, int h,
                            int w, int Hm, int Wm, float* loss, float* dpred, long long dpred_bstride, void* workspace) {
  if (!pred || !gt || !loss || !workspace || B <= 0 || h <= 0 || w <= 0 || Hm <= 0 || Wm <= 0) return CP_ERR_INVALID;
  const long long per = (long long)h * w;
  if (pred_bstride < per || (dpred && dpred_bstride < per)) return CP_ERR_INVALID;
  if (!cp_aligned16(workspace)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  LossWs* ws = (LossWs*)workspace;
  if (hipMemsetAsync(&ws->ticket, 0, sizeof(uint32_t), st) != hipSuccess) return CP_ERR_HIP;
  MaskLossArgs a{pred, gt, dpred, pred_bstride, dpred_bstride, B, h, w, Hm, Wm};
  const long long total = per * B;
  const int blocks = (int)((total + 255) / 256 < LOSS_BLOCKS ? (total + 255) / 256 : LOSS_BLOCKS);
  CP_LAUNCH(mask_loss_kernel, dim3(blocks), dim3(256), 0, st, a, ws, loss);
  if (dpred) CP_LAUNCH(mask_loss_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, ws);
  return cp_check_launch();
}
