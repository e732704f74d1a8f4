// Training-side helpers of the dense layers: train-mode BatchNorm2d (batch statistics, running
// stat update, backward), bias/activation backward, data-gradient weight transform, and the backward of the two
// resampling ops (bilinear x2 align_corners, HRNet nearest-upsample fuse sum).  All memory-bound: one thread per
// 16-byte channel group, per-channel sums are reduced in fp64 (block partials -> one finalize launch; no atomics, so
// the statistics are run-to-run deterministic).
//
// Replaces (reference /root/reference/checkerpose): nn.BatchNorm2d in .train() mode inside timm's hrnet
// (model/backbone.py:48), model/init.py:60 + model/pipeline.py:51 (EdgeConv BN), pipeline.py:189,194,203,208 (decoder
// BN) and their autograd; nn.UpsamplingBilinear2d backward (pipeline.py:199).
#include <stdlib.h>

#include "common.h"
#include <cstring>

// ------------------------------------------------------------------------------------------------ small plumbing
// zero fill / device copy as plain kernels: they sit inside the captured training graphs between kernel nodes (memset /
// memcpy graph nodes replayed wrongly here: measured NaNs on the second replay), 16 bytes per thread + byte tail.
__global__ void fill_zero_kernel(unsigned char* __restrict__ p, size_t nvec, size_t nbytes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nvec) ((u32x4*)p)[i] = u32x4{0u, 0u, 0u, 0u};
  if (i == 0) for (size_t b = nvec * 16; b < nbytes; ++b) p[b] = 0;
}
__global__ void copy_bytes_kernel(unsigned char* __restrict__ d, const unsigned char* __restrict__ s, size_t nvec, size_t nbytes, int vec) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    if (i < nvec) ((u32x4*)d)[i] = ((const u32x4*)s)[i];
    if (i == 0) for (size_t b = nvec * 16; b < nbytes; ++b) d[b] = s[b];
  } else if (i < nbytes) d[i] = s[i];
}

extern "C" int cp_memset_zero(cp_stream_t stream, void* p, size_t nbytes) {
  if (!p || ((uintptr_t)p & 15)) return p ? CP_ERR_ALIGN : CP_ERR_INVALID;
  if (nbytes == 0) return CP_OK;
  const size_t nvec = nbytes / 16;
  const size_t n = nvec > 0 ? nvec : 1;
  CP_LAUNCH(fill_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, nvec, nbytes);
  return cp_check_launch();
}

__global__ void weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int R, int S,
                                    size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over wt (Cin, Cout, R, S)
  if (i >= total) return;
  const int s = (int)(i % S);
  size_t t = i / S;
  const int r = (int)(t % R); t /= R;
  const int co = (int)(t % Cout);
  const int ci = (int)(t / Cout);
  wt[i] = w[(((size_t)co * Cin + ci) * R + (R - 1 - r)) * S + (S - 1 - s)];
}

extern "C" int cp_weight_dgrad(cp_stream_t stream, const float* w, int Cout, int Cin, int R, int S, float* wt) {
  if (!w || !wt || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return CP_ERR_INVALID;
  const size_t total = (size_t)Cout * Cin * R * S;
  CP_LAUNCH(weight_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin,
            R, S, total);
  return cp_check_launch();
}

// EdgeConv weight views (init.py:58-62): w (C', 2C) = [W1 | W2].
//   mode 0 (forward / weight-gradient view): out (2C', C) = [W1 ; W2 - W1]
//   mode 1 (data-gradient view):             out (C, 2C') : out[c][c'] = W1[c'][c], out[c][C'+c'] = W2[c'][c]
__global__ void edge_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int mode, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (mode == 0) {
    const int c = (int)(i % Ci);
    const int row = (int)(i / Ci);
    const int co = row < Co ? row : row - Co;
    const float w1 = w[(size_t)co * 2 * Ci + c], w2 = w[(size_t)co * 2 * Ci + Ci + c];
    out[i] = row < Co ? w1 : w2 - w1;
  } else {
    const int col = (int)(i % (2 * Co));
    const int c = (int)(i / (2 * Co));
    out[i] = col < Co ? w[(size_t)col * 2 * Ci + c] : w[(size_t)(col - Co) * 2 * Ci + Ci + c];
  }
}

extern "C" int cp_edge_weight_view(cp_stream_t stream, const float* w, int Cout, int Cin, int mode, float* out) {
  if (!w || !out || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return CP_ERR_INVALID;
  const size_t total = (size_t)2 * Cout * Cin;
  CP_LAUNCH(edge_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin,
            mode, total);
  return cp_check_launch();
}

// tensor (fp32 or `dtype`) with arbitrary element strides -> channels-last `dtype` (B, HW, Cphys), padded channels zero.
template <typename Tag, typename STag>
__global__ void strided_to_nhwc_kernel(const void* __restrict__ src, long long base, long long sb, long long sp, long long sc,
                                       void* __restrict__ out, int HW, int C, int Cphys, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*HW*Cphys
  if (i >= total) return;
  const int c = (int)(i % Cphys);
  const size_t t = i / Cphys;
  const int pix = (int)(t % HW);
  const size_t b = t / HW;
  const float v = c < C ? load_elem<STag>(src, (size_t)(base + (long long)b * sb + (long long)pix * sp + (long long)c * sc)) : 0.f;
  store_elem<Tag>(out, i, v);
}

extern "C" int cp_strided_to_nhwc(cp_stream_t stream, int dtype, const void* src, int src_dtype, long long base, long long sb,
                                  long long sp, long long sc, void* out, int B, int HW, int C, int Cphys) {
  if (!src || !out || B <= 0 || HW <= 0 || C <= 0 || Cphys < C) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (src_dtype != CP_F32 && src_dtype != dtype) return CP_ERR_INVALID;
  const size_t total = (size_t)B * HW * Cphys;
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CP_F32)
    CP_LAUNCH((strided_to_nhwc_kernel<F32Tag, F32Tag>), grid, blk, 0, st, src, base, sb, sp, sc, out, HW, C, Cphys, total);
  else if (src_dtype == CP_F32)
    CP_LAUNCH((strided_to_nhwc_kernel<BF16Tag, F32Tag>), grid, blk, 0, st, src, base, sb, sp, sc, out, HW, C, Cphys, total);
  else
    CP_LAUNCH((strided_to_nhwc_kernel<BF16Tag, BF16Tag>), grid, blk, 0, st, src, base, sb, sp, sc, out, HW, C, Cphys, total);
  return cp_check_launch();
}

extern "C" int cp_memcpy_d2d(cp_stream_t stream, void* dst, const void* src, size_t nbytes) {
  if (!dst || !src) return CP_ERR_INVALID;
  if (nbytes == 0) return CP_OK;
  const int vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
  const size_t nvec = nbytes / 16;
  const size_t n = vec ? (nvec > 0 ? nvec : 1) : nbytes;
  CP_LAUNCH(copy_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst,
            (const unsigned char*)src, nvec, nbytes, vec);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------ column sums
// Per-channel sums of two derived quantities over the M rows of a channels-last tensor, fp64 block partials:
//   mode 0: (x, x^2)                                   -- BatchNorm batch statistics
//   mode 1: (dz, dz * xhat), dz = dy * act'(y)         -- BatchNorm / bias backward reductions
// ---- per-channel finalize of the fp64 block partials (one WAVE per channel: lanes stride over the partial blocks, shuffle
// reduction), its own launch of ceil(C/4) blocks.  (Fusing it into the last block of the column-sum kernel behind a ticket
// counter was measured 5x SLOWER: one block then walks all C x 512 partials alone.)
struct FinArgs {
  int nblk, CP, C, Cvec, has_bn;
  double count;
  const float* gamma; const float* beta; const float* mean_in; const float* rstd_in;
  float eps, momentum;
  float *rmean, *rvar, *scale, *shift, *mean, *rstd;     // forward outputs
  float *coef, *dgamma, *dbeta;                           // backward outputs
};

__device__ __forceinline__ void fin_sums(const double* partial, int nblk, int CP, int c, int lane, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
  for (int b = lane; b < nblk; b += 64) { s1 += partial[((size_t)b * 2) * CP + c]; s2 += partial[((size_t)b * 2 + 1) * CP + c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
}

__device__ __forceinline__ void fin_fwd_channel(const double* partial, const FinArgs& f, int c, int lane) {
  if (c >= f.C) { if (lane == 0) { f.scale[c] = 0.f; f.shift[c] = 0.f; f.mean[c] = 0.f; f.rstd[c] = 0.f; } return; }
  double s1, s2;
  fin_sums(partial, f.nblk, f.CP, c, lane, s1, s2);
  if (lane != 0) return;
  const double mu = s1 / f.count;
  double var = s2 / f.count - mu * mu;
  var = var > 0.0 ? var : 0.0;
  const float rs = (float)(1.0 / sqrt(var + (double)f.eps));
  const float g = f.gamma ? f.gamma[c] : 1.f, bt = f.beta ? f.beta[c] : 0.f;
  f.mean[c] = (float)mu;
  f.rstd[c] = rs;
  f.scale[c] = g * rs;
  f.shift[c] = bt - (float)mu * g * rs;
  if (f.rmean) f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * (float)mu;
  if (f.rvar) f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)(f.count > 1.0 ? var * f.count / (f.count - 1.0) : var);
}

// coef[0..3][Cvec]: a = gamma*rstd, b = mean(dz), c = mean(dz*xhat)*rstd, mu ; dgamma / dbeta written for c < C
__device__ __forceinline__ void fin_bwd_channel(const double* partial, const FinArgs& f, int c, int lane) {
  const int Cvec = f.Cvec;
  if (c >= f.C) { if (lane == 0) { f.coef[c] = 0.f; f.coef[Cvec + c] = 0.f; f.coef[2 * Cvec + c] = 0.f; f.coef[3 * Cvec + c] = 0.f; } return; }
  double s1, s2;
  fin_sums(partial, f.nblk, f.CP, c, lane, s1, s2);
  if (lane != 0) return;
  if (f.dbeta) f.dbeta[c] = (float)s1;
  if (f.has_bn) {
    if (f.dgamma) f.dgamma[c] = (float)s2;
    const float g = f.gamma ? f.gamma[c] : 1.f;
    f.coef[c] = g * f.rstd_in[c];
    f.coef[Cvec + c] = (float)(s1 / f.count);
    f.coef[2 * Cvec + c] = (float)(s2 / f.count) * f.rstd_in[c];
    f.coef[3 * Cvec + c] = f.mean_in[c];
  } else {
    f.coef[c] = 1.f; f.coef[Cvec + c] = 0.f; f.coef[2 * Cvec + c] = 0.f; f.coef[3 * Cvec + c] = 0.f;
  }
}

// the fp64 accumulators of the two-launch forms are replicated CP_BN_ACC_SETS times (block b adds into set b % SETS, the
// consumer sums the sets): 512 blocks hitting the same 48 addresses serialised in the L2 atomic unit
// Deterministic mode (cp_set_deterministic): CP_BN_DET_SETS sets and at most as many blocks per launch -- every block then owns
// its set (one add into a zeroed slot is exact), and the consumers add the sets in index order.
constexpr int CP_BN_ACC_SETS = 8;
constexpr int CP_BN_DET_SETS = 64;
static inline int bn_acc_sets() { return cp_deterministic() ? CP_BN_DET_SETS : CP_BN_ACC_SETS; }

struct ColsumParams {
  const void* a; int a_cs, a_coff;      // mode 0: x ; mode 1: dy
  const void* y; int y_cs, y_coff;      // mode 1: post-activation output (NULL: no activation)
  const void* x; int x_cs, x_coff;      // mode 1: raw BN input (NULL: xhat := 0)
  const float* mean; const float* rstd;
  float slope; int act;
  int M, G, RL, rpb;
  double* partial;                      // [nblk][2][G*E]
  double* acc; int acc_stride;          // non-NULL: fp64 atomics into acc[which*acc_stride + c] instead of block partials
  int sets;                             // ... replicated `sets` times (a power of two): block b adds into set b % sets
};

template <typename Tag, int MODE>
__device__ __forceinline__ void colsum2_body(const ColsumParams& p, double* red, unsigned bid) {
  constexpr int E = Tag::E;
  const int tid = threadIdx.x;
  const int rl = tid / p.G, piece = tid - rl * p.G;
  double s1[E], s2[E];
#pragma unroll
  for (int j = 0; j < E; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
  if (rl < p.RL) {
    const int m_end = min((int)(bid + 1) * p.rpb, p.M);
    float mu[E], rs[E];
    if (MODE == 1 && p.x) {
#pragma unroll
      for (int j = 0; j < E; ++j) { mu[j] = p.mean[piece * E + j]; rs[j] = p.rstd[piece * E + j]; }
    }
#pragma unroll 4
    for (int m = bid * p.rpb + rl; m < m_end; m += p.RL) {
      float a[E];
      Vec16<Tag>::unpack(*(const u32x4*)((const typename Tag::elem*)p.a + (size_t)m * p.a_cs + p.a_coff + piece * E), a);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < E; ++j) { s1[j] += (double)a[j]; s2[j] += (double)a[j] * (double)a[j]; }
      } else {
        if (p.y) {
          float yv[E];
          Vec16<Tag>::unpack(*(const u32x4*)((const typename Tag::elem*)p.y + (size_t)m * p.y_cs + p.y_coff + piece * E), yv);
#pragma unroll
          for (int j = 0; j < E; ++j) a[j] = yv[j] > 0.f ? a[j] : a[j] * p.slope;
        }
        if (p.x) {
          float xv[E];
          Vec16<Tag>::unpack(*(const u32x4*)((const typename Tag::elem*)p.x + (size_t)m * p.x_cs + p.x_coff + piece * E), xv);
#pragma unroll
          for (int j = 0; j < E; ++j) { s1[j] += (double)a[j]; s2[j] += (double)a[j] * (double)((xv[j] - mu[j]) * rs[j]); }
        } else {
#pragma unroll
          for (int j = 0; j < E; ++j) s1[j] += (double)a[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < E; ++j) { red[tid * 2 * E + j] = s1[j]; red[tid * 2 * E + E + j] = s2[j]; }
  __syncthreads();
  const int CP = p.G * E;
  if (2 * CP <= 128) {
    // narrow layers (<= 64 channels: RL = 256 / G is 32 .. 128 rows): a serial loop over RL rows by 2*CP threads was the longest
    // phase of these latency-bound launches -- T = 256 / (2*CP) threads share each column's rows, a second step adds their T sums
    __shared__ double red2[256];
    const int T = 256 / (2 * CP);
    const int o = tid % (2 * CP), part = tid / (2 * CP);
    const int which = o / CP, c = o - which * CP;
    const int pc = c / E, j = c - pc * E;
    if (part < T) {
      double s = 0.0;
      for (int r = part; r < p.RL; r += T) s += red[(r * p.G + pc) * 2 * E + which * E + j];
      red2[part * 2 * CP + o] = s;
    }
    __syncthreads();
    if (tid < 2 * CP) {
      double s = 0.0;
      for (int k = 0; k < T; ++k) s += red2[k * 2 * CP + tid];
      if (p.acc) unsafeAtomicAdd(p.acc + ((bid & (p.sets - 1)) * 2 + which) * p.acc_stride + c, s);
      else p.partial[((size_t)bid * 2 + which) * CP + c] = s;
    }
    return;
  }
  for (int o = tid; o < 2 * CP; o += 256) {
    const int which = o / CP, c = o - which * CP;
    const int pc = c / E, j = c - pc * E;
    double s = 0.0;
    for (int r = 0; r < p.RL; ++r) s += red[(r * p.G + pc) * 2 * E + which * E + j];
    if (p.acc) unsafeAtomicAdd(p.acc + ((bid & (p.sets - 1)) * 2 + which) * p.acc_stride + c, s);
    else p.partial[((size_t)bid * 2 + which) * CP + c] = s;
  }
}

template <typename Tag, int MODE>
__global__ __launch_bounds__(256) void colsum2_kernel(const ColsumParams p) {
  __shared__ double red[256 * 2 * Tag::E];
  colsum2_body<Tag, MODE>(p, red, blockIdx.x);
}

static int colsum_plan(int M, int Cphys, int E, int* G, int* RL, int* nblk, int* rpb) {
  *G = Cphys / E;
  if (*G > 256) return CP_ERR_INVALID;
  *RL = 256 / *G;
  int nb = M / 64;
  nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
  *rpb = (M + nb - 1) / nb;
  *nblk = (M + *rpb - 1) / *rpb;
  return CP_OK;
}

extern "C" size_t cp_bn_workspace_bytes(int C) { return (size_t)512 * 2 * ((size_t)(C + 15) / 16 * 16) * sizeof(double); }

// forward finalize: mean / biased var -> scale, shift, mean, rstd; running stats (momentum, unbiased var)
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* __restrict__ partial, const FinArgs f) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c < f.Cvec) fin_fwd_channel(partial, f, c, threadIdx.x & 63);
}

// shared by cp_bn_train_stats and the EdgeConv statistics (train_edge.hip)
static FinArgs fin_fwd_args(int nblk, int CP, int C, double count, const float* gamma, const float* beta, float eps, float momentum,
                            float* rmean, float* rvar, float* scale, float* shift, float* mean, float* rstd) {
  FinArgs f = {};
  f.nblk = nblk; f.CP = CP; f.C = C; f.Cvec = (C + 15) / 16 * 16; f.count = count; f.gamma = gamma; f.beta = beta; f.eps = eps;
  f.momentum = momentum; f.rmean = rmean; f.rvar = rvar; f.scale = scale; f.shift = shift; f.mean = mean; f.rstd = rstd;
  return f;
}

int cp_bn_finalize_launch(hipStream_t st, const double* partial, int nblk, int CP, int C, double count, const float* gamma,
                          const float* beta, float eps, float momentum, float* rmean, float* rvar, float* scale, float* shift,
                          float* mean, float* rstd) {
  const FinArgs f = fin_fwd_args(nblk, CP, C, count, gamma, beta, eps, momentum, rmean, rvar, scale, shift, mean, rstd);
  CP_LAUNCH(bn_fwd_finalize_kernel, dim3((f.Cvec + 3) / 4), dim3(256), 0, st, partial, f);
  return cp_check_launch();
}

static int check_cl(int dtype, const void* p, int cs, int coff, int Cphys) {
  const int E = cp_chan_align(dtype);
  if (!p) return CP_ERR_INVALID;
  if (cs % E || coff % E || coff + Cphys > cs || !cp_aligned16(p)) return CP_ERR_ALIGN;
  return CP_OK;
}

extern "C" int cp_bn_train_stats(cp_stream_t stream, int dtype, const void* x, int M, int C, int x_cstride, int x_coff,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                                 void* workspace) {
  if (!scale || !shift || !mean || !rstd || !workspace || M <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E;
  int rc = check_cl(dtype, x, x_cstride, x_coff, Cphys);
  if (rc) return rc;
  ColsumParams p = {};
  int nblk;
  if ((rc = colsum_plan(M, Cphys, E, &p.G, &p.RL, &nblk, &p.rpb))) return rc;
  p.a = x; p.a_cs = x_cstride; p.a_coff = x_coff; p.M = M; p.partial = (double*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CP_F32) CP_LAUNCH((colsum2_kernel<F32Tag, 0>), dim3(nblk), dim3(256), 0, st, p);
  else CP_LAUNCH((colsum2_kernel<BF16Tag, 0>), dim3(nblk), dim3(256), 0, st, p);
  if ((rc = cp_check_launch())) return rc;
  return cp_bn_finalize_launch(st, p.partial, nblk, Cphys, C, (double)M, gamma, beta, eps, momentum, running_mean, running_var,
                               scale, shift, mean, rstd);
}

// ------------------------------------------------------------------------------------------------ y = act(x*s + t + res)
struct AffineParams {
  const void* x; int x_cs, x_coff;
  const void* res; int r_cs, r_coff;
  void* y; int y_cs, y_coff;
  const float* scale; const float* shift;
  int G, act; float slope; size_t total;
};

template <typename Tag>
__global__ void affine_act_kernel(const AffineParams p) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over M*G
  if (i >= p.total) return;
  const int g = (int)(i % p.G);
  const size_t m = i / p.G;
  float v[E], r[E];
  Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.x + m * p.x_cs + p.x_coff + g * E), v);
#pragma unroll
  for (int j = 0; j < E; ++j) v[j] = v[j] * p.scale[g * E + j] + p.shift[g * E + j];
  if (p.res) {
    Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.res + m * p.r_cs + p.r_coff + g * E), r);
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] += r[j];
  }
#pragma unroll
  for (int j = 0; j < E; ++j) {
    if (p.act == CP_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
    else if (p.act == CP_ACT_LEAKY) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
  }
  *(u32x4*)((T*)p.y + m * p.y_cs + p.y_coff + g * E) = Vec16<Tag>::pack(v);
}

extern "C" int cp_affine_act(cp_stream_t stream, int dtype, const void* x, int x_cstride, int x_coff, const float* scale,
                             const float* shift, const void* res, int res_cstride, int res_coff, void* y, int y_cstride,
                             int y_coff, int M, int C, int act, float slope) {
  if (!scale || !shift || M <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E;
  int rc;
  if ((rc = check_cl(dtype, x, x_cstride, x_coff, Cphys)) || (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (res && (rc = check_cl(dtype, res, res_cstride, res_coff, Cphys))) return rc;
  AffineParams p;
  p.x = x; p.x_cs = x_cstride; p.x_coff = x_coff; p.res = res; p.r_cs = res_cstride; p.r_coff = res_coff;
  p.y = y; p.y_cs = y_cstride; p.y_coff = y_coff; p.scale = scale; p.shift = shift;
  p.G = Cphys / E; p.act = act; p.slope = slope; p.total = (size_t)M * p.G;
  const unsigned blocks = (unsigned)((p.total + 255) / 256);
  if (dtype == CP_F32) CP_LAUNCH(affine_act_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else CP_LAUNCH(affine_act_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------ BN / bias backward
// coef[0..3][Cvec]: a = gamma*rstd, b = mean(dz), c = mean(dz*xhat)*rstd, mu ; dgamma / dbeta written for c < C
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, const FinArgs f) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per channel
  if (c < f.Cvec) fin_bwd_channel(partial, f, c, threadIdx.x & 63);
}

static FinArgs fin_bwd_args(int nblk, int CP, int C, double count, const float* gamma, const float* mean, const float* rstd, int has_bn,
                            float* coef, float* dgamma, float* dbeta) {
  FinArgs f = {};
  f.nblk = nblk; f.CP = CP; f.C = C; f.Cvec = (C + 15) / 16 * 16; f.count = count; f.gamma = gamma; f.mean_in = mean; f.rstd_in = rstd;
  f.has_bn = has_bn; f.coef = coef; f.dgamma = dgamma; f.dbeta = dbeta;
  return f;
}

int cp_bn_bwd_finalize_launch(hipStream_t st, const double* partial, int nblk, int CP, int C, double count, const float* gamma,
                              const float* mean, const float* rstd, float* coef, float* dgamma, float* dbeta) {
  const FinArgs f = fin_bwd_args(nblk, CP, C, count, gamma, mean, rstd, 1, coef, dgamma, dbeta);
  CP_LAUNCH(bn_bwd_finalize_kernel, dim3((f.Cvec + 3) / 4), dim3(256), 0, st, partial, f);
  return cp_check_launch();
}

struct BnBwdParams {
  const void* dy; int dy_cs, dy_coff;
  const void* y; int y_cs, y_coff;
  const void* x; int x_cs, x_coff;
  void* dx; int dx_cs, dx_coff;
  void* dres; int dr_cs, dr_coff, dr_acc;
  const float* coef; int Cvec;
  int G; float slope; size_t total;
};

template <typename Tag>
__global__ void bn_bwd_dx_kernel(const BnBwdParams p) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.total) return;
  const int g = (int)(i % p.G);
  const size_t m = i / p.G;
  float dz[E], t[E];
  Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.dy + m * p.dy_cs + p.dy_coff + g * E), dz);
  if (p.y) {
    Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.y + m * p.y_cs + p.y_coff + g * E), t);
#pragma unroll
    for (int j = 0; j < E; ++j) dz[j] = t[j] > 0.f ? dz[j] : dz[j] * p.slope;
  }
  if (p.dres) {
    float o[E];
    if (p.dr_acc) {
      Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.dres + m * p.dr_cs + p.dr_coff + g * E), o);
#pragma unroll
      for (int j = 0; j < E; ++j) o[j] += dz[j];
    } else {
#pragma unroll
      for (int j = 0; j < E; ++j) o[j] = dz[j];
    }
    *(u32x4*)((T*)p.dres + m * p.dr_cs + p.dr_coff + g * E) = Vec16<Tag>::pack(o);
  }
  float xv[E];
  if (p.x) Vec16<Tag>::unpack(*(const u32x4*)((const T*)p.x + m * p.x_cs + p.x_coff + g * E), xv);
  float o[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int c = g * E + j;
    const float a = p.coef[c], b = p.coef[p.Cvec + c], cr = p.coef[2 * p.Cvec + c], mu = p.coef[3 * p.Cvec + c];
    o[j] = p.x ? a * (dz[j] - b - (xv[j] - mu) * cr) : dz[j];
  }
  *(u32x4*)((T*)p.dx + m * p.dx_cs + p.dx_coff + g * E) = Vec16<Tag>::pack(o);
}

extern "C" size_t cp_bn_bwd_workspace_bytes(int C) {
  return cp_bn_workspace_bytes(C) + (size_t)4 * ((size_t)(C + 15) / 16 * 16) * sizeof(float);
}

extern "C" int cp_bn_train_bwd(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y,
                               int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* mean,
                               const float* rstd, const float* gamma, int M, int C, int act, float slope, void* dx,
                               int dx_cstride, int dx_coff, void* dres, int dres_cstride, int dres_coff, int dres_accumulate,
                               float* dgamma, float* dbeta, void* workspace) {
  if (!workspace || M <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (x && (!mean || !rstd)) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E, Cvec = (C + 15) / 16 * 16;
  int rc;
  if ((rc = check_cl(dtype, dy, dy_cstride, dy_coff, Cphys)) || (rc = check_cl(dtype, dx, dx_cstride, dx_coff, Cphys))) return rc;
  const void* yy = act == CP_ACT_NONE ? nullptr : y;
  if (act != CP_ACT_NONE && (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (x && (rc = check_cl(dtype, x, x_cstride, x_coff, Cphys))) return rc;
  if (dres && (rc = check_cl(dtype, dres, dres_cstride, dres_coff, Cphys))) return rc;
  const float sl = act == CP_ACT_RELU ? 0.f : slope;
  ColsumParams p = {};
  int nblk;
  if ((rc = colsum_plan(M, Cphys, E, &p.G, &p.RL, &nblk, &p.rpb))) return rc;
  p.a = dy; p.a_cs = dy_cstride; p.a_coff = dy_coff; p.y = yy; p.y_cs = y_cstride; p.y_coff = y_coff;
  p.x = x; p.x_cs = x_cstride; p.x_coff = x_coff; p.mean = mean; p.rstd = rstd; p.slope = sl; p.act = act;
  p.M = M; p.partial = (double*)workspace;
  float* coef = (float*)((char*)workspace + cp_bn_workspace_bytes(C));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CP_F32) CP_LAUNCH((colsum2_kernel<F32Tag, 1>), dim3(nblk), dim3(256), 0, st, p);
  else CP_LAUNCH((colsum2_kernel<BF16Tag, 1>), dim3(nblk), dim3(256), 0, st, p);
  if ((rc = cp_check_launch())) return rc;
  {
    const FinArgs f = fin_bwd_args(nblk, Cphys, C, (double)M, gamma, mean, rstd, x ? 1 : 0, coef, dgamma, dbeta);
    CP_LAUNCH(bn_bwd_finalize_kernel, dim3((f.Cvec + 3) / 4), dim3(256), 0, st, p.partial, f);
    if ((rc = cp_check_launch())) return rc;
  }
  BnBwdParams q;
  q.dy = dy; q.dy_cs = dy_cstride; q.dy_coff = dy_coff; q.y = yy; q.y_cs = y_cstride; q.y_coff = y_coff;
  q.x = x; q.x_cs = x_cstride; q.x_coff = x_coff; q.dx = dx; q.dx_cs = dx_cstride; q.dx_coff = dx_coff;
  q.dres = dres; q.dr_cs = dres_cstride; q.dr_coff = dres_coff; q.dr_acc = dres_accumulate;
  q.coef = coef; q.Cvec = Cvec; q.G = Cphys / E; q.slope = sl; q.total = (size_t)M * q.G;
  const unsigned blocks = (unsigned)((q.total + 255) / 256);
  if (dtype == CP_F32) CP_LAUNCH(bn_bwd_dx_kernel<F32Tag>, dim3(blocks), dim3(256), 0, st, q);
  else CP_LAUNCH(bn_bwd_dx_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, st, q);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------ resampling backward
// Adjoint of cp_upsample2x_bilinear_ac in gather form (deterministic): input pixel (y,x) collects every output pixel
// whose 2x2 footprint touches it, with the forward's own fp32 weights.
template <typename Tag>
__global__ void upsample2x_bwd_kernel(const void* __restrict__ dout, void* __restrict__ din, int H, int W, int CG, int o_cs,
                                      int o_coff, int i_cs, int i_coff, float sy, float sx, int accumulate, size_t total) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*W*CG
  if (i >= total) return;
  const int g = (int)(i % CG);
  size_t t = i / CG;
  const int x = (int)(t % W); t /= W;
  const int y = (int)(t % H);
  const size_t b = t / H;
  float acc[E];
#pragma unroll
  for (int j = 0; j < E; ++j) acc[j] = 0.f;
  // candidate outputs: src = s*o in (y-1, y+1)  ->  o in ((y-1)/s, (y+1)/s); scan a safe integer window
  const int oy_lo = max(0, (int)floorf((y - 1) / fmaxf(sy, 1e-20f)) - 1), oy_hi = min(2 * H - 1, (int)ceilf((y + 1) / fmaxf(sy, 1e-20f)) + 1);
  const int ox_lo = max(0, (int)floorf((x - 1) / fmaxf(sx, 1e-20f)) - 1), ox_hi = min(2 * W - 1, (int)ceilf((x + 1) / fmaxf(sx, 1e-20f)) + 1);
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float fy = sy * oy;
    const int y0 = (int)fy, y1 = y0 + (y0 < H - 1);
    const float ly1 = fy - y0, ly0 = 1.f - ly1;
    const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
    if (y0 != y && y1 != y) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = sx * ox;
      const int x0 = (int)fx, x1 = x0 + (x0 < W - 1);
      const float lx1 = fx - x0, lx0 = 1.f - lx1;
      const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
      if (x0 != x && x1 != x) continue;
      float d[E];
      Vec16<Tag>::unpack(*(const u32x4*)((const T*)dout + ((b * 2 * H + oy) * 2 * W + ox) * o_cs + o_coff + (size_t)g * E), d);
      const float w = wy * wx;
#pragma unroll
      for (int j = 0; j < E; ++j) acc[j] += w * d[j];
    }
  }
  T* dst = (T*)din + ((b * H + y) * W + x) * i_cs + i_coff + (size_t)g * E;
  if (accumulate) {
    float o[E];
    Vec16<Tag>::unpack(*(const u32x4*)dst, o);
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] += o[j];
  }
  *(u32x4*)dst = Vec16<Tag>::pack(acc);
}

extern "C" int cp_upsample2x_bilinear_ac_bwd(cp_stream_t stream, int dtype, const void* dout, void* din, int B, int H, int W,
                                             int C, int out_cstride, int out_coff, int in_cstride, int in_coff, int accumulate) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E) return CP_ERR_ALIGN;
  int rc;
  if ((rc = check_cl(dtype, dout, out_cstride, out_coff, C)) || (rc = check_cl(dtype, din, in_cstride, in_coff, C))) return rc;
  const float sy = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
  const float sx = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  const int CG = C / E;
  const size_t total = (size_t)B * H * W * CG;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(upsample2x_bwd_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, din, H, W, CG, out_cstride,
              out_coff, in_cstride, in_coff, sy, sx, accumulate, total);
  else
    CP_LAUNCH(upsample2x_bwd_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, din, H, W, CG, out_cstride,
              out_coff, in_cstride, in_coff, sy, sx, accumulate, total);
  return cp_check_launch();
}

// Backward of cp_fuse_sum_act for ONE source: dsrc[b,y,x,:] (+)= sum over the 2^sh x 2^sh block of dout * [out > 0]
struct FuseBwdParams {       // == CpFuseBwdItem
  const void* dout; const void* out; void* dsrc;
  int Hs, Ws, CG, sh, relu, accumulate;
  unsigned long long total;
};
static_assert(sizeof(FuseBwdParams) == sizeof(CpFuseBwdItem), "CpFuseBwdItem layout");

template <typename Tag>
__device__ __forceinline__ void fuse_sum_bwd_elem(const FuseBwdParams& p, size_t i) {      // i over B*Hs*Ws*CG
  constexpr int E = Tag::E;
  const int g = (int)(i % p.CG);
  size_t t = i / p.CG;
  const int x = (int)(t % p.Ws); t /= p.Ws;
  const int y = (int)(t % p.Hs);
  const size_t b = t / p.Hs;
  const int sh = p.sh, n = 1 << sh, H = p.Hs << sh, W = p.Ws << sh;
  float acc[E];
#pragma unroll
  for (int j = 0; j < E; ++j) acc[j] = 0.f;
  for (int dy = 0; dy < n; ++dy)
    for (int dx = 0; dx < n; ++dx) {
      const size_t v = ((b * H + (y << sh) + dy) * W + (x << sh) + dx) * p.CG + g;
      float d[E], o[E];
      Vec16<Tag>::unpack(((const u32x4*)p.dout)[v], d);
      if (p.relu) {
        Vec16<Tag>::unpack(((const u32x4*)p.out)[v], o);
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] += o[j] > 0.f ? d[j] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] += d[j];
      }
    }
  if (p.accumulate) {
    float o[E];
    Vec16<Tag>::unpack(((const u32x4*)p.dsrc)[i], o);
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] += o[j];
  }
  ((u32x4*)p.dsrc)[i] = Vec16<Tag>::pack(acc);
}

template <typename Tag>
__global__ void fuse_sum_bwd_kernel(const FuseBwdParams p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.total) fuse_sum_bwd_elem<Tag>(p, i);
}

// every (output, term) pair of an HRNet module's fuse layer in one launch: the pairs write distinct gradient tensors
template <typename Tag>
__global__ __launch_bounds__(256) void fuse_sum_bwd_group_kernel(const FuseBwdParams* __restrict__ items, const uint32_t* __restrict__ prefix, int n) {
  int k = 0;
  while (k + 1 < n && blockIdx.x >= prefix[k + 1]) ++k;
  const FuseBwdParams p = items[k];
  const size_t i = (size_t)(blockIdx.x - prefix[k]) * 256 + threadIdx.x;
  if (i < p.total) fuse_sum_bwd_elem<Tag>(p, i);
}

static int build_fuse_bwd(int dtype, const void* dout, const void* out, void* dsrc, int B, int Hs, int Ws, int C, int shift, int relu,
                          int accumulate, FuseBwdParams* p) {
  if (!dout || !dsrc || (relu && !out) || B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || shift < 0 || shift > 5) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || !cp_aligned16(dout) || !cp_aligned16(dsrc) || (out && !cp_aligned16(out))) return CP_ERR_ALIGN;
  p->dout = dout; p->out = out; p->dsrc = dsrc; p->Hs = Hs; p->Ws = Ws; p->CG = C / E; p->sh = shift; p->relu = relu; p->accumulate = accumulate;
  p->total = (unsigned long long)B * Hs * Ws * p->CG;
  return CP_OK;
}

extern "C" int cp_fuse_sum_act_bwd(cp_stream_t stream, int dtype, const void* dout, const void* out, void* dsrc, int B, int Hs,
                                   int Ws, int C, int shift, int relu, int accumulate) {
  FuseBwdParams p;
  const int rc = build_fuse_bwd(dtype, dout, out, dsrc, B, Hs, Ws, C, shift, relu, accumulate, &p);
  if (rc) return rc;
  const unsigned blocks = (unsigned)((p.total + 255) / 256);
  if (dtype == CP_F32) CP_LAUNCH(fuse_sum_bwd_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else CP_LAUNCH(fuse_sum_bwd_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}

extern "C" int cp_fuse_sum_act_bwd_item(int dtype, const void* dout, const void* out, void* dsrc, int B, int Hs, int Ws, int C, int shift,
                                        int relu, int accumulate, CpFuseBwdItem* item, uint32_t* blocks) {
  if (!item || !blocks) return CP_ERR_INVALID;
  FuseBwdParams p;
  const int rc = build_fuse_bwd(dtype, dout, out, dsrc, B, Hs, Ws, C, shift, relu, accumulate, &p);
  if (rc) return rc;
  memcpy(item, &p, sizeof(p));
  *blocks = (uint32_t)((p.total + 255) / 256);
  return CP_OK;
}

extern "C" int cp_fuse_sum_act_bwd_group(cp_stream_t stream, int dtype, const CpFuseBwdItem* items_dev, const uint32_t* prefix_dev, int n_items,
                                         uint32_t total_blocks) {
  if (!items_dev || !prefix_dev || n_items <= 0 || n_items > 64 || total_blocks == 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (dtype == CP_F32)
    CP_LAUNCH(fuse_sum_bwd_group_kernel<F32Tag>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const FuseBwdParams*)items_dev, prefix_dev, n_items);
  else
    CP_LAUNCH(fuse_sum_bwd_group_kernel<BF16Tag>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const FuseBwdParams*)items_dev, prefix_dev, n_items);
  return cp_check_launch();
}

// Backward of cp_maxpool3x3s2 (F.max_pool2d(x, 3, 2, 1), resnet34 stem) in gather form: an input pixel belongs to at most
// four windows; for each it re-scans the window in row-major order (first maximum wins, like ATen's CPU kernel -- ties are
// common behind a ReLU) and takes the window's gradient if it is the arg-max.
template <typename Tag>
__global__ void maxpool3x3s2_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dout, void* __restrict__ din, int H, int W,
                                        int CG, int accumulate, size_t total) {
  constexpr int E = Tag::E;
  const int Ho = H / 2, Wo = W / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*W*CG
  if (i >= total) return;
  const int g = (int)(i % CG);
  size_t t = i / CG;
  const int xx = (int)(t % W); t /= W;
  const int yy = (int)(t % H);
  const size_t b = t / H;
  float acc[E];
#pragma unroll
  for (int j = 0; j < E; ++j) acc[j] = 0.f;
  for (int oy = yy / 2; oy <= (yy + 1) / 2; ++oy) {
    if (oy >= Ho) continue;
    for (int ox = xx / 2; ox <= (xx + 1) / 2; ++ox) {
      if (ox >= Wo) continue;
      float m[E], f[E];
      int arg[E];
#pragma unroll
      for (int j = 0; j < E; ++j) { m[j] = -INFINITY; arg[j] = -1; }
      for (int r = 0; r < 3; ++r) {
        const int y = 2 * oy - 1 + r;
        if ((unsigned)y >= (unsigned)H) continue;
        for (int s = 0; s < 3; ++s) {
          const int x2 = 2 * ox - 1 + s;
          if ((unsigned)x2 >= (unsigned)W) continue;
          Vec16<Tag>::unpack(((const u32x4*)x)[((b * H + y) * W + x2) * CG + g], f);
#pragma unroll
          for (int j = 0; j < E; ++j)
            if (f[j] > m[j]) { m[j] = f[j]; arg[j] = y * W + x2; }
        }
      }
      float d[E];
      Vec16<Tag>::unpack(((const u32x4*)dout)[((b * Ho + oy) * Wo + ox) * CG + g], d);
#pragma unroll
      for (int j = 0; j < E; ++j)
        if (arg[j] == yy * W + xx) acc[j] += d[j];
    }
  }
  if (accumulate) {
    float o[E];
    Vec16<Tag>::unpack(((const u32x4*)din)[i], o);
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] += o[j];
  }
  ((u32x4*)din)[i] = Vec16<Tag>::pack(acc);
}

extern "C" int cp_maxpool3x3s2_bwd(cp_stream_t stream, int dtype, const void* x, const void* dout, void* din, int B, int H, int W,
                                   int C, int accumulate) {
  if (!x || !dout || !din || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || !cp_aligned16(x) || !cp_aligned16(dout) || !cp_aligned16(din)) return CP_ERR_ALIGN;
  const int CG = C / E;
  const size_t total = (size_t)B * H * W * CG;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(maxpool3x3s2_bwd_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, dout, din, H, W, CG, accumulate, total);
  else
    CP_LAUNCH(maxpool3x3s2_bwd_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, dout, din, H, W, CG, accumulate, total);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------ two-launch BatchNorm
// Variant without the finalize launch (the training program's 670 finalize launches were 8.5 % of a step): the column sums
// go straight into a per-layer fp64 accumulator pair (hardware fp64 atomics, <= 128 blocks per launch; the caller zeroes
// all accumulators of a step with ONE fill), and the consumer kernel derives the per-channel coefficients in its prologue
// (the thread that owns row 0 of a channel group also writes the saved statistics / running stats / dgamma, dbeta).
static int colsum_plan_acc(int M, int Cphys, int E, int* G, int* RL, int* nblk, int* rpb) {
  *G = Cphys / E;
  if (*G > 256) return CP_ERR_INVALID;
  *RL = 256 / *G;
  static const int max_blocks = cp_knob("CP_BN_ACC_BLOCKS") ? atoi(cp_knob("CP_BN_ACC_BLOCKS")) : 256;
  static const int rows_per_block = cp_knob("CP_BN_ACC_ROWS") ? atoi(cp_knob("CP_BN_ACC_ROWS")) : 64;
  int nb = M / rows_per_block;                            // as many blocks as the partial-sum variant ...
  int cap = 65536 / (2 * Cphys);                          // ... but at most ~64k atomics per launch (~30 G atomics/s)
  // ... unless the tensor is big: 128 blocks on 256 CUs streamed the 64 x 64 x 256-channel maps of layer1 at 3.5 TB/s (57 us per
  // backward pass); from `elems_per_block` elements per block on, more blocks pay for their atomics.  Measured on the B = 32 training
  // step (KNOBS=1 build): narrow layers 512 -> 256 blocks -0.55 ms, big-tensor rule at 16 K elements / <= 768 blocks another -0.4 ms
  static const long long elems_per_block = cp_knob("CP_BN_ACC_ELEMS") ? atoll(cp_knob("CP_BN_ACC_ELEMS")) : 16384;
  static const int big_blocks = cp_knob("CP_BN_ACC_BIGBLOCKS") ? atoi(cp_knob("CP_BN_ACC_BIGBLOCKS")) : 768;
  const long long by_size = (long long)M * Cphys / elems_per_block;
  cap = cap < 32 ? 32 : (cap > max_blocks ? max_blocks : cap);
  if (by_size > cap) cap = (int)(by_size > big_blocks ? big_blocks : by_size);
  nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
  if (cp_deterministic() && nb > CP_BN_DET_SETS) nb = CP_BN_DET_SETS;      // one block per accumulator set
  *rpb = (M + nb - 1) / nb;
  *nblk = (M + *rpb - 1) / *rpb;
  return CP_OK;
}

extern "C" size_t cp_bn_acc_doubles(int C) { return (size_t)bn_acc_sets() * 2 * ((size_t)(C + 15) / 16 * 16); }

static int build_bn_stats(int dtype, const void* x, int M, int C, int x_cstride, int x_coff, double* acc, ColsumParams* out, int* nblk) {
  if (!acc || M <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E, Cvec = (C + 15) / 16 * 16;
  int rc = check_cl(dtype, x, x_cstride, x_coff, Cphys);
  if (rc) return rc;
  ColsumParams p = {};
  if ((rc = colsum_plan_acc(M, Cphys, E, &p.G, &p.RL, nblk, &p.rpb))) return rc;
  p.a = x; p.a_cs = x_cstride; p.a_coff = x_coff; p.M = M; p.acc = acc; p.acc_stride = Cvec; p.sets = bn_acc_sets();
  *out = p;
  return CP_OK;
}

extern "C" int cp_bn_stats_accumulate(cp_stream_t stream, int dtype, const void* x, int M, int C, int x_cstride, int x_coff,
                                      double* acc) {
  ColsumParams p;
  int nblk;
  const int rc = build_bn_stats(dtype, x, M, C, x_cstride, x_coff, acc, &p, &nblk);
  if (rc) return rc;
  if (dtype == CP_F32) CP_LAUNCH((colsum2_kernel<F32Tag, 0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  else CP_LAUNCH((colsum2_kernel<BF16Tag, 0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}

// per-channel coefficient tables in LDS: a lane reads the 8 (bf16) / 4 (fp32) coefficients of its 16-byte piece with ds_read_b128; at a piece
// stride of 8 floats the 16 lanes of a read phase fall on 16 x 4 banks of which only 8 x 4 are distinct (pieces g and g + 8 collide:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.82 in profiles/r03_sq_counters_train.csv) -- bf16 pieces are therefore 12 floats apart
template <typename Tag> struct CoefPitch { static constexpr int PS = Tag::E == 8 ? 12 : Tag::E; };
static inline size_t coef_floats(int dtype, int Cphys) { const int E = cp_chan_align(dtype); return (size_t)(Cphys / E) * (E == 8 ? 12 : E); }

struct BnApplyParams {
  const void* x; int x_cs, x_coff;
  const void* res; int r_cs, r_coff;
  void* y; int y_cs, y_coff;
  const double* acc; int Cvec, C, sets;
  double count;
  const float* gamma; const float* beta;
  float eps, momentum;
  float *rmean, *rvar, *mean, *rstd;
  int G, act; float slope; size_t total;
};

template <typename Tag>
__device__ __forceinline__ void bn_apply_coef(const BnApplyParams& p, float* s_coef, bool first) {      // [2][Cphys]: scale | shift, once per block
  constexpr int E = Tag::E;
  const int Cphys = p.G * E;
  const double inv = 1.0 / p.count;
  for (int c = threadIdx.x; c < Cphys; c += 256) {
    float sc = 0.f, sh = 0.f;
    if (c < p.C) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
      for (int k = 0; k < p.sets; ++k) { a1 += p.acc[(2 * k) * p.Cvec + c]; a2 += p.acc[(2 * k + 1) * p.Cvec + c]; }
      const double mu = a1 * inv;
      double var = a2 * inv - mu * mu;
      var = var > 0.0 ? var : 0.0;
      const float rs = 1.0f / sqrtf((float)var + p.eps);
      const float gm = p.gamma ? p.gamma[c] : 1.f, bt = p.beta ? p.beta[c] : 0.f;
      sc = gm * rs;
      sh = bt - (float)mu * sc;
      if (first) {                                 // one block writes the saved statistics + running stats
        p.mean[c] = (float)mu;
        p.rstd[c] = rs;
        if (p.rmean) p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * (float)mu;
        if (p.rvar) p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)(p.count > 1.0 ? var * p.count / (p.count - 1.0) : var);
      }
    }
    const int ci = (c / E) * CoefPitch<Tag>::PS + c % E;
    s_coef[ci] = sc;
    s_coef[p.G * CoefPitch<Tag>::PS + ci] = sh;
  }
}

// a piece's operands are fetched BEFORE the coefficient prologue (whose accumulator loads are a memory round trip of their own): the
// two latencies overlap instead of adding up -- these launches are a few microseconds long
template <typename Tag>
struct BnApplyOperands { u32x4 x, r; };

template <typename Tag>
__device__ __forceinline__ void bn_apply_fetch(const BnApplyParams& p, size_t i, BnApplyOperands<Tag>& o) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const int g = (int)(i % p.G);
  const size_t m = i / p.G;
  o.x = *(const u32x4*)((const T*)p.x + m * p.x_cs + p.x_coff + g * E);
  if (p.res) o.r = *(const u32x4*)((const T*)p.res + m * p.r_cs + p.r_coff + g * E);
}

template <typename Tag>
__device__ __forceinline__ void bn_apply_finish(const BnApplyParams& p, const float* s_coef, size_t i, const BnApplyOperands<Tag>& o) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const int g = (int)(i % p.G);
  const size_t m = i / p.G;
  float v[E], r[E];
  Vec16<Tag>::unpack(o.x, v);
#pragma unroll
  for (int j = 0; j < E; ++j) v[j] = v[j] * s_coef[g * CoefPitch<Tag>::PS + j] + s_coef[(p.G + g) * CoefPitch<Tag>::PS + j];
  if (p.res) {
    Vec16<Tag>::unpack(o.r, r);
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] += r[j];
  }
#pragma unroll
  for (int j = 0; j < E; ++j) {
    if (p.act == CP_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
    else if (p.act == CP_ACT_LEAKY) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
  }
  *(u32x4*)((T*)p.y + m * p.y_cs + p.y_coff + g * E) = Vec16<Tag>::pack(v);
}

template <typename Tag>
__device__ __forceinline__ void bn_apply_piece(const BnApplyParams& p, const float* s_coef, size_t i) {      // piece i of M * G
  BnApplyOperands<Tag> o;
  bn_apply_fetch<Tag>(p, i, o);
  bn_apply_finish<Tag>(p, s_coef, i, o);
}

template <typename Tag>
__device__ __forceinline__ void bn_apply_block(const BnApplyParams& p, float* s_coef, unsigned bid) {
  const size_t i = (size_t)bid * 256 + threadIdx.x;   // over M*G
  BnApplyOperands<Tag> o;
  if (i < p.total) bn_apply_fetch<Tag>(p, i, o);
  bn_apply_coef<Tag>(p, s_coef, bid == 0);
  __syncthreads();
  if (i < p.total) bn_apply_finish<Tag>(p, s_coef, i, o);
}

template <typename Tag>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyParams p) {
  extern __shared__ float s_coef[];
  bn_apply_block<Tag>(p, s_coef, blockIdx.x);
}

// ---- statistics + apply in ONE launch (the training step runs ~670 such pairs of 6-13 us launches): every block adds its rows'
// column sums to the layer's accumulators, all blocks meet at a grid barrier, then every block derives scale / shift from the
// complete sums and applies them to THE SAME rows (still in the L2).  The grid (<= 256 blocks of 256 threads, 32 KB of LDS) is
// always co-resident on the 256 CUs, and nothing a waiting block depends on can be blocked by it; the spin is bounded all the same.
// MEASURED NEGATIVE (round 3, bench_train.py, B = 32): 42.3 ms per step (256 blocks, poll every ~0.85 us) / 44.1 (128 blocks) /
// 52.5 (64) / 50.5 (512 blocks, fast polling) against 35.7 ms for the two-launch forms -- the barrier (an atomic arrival + polls
// of one L2 word by every block) costs >= 10 us per launch, more than the second launch it saves.  Kept as an opt-in
// (CHECKERPOSE_AMD_BN_FUSED=1) and as a tested entry point; the training program uses the two-launch forms.
__device__ __forceinline__ void cp_grid_barrier(uint32_t* counter, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    unsigned spins = 0;
#ifndef CP_BAR_SLEEP
#define CP_BAR_SLEEP 32
#endif
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(CP_BAR_SLEEP);
    __threadfence();
  }
  __syncthreads();
}

template <typename Tag>
__global__ __launch_bounds__(256) void bn_fused_fwd_kernel(const ColsumParams cs, const BnApplyParams ap, uint32_t* counter) {
  __shared__ double red[256 * 2 * Tag::E];
  extern __shared__ float s_coef[];
  colsum2_body<Tag, 0>(cs, red, blockIdx.x);
  cp_grid_barrier(counter, gridDim.x);
  bn_apply_coef<Tag>(ap, s_coef, blockIdx.x == 0);
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * cs.rpb * ap.G;
  size_t hi = lo + (size_t)cs.rpb * ap.G;
  hi = hi < ap.total ? hi : ap.total;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) bn_apply_piece<Tag>(ap, s_coef, i);
}

static int build_bn_apply(int dtype, const void* x, int x_cstride, int x_coff, const double* acc, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, const void* res, int res_cstride,
                          int res_coff, void* y, int y_cstride, int y_coff, int M, int C, int act, float slope, float* mean, float* rstd,
                          BnApplyParams* out, unsigned* blocks, int* Cphys_out) {
  if (!acc || !mean || !rstd || M <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E;
  int rc;
  if ((rc = check_cl(dtype, x, x_cstride, x_coff, Cphys)) || (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (res && (rc = check_cl(dtype, res, res_cstride, res_coff, Cphys))) return rc;
  BnApplyParams p;
  p.x = x; p.x_cs = x_cstride; p.x_coff = x_coff; p.res = res; p.r_cs = res_cstride; p.r_coff = res_coff;
  p.y = y; p.y_cs = y_cstride; p.y_coff = y_coff; p.acc = acc; p.Cvec = (C + 15) / 16 * 16; p.C = C; p.count = (double)M; p.sets = bn_acc_sets();
  p.gamma = gamma; p.beta = beta; p.eps = eps; p.momentum = momentum; p.rmean = running_mean; p.rvar = running_var;
  p.mean = mean; p.rstd = rstd; p.G = Cphys / E; p.act = act; p.slope = slope; p.total = (size_t)M * p.G;
  *out = p;
  *blocks = (unsigned)((p.total + 255) / 256);
  *Cphys_out = Cphys;
  return CP_OK;
}

extern "C" int cp_bn_apply(cp_stream_t stream, int dtype, const void* x, int x_cstride, int x_coff, const double* acc,
                           const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                           float eps, const void* res, int res_cstride, int res_coff, void* y, int y_cstride, int y_coff, int M,
                           int C, int act, float slope, float* mean, float* rstd) {
  BnApplyParams p;
  unsigned blocks;
  int Cphys;
  const int rc = build_bn_apply(dtype, x, x_cstride, x_coff, acc, gamma, beta, running_mean, running_var, momentum, eps, res, res_cstride,
                                res_coff, y, y_cstride, y_coff, M, C, act, slope, mean, rstd, &p, &blocks, &Cphys);
  if (rc) return rc;
  const size_t lds = 2 * coef_floats(dtype, Cphys) * sizeof(float);
  if (dtype == CP_F32) CP_LAUNCH(bn_apply_kernel<F32Tag>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, p);
  else CP_LAUNCH(bn_apply_kernel<BF16Tag>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, p);
  return cp_check_launch();
}

static int build_bn_bwd_sums(int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride, int y_coff, const void* x,
                             int x_cstride, int x_coff, const float* mean, const float* rstd, int M, int C, int act, float slope, double* acc,
                             ColsumParams* out, int* nblk) {
  if (!acc || M <= 0 || C <= 0 || (x && (!mean || !rstd))) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E, Cvec = (C + 15) / 16 * 16;
  int rc;
  if ((rc = check_cl(dtype, dy, dy_cstride, dy_coff, Cphys))) return rc;
  const void* yy = act == CP_ACT_NONE ? nullptr : y;
  if (act != CP_ACT_NONE && (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (x && (rc = check_cl(dtype, x, x_cstride, x_coff, Cphys))) return rc;
  ColsumParams p = {};
  if ((rc = colsum_plan_acc(M, Cphys, E, &p.G, &p.RL, nblk, &p.rpb))) return rc;
  p.a = dy; p.a_cs = dy_cstride; p.a_coff = dy_coff; p.y = yy; p.y_cs = y_cstride; p.y_coff = y_coff;
  p.x = x; p.x_cs = x_cstride; p.x_coff = x_coff; p.mean = mean; p.rstd = rstd; p.slope = act == CP_ACT_RELU ? 0.f : slope; p.act = act;
  p.M = M; p.acc = acc; p.acc_stride = Cvec; p.sets = bn_acc_sets();
  *out = p;
  return CP_OK;
}

extern "C" int cp_bn_bwd_accumulate(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y,
                                    int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* mean,
                                    const float* rstd, int M, int C, int act, float slope, double* acc) {
  ColsumParams p;
  int nblk;
  const int rc = build_bn_bwd_sums(dtype, dy, dy_cstride, dy_coff, y, y_cstride, y_coff, x, x_cstride, x_coff, mean, rstd, M, C, act, slope,
                                   acc, &p, &nblk);
  if (rc) return rc;
  if (dtype == CP_F32) CP_LAUNCH((colsum2_kernel<F32Tag, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  else CP_LAUNCH((colsum2_kernel<BF16Tag, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}

struct BnBwdApplyParams {
  BnBwdParams q;
  const double* acc; int C, sets; double count;
  const float* gamma; const float* mean; const float* rstd;
  float* dgamma; float* dbeta;
};

template <typename Tag>
__device__ __forceinline__ void bn_bwd_coef(const BnBwdApplyParams& pp, float* s_cf, bool first) {       // [4][Cphys]: a | b | cr | mu, once per block
  constexpr int E = Tag::E;
  const BnBwdParams& p = pp.q;
  const int Cphys = p.G * E;
  const double inv = 1.0 / pp.count;
  for (int c = threadIdx.x; c < Cphys; c += 256) {
    float ca = 0.f, cb = 0.f, cr = 0.f, mu = 0.f;
    if (c < pp.C) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
      for (int k = 0; k < pp.sets; ++k) { s1 += pp.acc[(2 * k) * p.Cvec + c]; s2 += pp.acc[(2 * k + 1) * p.Cvec + c]; }
      if (p.x) {
        const float rs = pp.rstd[c];
        ca = (pp.gamma ? pp.gamma[c] : 1.f) * rs;
        cb = (float)(s1 * inv);
        cr = (float)(s2 * inv) * rs;
        mu = pp.mean[c];
      }
      if (first) {
        if (pp.dbeta) pp.dbeta[c] = (float)s1;
        if (p.x && pp.dgamma) pp.dgamma[c] = (float)s2;
      }
    }
    const int ci = (c / E) * CoefPitch<Tag>::PS + c % E, AS = p.G * CoefPitch<Tag>::PS;
    s_cf[ci] = ca; s_cf[AS + ci] = cb; s_cf[2 * AS + ci] = cr; s_cf[3 * AS + ci] = mu;
  }
}

template <typename Tag>
struct BnBwdOperands { u32x4 dy, y, dres, x; };

template <typename Tag>
__device__ __forceinline__ void bn_bwd_fetch(const BnBwdParams& p, size_t i, BnBwdOperands<Tag>& o) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const int g = (int)(i % p.G);
  const size_t m = i / p.G;
  o.dy = *(const u32x4*)((const T*)p.dy + m * p.dy_cs + p.dy_coff + g * E);
  if (p.y) o.y = *(const u32x4*)((const T*)p.y + m * p.y_cs + p.y_coff + g * E);
  if (p.dres && p.dr_acc) o.dres = *(const u32x4*)((const T*)p.dres + m * p.dr_cs + p.dr_coff + g * E);
  if (p.x) o.x = *(const u32x4*)((const T*)p.x + m * p.x_cs + p.x_coff + g * E);
}

template <typename Tag>
__device__ __forceinline__ void bn_bwd_finish(const BnBwdParams& p, const float* s_cf, size_t i, const BnBwdOperands<Tag>& ops) {
  constexpr int E = Tag::E;
  using T = typename Tag::elem;
  const int g = (int)(i % p.G);
  const size_t m = i / p.G;
  float dz[E], t[E];
  Vec16<Tag>::unpack(ops.dy, dz);
  if (p.y) {
    Vec16<Tag>::unpack(ops.y, t);
#pragma unroll
    for (int j = 0; j < E; ++j) dz[j] = t[j] > 0.f ? dz[j] : dz[j] * p.slope;
  }
  if (p.dres) {
    float o[E];
    if (p.dr_acc) {
      Vec16<Tag>::unpack(ops.dres, o);
#pragma unroll
      for (int j = 0; j < E; ++j) o[j] += dz[j];
    } else {
#pragma unroll
      for (int j = 0; j < E; ++j) o[j] = dz[j];
    }
    *(u32x4*)((T*)p.dres + m * p.dr_cs + p.dr_coff + g * E) = Vec16<Tag>::pack(o);
  }
  float o[E];
  if (p.x) {
    float xv[E];
    Vec16<Tag>::unpack(ops.x, xv);
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const int c = g * CoefPitch<Tag>::PS + j, AS = p.G * CoefPitch<Tag>::PS;
      o[j] = s_cf[c] * (dz[j] - s_cf[AS + c] - (xv[j] - s_cf[3 * AS + c]) * s_cf[2 * AS + c]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < E; ++j) o[j] = dz[j];
  }
  *(u32x4*)((T*)p.dx + m * p.dx_cs + p.dx_coff + g * E) = Vec16<Tag>::pack(o);
}

template <typename Tag>
__device__ __forceinline__ void bn_bwd_piece(const BnBwdParams& p, const float* s_cf, size_t i) {
  BnBwdOperands<Tag> o;
  bn_bwd_fetch<Tag>(p, i, o);
  bn_bwd_finish<Tag>(p, s_cf, i, o);
}

template <typename Tag>
__device__ __forceinline__ void bn_bwd_block(const BnBwdApplyParams& pp, float* s_cf, unsigned bid) {
  const size_t i = (size_t)bid * 256 + threadIdx.x;
  BnBwdOperands<Tag> o;
  if (i < pp.q.total) bn_bwd_fetch<Tag>(pp.q, i, o);        // all loads (dx / dres may alias dy: in place) before any store of this thread
  bn_bwd_coef<Tag>(pp, s_cf, bid == 0);
  __syncthreads();
  if (i < pp.q.total) bn_bwd_finish<Tag>(pp.q, s_cf, i, o);
}

template <typename Tag>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApplyParams pp) {
  extern __shared__ float s_cf[];
  bn_bwd_block<Tag>(pp, s_cf, blockIdx.x);
}

// backward twin of bn_fused_fwd_kernel: sums of dz and dz * xhat over this block's rows -> barrier -> dx in place over the same rows
template <typename Tag>
__global__ __launch_bounds__(256) void bn_fused_bwd_kernel(const ColsumParams cs, const BnBwdApplyParams pp, uint32_t* counter) {
  __shared__ double red[256 * 2 * Tag::E];
  extern __shared__ float s_cf[];
  colsum2_body<Tag, 1>(cs, red, blockIdx.x);
  cp_grid_barrier(counter, gridDim.x);
  bn_bwd_coef<Tag>(pp, s_cf, blockIdx.x == 0);
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * cs.rpb * pp.q.G;
  size_t hi = lo + (size_t)cs.rpb * pp.q.G;
  hi = hi < pp.q.total ? hi : pp.q.total;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) bn_bwd_piece<Tag>(pp.q, s_cf, i);
}

static int build_bn_bwd_apply(int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride, int y_coff, const void* x,
                              int x_cstride, int x_coff, const float* mean, const float* rstd, const float* gamma, const double* acc, int M,
                              int C, int act, float slope, void* dx, int dx_cstride, int dx_coff, void* dres, int dres_cstride,
                              int dres_coff, int dres_accumulate, float* dgamma, float* dbeta, BnBwdApplyParams* out, unsigned* blocks,
                              int* Cphys_out) {
  if (!acc || M <= 0 || C <= 0 || (x && (!mean || !rstd))) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E;
  int rc;
  if ((rc = check_cl(dtype, dy, dy_cstride, dy_coff, Cphys)) || (rc = check_cl(dtype, dx, dx_cstride, dx_coff, Cphys))) return rc;
  const void* yy = act == CP_ACT_NONE ? nullptr : y;
  if (act != CP_ACT_NONE && (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (x && (rc = check_cl(dtype, x, x_cstride, x_coff, Cphys))) return rc;
  if (dres && (rc = check_cl(dtype, dres, dres_cstride, dres_coff, Cphys))) return rc;
  BnBwdApplyParams pp;
  BnBwdParams& q = pp.q;
  q.dy = dy; q.dy_cs = dy_cstride; q.dy_coff = dy_coff; q.y = yy; q.y_cs = y_cstride; q.y_coff = y_coff;
  q.x = x; q.x_cs = x_cstride; q.x_coff = x_coff; q.dx = dx; q.dx_cs = dx_cstride; q.dx_coff = dx_coff;
  q.dres = dres; q.dr_cs = dres_cstride; q.dr_coff = dres_coff; q.dr_acc = dres_accumulate;
  q.coef = nullptr; q.Cvec = (C + 15) / 16 * 16; q.G = Cphys / E; q.slope = act == CP_ACT_RELU ? 0.f : slope; q.total = (size_t)M * q.G;
  pp.acc = acc; pp.C = C; pp.sets = bn_acc_sets(); pp.count = (double)M; pp.gamma = gamma; pp.mean = mean; pp.rstd = rstd; pp.dgamma = dgamma; pp.dbeta = dbeta;
  *out = pp;
  *blocks = (unsigned)((q.total + 255) / 256);
  *Cphys_out = Cphys;
  return CP_OK;
}

extern "C" int cp_bn_bwd_apply(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y,
                               int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* mean,
                               const float* rstd, const float* gamma, const double* acc, int M, int C, int act, float slope,
                               void* dx, int dx_cstride, int dx_coff, void* dres, int dres_cstride, int dres_coff,
                               int dres_accumulate, float* dgamma, float* dbeta) {
  BnBwdApplyParams pp;
  unsigned blocks;
  int Cphys;
  const int rc = build_bn_bwd_apply(dtype, dy, dy_cstride, dy_coff, y, y_cstride, y_coff, x, x_cstride, x_coff, mean, rstd, gamma, acc, M, C,
                                    act, slope, dx, dx_cstride, dx_coff, dres, dres_cstride, dres_coff, dres_accumulate, dgamma, dbeta,
                                    &pp, &blocks, &Cphys);
  if (rc) return rc;
  const size_t lds = 4 * coef_floats(dtype, Cphys) * sizeof(float);
  if (dtype == CP_F32) CP_LAUNCH(bn_bwd_apply_kernel<F32Tag>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, pp);
  else CP_LAUNCH(bn_bwd_apply_kernel<BF16Tag>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, pp);
  return cp_check_launch();
}

// ---- grouped launches: the BatchNorm passes of INDEPENDENT layers (the 2-4 branches of an HRNet module at the same depth, the fuse-layer
// convs of a module) in ONE launch each.  At the training batch (32 crops) a single pass is a 5-13 us launch over 0.3-5 MB -- latency,
// not bandwidth; the step ran ~1 340 of them.  An item is the parameter block of one layer's pass, built on the host by cp_bn_item_*
// (same arguments and checks as the single-layer entry points), the table lives in device memory, block b belongs to the item k with
// prefix[k] <= b < prefix[k + 1].
union BnItemU {
  ColsumParams cs;
  BnApplyParams ap;
  BnBwdApplyParams bp;
  unsigned long long pad[CP_BN_ITEM_BYTES / 8];
};
static_assert(sizeof(BnItemU) == CP_BN_ITEM_BYTES, "CpBnItem too small for the parameter blocks");

template <typename Tag, int KIND>
__global__ __launch_bounds__(256) void bn_group_kernel(const CpBnItem* __restrict__ items, const uint32_t* __restrict__ prefix, int n) {
  int k = 0;
  while (k + 1 < n && blockIdx.x >= prefix[k + 1]) ++k;
  const unsigned bid = blockIdx.x - prefix[k];
  const BnItemU* it = (const BnItemU*)items[k].params;
  if constexpr (KIND == CP_BN_ITEM_STATS || KIND == CP_BN_ITEM_BWD_SUMS) {
    __shared__ double red[256 * 2 * Tag::E];
    const ColsumParams p = it->cs;                 // block-uniform copy (SGPRs)
    colsum2_body<Tag, KIND == CP_BN_ITEM_STATS ? 0 : 1>(p, red, bid);
  } else if constexpr (KIND == CP_BN_ITEM_APPLY) {
    extern __shared__ float s_coef[];
    const BnApplyParams p = it->ap;
    bn_apply_block<Tag>(p, s_coef, bid);
  } else {
    extern __shared__ float s_coef[];
    const BnBwdApplyParams p = it->bp;
    bn_bwd_block<Tag>(p, s_coef, bid);
  }
}

extern "C" int cp_bn_item_stats(int dtype, const void* x, int M, int C, int x_cstride, int x_coff, double* acc, CpBnItem* item) {
  if (!item) return CP_ERR_INVALID;
  BnItemU u = {};
  int nblk;
  const int rc = build_bn_stats(dtype, x, M, C, x_cstride, x_coff, acc, &u.cs, &nblk);
  if (rc) return rc;
  memcpy(item->params, &u, sizeof(u));
  item->kind = CP_BN_ITEM_STATS; item->dtype = dtype; item->blocks = (uint32_t)nblk; item->lds_bytes = 0;
  return CP_OK;
}

extern "C" int cp_bn_item_apply(int dtype, const void* x, int x_cstride, int x_coff, const double* acc, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, const void* res, int res_cstride,
                                int res_coff, void* y, int y_cstride, int y_coff, int M, int C, int act, float slope, float* mean,
                                float* rstd, CpBnItem* item) {
  if (!item) return CP_ERR_INVALID;
  BnItemU u = {};
  unsigned blocks;
  int Cphys;
  const int rc = build_bn_apply(dtype, x, x_cstride, x_coff, acc, gamma, beta, running_mean, running_var, momentum, eps, res, res_cstride,
                                res_coff, y, y_cstride, y_coff, M, C, act, slope, mean, rstd, &u.ap, &blocks, &Cphys);
  if (rc) return rc;
  memcpy(item->params, &u, sizeof(u));
  item->kind = CP_BN_ITEM_APPLY; item->dtype = dtype; item->blocks = blocks; item->lds_bytes = (uint32_t)(2 * coef_floats(dtype, Cphys) * sizeof(float));
  return CP_OK;
}

extern "C" int cp_bn_item_bwd_sums(int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride, int y_coff,
                                   const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd, int M, int C, int act,
                                   float slope, double* acc, CpBnItem* item) {
  if (!item) return CP_ERR_INVALID;
  BnItemU u = {};
  int nblk;
  const int rc = build_bn_bwd_sums(dtype, dy, dy_cstride, dy_coff, y, y_cstride, y_coff, x, x_cstride, x_coff, mean, rstd, M, C, act, slope,
                                   acc, &u.cs, &nblk);
  if (rc) return rc;
  memcpy(item->params, &u, sizeof(u));
  item->kind = CP_BN_ITEM_BWD_SUMS; item->dtype = dtype; item->blocks = (uint32_t)nblk; item->lds_bytes = 0;
  return CP_OK;
}

extern "C" int cp_bn_item_bwd_apply(int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride, int y_coff,
                                    const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd, const float* gamma,
                                    const double* acc, int M, int C, int act, float slope, void* dx, int dx_cstride, int dx_coff,
                                    void* dres, int dres_cstride, int dres_coff, int dres_accumulate, float* dgamma, float* dbeta,
                                    CpBnItem* item) {
  if (!item) return CP_ERR_INVALID;
  BnItemU u = {};
  unsigned blocks;
  int Cphys;
  const int rc = build_bn_bwd_apply(dtype, dy, dy_cstride, dy_coff, y, y_cstride, y_coff, x, x_cstride, x_coff, mean, rstd, gamma, acc, M, C,
                                    act, slope, dx, dx_cstride, dx_coff, dres, dres_cstride, dres_coff, dres_accumulate, dgamma, dbeta,
                                    &u.bp, &blocks, &Cphys);
  if (rc) return rc;
  memcpy(item->params, &u, sizeof(u));
  item->kind = CP_BN_ITEM_BWD_APPLY; item->dtype = dtype; item->blocks = blocks; item->lds_bytes = (uint32_t)(4 * coef_floats(dtype, Cphys) * sizeof(float));
  return CP_OK;
}

template <typename Tag>
static void launch_bn_group(int kind, unsigned blocks, size_t lds, hipStream_t st, const CpBnItem* items, const uint32_t* prefix, int n) {
  switch (kind) {
    case CP_BN_ITEM_STATS: CP_LAUNCH((bn_group_kernel<Tag, CP_BN_ITEM_STATS>), dim3(blocks), dim3(256), 0, st, items, prefix, n); break;
    case CP_BN_ITEM_APPLY: CP_LAUNCH((bn_group_kernel<Tag, CP_BN_ITEM_APPLY>), dim3(blocks), dim3(256), lds, st, items, prefix, n); break;
    case CP_BN_ITEM_BWD_SUMS: CP_LAUNCH((bn_group_kernel<Tag, CP_BN_ITEM_BWD_SUMS>), dim3(blocks), dim3(256), 0, st, items, prefix, n); break;
    default: CP_LAUNCH((bn_group_kernel<Tag, CP_BN_ITEM_BWD_APPLY>), dim3(blocks), dim3(256), lds, st, items, prefix, n); break;
  }
}

extern "C" int cp_bn_group(cp_stream_t stream, int dtype, int kind, const CpBnItem* items_dev, const uint32_t* prefix_dev, int n_items,
                           uint32_t total_blocks, uint32_t lds_bytes) {
  if (!items_dev || !prefix_dev || n_items <= 0 || n_items > CP_BN_GROUP_MAX || total_blocks == 0 || lds_bytes > 48 * 1024)
    return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (kind < CP_BN_ITEM_STATS || kind > CP_BN_ITEM_BWD_APPLY) return CP_ERR_INVALID;
  if (!cp_aligned16(items_dev)) return CP_ERR_ALIGN;
  if (dtype == CP_F32) launch_bn_group<F32Tag>(kind, total_blocks, lds_bytes, (hipStream_t)stream, items_dev, prefix_dev, n_items);
  else launch_bn_group<BF16Tag>(kind, total_blocks, lds_bytes, (hipStream_t)stream, items_dev, prefix_dev, n_items);
  return cp_check_launch();
}


#ifndef CP_BAR_BLOCKS
#define CP_BAR_BLOCKS 256
#endif
// fewer, fatter blocks than the two-launch forms: every block polls the barrier counter, and hundreds of pollers on one L2 word
// slow the arrivals down
static int colsum_plan_fused(int M, int Cphys, int E, int* G, int* RL, int* nblk, int* rpb) {
  *G = Cphys / E;
  if (*G > 256) return CP_ERR_INVALID;
  *RL = 256 / *G;
  int nb = M / 64;
  nb = nb < 1 ? 1 : (nb > CP_BAR_BLOCKS ? CP_BAR_BLOCKS : nb);
  *rpb = (M + nb - 1) / nb;
  *nblk = (M + *rpb - 1) / *rpb;
  return CP_OK;
}

// ---- fused entry points: the statistics pass and the apply pass of a train-mode BatchNorm in ONE launch (grid barrier between them).
// `acc` as above (zeroed by the caller once per step); `counter`: 4 zeroed bytes of the same arena, one per call.
extern "C" int cp_bn_train_fused(cp_stream_t stream, int dtype, const void* x, int x_cstride, int x_coff, double* acc, uint32_t* counter,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                 float eps, const void* res, int res_cstride, int res_coff, void* y, int y_cstride, int y_coff, int M,
                                 int C, int act, float slope, float* mean, float* rstd) {
  if (!acc || !counter || !mean || !rstd || M <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (cp_deterministic()) return CP_ERR_INVALID;      // the one-launch forms add many blocks per set: use cp_bn_stats_accumulate + cp_bn_apply
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E, Cvec = (C + 15) / 16 * 16;
  int rc;
  if ((rc = check_cl(dtype, x, x_cstride, x_coff, Cphys)) || (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (res && (rc = check_cl(dtype, res, res_cstride, res_coff, Cphys))) return rc;
  ColsumParams cs = {};
  int nblk;
  if ((rc = colsum_plan_fused(M, Cphys, E, &cs.G, &cs.RL, &nblk, &cs.rpb))) return rc;
  cs.a = x; cs.a_cs = x_cstride; cs.a_coff = x_coff; cs.M = M; cs.acc = acc; cs.acc_stride = Cvec; cs.sets = CP_BN_ACC_SETS;
  BnApplyParams p;
  p.sets = CP_BN_ACC_SETS;
  p.x = x; p.x_cs = x_cstride; p.x_coff = x_coff; p.res = res; p.r_cs = res_cstride; p.r_coff = res_coff;
  p.y = y; p.y_cs = y_cstride; p.y_coff = y_coff; p.acc = acc; p.Cvec = Cvec; p.C = C; p.count = (double)M;
  p.gamma = gamma; p.beta = beta; p.eps = eps; p.momentum = momentum; p.rmean = running_mean; p.rvar = running_var;
  p.mean = mean; p.rstd = rstd; p.G = Cphys / E; p.act = act; p.slope = slope; p.total = (size_t)M * p.G;
  const size_t lds = 2 * coef_floats(dtype, Cphys) * sizeof(float);
  if (dtype == CP_F32) CP_LAUNCH(bn_fused_fwd_kernel<F32Tag>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, cs, p, counter);
  else CP_LAUNCH(bn_fused_fwd_kernel<BF16Tag>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, cs, p, counter);
  return cp_check_launch();
}

extern "C" int cp_bn_bwd_fused(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride,
                               int y_coff, const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd,
                               const float* gamma, double* acc, uint32_t* counter, int M, int C, int act, float slope, void* dx,
                               int dx_cstride, int dx_coff, void* dres, int dres_cstride, int dres_coff, int dres_accumulate,
                               float* dgamma, float* dbeta) {
  if (!acc || !counter || M <= 0 || C <= 0 || (x && (!mean || !rstd))) return CP_ERR_INVALID;
  if ((dtype != CP_F32 && dtype != CP_BF16) || cp_deterministic()) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype), Cphys = (C + E - 1) / E * E, Cvec = (C + 15) / 16 * 16;
  int rc;
  if ((rc = check_cl(dtype, dy, dy_cstride, dy_coff, Cphys)) || (rc = check_cl(dtype, dx, dx_cstride, dx_coff, Cphys))) return rc;
  const void* yy = act == CP_ACT_NONE ? nullptr : y;
  if (act != CP_ACT_NONE && (rc = check_cl(dtype, y, y_cstride, y_coff, Cphys))) return rc;
  if (x && (rc = check_cl(dtype, x, x_cstride, x_coff, Cphys))) return rc;
  if (dres && (rc = check_cl(dtype, dres, dres_cstride, dres_coff, Cphys))) return rc;
  ColsumParams cs = {};
  int nblk;
  if ((rc = colsum_plan_fused(M, Cphys, E, &cs.G, &cs.RL, &nblk, &cs.rpb))) return rc;
  cs.a = dy; cs.a_cs = dy_cstride; cs.a_coff = dy_coff; cs.y = yy; cs.y_cs = y_cstride; cs.y_coff = y_coff;
  cs.x = x; cs.x_cs = x_cstride; cs.x_coff = x_coff; cs.mean = mean; cs.rstd = rstd; cs.slope = act == CP_ACT_RELU ? 0.f : slope; cs.act = act;
  cs.M = M; cs.acc = acc; cs.acc_stride = Cvec; cs.sets = CP_BN_ACC_SETS;
  BnBwdApplyParams pp;
  BnBwdParams& q = pp.q;
  q.dy = dy; q.dy_cs = dy_cstride; q.dy_coff = dy_coff; q.y = yy; q.y_cs = y_cstride; q.y_coff = y_coff;
  q.x = x; q.x_cs = x_cstride; q.x_coff = x_coff; q.dx = dx; q.dx_cs = dx_cstride; q.dx_coff = dx_coff;
  q.dres = dres; q.dr_cs = dres_cstride; q.dr_coff = dres_coff; q.dr_acc = dres_accumulate;
  q.coef = nullptr; q.Cvec = Cvec; q.G = Cphys / E; q.slope = act == CP_ACT_RELU ? 0.f : slope; q.total = (size_t)M * q.G;
  pp.acc = acc; pp.C = C; pp.sets = CP_BN_ACC_SETS; pp.count = (double)M; pp.gamma = gamma; pp.mean = mean; pp.rstd = rstd; pp.dgamma = dgamma; pp.dbeta = dbeta;
  const size_t lds = 4 * coef_floats(dtype, Cphys) * sizeof(float);
  if (dtype == CP_F32) CP_LAUNCH(bn_fused_bwd_kernel<F32Tag>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, cs, pp, counter);
  else CP_LAUNCH(bn_fused_bwd_kernel<BF16Tag>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, cs, pp, counter);
  return cp_check_launch();
}
