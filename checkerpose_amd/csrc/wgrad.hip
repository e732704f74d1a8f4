// Weight gradient of a convolution / linear layer on gfx950 MFMA (training side).
//
//   dW[co][ci][r][s] = sum over output pixels m=(b,oy,ox) of  dy[m][co] * x[b, oy*stride-pad+r, ox*stride-pad+s][ci]
//
// GEMM view per tap: D[co][ci] = sum_m DY^T[co][m] * X_tap[m][ci] -- the contraction runs over PIXELS, which are the
// slow (strided) axis of both channels-last operands, so both MFMA operands need a transpose.  bf16: the pixel rows are
// staged in LDS exactly as they lie in memory ([pixel][64 channels], 16-byte pieces, coalesced) and read back with
// gfx950's transposing LDS read `ds_read_b64_tr_b16` (per 16-lane group: 4 pixel rows x 16 channels delivered
// channel-major), two reads per 16x16x32 operand.  The k <-> pixel assignment is free as long as both operands use the
// same one: lane group g takes pixels 4g..4g+3 and 16+4g..16+4g+3 of a 32-pixel sub-chunk, so a 32-lane half reads 8
// consecutive rows, and with a row pitch of 160 bytes (= 5 * 32: odd multiple of 8 banks) those 8 rows fall on disjoint
// bank octets -> conflict-free.  fp32: v_mfma_f32_16x16x4_f32 takes ONE k per lane, so lane (x,q) reads channel x of
// pixel 4j+q with a plain ds_read_b32 (pitch 320 B: the two rows of a half land 16 banks apart).
//
// Block = 64 (co) x 64 (ci) x one tap x one pixel slice; 4 waves own 32x32 quadrants (2x2 tiles).  Stages of 64 pixels
// go global -> registers -> LDS (single buffer, the next stage's global loads are in flight during the MFMAs).
// Out-of-image taps and channel / pixel tails are zero rows.  Partial sums of the pixel slices are combined with fp32
// hardware atomics into dW (caller zeroes it; addition order across slices is not fixed -> last-bit run-to-run noise).
//
// Replaces: autograd of every nn.Conv2d / nn.ConvTranspose2d / nn.Linear weight on the path (reference
// checkerpose/train.py:319 `loss.backward()`).
#include <stdlib.h>

#include <string.h>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradParams {
  const void* dy; const void* x; float* dw;
  int M, HoWo, Wo, H, W;
  int Cout, dy_cs, dy_coff;       // logical channels of dy, its pixel stride / channel offset (elements)
  int Cin, x_cs, x_coff;
  int R, S, stride, pad;
  int co_blocks, ci_blocks, slice; // pixels per block slice (multiple of 64)
  long long dw_base, dw_sco, dw_sci, dw_sr, dw_ss;
  float* ws;                       // per-block partial tiles [slice][blockIdx.y][64][64] (NULL: fp32 atomics into dw)
};

template <typename Tag> struct WgCfg;
template <> struct WgCfg<BF16Tag> { static constexpr int PITCH = 160, PPR = 8; };    // bytes per LDS row, 16-B pieces per row
template <> struct WgCfg<F32Tag> { static constexpr int PITCH = 320, PPR = 16; };

template <typename Tag>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const unsigned bx, const unsigned by, const unsigned gy) {
  constexpr int E = Tag::E;
  constexpr int PITCH = WgCfg<Tag>::PITCH, PPR = WgCfg<Tag>::PPR;
  constexpr int NLD = 64 * PPR / 256;           // 16-byte pieces per thread per operand per stage (2 bf16 / 4 f32)
  constexpr int ES = 16 / E;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 64 * PITCH];
  unsigned char* ldy = lds;
  unsigned char* lx = lds + 64 * PITCH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xl = lane & 15, g = lane >> 4;
  int t = (int)by;
  const int cob = t % p.co_blocks; t /= p.co_blocks;
  const int cib = t % p.ci_blocks; t /= p.ci_blocks;
  const int r = t / p.S, s = t - r * p.S;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int m_begin = (int)bx * p.slice;
  const int m_end = min(m_begin + p.slice, p.M);

  // this thread's pieces: row = i*(256/PPR) + tid/PPR, piece = tid % PPR
  const int prow = tid / PPR, pc = tid - prow * PPR;
  const bool dy_cok = co0 + pc * E < p.Cout;     // piece inside the logical channels (tails are whole zero pieces or
  const bool x_cok = ci0 + pc * E < p.Cin;       // hold zero-padded physical channels)

  u32x4 rdy[NLD], rx[NLD];
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int m = m0 + i * (256 / PPR) + prow;
      rdy[i] = u32x4{0u, 0u, 0u, 0u};
      rx[i] = u32x4{0u, 0u, 0u, 0u};
      if (m < m_end) {
        const int b = m / p.HoWo;
        const int rem = m - b * p.HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        if (dy_cok)
          rdy[i] = *(const u32x4*)((const unsigned char*)p.dy + ((size_t)m * p.dy_cs + p.dy_coff + co0 + pc * E) * ES);
        const int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s;
        if (x_cok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          rx[i] = *(const u32x4*)((const unsigned char*)p.x +
                                  (((size_t)(b * p.H + iy) * p.W + ix) * p.x_cs + p.x_coff + ci0 + pc * E) * ES);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int row = i * (256 / PPR) + prow;
      *(u32x4*)(ldy + row * PITCH + pc * 16) = rdy[i];
      *(u32x4*)(lx + row * PITCH + pc * 16) = rx[i];
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;   // quadrant of the 64x64 block tile

  gload(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    __syncthreads();                 // previous stage's LDS reads are done
    lstore();
    __syncthreads();
    gload(m0 + 64);                  // next stage in flight under the MFMAs (rows >= m_end load nothing)
    if constexpr (E == 8) {
      // lane 4q+pp of group g supplies row (4g+q [+16]) , channels 4pp..4pp+3 of the 16-channel tile
      const int q = xl >> 2, pp = xl & 3;
#pragma unroll
      for (int sc = 0; sc < 2; ++sc) {            // two 32-pixel sub-chunks per stage
        const int row_lo = sc * 32 + 4 * g + q;
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const unsigned char* base = ldy + row_lo * PITCH + (wco + a * 16 + 4 * pp) * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 16 * PITCH));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          fa[a] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const unsigned char* base = lx + row_lo * PITCH + (wci + b * 16 + 4 * pp) * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 16 * PITCH));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          fb[b] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int j = 0; j < 16; ++j) {              // 16 MFMA k-steps of 4 pixels
        const int row = 4 * j + g;
        float fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = *(const float*)(ldy + row * PITCH + (wco + a * 16 + xl) * 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = *(const float*)(lx + row * PITCH + (wci + b * 16 + xl) * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
  }

  // D[co][ci]: lane holds rows co = 4g+j (j = 0..3), column ci = xl of every tile
  if (p.ws) {        // plain stores of the block's partial tile; wgrad_reduce_kernel sums the slices (no atomics)
    float* tile = p.ws + ((size_t)bx * gy + by) * 4096;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j)      // only the layer's own (co, ci): the reduction reads nothing else, and an 18 x 18 layer
          if (co0 + wco + a * 16 + 4 * g + j < p.Cout && ci0 + wci + b * 16 + xl < p.Cin)      // would write 92 % padding
            tile[(wco + a * 16 + 4 * g + j) * 64 + wci + b * 16 + xl] = acc[a][b][j];
    return;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ci = ci0 + wci + b * 16 + xl;
      if (ci >= p.Cin) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + wco + a * 16 + 4 * g + j;
        if (co >= p.Cout) continue;
        float* dst = p.dw + p.dw_base + (long long)co * p.dw_sco + (long long)ci * p.dw_sci + (long long)r * p.dw_sr +
                     (long long)s * p.dw_ss;
        unsafeAtomicAdd(dst, acc[a][b][j]);
      }
    }
}

template <typename Tag>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) { wgrad_body<Tag>(p, blockIdx.x, blockIdx.y, gridDim.y); }

// dw[co][ci][r][s] += sum over the pixel slices of the partial tiles (deterministic order).  Thread = one dw element,
// ci fastest (coalesced reads of the 64-float tile rows).
struct WgradReduceParams {
  const float* ws; float* dw;
  int S, GY, co_blocks, ci_blocks, R, Ssz, Cout, Cin, taps_in_block;   // taps_in_block: 9 (all-taps kernel) or 1
  long long dw_base, dw_sco, dw_sci, dw_sr, dw_ss;
};
// slice lanes per dw element: 16 for layers cut into many slices; the grouped launches cut most layers into 1-8 slices, where 16
// lanes per element left 15 of 16 threads idle and the step's reductions ran 1.3 M blocks of 16 elements at < 1 TB/s
__host__ __device__ __forceinline__ int wgrad_reduce_lanes(int S) { return S >= 16 ? 16 : (S > 8 ? 16 : (S > 4 ? 8 : (S > 2 ? 4 : (S > 1 ? 2 : 1)))); }

__device__ __forceinline__ void wgrad_reduce_block(const WgradReduceParams& p, unsigned block) {
  // block = EPB = 256 / L consecutive dw elements (ci fastest: coalesced reads of the 64-float tile rows) x L slice lanes; the lanes
  // stride over the slices (a serial loop over up to 512 slices per element was latency-bound), then the L partial sums are added
  __shared__ float red[272];
  const int L = wgrad_reduce_lanes(p.S), EPB = 256 / L;
  const int e = threadIdx.x % EPB, sl0 = threadIdx.x / EPB;
  const size_t i = (size_t)block * EPB + e;
  const size_t total = (size_t)p.R * p.Ssz * p.Cout * p.Cin;
  const bool ok = i < total;
  const size_t ii = ok ? i : 0;
  const int ci = (int)(ii % p.Cin);
  size_t t = ii / p.Cin;
  const int co = (int)(t % p.Cout);
  const int tap = (int)(t / p.Cout);
  const int cob = co >> 6, cib = ci >> 6;
  size_t off, stride;
  if (p.taps_in_block == 1) {
    const int y = cob + p.co_blocks * (cib + p.ci_blocks * tap);
    off = (size_t)y * 4096;
    stride = (size_t)p.GY * 4096;
  } else {
    const int y = cob + p.co_blocks * cib;
    off = ((size_t)y * 9 + tap) * 4096;
    stride = (size_t)p.GY * 9 * 4096;
  }
  off += (size_t)(co & 63) * 64 + (ci & 63);
  // 8 slices' loads in flight per lane (one dependent load per iteration made the kernel a chain of memory latencies: 16
  // round trips at 256 slices); the order of the additions is fixed by S alone
  float acc = 0.f;
  if (ok) {
    int sl = sl0;
    for (; sl + 7 * L < p.S; sl += 8 * L) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p.ws[off + (size_t)(sl + L * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sl < p.S; sl += L) acc += p.ws[off + (size_t)sl * stride];
  }
  float* dst = p.dw + p.dw_base + (long long)co * p.dw_sco + (long long)ci * p.dw_sci + (long long)(tap / p.Ssz) * p.dw_sr +
               (long long)(tap % p.Ssz) * p.dw_ss;
  if (L == 1) {                      // block-uniform
    if (ok) *dst += acc;
    return;
  }
  red[sl0 * (EPB + 1) + e] = acc;
  __syncthreads();
  if (sl0 == 0 && ok) {
    float s_ = 0.f;
    for (int k = 0; k < L; ++k) s_ += red[k * (EPB + 1) + e];
    *dst += s_;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceParams p) { wgrad_reduce_block(p, blockIdx.x); }

// every pending reduction of a stretch of the backward in ONE launch (358 launches of ~27 us, each a handful of latency-bound
// blocks, were 9.7 of the step's 35.7 ms): block -> item by binary search in the exclusive prefix sum of the items' block counts
static_assert(sizeof(WgradReduceParams) == sizeof(CpWgradReduceItem), "CpWgradReduceItem is the launch-parameter block of one reduction");
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceParams* __restrict__ items, const uint32_t* __restrict__ prefix, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const WgradReduceParams p = items[lo];
  wgrad_reduce_block(p, blockIdx.x - prefix[lo]);
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 specialisation (bf16): ALL NINE TAPS in one block.  The per-tap kernel above re-reads dy and x
// once per tap (9x the traffic; measured 40 TFLOP/s on the 256 -> 256 decoder convs).  Here a block walks tiles of 64
// output pixels (TH x TW, TH*TW = 64), stages the dy tile and the (TH+2) x (TW+2) x-halo ONCE in LDS and every tap reads
// its operand fragments from the halo at a shifted row (each lane supplies its own row address to the transposing
// read, so a shift costs nothing).  A wave owns a 32 x 32 (co x ci) quadrant for all taps: 36 accumulator tiles
// (144 VGPRs); per 32-pixel sub-chunk 4 + 36 transposed reads feed 36 MFMAs.  Waves whose quadrant lies outside the
// layer's channels (18/36-channel HRNet branches) skip the MFMAs but keep staging.
struct Wgrad3Params {
  const void* dy; const void* x; float* dw;
  int H, W, TH, TW, tiles_x, tiles_img, n_tiles, tiles_per_block;   // H, W: the OUTPUT map (dy); tiles of TH x TW = 64 output pixels
  int Hi, Wi;                                                        // the input map (x): H, W at stride 1, 2H, 2W at stride 2
  int Cout, dy_cs, dy_coff, Cin, x_cs, x_coff;
  int co_blocks, ci_blocks;
  long long dw_base, dw_sco, dw_sci, dw_sr, dw_ss;
  float* ws;                       // [slice][blockIdx.y][tap][64][64]
};

__device__ __forceinline__ void wgrad3x3_body(const Wgrad3Params& p, const unsigned bx, const unsigned by, const unsigned gy) {
  constexpr int PITCH = 160, XROWS = 200, NX = 7;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(64 + XROWS) * PITCH];
  unsigned char* ldy = lds;
  unsigned char* lx = lds + 64 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xl = lane & 15, g = lane >> 4, q = xl >> 2, pp = xl & 3;
  const int cob = (int)by % p.co_blocks, cib = (int)by / p.co_blocks;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
  const bool active = (co0 + wco < p.Cout) && (ci0 + wci < p.Cin);          // wave-uniform
  const int TW2 = p.TW + 2, HR = (p.TH + 2) * TW2;
  const int t_begin = (int)bx * p.tiles_per_block;
  const int t_end = min(t_begin + p.tiles_per_block, p.n_tiles);

  const int prow = tid >> 3, pc = tid & 7;
  const bool dy_cok = co0 + pc * 8 < p.Cout, x_cok = ci0 + pc * 8 < p.Cin;
  u32x4 rdy[2], rx[NX];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) rdy[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NX; ++i) rx[i] = u32x4{0u, 0u, 0u, 0u};
    if (t >= t_end) return;
    const int b = t / p.tiles_img;
    const int rem = t - b * p.tiles_img;
    const int ty0 = (rem / p.tiles_x) * p.TH, tx0 = (rem % p.tiles_x) * p.TW;
    if (dy_cok) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int px = i * 32 + prow;
        const int ty = px / p.TW, tx = px - ty * p.TW;
        rdy[i] = *(const u32x4*)((const uint16_t*)p.dy + ((size_t)(b * p.H + ty0 + ty) * p.W + tx0 + tx) * p.dy_cs + p.dy_coff + co0 + pc * 8);
      }
    }
    if (x_cok) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int row = i * 32 + prow;
        if (row < HR) {
          const int hy = row / TW2, hx = row - hy * TW2;
          const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
          if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
            rx[i] = *(const u32x4*)((const uint16_t*)p.x + ((size_t)(b * p.Hi + iy) * p.Wi + ix) * p.x_cs + p.x_coff + ci0 + pc * 8);
        }
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4*)(ldy + (i * 32 + prow) * PITCH + pc * 16) = rdy[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int row = i * 32 + prow;
      if (row < XROWS) *(u32x4*)(lx + row * PITCH + pc * 16) = rx[i];
    }
  };

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  gload(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    lstore();
    __syncthreads();
    gload(t + 1);
    if (active) {
#pragma unroll
      for (int sc = 0; sc < 2; ++sc) {
        const int p_lo = sc * 32 + 4 * g + q, p_hi = p_lo + 16;
        const int h_lo = (p_lo / p.TW) * TW2 + (p_lo % p.TW), h_hi = (p_hi / p.TW) * TW2 + (p_hi % p.TW);
        bf16x8 fa[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int cb = (wco + a * 16 + 4 * pp) * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ldy + p_lo * PITCH + cb));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ldy + p_hi * PITCH + cb));
          fa[a] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int sh = (tap / 3) * TW2 + (tap % 3);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int cb = (wci + b * 16 + 4 * pp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lx + (h_lo + sh) * PITCH + cb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lx + (h_hi + sh) * PITCH + cb));
            const bf16x8 fb = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[tap][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[tap][a][b], 0, 0, 0);
          }
        }
      }
    }
  }
  if (p.ws) {        // partial tile of this block; quadrants outside the layer's channels are never read back
    if (!active) return;
    float* tile = p.ws + ((size_t)bx * gy + by) * 9 * 4096;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + wco + a * 16 + 4 * g + j < p.Cout && ci0 + wci + b * 16 + xl < p.Cin)
              tile[tap * 4096 + (wco + a * 16 + 4 * g + j) * 64 + wci + b * 16 + xl] = acc[tap][a][b][j];
    return;
  }
  if (!active) return;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ci = ci0 + wci + b * 16 + xl;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + wco + a * 16 + 4 * g + j;
          if (co >= p.Cout) continue;
          unsafeAtomicAdd(p.dw + p.dw_base + (long long)co * p.dw_sco + (long long)ci * p.dw_sci + (long long)(tap / 3) * p.dw_sr +
                              (long long)(tap % 3) * p.dw_ss, acc[tap][a][b][j]);
        }
      }
}

__global__ __launch_bounds__(256) void wgrad3x3_kernel(const Wgrad3Params p) { wgrad3x3_body(p, blockIdx.x, blockIdx.y, gridDim.y); }

// <= 32 x 32 channel layers (the 18-channel HRNet branch): in the kernel above only ONE wave has a non-empty quadrant.  Here
// the four waves split the nine TAPS of the single 32 x 32 quadrant (wave w: taps w, w+4, w+8), rows are 32 channels wide
// (pitch 96 B: 8 consecutive rows still fall on disjoint bank octets), 12 accumulator tiles per wave -> ~100 VGPRs and
// 25 KB of LDS, so several blocks share a CU and hide each other's global -> LDS latency.
// STRIDE 2 (round 3: the 18 -> 18 down-sampling convs of the fuse layers): the x halo of a tile is (2 TH + 1) x (2 TW + 1) input pixels
// (TW <= 16: <= 297 rows), an output pixel's operand row for tap (r, s) is halo row (2 y + r, 2 x + s) -- still a per-lane row address.
template <int STRIDE>
__device__ __forceinline__ void wgrad3x3_small_body(const Wgrad3Params& p, const unsigned bx) {
  constexpr int PITCH = 96, XROWS = STRIDE == 1 ? 200 : 304, NX = STRIDE == 1 ? 4 : 5;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(64 + XROWS) * PITCH];
  unsigned char* ldy = lds;
  unsigned char* lx = lds + 64 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xl = lane & 15, g = lane >> 4, q = xl >> 2, pp = xl & 3;
  const int TW2 = STRIDE * (p.TW - 1) + 3, HR = (STRIDE * (p.TH - 1) + 3) * TW2;
  const int t_begin = (int)bx * p.tiles_per_block;
  const int t_end = min(t_begin + p.tiles_per_block, p.n_tiles);
  const int prow = tid >> 2, pc = tid & 3;              // 4 pieces (32 channels) per row
  const bool dy_cok = pc * 8 < p.Cout, x_cok = pc * 8 < p.Cin;
  u32x4 rdy, rx[NX];
  auto gload = [&](int t) {
    rdy = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NX; ++i) rx[i] = u32x4{0u, 0u, 0u, 0u};
    if (t >= t_end) return;
    const int b = t / p.tiles_img;
    const int rem = t - b * p.tiles_img;
    const int ty0 = (rem / p.tiles_x) * p.TH, tx0 = (rem % p.tiles_x) * p.TW;
    if (dy_cok) {
      const int ty = prow / p.TW, tx = prow - ty * p.TW;
      rdy = *(const u32x4*)((const uint16_t*)p.dy + ((size_t)(b * p.H + ty0 + ty) * p.W + tx0 + tx) * p.dy_cs + p.dy_coff + pc * 8);
    }
    if (x_cok) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int row = i * 64 + prow;
        if (row < HR) {
          const int hy = row / TW2, hx = row - hy * TW2;
          const int iy = STRIDE * ty0 - 1 + hy, ix = STRIDE * tx0 - 1 + hx;
          if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
            rx[i] = *(const u32x4*)((const uint16_t*)p.x + ((size_t)(b * p.Hi + iy) * p.Wi + ix) * p.x_cs + p.x_coff + pc * 8);
        }
      }
    }
  };
  auto lstore = [&]() {
    *(u32x4*)(ldy + prow * PITCH + pc * 16) = rdy;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int row = i * 64 + prow;
      if (row < XROWS) *(u32x4*)(lx + row * PITCH + pc * 16) = rx[i];
    }
  };
  f32x4 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  gload(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    lstore();
    __syncthreads();
    gload(t + 1);
#pragma unroll
    for (int sc = 0; sc < 2; ++sc) {
      const int p_lo = sc * 32 + 4 * g + q, p_hi = p_lo + 16;
      const int h_lo = STRIDE * ((p_lo / p.TW) * TW2 + (p_lo % p.TW)), h_hi = STRIDE * ((p_hi / p.TW) * TW2 + (p_hi % p.TW));
      bf16x8 fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int cb = (a * 16 + 4 * pp) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ldy + p_lo * PITCH + cb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ldy + p_hi * PITCH + cb));
        fa[a] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int tap = wave + 4 * i;                  // wave-uniform
        if (tap < 9) {
          const int sh = (tap / 3) * TW2 + (tap % 3);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int cb = (b * 16 + 4 * pp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lx + (h_lo + sh) * PITCH + cb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lx + (h_hi + sh) * PITCH + cb));
            const bf16x8 fb = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[i][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[i][a][b], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tap = wave + 4 * i;
    if (tap >= 9) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ci = b * 16 + xl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = a * 16 + 4 * g + j;
          if (p.ws) { if (co < p.Cout && ci < p.Cin) p.ws[((size_t)bx * 9 + tap) * 4096 + co * 64 + ci] = acc[i][a][b][j]; }
          else if (co < p.Cout && ci < p.Cin)
            unsafeAtomicAdd(p.dw + p.dw_base + (long long)co * p.dw_sco + (long long)ci * p.dw_sci + (long long)(tap / 3) * p.dw_sr +
                                (long long)(tap % 3) * p.dw_ss, acc[i][a][b][j]);
        }
      }
  }
}

__global__ __launch_bounds__(256, 2) void wgrad3x3_small_kernel(const Wgrad3Params p) { wgrad3x3_small_body<1>(p, blockIdx.x); }
__global__ __launch_bounds__(256, 2) void wgrad3x3_s2_small_kernel(const Wgrad3Params p) { wgrad3x3_small_body<2>(p, blockIdx.x); }

// mode 0: partial kernel + its reduction (cp_conv2d_wgrad_ws); 1: partial kernel only, *item = the reduction still owed
// (cp_conv2d_wgrad_deferred); 2: no launch at all, *item as mode 1 would fill it (cp_conv2d_wgrad_plan)
static int launch_reduce(hipStream_t st, const float* ws, float* dw, int S, int GY, int co_blocks, int ci_blocks, const CpWgradDesc* d,
                         int taps_in_block, int mode, CpWgradReduceItem* item) {
  WgradReduceParams r;
  r.ws = ws; r.dw = dw; r.S = S; r.GY = GY; r.co_blocks = co_blocks; r.ci_blocks = ci_blocks; r.R = d->R; r.Ssz = d->S;
  r.Cout = d->Cout; r.Cin = d->Cin; r.taps_in_block = taps_in_block;
  r.dw_base = d->dw_base; r.dw_sco = d->dw_sco; r.dw_sci = d->dw_sci; r.dw_sr = d->dw_sr; r.dw_ss = d->dw_ss;
  const size_t total = (size_t)d->R * d->S * d->Cout * d->Cin;
  if (mode != 0) {
    memcpy(item, &r, sizeof(r));
    return CP_OK;
  }
  const size_t epb = 256 / wgrad_reduce_lanes(S);
  CP_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((total + epb - 1) / epb)), dim3(256), 0, st, r);
  return cp_check_launch();
}

static int wgrad_impl(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace,
                      size_t workspace_bytes, int mode, CpWgradReduceItem* item, int target_blocks = 0, CpWgradItem* citem = nullptr);

extern "C" int cp_conv2d_wgrad(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw) {
  return cp_conv2d_wgrad_ws(stream, d, dy, x, dw, nullptr, 0);
}

extern "C" int cp_conv2d_wgrad_ws(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw,
                                  void* workspace, size_t workspace_bytes) {
  return wgrad_impl(stream, d, dy, x, dw, workspace, workspace_bytes, 0, nullptr);
}

extern "C" int cp_conv2d_wgrad_deferred(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw,
                                        void* workspace, size_t workspace_bytes, CpWgradReduceItem* item) {
  if (!item) return CP_ERR_INVALID;
  memset(item, 0, sizeof(*item));
  return wgrad_impl(stream, d, dy, x, dw, workspace, workspace_bytes, 1, item);
}

extern "C" int cp_conv2d_wgrad_plan(const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace,
                                    size_t workspace_bytes, CpWgradReduceItem* item) {
  if (!item) return CP_ERR_INVALID;
  memset(item, 0, sizeof(*item));
  return wgrad_impl(nullptr, d, dy, x, dw, workspace, workspace_bytes, 2, item);
}

// 3x3 / pad 1 layers in bf16 on power-of-two OUTPUT maps, stride 1 (same size) or 2 (half size): the all-taps kernels.  TW x TH = 64
// output pixels per tile; stride 2 keeps TW <= 16 so that the (2 TH + 1) x (2 TW + 1) input halo fits the LDS rows.
static bool wgrad_all_taps(const CpWgradDesc* d, int* TW_out, int* TH_out) {
  if (d->dtype != CP_BF16 || d->R != 3 || d->S != 3 || d->pad != 1 || cp_knob("CP_WGRAD_GENERIC")) return false;
  const bool s1 = d->stride == 1 && d->Ho == d->H && d->Wo == d->W;
  // stride 2: only the <= 32 x 32-channel layers -- measured on the training step, the one-block-per-CU kernel on the wider stride-2
  // layers (18 -> 36 .. 72 -> 144, 3 -> 64: mostly padding in 64 x 64 channel tiles) lost 0.2 ms against the per-tap kernel, whose
  // re-reads the L2 absorbs; that variant was removed again
  const bool s2 = d->stride == 2 && d->H == 2 * d->Ho && d->W == 2 * d->Wo && d->Cout <= 32 && d->Cin <= 32 && !cp_knob("CP_WGRAD_NO_S2");
  if (!s1 && !s2) return false;
  if (d->Wo < 8 || (d->Wo & (d->Wo - 1))) return false;
  const int cap = s1 ? 64 : 16;
  const int TW = d->Wo < cap ? d->Wo : cap, TH = 64 / TW;
  if (d->Ho % TH) return false;
  if (TW_out) { *TW_out = TW; *TH_out = TH; }
  return true;
}

extern "C" size_t cp_conv2d_wgrad_scratch_bytes(const CpWgradDesc* d) {
  if (!d || d->Cout <= 0 || d->Cin <= 0 || d->R <= 0 || d->S <= 0) return 0;
  const size_t cob = (d->Cout + 63) / 64, cib = (d->Cin + 63) / 64;
  // the slice counts the launchers aim for (512 / 256 / 1024 blocks over the tile grid) times one slice of 64 x 64 fp32 tiles
  const bool taps9 = wgrad_all_taps(d, nullptr, nullptr);
  if (taps9) {
    const size_t tb = cob * cib, per_slice = tb * 9 * 4096 * sizeof(float);
    const size_t S = (d->Cout <= 32 && d->Cin <= 32) ? 512 : (256 / tb ? 256 / tb : 1);
    return S * per_slice;
  }
  const size_t tiles = cob * cib * d->R * d->S, per_slice = tiles * 4096 * sizeof(float);
  return ((1024 + tiles - 1) / tiles) * per_slice;
}

extern "C" int cp_wgrad_reduce_batch(cp_stream_t stream, const CpWgradReduceItem* items_dev, const uint32_t* block_prefix_dev, int n_items,
                                     uint32_t total_blocks) {
  if (n_items < 0) return CP_ERR_INVALID;
  if (n_items == 0 || total_blocks == 0) return CP_OK;
  if (!items_dev || !block_prefix_dev) return CP_ERR_INVALID;
  CP_LAUNCH(wgrad_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const WgradReduceParams*)items_dev, block_prefix_dev, n_items);
  return cp_check_launch();
}

extern "C" uint32_t cp_wgrad_reduce_item_blocks(const CpWgradReduceItem* item) {
  if (!item || !item->ws) return 0;
  const size_t epb = 256 / wgrad_reduce_lanes(item->S);
  return (uint32_t)(((size_t)item->R * item->Ssz * item->Cout * item->Cin + epb - 1) / epb);
}

// mode 3: nothing is launched; *citem describes the partial-sum launch (for cp_wgrad_group), *item the reduction it owes.
// target_blocks > 0: the pixel slices are cut for about that many workgroups instead of a whole GPU's worth (a grouped launch fills
// the GPU with SEVERAL layers: fewer, longer slices per layer -> less partial-sum traffic, the fixed cost of a block amortised)
static int wgrad_impl(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace,
                      size_t workspace_bytes, int mode, CpWgradReduceItem* item, int target_blocks, CpWgradItem* citem) {
  if (!d || !dy || !x || !dw) return CP_ERR_INVALID;
  const bool launch = mode == 0 || mode == 1;
  if (d->dtype != CP_F32 && d->dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->R <= 0 || d->S <= 0 || d->stride <= 0 ||
      d->Cout <= 0 || d->Cin <= 0)
    return CP_ERR_INVALID;
  if (d->dy_cstride % E || d->dy_coff % E || d->x_cstride % E || d->x_coff % E) return CP_ERR_ALIGN;
  // every 16-byte piece that starts inside the logical channels must lie inside the pixel's row
  if (d->dy_coff + (d->Cout + E - 1) / E * E > d->dy_cstride || d->x_coff + (d->Cin + E - 1) / E * E > d->x_cstride)
    return CP_ERR_ALIGN;
  if (!cp_aligned16(dy) || !cp_aligned16(x) || ((uintptr_t)dw & 3)) return CP_ERR_ALIGN;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  if (M >= (1LL << 31) || (long long)d->B * d->H * d->W >= (1LL << 31)) return CP_ERR_RANGE;
  // 3x3 / p1 in bf16 on power-of-two maps, stride 1 or 2: all-taps kernels
  int TW, TH;
  if (wgrad_all_taps(d, &TW, &TH)) {
    {
      const bool s2 = d->stride == 2;
      Wgrad3Params q;
      q.dy = dy; q.x = x; q.dw = dw; q.H = d->Ho; q.W = d->Wo; q.Hi = d->H; q.Wi = d->W; q.TH = TH; q.TW = TW;
      q.tiles_x = d->Wo / TW; q.tiles_img = q.tiles_x * (d->Ho / TH); q.n_tiles = d->B * q.tiles_img;
      q.Cout = d->Cout; q.dy_cs = d->dy_cstride; q.dy_coff = d->dy_coff; q.Cin = d->Cin; q.x_cs = d->x_cstride; q.x_coff = d->x_coff;
      q.co_blocks = (d->Cout + 63) / 64; q.ci_blocks = (d->Cin + 63) / 64;
      q.dw_base = d->dw_base; q.dw_sco = d->dw_sco; q.dw_sci = d->dw_sci; q.dw_sr = d->dw_sr; q.dw_ss = d->dw_ss;
      const int tb = q.co_blocks * q.ci_blocks;
      const size_t per_slice = (size_t)tb * 9 * 4096 * sizeof(float);
      const bool use_ws = workspace && cp_aligned16(workspace) && workspace_bytes >= per_slice;
      if (!use_ws && cp_deterministic()) return CP_ERR_INVALID;       // no atomics in deterministic mode: partial tiles need a workspace
      int S = (target_blocks > 0 ? target_blocks : (use_ws ? 256 : 1024)) / tb;              // one block per CU (400 VGPRs): one round of blocks
      if (S > q.n_tiles / 2) S = q.n_tiles / 2;
      if (use_ws && (size_t)S * per_slice > workspace_bytes) S = (int)(workspace_bytes / per_slice);
      if (S < 1) S = 1;
      q.tiles_per_block = (q.n_tiles + S - 1) / S;
      S = (q.n_tiles + q.tiles_per_block - 1) / q.tiles_per_block;
      q.ws = use_ws ? (float*)workspace : nullptr;
      const bool small = d->Cout <= 32 && d->Cin <= 32 && !cp_knob("CP_WGRAD_NO_SMALL");
      if (small) {                                    // tap-split variant: more, lighter blocks (several per CU)
        S = target_blocks > 0 ? target_blocks : (use_ws ? 512 : 1024);
        if (S > q.n_tiles / 2) S = q.n_tiles / 2;
        if (use_ws && (size_t)S * per_slice > workspace_bytes) S = (int)(workspace_bytes / per_slice);
        if (S < 1) S = 1;
        q.tiles_per_block = (q.n_tiles + S - 1) / S;
        S = (q.n_tiles + q.tiles_per_block - 1) / q.tiles_per_block;
        if (launch) {
          if (s2) CP_LAUNCH(wgrad3x3_s2_small_kernel, dim3((unsigned)S, 1), dim3(256), 0, (hipStream_t)stream, q);
          else CP_LAUNCH(wgrad3x3_small_kernel, dim3((unsigned)S, 1), dim3(256), 0, (hipStream_t)stream, q);
        }
      } else if (launch)
        CP_LAUNCH(wgrad3x3_kernel, dim3((unsigned)S, (unsigned)tb), dim3(256), 0, (hipStream_t)stream, q);
      if (mode == 3) {
        static_assert(sizeof(Wgrad3Params) <= CP_WGRAD_ITEM_BYTES, "CpWgradItem too small");
        memset(citem, 0, sizeof(*citem));
        citem->kind = s2 ? CP_WGRAD_ITEM_3X3_S2_SMALL : (small ? CP_WGRAD_ITEM_3X3_SMALL : CP_WGRAD_ITEM_3X3);      // s2 implies small
        citem->gx = (uint32_t)S; citem->gy = small ? 1u : (uint32_t)tb; citem->blocks = citem->gx * citem->gy;
        memcpy(citem->params, &q, sizeof(q));
      }
      int rc3 = launch ? cp_check_launch() : CP_OK;
      if (rc3 || !use_ws) return rc3;
      return launch_reduce((hipStream_t)stream, q.ws, dw, S, tb, q.co_blocks, q.ci_blocks, d, 9, mode, item);
    }
  }
  WgradParams p;
  p.dy = dy; p.x = x; p.dw = dw;
  p.M = (int)M; p.HoWo = d->Ho * d->Wo; p.Wo = d->Wo; p.H = d->H; p.W = d->W;
  p.Cout = d->Cout; p.dy_cs = d->dy_cstride; p.dy_coff = d->dy_coff;
  p.Cin = d->Cin; p.x_cs = d->x_cstride; p.x_coff = d->x_coff;
  p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
  p.co_blocks = (d->Cout + 63) / 64; p.ci_blocks = (d->Cin + 63) / 64;
  p.dw_base = d->dw_base; p.dw_sco = d->dw_sco; p.dw_sci = d->dw_sci; p.dw_sr = d->dw_sr; p.dw_ss = d->dw_ss;
  // pixel slices: enough blocks to fill 256 CUs a few times over, at least 256 pixels (4 stages) per slice
  const long long tiles = (long long)p.co_blocks * p.ci_blocks * d->R * d->S;
  const size_t per_slice = (size_t)tiles * 4096 * sizeof(float);
  bool use_ws = workspace && cp_aligned16(workspace) && workspace_bytes >= per_slice;
  long long want = ((target_blocks > 0 ? target_blocks : (use_ws ? 1024 : 2048)) + tiles - 1) / tiles;  // slices wanted
  if (use_ws && (size_t)want * per_slice > workspace_bytes) want = (long long)(workspace_bytes / per_slice);
  long long slice = (M + want - 1) / want;
  slice = (slice + 63) / 64 * 64;
  if (slice < 256) slice = 256;
  p.slice = (int)slice;
  const unsigned nslice = (unsigned)((M + slice - 1) / slice);
  // small layers: with few valid elements the atomics are cheaper than a second (reduction) launch -- the ~30 G atomics/s
  // limit only bites from ~1e5 atomics per launch on
  static const long long atomics_max = cp_knob("CP_WGRAD_ATOMICS_MAX") ? atoll(cp_knob("CP_WGRAD_ATOMICS_MAX")) : 32768;
  if (use_ws && !cp_deterministic() && (long long)d->Cout * d->Cin * d->R * d->S * nslice <= atomics_max) use_ws = false;
  if (!use_ws && cp_deterministic()) return CP_ERR_INVALID;           // (see cp_set_deterministic)
  if (tiles > 65535) return CP_ERR_RANGE;
  p.ws = use_ws ? (float*)workspace : nullptr;
  dim3 grid(nslice, (unsigned)tiles);
  if (mode == 3) {
    static_assert(sizeof(WgradParams) <= CP_WGRAD_ITEM_BYTES, "CpWgradItem too small");
    memset(citem, 0, sizeof(*citem));
    citem->kind = d->dtype == CP_F32 ? CP_WGRAD_ITEM_GENERIC_F32 : CP_WGRAD_ITEM_GENERIC_BF16;
    citem->gx = nslice; citem->gy = (uint32_t)tiles; citem->blocks = citem->gx * citem->gy;
    memcpy(citem->params, &p, sizeof(p));
  } else if (!launch) { /* plan only */ }
  else if (d->dtype == CP_F32) CP_LAUNCH(wgrad_kernel<F32Tag>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else CP_LAUNCH(wgrad_kernel<BF16Tag>, grid, dim3(256), 0, (hipStream_t)stream, p);
  int rcg = launch ? cp_check_launch() : CP_OK;
  if (rcg || !use_ws) return rcg;
  return launch_reduce((hipStream_t)stream, p.ws, dw, (int)nslice, (int)tiles, p.co_blocks, p.ci_blocks, d, 1, mode, item);
}

// ---- grouped launches: the partial-sum kernels of SEVERAL layers in one launch (a device table of parameter blocks built by
// cp_conv2d_wgrad_item; block b belongs to item k with prefix[k] <= b < prefix[k+1]; inside the item blockIdx.x runs fastest).
// One launch per kernel kind.  The ~340 weight-gradient launches of the B = 32 training step were 9-23 us each for 1-8 tiles of work
// per block (fixed cost: cold instruction cache, prologue / epilogue of a 400-VGPR block) and every layer cut its pixels into a
// whole GPU's worth of slices (2.4 GB of partial tiles per step).
template <int KIND>
__global__ __launch_bounds__(256, (KIND == CP_WGRAD_ITEM_3X3_SMALL || KIND == CP_WGRAD_ITEM_3X3_S2_SMALL) ? 2 : 1) void wgrad_group_kernel(const CpWgradItem* __restrict__ items,
                                                                                                const uint32_t* __restrict__ prefix, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const unsigned b = blockIdx.x - prefix[lo], gx = items[lo].gx, gy = items[lo].gy;
  const unsigned by = b / gx, bx = b - by * gx;
  if constexpr (KIND == CP_WGRAD_ITEM_3X3) {
    const Wgrad3Params p = *(const Wgrad3Params*)items[lo].params;
    wgrad3x3_body(p, bx, by, gy);
  } else if constexpr (KIND == CP_WGRAD_ITEM_3X3_SMALL || KIND == CP_WGRAD_ITEM_3X3_S2_SMALL) {
    const Wgrad3Params p = *(const Wgrad3Params*)items[lo].params;
    wgrad3x3_small_body<KIND == CP_WGRAD_ITEM_3X3_SMALL ? 1 : 2>(p, bx);
  } else if constexpr (KIND == CP_WGRAD_ITEM_GENERIC_BF16) {
    const WgradParams p = *(const WgradParams*)items[lo].params;
    wgrad_body<BF16Tag>(p, bx, by, gy);
  } else {
    const WgradParams p = *(const WgradParams*)items[lo].params;
    wgrad_body<F32Tag>(p, bx, by, gy);
  }
}

extern "C" int cp_conv2d_wgrad_item(const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace, size_t workspace_bytes,
                                    int target_blocks, CpWgradItem* compute, CpWgradReduceItem* reduce) {
  if (!compute || !reduce || target_blocks < 0) return CP_ERR_INVALID;
  memset(reduce, 0, sizeof(*reduce));
  return wgrad_impl(nullptr, d, dy, x, dw, workspace, workspace_bytes, 3, reduce, target_blocks, compute);
}

extern "C" int cp_wgrad_group(cp_stream_t stream, int kind, const CpWgradItem* items_dev, const uint32_t* prefix_dev, int n_items,
                              uint32_t total_blocks) {
  if (!items_dev || !prefix_dev || n_items <= 0 || total_blocks == 0) return CP_ERR_INVALID;
  if (!cp_aligned16(items_dev)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case CP_WGRAD_ITEM_3X3: CP_LAUNCH(wgrad_group_kernel<CP_WGRAD_ITEM_3X3>, dim3(total_blocks), dim3(256), 0, st, items_dev, prefix_dev, n_items); break;
    case CP_WGRAD_ITEM_3X3_SMALL: CP_LAUNCH(wgrad_group_kernel<CP_WGRAD_ITEM_3X3_SMALL>, dim3(total_blocks), dim3(256), 0, st, items_dev, prefix_dev, n_items); break;
    case CP_WGRAD_ITEM_GENERIC_BF16: CP_LAUNCH(wgrad_group_kernel<CP_WGRAD_ITEM_GENERIC_BF16>, dim3(total_blocks), dim3(256), 0, st, items_dev, prefix_dev, n_items); break;
    case CP_WGRAD_ITEM_GENERIC_F32: CP_LAUNCH(wgrad_group_kernel<CP_WGRAD_ITEM_GENERIC_F32>, dim3(total_blocks), dim3(256), 0, st, items_dev, prefix_dev, n_items); break;
    case CP_WGRAD_ITEM_3X3_S2_SMALL: CP_LAUNCH(wgrad_group_kernel<CP_WGRAD_ITEM_3X3_S2_SMALL>, dim3(total_blocks), dim3(256), 0, st, items_dev, prefix_dev, n_items); break;
    default: return CP_ERR_INVALID;
  }
  return cp_check_launch();
}
