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
};

template <typename Tag> struct WgCfg;
template <> struct WgCfg<BF16Tag> { static constexpr int PITCH = 160, PPR = 8; };    // bytes per LDS row, 16-B pieces per row
template <> struct WgCfg<F32Tag> { static constexpr int PITCH = 320, PPR = 16; };

template <typename Tag>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
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
  int t = blockIdx.y;
  const int cob = t % p.co_blocks; t /= p.co_blocks;
  const int cib = t % p.ci_blocks; t /= p.ci_blocks;
  const int r = t / p.S, s = t - r * p.S;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int m_begin = blockIdx.x * p.slice;
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

extern "C" int cp_conv2d_wgrad(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw) {
  if (!d || !dy || !x || !dw) return CP_ERR_INVALID;
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
  long long want = (2048 + tiles - 1) / tiles;                    // slices wanted
  long long slice = (M + want - 1) / want;
  slice = (slice + 63) / 64 * 64;
  if (slice < 256) slice = 256;
  p.slice = (int)slice;
  const unsigned nslice = (unsigned)((M + slice - 1) / slice);
  if (tiles > 65535) return CP_ERR_RANGE;
  dim3 grid(nslice, (unsigned)tiles);
  if (d->dtype == CP_F32) CP_LAUNCH(wgrad_kernel<F32Tag>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else CP_LAUNCH(wgrad_kernel<BF16Tag>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}
