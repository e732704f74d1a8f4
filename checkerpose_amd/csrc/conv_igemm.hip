// Implicit-GEMM convolution / linear layer on gfx950 MFMA.
//
// GEMM view:  D^T[channel][pixel] = sum_k Wp[channel][k] * im2col(X)[pixel][k]
//   pixels  m = (b, oy, ox) flattened, channels-last input  X (B,H,W,Cstride)
//   k       = (r, s, cin), cin padded to the physical channel count (multiple of E = 16B / sizeof(T))
// One wave owns a (MT*16 pixel) x (NT*16 channel) tile; a 256-thread block = 4 waves stacked along
// the pixel axis that share the same channel tiles (weight fragments hit in L1).
//
// Fragment shape (both dtypes): a K-chunk is 64 bytes per row (16 f32 / 32 bf16); lane (x = l&15,
// q = l>>4) loads the 16-byte piece q of row x -- for activations straight from the channels-last
// tensor (contiguous in cin, bounds-checked buffer load -> zero padding for free), for weights from
// the pre-packed fragment-major image (one fully coalesced 1 KiB wave load per fragment).
//   bf16: one v_mfma_f32_16x16x32_bf16 per fragment pair (lane holds k = 8q..8q+7, exactly its layout)
//   f32 : four v_mfma_f32_16x16x4_f32; instruction j contracts k = {4q + j}: the same K permutation is
//         applied to both operands, so the sum is unchanged and every load stays 16 bytes wide.
// The weight fragment is the MFMA A operand and the activation fragment the B operand, so the
// accumulator holds 4 CONSECUTIVE CHANNELS of one pixel per lane -> 16-byte (f32) / 8-byte (bf16)
// epilogue stores into the channels-last output.
//
// Replaces (reference, relative to /root/reference/checkerpose): every nn.Conv2d/BatchNorm2d/ReLU,
// nn.Linear/LeakyReLU on the path -- see include/checkerpose_hip.h for the file:line list.
#include <stdlib.h>

#include "common.h"

struct ConvParams {
  const void* in; const void* w; const float* scale; const float* shift; const void* res; void* out;
  int M, H, W, HoWo, Wo;
  int Cin, in_cs, in_coff;
  int R, S, stride, pad;
  int KC, Cout, n_tiles, m_blocks, n_blocks, old_map;
  int act; float slope; int out_f32; int epi_lds;
  uint32_t in_bytes, w_bytes;
  long long o_base, o_sb, o_sy, o_sx, o_sc;
};

template <typename Tag> struct Mma;
template <> struct Mma<F32Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};
template <> struct Mma<BF16Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
  }
};

template <> struct Mma<F16Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) { acc = cp_mma16<true>(w, a, acc); }
};

template <typename Tag, int MT, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  if (p.out_f32 == 2 || Tag::dtype == CP_F16) cp_f16_saturate_on();      // half output rows saturate at +-65504 (common.h)
  constexpr int E = Tag::E;
  constexpr int KCH = 4 * E;
  constexpr int ES = 16 / E;  // sizeof(elem)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = lane & 15, q = lane >> 4;
  // block -> (pixel block, channel block): channel blocks are the FAST index and all channel blocks of one pixel
  // block share a blockIdx % 8 label (one XCD): they run back to back on one L2, so the activation tile they all
  // read is fetched from HBM once and the 128-byte channel slices they each write/read of the same pixel rows are
  // touched together (full DRAM rows instead of four strided sweeps over the output / residual tensors).
  const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int nblk = slot % p.n_blocks;
  int mblk = (slot / p.n_blocks) * 8 + label;
  if (p.old_map) { const int m8 = (p.m_blocks + 7) / 8 * 8; mblk = blockIdx.x % m8; nblk = blockIdx.x / m8; }   // A/B only
  if (mblk >= p.m_blocks) return;
  const int m_wave = mblk * (4 * MT * 16) + wave * (MT * 16);
  const int nt0 = nblk * NT;

  // ---- per-(lane, mt) pixel decode: input window origin and element offset of (iy0, ix0)
  int iy0[MT], ix0[MT], rowbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_wave + mt * 16 + x;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / p.HoWo;
    const int rem = mm - b * p.HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    iy0[mt] = ok ? oy * p.stride - p.pad : -(1 << 28);   // out-of-range rows never pass the bounds test
    ix0[mt] = ox * p.stride - p.pad;
    rowbase[mt] = ((b * p.H + (ok ? iy0[mt] : 0)) * p.W + ix0[mt]) * p.in_cs + p.in_coff;
  }
  // ---- per-lane K state: this lane's 16-byte piece starts at k = kc*KCH + q*E -> (tap r,s ; cin c)
  int kk = q * E;
  int tap = kk / p.Cin;
  int c = kk - tap * p.Cin;
  int r = tap / p.S;
  int s = tap - r * p.S;

  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  // weights through the SAME load kind (buffer loads): hipcc then counts vmcnt across both operand streams
  // instead of draining to 0 (mixed global/buffer loads made it wait for the prefetched chunk too).
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  // weight fragment byte offsets (tile index clamped: a partial last channel block recomputes a valid tile)
  uint32_t woff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int t = nt0 + nt;
    t = t < p.n_tiles ? t : p.n_tiles - 1;
    woff[nt] = ((uint32_t)t * p.KC * 64 + lane) * 16u;
  }

  // ---- coalesced-epilogue bookkeeping + residual prefetch (see the epilogue): issued BEFORE the main loop so the
  // residual's HBM round trip overlaps the operand loads instead of following the MFMAs (these launches are
  // latency-bound: PMC showed waves parked on memory 57 % of their life with ~11 waves resident per CU).
  constexpr bool EPI = (MT == 2 && (NT == 1 || NT == 2 || NT == 4));
  constexpr bool f32o = (E == 4);                            // host enables this path only when out dtype == T
  constexpr int CPL = f32o ? 4 : 8;                          // channels per 16-byte lane access
  constexpr int LPP = EPI ? NT * 16 / CPL : 1;               // lanes per pixel
  constexpr int PPP = 64 / LPP;                              // pixels per pass
  constexpr int NPASS = EPI ? MT * 16 / PPP : 1;
  const int pl0 = lane / LPP, cg = lane - pl0 * LPP;
  const int cch = nt0 * 16 + cg * CPL;                       // first channel of this lane
  u32x4 rres[NPASS];
  long long ofs[NPASS];
  bool okp[NPASS];
  if constexpr (EPI) if (p.epi_lds) {
    const bool cok = cch < p.Cout;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int m = m_wave + ps * PPP + pl0;
      okp[ps] = (m < p.M) & cok;
      const int mm = m < p.M ? m : 0;
      const int b = mm / p.HoWo;
      const int rem = mm - b * p.HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      ofs[ps] = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + cch;
      rres[ps] = u32x4{0u, 0u, 0u, 0u};
      if (p.res && okp[ps])
        rres[ps] = f32o ? *(const u32x4*)((const float*)p.res + ofs[ps]) : *(const u32x4*)((const uint16_t*)p.res + ofs[ps]);
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool wide_c = p.Cin >= KCH;   // wave-uniform: at most one tap step per chunk
  auto load_chunk = [&](u32x4* a, u32x4* w, int kc) {
    const int tapoff = (r * p.W + s) * p.in_cs + c;
    const bool tap_ok = r < p.R;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = tap_ok & ((unsigned)(iy0[mt] + r) < (unsigned)p.H) & ((unsigned)(ix0[mt] + s) < (unsigned)p.W);
      const uint32_t off = ok ? (uint32_t)(rowbase[mt] + tapoff) * ES : 0x80000000u;
      a[mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      w[nt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff[nt] + (uint32_t)kc * 1024u, 0, 0));
    // advance the lane's K state by one chunk (branch-free when Cin >= KCH)
    c += KCH;
    if (wide_c) {
      const bool wrap = c >= p.Cin;
      c = wrap ? c - p.Cin : c;
      s += wrap ? 1 : 0;
      const bool swrap = s == p.S;
      s = swrap ? 0 : s;
      r += swrap ? 1 : 0;
    } else {
      while (c >= p.Cin) {
        c -= p.Cin;
        if (++s == p.S) { s = 0; ++r; }
      }
    }
  };
  auto compute = [&](const u32x4* a, const u32x4* w) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) Mma<Tag>::run(w[nt], a[mt], acc[mt][nt]);
  };

  // ---- main loop, register double-buffered and STRAIGHT-LINE: two chunks are always in flight.  Loads past the
  // last chunk are issued unconditionally (their offsets are out of range -> the buffer unit returns zeros, and
  // the tap index is past R -> invalid), so there is no control-flow join between a load and its use and hipcc
  // emits counted `s_waitcnt vmcnt(8)` instead of draining the prefetch (a conditional prefetch made it wait to 0).
  u32x4 a0[MT], w0[NT], a1[MT], w1[NT];
  load_chunk(a0, w0, 0);
  load_chunk(a1, w1, 1);
  for (int kc = 0; kc < p.KC; kc += 2) {
    compute(a0, w0);
    load_chunk(a0, w0, kc + 2);
    if (kc + 1 < p.KC) compute(a1, w1);     // wave-uniform; no loads inside the branch
    load_chunk(a1, w1, kc + 3);
  }

  // ---- coalesced epilogue through wave-private LDS (channels-last output with NT in {1,2,4}).
  // Measured (rocprofv3 PMC, 64->256 1x1 conv + residual): the texture-address unit was 72 % busy and its cost is
  // ~4 cycles per distinct 128-byte line an instruction touches; the MFMA-layout epilogue below touches 16 lines
  // per 512-byte store / residual load (32-byte pieces of 16 pixel rows) and capped the kernel at ~2.4 TB/s.
  // Here the scaled accumulators are transposed through LDS so that every lane moves 16 contiguous bytes and
  // consecutive lanes walk one pixel's channels: full-line accesses, 4x fewer line touches for residual + store.
  if constexpr (EPI) if (p.epi_lds) {
    extern __shared__ __attribute__((aligned(16))) float epi_smem[];
    constexpr int ROWF = NT * 16 + 4;                       // floats per pixel row (+4: spreads the banks)
    float* my = epi_smem + wave * (MT * 16 * ROWF);
    // scaled accumulators -> LDS [pixel][channel] (fp32)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = (nt0 + nt) * 16 + q * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nt0 + nt < p.n_tiles && n < p.Cout) {
          const f32x4 sc = *(const f32x4*)(p.scale + n);
          const f32x4 sh = *(const f32x4*)(p.shift + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * sc[j] + sh[j];
        }
        *(f32x4*)(my + (mt * 16 + x) * ROWF + nt * 16 + q * 4) = v;
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const float* src = my + (ps * PPP + pl0) * ROWF + cg * CPL;
      float v[8];
      const f32x4 v0 = *(const f32x4*)src;
      v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
      if (!f32o) { const f32x4 v1 = *(const f32x4*)(src + 4); v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3]; }
      if (!okp[ps]) continue;
      if (p.res) {
        if (f32o) {
          v[0] += __uint_as_float(rres[ps].x); v[1] += __uint_as_float(rres[ps].y);
          v[2] += __uint_as_float(rres[ps].z); v[3] += __uint_as_float(rres[ps].w);
        } else {
          float r8[8];
          Vec16<BF16Tag>::unpack(rres[ps], r8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += r8[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
      }
      if (f32o) *(f32x4*)((float*)p.out + ofs[ps]) = f32x4{v[0], v[1], v[2], v[3]};
      else *(u32x4*)((uint16_t*)p.out + ofs[ps]) = Vec16<BF16Tag>::pack(v);
    }
    return;
  }

  // ---- epilogue: lane holds pixel (lane&15) x channels 4q..4q+3 of every (mt, nt) tile.
  // Residuals are ALL loaded before the first store: `res` may alias `out` (no __restrict__), so a load issued
  // after a store would be ordered behind it and the epilogue would degrade into MT*NT serial memory round trips.
  const bool vec = (p.o_sc == 1);
  const bool f32io = p.out_f32 == 1 || E == 4;
  const bool h16out = p.out_f32 == 2 || (Tag::dtype == CP_F16 && p.out_f32 == 0);      // IEEE-half output rows (no residual: host check)
  long long pix[MT];
  bool pok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_wave + mt * 16 + x;
    pok[mt] = m < p.M;
    const int mm = pok[mt] ? m : 0;
    const int b = mm / p.HoWo;
    const int rem = mm - b * p.HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    pix[mt] = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx;
  }
  f32x4 rv[MT][NT];
  if (vec && p.res) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = (nt0 + nt) * 16 + q * 4;
        rv[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pok[mt] && nt0 + nt < p.n_tiles && n < p.Cout) {
          const long long o = pix[mt] + n;
          if (f32io) rv[mt][nt] = *(const f32x4*)((const float*)p.res + o);
          else {
            const u32x2 r2 = *(const u32x2*)((const uint16_t*)p.res + o);
            rv[mt][nt] = f32x4{__uint_as_float(r2.x << 16), __uint_as_float(r2.x & 0xffff0000u),
                               __uint_as_float(r2.y << 16), __uint_as_float(r2.y & 0xffff0000u)};
          }
        }
      }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!pok[mt]) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (nt0 + nt) * 16 + q * 4;
      if (nt0 + nt >= p.n_tiles || n >= p.Cout) continue;
      const f32x4 sc = *(const f32x4*)(p.scale + n);
      const f32x4 sh = *(const f32x4*)(p.shift + n);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * sc[j] + sh[j];
      if (vec) {
        const long long o = pix[mt] + n;
        if (p.res) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += rv[mt][nt][j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
        }
        if (f32io) {
          *(f32x4*)((float*)p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else if (h16out) {
          u32x2 pk; pk.x = cp_pack2<true>(v[0], v[1]); pk.y = cp_pack2<true>(v[2], v[3]);
          *(u32x2*)((uint16_t*)p.out + o) = pk;
        } else {
          u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
          *(u32x2*)((uint16_t*)p.out + o) = pk;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n + j >= p.Cout) continue;
          const long long o = pix[mt] + (long long)(n + j) * p.o_sc;
          float y = v[j];
          if (p.res) y += f32io ? ((const float*)p.res)[o] : bf16_bits_to_f32(((const uint16_t*)p.res)[o]);
          y = cp_act_apply(y, cp_act_slope(p.act, p.slope));
          if (f32io) ((float*)p.out)[o] = y;
          else if (h16out) ((uint16_t*)p.out)[o] = (uint16_t)f32_to_f16_bits_sat(y);
          else ((uint16_t*)p.out)[o] = (uint16_t)f32_to_bf16_bits(y);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ split-K variant
// Small-batch regime (the reference's own inference batch is 1, test.py:198): a 3x3 conv on an 8x8x144 map is 64 pixels x
// 144 channels x K = 1296 -- two or three workgroups of the tiled kernel above, each walking 41 K chunks one dependent
// L2 round trip after the other (13 us for 0.02 GFLOP).  Here the KS waves of a workgroup share ONE 32-pixel x NT*16-
// channel tile and split its K chunks between them (wave w: chunks w, w + KS, ...), with all of a wave's operand loads
// in flight at once; the partial accumulators meet in LDS and the (pixel, 4-channel) items of the tile are finished by
// the whole workgroup, one item per thread.  Same packed weights, same K order per chunk, same epilogue arithmetic as
// the tiled kernel; the fp32 partial sums are added in a different order (wave-major), so results agree with it to
// fp32 rounding of the accumulator, not bit for bit.
template <typename Tag, int NT, int KS>
__global__ __launch_bounds__(KS * 64) void conv_igemm_splitk_kernel(const ConvParams p) {
  if (p.out_f32 == 2 || Tag::dtype == CP_F16) cp_f16_saturate_on();      // half output rows saturate at +-65504 (common.h)
  constexpr int E = Tag::E, KCH = 4 * E, ES = 16 / E, MT = 2, T = MT * NT;
  constexpr int U = NT == 1 ? 8 : (NT == 2 ? 6 : 4);        // chunks of one wave in flight together
  __shared__ __attribute__((aligned(16))) float red[KS * T * 64 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int nblk = blockIdx.x % p.n_blocks, mblk = blockIdx.x / p.n_blocks;
  const int m0 = mblk * (MT * 16), nt0 = nblk * NT;

  int iy0[MT], ix0[MT], rowbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + mt * 16 + x;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / p.HoWo;
    const int rem = mm - b * p.HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    iy0[mt] = ok ? oy * p.stride - p.pad : -(1 << 28);
    ix0[mt] = ox * p.stride - p.pad;
    rowbase[mt] = ((b * p.H + (ok ? iy0[mt] : 0)) * p.W + ix0[mt]) * p.in_cs + p.in_coff;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  uint32_t woff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int t = nt0 + nt;
    t = t < p.n_tiles ? t : p.n_tiles - 1;
    woff[nt] = ((uint32_t)t * p.KC * 64 + lane) * 16u;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kc0 = wave; kc0 < p.KC; kc0 += KS * U) {
    u32x4 a[U][MT], w[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kc = kc0 + u * KS;                          // past the last chunk: tap >= R*S and the weight offset is out of range -> zeros
      const int kk = kc * KCH + q * E;
      const int tap = kk / p.Cin, c = kk - tap * p.Cin;
      const int r = tap / p.S, s_ = tap - r * p.S;
      const int tapoff = (r * p.W + s_) * p.in_cs + c;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool ok = (r < p.R) & ((unsigned)(iy0[mt] + r) < (unsigned)p.H) & ((unsigned)(ix0[mt] + s_) < (unsigned)p.W);
        const uint32_t off = ok ? (uint32_t)(rowbase[mt] + tapoff) * ES : 0x80000000u;
        a[u][mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint32_t off = kc < p.KC ? woff[nt] + (uint32_t)kc * 1024u : 0x80000000u;
        w[u][nt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Mma<Tag>::run(w[u][nt], a[u][mt], acc[mt][nt]);
  }

  // ---- partial sums meet in LDS: [wave][tile][lane] f32x4
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *(f32x4*)(red + ((wave * T + mt * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();

  // ---- one (pixel, 4 channels) item per thread: sum the KS partials in wave order, then the tiled kernel's epilogue
  const bool f32io = p.out_f32 == 1 || E == 4;
  const bool h16out = p.out_f32 == 2 || (Tag::dtype == CP_F16 && p.out_f32 == 0);      // IEEE-half output rows (no residual: host check)
  for (int it = threadIdx.x; it < T * 64; it += KS * 64) {
    const int t = it >> 6, l = it & 63;
    const int mt = t / NT, nt = t - mt * NT;
    f32x4 sum = *(const f32x4*)(red + (t * 64 + l) * 4);
#pragma unroll
    for (int wv = 1; wv < KS; ++wv) {
      const f32x4 v = *(const f32x4*)(red + ((wv * T + t) * 64 + l) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) sum[j] += v[j];
    }
    const int m = m0 + mt * 16 + (l & 15);
    const int n = (nt0 + nt) * 16 + (l >> 4) * 4;
    if (m >= p.M || nt0 + nt >= p.n_tiles || n >= p.Cout) continue;
    const int b = m / p.HoWo;
    const int rem = m - b * p.HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const long long pix = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx;
    const f32x4 sc = *(const f32x4*)(p.scale + n);
    const f32x4 sh = *(const f32x4*)(p.shift + n);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sum[j] * sc[j] + sh[j];
    if (p.o_sc == 1) {
      const long long o = pix + n;
      if (p.res) {
        if (f32io) {
          const f32x4 r4 = *(const f32x4*)((const float*)p.res + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += r4[j];
        } else {
          const u32x2 r2 = *(const u32x2*)((const uint16_t*)p.res + o);
          v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
          v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
      }
      if (f32io) {
        *(f32x4*)((float*)p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
      } else if (h16out) {
        u32x2 pk; pk.x = cp_pack2<true>(v[0], v[1]); pk.y = cp_pack2<true>(v[2], v[3]);
        *(u32x2*)((uint16_t*)p.out + o) = pk;
      } else {
        u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
        *(u32x2*)((uint16_t*)p.out + o) = pk;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (n + j >= p.Cout) continue;
        const long long o = pix + (long long)(n + j) * p.o_sc;
        float y = v[j];
        if (p.res) y += f32io ? ((const float*)p.res)[o] : bf16_bits_to_f32(((const uint16_t*)p.res)[o]);
        y = cp_act_apply(y, cp_act_slope(p.act, p.slope));
        if (f32io) ((float*)p.out)[o] = y;
        else if (h16out) ((uint16_t*)p.out)[o] = (uint16_t)f32_to_f16_bits_sat(y);
        else ((uint16_t*)p.out)[o] = (uint16_t)f32_to_bf16_bits(y);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
// split-K plan: bf16, a K walk worth splitting, and a tiled grid that would leave most CUs idle.  Channel tiles per
// workgroup: as many (<= 3: the pixel fragments are then read once for 48 channels) as still leave a workgroup per CU.
static int splitk_plan(int dtype, long long M, int KC, int n_tiles, int* NTs) {
  if ((dtype != CP_BF16 && dtype != CP_F16) || KC < 6 || cp_knob("CP_NO_SPLITK")) return 0;
  int NT = 1, best = 1 << 30;
  for (int nt = 5; nt >= 1; --nt) {
    const int padded = (n_tiles + nt - 1) / nt * nt;
    if (padded < best) { best = padded; NT = nt; }
  }
  const long long blocks2 = ((M + 127) / 128) * ((n_tiles + NT - 1) / NT);     // the tiled kernel's grid at MT = 2
  if (blocks2 > (KC >= 64 ? 192 : 96)) return 0;                               // a long K walk (decoder convs at batch 1-2) pays a little longer
  const long long m32 = (M + 31) / 32;
  *NTs = 1;
  for (int nt = 3; nt >= 2; --nt)
    if (m32 * ((n_tiles + nt - 1) / nt) >= 256) { *NTs = nt; break; }
  return KC >= 24 ? 8 : 4;
}

template <int NT, int KS>
static void launch_splitk(ConvParams p, hipStream_t st, bool half_in) {
  p.m_blocks = (p.M + 31) / 32;
  p.n_blocks = (p.n_tiles + NT - 1) / NT;
  cp_mark_kernel("conv_igemm_splitk_kernel<%s, %d, %d>", half_in ? "F16Tag" : "BF16Tag", NT, KS);
  if (half_in) hipLaunchKernelGGL((conv_igemm_splitk_kernel<F16Tag, NT, KS>), dim3((unsigned)(p.m_blocks * p.n_blocks)), dim3(KS * 64), 0, st, p);
  else hipLaunchKernelGGL((conv_igemm_splitk_kernel<BF16Tag, NT, KS>), dim3((unsigned)(p.m_blocks * p.n_blocks)), dim3(KS * 64), 0, st, p);
}

// 0, or the number of waves the K walk of this conv would be split over (the engine then routes 3x3 / 1x1 convs that have
// faster large-batch kernels to cp_conv2d_igemm with the generic weight pack)
extern "C" int cp_conv2d_igemm_splitk(int dtype, long long M, int K, int Cout) {
  if (M <= 0 || K <= 0 || Cout <= 0 || (dtype != CP_F32 && dtype != CP_BF16 && dtype != CP_F16)) return 0;
  const int KCH = 4 * cp_chan_align(dtype);
  int nts = 1;
  return splitk_plan(dtype, M, (K + KCH - 1) / KCH, (Cout + 15) / 16, &nts);
}

template <typename Tag, int MT, int NT>
static void launch(ConvParams p, hipStream_t st) {
  p.m_blocks = (p.M + 4 * MT * 16 - 1) / (4 * MT * 16);
  p.n_blocks = (p.n_tiles + NT - 1) / NT;
  p.old_map = cp_knob("CP_OLD_MAP") ? 1 : 0;
  dim3 grid((unsigned)((p.m_blocks + 7) / 8 * 8) * p.n_blocks);
  const size_t lds = p.epi_lds ? (size_t)4 * MT * 16 * (NT * 16 + 4) * sizeof(float) : 0;
  cp_mark_kernel("conv_igemm_kernel<%s, %d, %d>", Tag::dtype == CP_BF16 ? "BF16Tag" : (Tag::dtype == CP_F16 ? "F16Tag" : "F32Tag"), MT, NT);
  hipLaunchKernelGGL((conv_igemm_kernel<Tag, MT, NT>), grid, dim3(256), lds, st, p);
}

template <typename Tag, int MT>
static void dispatch_nt(const ConvParams& p, int NT, hipStream_t st) {
  switch (NT) {
    case 1: launch<Tag, MT, 1>(p, st); break;
    case 2: launch<Tag, MT, 2>(p, st); break;
    case 3: launch<Tag, MT, 3>(p, st); break;
    case 4: launch<Tag, MT, 4>(p, st); break;
    default: launch<Tag, MT, 5>(p, st); break;
  }
}

extern "C" int cp_conv2d_igemm(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                               const float* scale, const float* shift, const void* residual, void* out) {
  if (!d || !in || !packed_w || !scale || !shift || !out) return CP_ERR_INVALID;
  if ((d->dtype != CP_F32 && d->dtype != CP_BF16 && d->dtype != CP_F16) || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  // IEEE half at the keypoint side's edges: dtype CP_F16 = half input rows + half weights (cp_pack_conv_weight(CP_F16)); out_f32 = 2 =
  // half OUTPUT rows of a bf16 conv.  Neither takes a residual (the epilogue's residual decode is bf16 / fp32).
  const bool halfish = d->dtype == CP_F16 || d->out_f32 == 2;
  if (d->out_f32 < 0 || d->out_f32 > 2 || (d->out_f32 == 2 && d->dtype != CP_BF16) || (halfish && residual)) return CP_ERR_INVALID;
  const int E = cp_chan_align(d->dtype), es = cp_elem_size(d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->R <= 0 || d->S <= 0 || d->stride <= 0)
    return CP_ERR_INVALID;
  if (d->Cin <= 0 || d->Cin % E || d->in_coff % E || d->in_cstride % E || d->in_coff + d->Cin > d->in_cstride)
    return CP_ERR_ALIGN;
  if (d->Cout <= 0 || (d->o_sc == 1 && d->Cout % 4)) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift))
    return CP_ERR_ALIGN;
  const long long in_bytes = (long long)d->B * d->H * d->W * d->in_cstride * es;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  if (M >= (1LL << 31)) return CP_ERR_RANGE;
  if (d->o_sc == 1) {  // vector epilogue: 4 consecutive channels per store
    const int oes = d->out_f32 == 1 ? 4 : es;
    if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4)) return CP_ERR_ALIGN;
    if (((uintptr_t)out % (4 * oes)) || (residual && ((uintptr_t)residual % (4 * oes)))) return CP_ERR_ALIGN;
  }
  ConvParams p;
  p.in = in; p.w = packed_w; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
  p.M = (int)M; p.H = d->H; p.W = d->W; p.HoWo = d->Ho * d->Wo; p.Wo = d->Wo;
  p.Cin = d->Cin; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
  const int K = d->R * d->S * d->Cin, KCH = 4 * E;
  p.KC = (K + KCH - 1) / KCH;
  p.Cout = d->Cout; p.n_tiles = (d->Cout + 15) / 16;
  p.act = d->act; p.slope = d->slope; p.out_f32 = d->out_f32;
  p.in_bytes = (uint32_t)in_bytes;
  p.w_bytes = (uint32_t)((size_t)p.n_tiles * p.KC * 1024);
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx; p.o_sc = d->o_sc;

  int nts = 1;
  if (const int ks = d->ksplit == -1 ? 0 : splitk_plan(d->dtype, M, p.KC, p.n_tiles, &nts)) {
    hipStream_t st_ = (hipStream_t)stream;
    const bool hin = d->dtype == CP_F16;
    if (ks == 8) { if (nts == 3) launch_splitk<3, 8>(p, st_, hin); else if (nts == 2) launch_splitk<2, 8>(p, st_, hin); else launch_splitk<1, 8>(p, st_, hin); }
    else         { if (nts == 3) launch_splitk<3, 4>(p, st_, hin); else if (nts == 2) launch_splitk<2, 4>(p, st_, hin); else launch_splitk<1, 4>(p, st_, hin); }
    return cp_check_launch();
  }
  // tile choice: NT minimises padded channel tiles (ties -> wider), MT=4 (256 pixels/block) unless the
  // grid would leave most of the 256 CUs idle.
  int NT = 1, best = 1 << 30;
  for (int nt = 5; nt >= 1; --nt) {
    const int padded = (p.n_tiles + nt - 1) / nt * nt;
    if (padded < best) { best = padded; NT = nt; }
  }
  const long long blocks4 = ((M + 255) / 256) * ((p.n_tiles + NT - 1) / NT);
  // MT=4 (256 pixels/block) for MFMA-heavy shapes; MT=2 when the grid would be small OR the layer is memory-bound
  // (few K-chunks: more, lighter waves keep more bytes in flight -- measured +20..30 % on the 1x1 convs).
  int MT = (blocks4 >= 512 && p.KC > 4) ? 4 : 2;
  if (const char* e = cp_knob("CP_CONV_MT")) MT = atoi(e) == 2 ? 2 : 4;   // kernel-work A/B switch
  // coalesced LDS epilogue: channels-last vector output, NT in {1,2,4}, 16-byte alignment of every row piece
  const int oes_ = d->out_f32 == 1 ? 4 : es;
  const int cpl = 16 / oes_;
  p.epi_lds = (MT == 2 && d->o_sc == 1 && !d->out_f32 && !halfish && (NT == 1 || NT == 2 || NT == 4) && d->Cout % cpl == 0 && d->o_base % cpl == 0 &&
               d->o_sb % cpl == 0 && d->o_sy % cpl == 0 && d->o_sx % cpl == 0 && ((uintptr_t)out % 16) == 0 &&
               (!residual || ((uintptr_t)residual % 16) == 0) && !cp_knob("CP_NO_EPI_LDS")) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CP_F32) { if (MT == 4) dispatch_nt<F32Tag, 4>(p, NT, st); else dispatch_nt<F32Tag, 2>(p, NT, st); }
  else if (d->dtype == CP_F16) { if (MT == 4) dispatch_nt<F16Tag, 4>(p, NT, st); else dispatch_nt<F16Tag, 2>(p, NT, st); }
  else                    { if (MT == 4) dispatch_nt<BF16Tag, 4>(p, NT, st); else dispatch_nt<BF16Tag, 2>(p, NT, st); }
  return cp_check_launch();
}
