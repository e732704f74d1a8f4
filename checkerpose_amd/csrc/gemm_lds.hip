// 1x1 convolution / Linear layer as an LDS-staged GEMM (the EdgeConv node GEMMs 256->512, the 256-wide MLPs,
// conv1x1 1024->N, incre conv3): M = pixels or keypoints (rows, K contiguous), N = Cout >= 96, K = Cin >= 64.
//
// Same operand split as the 3x3 halo kernel (conv3x3_halo.hip): the block's 128 rows are staged in LDS in full
// 256-byte row segments (16 consecutive lanes read one row's 256 contiguous bytes: coalesced, each byte crosses
// the texture path once per block instead of once per wave and per fragment), the 4 waves own disjoint 32-channel
// groups so each packed weight fragment (1 KiB coalesced buffer load) is read by exactly one wave.
//   LDS image per K super-chunk (4 chunks of 64 B): [chunk][q = 16-B piece][row 0..127][16 B]  (32 KiB, x2 buffers)
//   ds_read_b128 of fragment (row block mt, chunk, q): rows consecutive -> conflict-free (plane pitch 2 KiB).
// Weights: four rotating register sets, three chunks ahead; row fragments: one chunk ahead (sched_barrier pinned).
// Packed-weight image = the halo kernel's with one tap: [32-ch group][chunk][tile][lane][16 B], rows permuted so a
// lane ends with 8 consecutive channels (16-byte bf16 / 2 x 16-byte f32 stores).
#include <stdlib.h>

#include "common.h"

struct GemmParams {
  const void* in; const void* w; const float* scale; const float* shift; const void* res; void* out;
  int M, HoWo, Wo, Cin, in_cs, in_coff, nchunk, nsuper, Cout, ngroups, NB, m_blocks;
  int act; float slope;
  uint32_t in_bytes, w_bytes;
  long long o_base, o_sb, o_sy, o_sx;
  int dense;                  // o_sy == Wo * o_sx and o_sb == HoWo * o_sx: output row m starts at o_base + m * o_sx
};

constexpr int GROWS = 128, GSUP = 4;                       // rows per block, chunks per super-chunk
constexpr int GPLANE = GROWS * 16, GCHUNK = 4 * GPLANE, GBUF = GSUP * GCHUNK;   // 2 KiB, 8 KiB, 32 KiB

template <typename Tag> struct MmaG;
template <> struct MmaG<F32Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};
template <> struct MmaG<BF16Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
  }
};

template <> struct MmaG<F16Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) { acc = cp_mma16<true>(w, a, acc); }
};

template <typename Tag, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const GemmParams p) {
  constexpr bool HF = Tag::dtype == CP_F16;                        // IEEE-half rows and weights (keypoint side, no residual)
  if constexpr (HF) cp_f16_saturate_on();
  constexpr int E = Tag::E;
  constexpr int KCH = 4 * E;
  constexpr int ES = 16 / E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * GBUF

  // block -> (row block, channel block); channel blocks of one row block share a blockIdx % 8 label (one XCD's L2)
  const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nblk = slot % p.NB;
  const int mb = (slot / p.NB) * 8 + label;
  if (mb >= p.m_blocks) return;
  const int m0 = mb * GROWS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int g = nblk * 4 + wave;
  const bool wave_active = g < p.ngroups;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  // staging map: piece i = tid + 256k (k = 0..7): row = i >> 4, piece-in-row pr = i & 15 (chunk = pr >> 2, q = pr & 3)
  const int s_pr = tid & 15;
  uint32_t s_goff[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = (tid >> 4) + 16 * k;
    const int m = m0 + row;
    s_goff[k] = m < p.M ? (uint32_t)(m * p.in_cs + p.in_coff + s_pr * E) : 0xFFFFFFFFu;
  }
  const uint32_t s_lds0 = (uint32_t)((s_pr >> 2) * GCHUNK + (s_pr & 3) * GPLANE + (tid >> 4) * 16);   // + k*16 rows

  auto stage_load = [&](u32x4* v, int sc) {
    const int c0 = sc * GSUP * KCH;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = (s_goff[k] != 0xFFFFFFFFu) & (c0 + s_pr * E < p.Cin);
      const uint32_t off = ok ? (s_goff[k] + (uint32_t)c0) * ES : 0x80000000u;
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };
  auto stage_write = [&](const u32x4* v, int buf) {
#pragma unroll
    for (int k = 0; k < 8; ++k) *(u32x4*)(smem + buf * GBUF + s_lds0 + k * 16 * 16) = v[k];
  };

  f32x4 acc[8][2];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const uint32_t wbase = ((uint32_t)(wave_active ? g : 0) * p.nchunk * 2u * 64u + lane) * 16u;   // [g][chunk][nt][lane]
  auto w_load = [&](u32x4* w, int chunk) {
    const uint32_t off = wbase + (uint32_t)chunk * 2048u;      // chunks past the end are out of range -> zeros
    w[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
    w[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 1024u, 0, 0));
  };
  const uint32_t a_lane = (uint32_t)(q * GPLANE + x * 16);
  auto a_load = [&](u32x4* a, int buf, int cc) {
    const unsigned char* base = smem + buf * GBUF + cc * GCHUNK + a_lane;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) a[mt] = *(const u32x4*)(base + mt * 256);
  };
  auto mma = [&](const u32x4* a, const u32x4* w) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      MmaG<Tag>::run(w[0], a[mt], acc[mt][0]);
      MmaG<Tag>::run(w[1], a[mt], acc[mt][1]);
    }
  };

  u32x4 sv[8];
  stage_load(sv, 0);
  u32x4 w0[2], w1[2], w2[2], w3[2], aA[8], aB[8];
  w_load(w0, 0); w_load(w1, 1); w_load(w2, 2);
  stage_write(sv, 0);
  __syncthreads();

#define CP_STEP(CC, ACUR, ANEXT, WCUR, WNEXT)                            \
  w_load(WNEXT, sc * GSUP + CC + 3);                                     \
  if (CC < GSUP - 1) a_load(ANEXT, buf, CC + 1);                         \
  __builtin_amdgcn_sched_barrier(0);                                     \
  mma(ACUR, WCUR);                                                       \
  __builtin_amdgcn_sched_barrier(0);
  for (int sc = 0; sc < p.nsuper; ++sc) {
    const int buf = sc & 1;
    stage_load(sv, sc + 1);              // past the end: every piece out of range -> zeros (never consumed)
    a_load(aA, buf, 0);
    CP_STEP(0, aA, aB, w0, w3)
    CP_STEP(1, aB, aA, w1, w0)
    CP_STEP(2, aA, aB, w2, w1)
    CP_STEP(3, aB, aA, w3, w2)
    stage_write(sv, buf ^ 1);
    __syncthreads();
  }
#undef CP_STEP

  if (!wave_active) return;
  // ---- epilogue: lane (x, q): rows m0 + 16*mt + x, channels g*32 + 8q + {0..7}
  const int ch = g * 32 + q * 8;
  if (ch >= p.Cout) return;
  const bool hi_ok = ch + 4 < p.Cout;
  float sc8[8], sh8[8];
  {
    const f32x4 s0 = *(const f32x4*)(p.scale + ch), t0 = *(const f32x4*)(p.shift + ch);
    const f32x4 s1 = *(const f32x4*)(p.scale + ch + 4), t1 = *(const f32x4*)(p.shift + ch + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc8[j] = s0[j]; sh8[j] = t0[j]; sc8[4 + j] = s1[j]; sh8[4 + j] = t1[j]; }
  }
  long long o[8];
  bool ok[8];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + mt * 16 + x;
    ok[mt] = m < p.M;
    const int mm = ok[mt] ? m : 0;
    const int b = mm / p.HoWo;
    const int rem = mm - b * p.HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    o[mt] = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + ch;
  }
  float rvv[HAS_RES ? 8 : 1][8];
  if constexpr (HAS_RES) {                // all residual loads before the first store (res may alias out)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rvv[mt][j] = 0.f;
      if (!ok[mt]) continue;
      if (E == 4) {
        const f32x4 r0 = *(const f32x4*)((const float*)p.res + o[mt]);
#pragma unroll
        for (int j = 0; j < 4; ++j) rvv[mt][j] = r0[j];
        if (hi_ok) {
          const f32x4 r1 = *(const f32x4*)((const float*)p.res + o[mt] + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) rvv[mt][4 + j] = r1[j];
        }
      } else {
        const u32x2 r0 = *(const u32x2*)((const uint16_t*)p.res + o[mt]);
        rvv[mt][0] = __uint_as_float(r0.x << 16); rvv[mt][1] = __uint_as_float(r0.x & 0xffff0000u);
        rvv[mt][2] = __uint_as_float(r0.y << 16); rvv[mt][3] = __uint_as_float(r0.y & 0xffff0000u);
        if (hi_ok) {
          const u32x2 r1 = *(const u32x2*)((const uint16_t*)p.res + o[mt] + 4);
          rvv[mt][4] = __uint_as_float(r1.x << 16); rvv[mt][5] = __uint_as_float(r1.x & 0xffff0000u);
          rvv[mt][6] = __uint_as_float(r1.y << 16); rvv[mt][7] = __uint_as_float(r1.y & 0xffff0000u);
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    if (!ok[mt]) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = acc[mt][0][j] * sc8[j] + sh8[j]; v[4 + j] = acc[mt][1][j] * sc8[4 + j] + sh8[4 + j]; }
    if constexpr (HAS_RES) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rvv[mt][j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
    }
    if (E == 4) {
      *(f32x4*)((float*)p.out + o[mt]) = f32x4{v[0], v[1], v[2], v[3]};
      if (hi_ok) *(f32x4*)((float*)p.out + o[mt] + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else if (hi_ok) {
      const u32x4 pk = cp_pack8<HF>(v);
      if ((((uintptr_t)((uint16_t*)p.out + o[mt])) & 15u) == 0) *(u32x4*)((uint16_t*)p.out + o[mt]) = pk;
      else { *(u32x2*)((uint16_t*)p.out + o[mt]) = u32x2{pk.x, pk.y}; *(u32x2*)((uint16_t*)p.out + o[mt] + 4) = u32x2{pk.z, pk.w}; }
    } else {
      u32x2 pk; pk.x = cp_pack2<HF>(v[0], v[1]); pk.y = cp_pack2<HF>(v[2], v[3]);
      *(u32x2*)((uint16_t*)p.out + o[mt]) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-stationary persistent variant (bf16, K <= 256, no residual): the EdgeConv node GEMMs 256 -> 512 and the
// 256-wide MLP layers (M = B*N rows).  The kernel above re-streams a block's 64 KB weight slice from L2 for every
// 128-row block -- at M = 131072 that is as many bytes through the texture path as the activations themselves, and
// the 128x128 tile caps the arithmetic intensity at ~65 flop/B (measured 0.28 PFLOP/s, 1.7 TB/s algorithmic).  Here a
// wave owns 64 output channels and keeps ALL of their weight fragments in registers (8 chunks x 4 tiles = 128 VGPRs)
// while its block walks row tiles of 64: per tile only the 32 KB of activation rows are loaded (double-buffered LDS,
// next tile's loads in flight during the MFMAs, one barrier per tile) and every fragment read feeds 4 MFMAs.
// Column-group blocks of one row stream share a blockIdx % 8 label: the rows they both read are an L2 hit.
// Two shapes: K <= 256: 4 waves, a wave owns 64 channels (2 groups, 4 tiles), two blocks per CU; 256 < K <= 512: 8 waves, a
// wave owns 32 channels (1 group, 2 tiles: the same 128 weight VGPRs), one block per CU -- both cover 256 channels per block.
constexpr int WS_ROWS = 64;
constexpr int WS_PITCH = WS_ROWS * 16 + 16;                 // plane [row][16 B]; +16: consecutive planes shift one slot
constexpr int ws_buf(int NCH) { return NCH * 4 * WS_PITCH; }   // 33 280 B (K <= 256) / 66 560 B (K <= 512)

template <int NCH, int NG, bool HF = false>                  // K chunks resident per wave, 32-channel groups per wave; HF: IEEE half
__global__ __launch_bounds__(512 / NG, NG) void gemm_rows_ws_kernel(const GemmParams p, const int ncg, const int n_rt) {
  if constexpr (HF) cp_f16_saturate_on();
  constexpr int NT = 2 * NG;                                 // MFMA tiles per wave
  constexpr int NWAVE = 8 / NG, THREADS = 64 * NWAVE;
  constexpr int PPR = NCH * 4;                               // 16-byte pieces per row
  constexpr int RPP = THREADS / PPR;                         // rows staged per pass (8)
  constexpr int MTP = 4 / NG;                                // 16-row fragments per accumulator pass (2 or 4)
  constexpr int BUF = ws_buf(NCH);
  static_assert(RPP * 8 == WS_ROWS, "8 staging loads per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * BUF
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % ncg, stream = slot / ncg;
  const int nstreams = (gridDim.x >> 3) / ncg;               // row streams per XCD
  const int g0 = (cg * NWAVE + wave) * NG;                   // first of this wave's NG 32-channel groups

  // ---- resident weights: [group][chunk][nt][lane][16 B]; chunks >= nchunk and groups past the end stay zero
  u32x4 W[NCH][NT];
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int g = g0 + (t >> 1);
      W[kc][t] = u32x4{0u, 0u, 0u, 0u};
      if (kc < p.nchunk && g < p.ngroups) W[kc][t] = ((const u32x4*)p.w)[((size_t)(g * p.nchunk + kc) * 2 + (t & 1)) * 64 + lane];
    }

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  // staging: piece i = tid + THREADS k (k = 0..7): row = i / PPR, piece-in-row pr = i % PPR (= chunk * 4 + q)
  const int s_pr = tid % PPR, s_row = tid / PPR;             // rows s_row + RPP k
  auto stage_load = [&](u32x4* v, int rt) {
    const int m0 = rt * WS_ROWS;
    const bool pok = (rt < n_rt) & (s_pr * 8 < p.Cin);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int m = m0 + s_row + RPP * k;
      const uint32_t off = (pok & (m < p.M)) ? (uint32_t)(m * p.in_cs + p.in_coff + s_pr * 8) * 2u : 0x80000000u;
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };
  auto stage_write = [&](const u32x4* v, int buf) {
    unsigned char* dst = smem + buf * BUF + s_pr * WS_PITCH + s_row * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) *(u32x4*)(dst + k * RPP * 16) = v[k];
  };

  const int rt_step = nstreams * 8;
  int rt = stream * 8 + xcd;
  u32x4 sv[8];
  stage_load(sv, rt);
  stage_write(sv, 0);
  __syncthreads();
  int buf = 0;
  for (; rt < n_rt; rt += rt_step, buf ^= 1) {
    stage_load(sv, rt + rt_step);                            // past the end: zeros, never consumed
    const int m0 = rt * WS_ROWS;
#pragma unroll 1
    for (int mh = 0; mh < 4 / MTP; ++mh) {                   // passes of 16 MTP rows: bounds the live accumulators (32 VGPRs)
    f32x4 acc[MTP][NT];
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* ab = smem + buf * BUF + q * WS_PITCH + (mh * MTP * 16 + x) * 16;
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      if (kc < p.nchunk) {                                   // block-uniform
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
          const u32x4 a = *(const u32x4*)(ab + kc * 4 * WS_PITCH + mt * 256);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[mt][t] = cp_mma16<HF>(W[kc][t], a, acc[mt][t]);
        }
      }
    }
    // ---- epilogue: lane (x, q): row m0 + 16 MTP mh + 16 mt + x, channels (g0 + h) * 32 + 8 q + {0..7}
#pragma unroll
    for (int h = 0; h < NG; ++h) {
      if (g0 + h >= p.ngroups) continue;
      const int ch = (g0 + h) * 32 + q * 8;
      if (ch >= p.Cout) continue;
      const bool hi_ok = ch + 4 < p.Cout;
      const f32x4 s0 = *(const f32x4*)(p.scale + ch), t0 = *(const f32x4*)(p.shift + ch);
      const f32x4 s1 = *(const f32x4*)(p.scale + ch + 4), t1 = *(const f32x4*)(p.shift + ch + 4);
#pragma unroll
      for (int mt = 0; mt < MTP; ++mt) {
        const int m = m0 + mh * MTP * 16 + mt * 16 + x;
        if (m >= p.M) continue;
        uint16_t* op;
        if (p.dense) op = (uint16_t*)p.out + p.o_base + (long long)m * p.o_sx + ch;   // row-major output: no (b, y, x) split (2 integer divisions per fragment)
        else {
          const int b = m / p.HoWo;
          const int rem = m - b * p.HoWo;
          const int oy = rem / p.Wo;
          const int ox = rem - oy * p.Wo;
          op = (uint16_t*)p.out + p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + ch;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = acc[mt][2 * h][j] * s0[j] + t0[j]; v[4 + j] = acc[mt][2 * h + 1][j] * s1[j] + t1[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
        }
        if (hi_ok) {
          const u32x4 pk = cp_pack8<HF>(v);
          if ((((uintptr_t)op) & 15u) == 0) *(u32x4*)op = pk;
          else { *(u32x2*)op = u32x2{pk.x, pk.y}; *(u32x2*)(op + 4) = u32x2{pk.z, pk.w}; }
        } else {
          u32x2 pk; pk.x = cp_pack2<HF>(v[0], v[1]); pk.y = cp_pack2<HF>(v[2], v[3]);
          *(u32x2*)op = pk;
        }
      }
    }
    }
    stage_write(sv, buf ^ 1);
    __syncthreads();
  }
}

// ---- packing: [group g (32 ch)][chunk][nt][lane][16 B]; tile row i = 4*qr + reg of tile nt -> channel g*32 + 8*qr + 4*nt + reg
template <typename Tag>
__global__ void pack_gemm_weight_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int nchunk,
                                        const int32_t* __restrict__ row_map, size_t total) {
  constexpr int E = Tag::E;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % E);
  const int lane = (int)((i / E) % 64);
  size_t blk = i / (E * 64);
  const int nt = (int)(blk % 2); blk /= 2;
  const int c = (int)(blk % nchunk);
  const int g = (int)(blk / nchunk);
  const int row = lane & 15, kq = lane >> 4;
  const int n = g * 32 + (row >> 2) * 8 + nt * 4 + (row & 3);
  const int cin = c * (4 * E) + kq * E + e;
  float v = 0.f;
  if (n < Cout && cin < Cin) v = w[(size_t)n * Cin + cin];
  store_elem<Tag>(out, i, v);
}

extern "C" size_t cp_packed_gemm_weight_bytes(int dtype, int Cout, int cin_phys) {
  const int E = cp_chan_align(dtype), KCH = 4 * E;
  const size_t nchunk = ((size_t)cin_phys + KCH - 1) / KCH;
  return (((size_t)Cout + 31) / 32) * nchunk * 2 * 1024;
}

extern "C" int cp_pack_gemm_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || cin_phys < Cin) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;      // CP_F16: the fused MLP kernels' half images
  const int E = cp_chan_align(dtype);
  if (cin_phys % E || !cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int nchunk = (cin_phys + 4 * E - 1) / (4 * E);
  const size_t total = cp_packed_gemm_weight_bytes(dtype, Cout, cin_phys) / cp_elem_size(dtype);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(pack_gemm_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, nullptr, total);
  else if (dtype == CP_F16)
    CP_LAUNCH(pack_gemm_weight_kernel<F16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, nullptr, total);
  else
    CP_LAUNCH(pack_gemm_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, nullptr, total);
  return cp_check_launch();
}

extern "C" int cp_pack_item_gemm(int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed, CpPackItem* it) {
  if (!w || !packed || !it || Cout <= 0 || Cin <= 0 || cin_phys < Cin) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys % E || !cp_aligned16(packed)) return CP_ERR_ALIGN;
  for (int k = 0; k < 9; ++k) it->a[k] = 0;
  it->kind = CP_PACK_GEMM; it->src = w; it->dst = packed; it->row_map = nullptr;
  it->a[0] = Cout; it->a[1] = Cin; it->a[2] = (cin_phys + 4 * E - 1) / (4 * E);
  it->total = cp_packed_gemm_weight_bytes(dtype, Cout, cin_phys) / cp_elem_size(dtype);
  return CP_OK;
}

extern "C" int cp_gemm_rows(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                            const float* scale, const float* shift, const void* residual, void* out) {
  if (!d || !in || !packed_w || !scale || !shift || !out) return CP_ERR_INVALID;
  if ((d->dtype != CP_F32 && d->dtype != CP_BF16 && d->dtype != CP_F16) || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  if (d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->Ho != d->H || d->Wo != d->W || d->out_f32 || d->o_sc != 1)
    return CP_ERR_INVALID;
  if (d->dtype == CP_F16 && residual) return CP_ERR_INVALID;          // IEEE-half rows (keypoint side): in, weights and out in half, no residual
  const int E = cp_chan_align(d->dtype), es = cp_elem_size(d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return CP_ERR_INVALID;
  if (d->Cin <= 0 || d->Cin % E || d->in_coff % E || d->in_cstride % E || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (d->Cout <= 0 || d->Cout % 4) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift)) return CP_ERR_ALIGN;
  if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4)) return CP_ERR_ALIGN;
  if (((uintptr_t)out % (4 * es)) || (residual && ((uintptr_t)residual % (4 * es)))) return CP_ERR_ALIGN;
  const long long M = (long long)d->B * d->H * d->W;
  const long long in_bytes = M * d->in_cstride * es;
  if (in_bytes >= (1LL << 31) || M >= (1LL << 31)) return CP_ERR_RANGE;
  GemmParams p;
  p.in = in; p.w = packed_w; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
  p.M = (int)M; p.HoWo = d->H * d->W; p.Wo = d->W;
  p.Cin = d->Cin; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.nchunk = (d->Cin + 4 * E - 1) / (4 * E);
  p.nsuper = (p.nchunk + GSUP - 1) / GSUP;
  p.Cout = d->Cout; p.ngroups = (d->Cout + 31) / 32; p.NB = (p.ngroups + 3) / 4;
  p.m_blocks = (int)((M + GROWS - 1) / GROWS);
  p.act = d->act; p.slope = d->slope;
  p.in_bytes = (uint32_t)in_bytes;
  const size_t wb = cp_packed_gemm_weight_bytes(d->dtype, d->Cout, d->Cin);
  if (wb >= (1ull << 31)) return CP_ERR_RANGE;
  p.w_bytes = (uint32_t)wb;
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  p.dense = (d->o_sy == (long long)d->Wo * d->o_sx && d->o_sb == (long long)d->Ho * d->Wo * d->o_sx) ? 1 : 0;
  const unsigned grid = (unsigned)(((p.m_blocks + 7) / 8) * 8 * p.NB);
  hipStream_t st = (hipStream_t)stream;
  if ((d->dtype == CP_BF16 || d->dtype == CP_F16) && !residual && p.nchunk <= 16 && M >= 16384 && !cp_knob("CP_NO_GEMM_WS")) {
    static CpDeviceOnce once;
    const int dev = cp_current_device();
    CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)gemm_rows_ws_kernel<8, 2>, 2 * ws_buf(8)) &&
                                    cp_set_max_lds((const void*)gemm_rows_ws_kernel<16, 1>, 2 * ws_buf(16)) &&
                                    cp_set_max_lds((const void*)gemm_rows_ws_kernel<8, 2, true>, 2 * ws_buf(8)) &&
                                    cp_set_max_lds((const void*)gemm_rows_ws_kernel<16, 1, true>, 2 * ws_buf(16)));
    const int n_cu = cp_num_cus();
    if (n_cu <= 0) return CP_ERR_HIP;
    const bool deep = p.nchunk > 8;                           // 256 < K <= 512: 8-wave blocks, one per CU
    const int ncg = (p.ngroups + 7) / 8;                      // column groups of 256 channels
    const int n_rt = (int)((M + WS_ROWS - 1) / WS_ROWS);
    int per_xcd = (deep ? 1 : 2) * n_cu / 8;                  // resident blocks per XCD
    int nstreams = per_xcd / ncg > 0 ? per_xcd / ncg : 1;
    const int need = (n_rt + 7) / 8;                          // row tiles one XCD label owns
    if (nstreams > need) nstreams = need;
    if (d->dtype == CP_F16) {
      if (deep) CP_LAUNCH((gemm_rows_ws_kernel<16, 1, true>), dim3((unsigned)(8 * nstreams * ncg)), dim3(512), 2 * ws_buf(16), st, p, ncg, n_rt);
      else CP_LAUNCH((gemm_rows_ws_kernel<8, 2, true>), dim3((unsigned)(8 * nstreams * ncg)), dim3(256), 2 * ws_buf(8), st, p, ncg, n_rt);
    } else if (deep) CP_LAUNCH((gemm_rows_ws_kernel<16, 1>), dim3((unsigned)(8 * nstreams * ncg)), dim3(512), 2 * ws_buf(16), st, p, ncg, n_rt);
    else CP_LAUNCH((gemm_rows_ws_kernel<8, 2>), dim3((unsigned)(8 * nstreams * ncg)), dim3(256), 2 * ws_buf(8), st, p, ncg, n_rt);
    return cp_check_launch();
  }
  if (d->dtype == CP_F32) {
    if (residual) CP_LAUNCH((gemm_rows_kernel<F32Tag, true>), dim3(grid), dim3(256), 2 * GBUF, st, p);
    else CP_LAUNCH((gemm_rows_kernel<F32Tag, false>), dim3(grid), dim3(256), 2 * GBUF, st, p);
  } else if (d->dtype == CP_F16) {
    CP_LAUNCH((gemm_rows_kernel<F16Tag, false>), dim3(grid), dim3(256), 2 * GBUF, st, p);
  } else {
    if (residual) CP_LAUNCH((gemm_rows_kernel<BF16Tag, true>), dim3(grid), dim3(256), 2 * GBUF, st, p);
    else CP_LAUNCH((gemm_rows_kernel<BF16Tag, false>), dim3(grid), dim3(256), 2 * GBUF, st, p);
  }
  return cp_check_launch();
}
