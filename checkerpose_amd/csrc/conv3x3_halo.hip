// 3x3 / stride 1 / pad 1 convolution with an LDS-staged input halo tile (the decoder + HRNet body convs:
// > 85 % of the forward's dense FLOPs).
//
// Why a second conv kernel: in the generic implicit GEMM (conv_igemm.hip) every tap re-reads its activation
// fragments from L2 in fragment shape (16 pixel rows x 64 B per wave instruction), so a 3x3 conv moves 9x its
// input through the texture path and bf16 MFMA sits at ~17 % of peak.  Here a block stages the (8+2)x(16+2) pixel
// halo of ONE 64-byte channel chunk in LDS once and all 9 taps read their fragments from it:
//   * block = 8x16 output pixels x 128 output channels; 4 waves, wave w owns channels [32w, 32w+32) of ALL 128
//     pixels -> 8 pixel fragments (one per tile row) x 2 channel tiles, 16 accumulators.
//   * activations: LDS image [q = 16-byte piece 0..3][pixel 0..191][16 B]; lane (x = l&15, q = l>>4) of fragment
//     (row mt, tap r,s) reads pixel (mt+r)*18 + s + x of plane q -> every 16-lane ds_read_b128 group covers 16
//     distinct 16-byte slots mod 16 (plane pitch 192 pixels == 0 mod 16): conflict-free.
//   * weights: NOT through LDS -- the 4 waves own disjoint channels, so each packed fragment (one coalesced
//     1 KiB buffer load) is read by exactly one wave: no redundancy, and the LDS port is left to the activations.
//   * epilogue: packed-weight rows are permuted so lane q ends with channels 8q..8q+7 of its 32-channel group
//     (tile nt -> channels 8q+4nt..+3): one 16-byte (bf16) / two 16-byte (f32) stores per pixel, 64/128
//     contiguous bytes per pixel across the four q lanes.
// One barrier per channel chunk (two LDS buffers); the next chunk's halo is prefetched into registers under the
// 9 x 16 MFMAs (bf16) / 9 x 64 MFMAs (f32) of the current one.
#include <stdlib.h>

#include "common.h"
#include <cstring>

struct HaloParams {
  const void* in; const void* w; const float* scale; const float* shift; const void* res; void* out;
  int B, H, W, Cin, in_cs, in_coff, nchunk, Cout, ngroups, NB;
  int tiles_x, tiles_y, total_tiles;
  int act; float slope;
  uint32_t in_bytes, w_bytes;
  long long o_base, o_sb, o_sy, o_sx;
  int Hs, Ws; float up_sy, up_sx;     // fused bilinear x2 (halo4 UP variant): source size, align_corners scales
  const float* seg_w; const float* seg_b; float* seg_out; int seg_S;   // fused 1x1 head (halo4 SEG variant): [S][Cout] weights, NCHW fp32 out
};

constexpr int HTH = 8, HTW = 16, HPW = HTW + 2, HPH = HTH + 2, HNPIX = 192;   // 180 halo pixels, plane padded to 192
constexpr int HPLANE = HNPIX * 16, HBUF = 4 * HPLANE;                          // bytes
// fused bilinear x2 (align_corners, scale < 1/2): the 10 x 18 halo of the upsampled map comes from at most 7 x 11 source pixels
constexpr int USRH = 7, USRW = 11, USPIX = 80, USBUF = USPIX * 64;             // source tile: [pixel][16-byte piece 0..3]

template <typename Tag> struct MmaH;
template <> struct MmaH<F32Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};
template <> struct MmaH<BF16Tag> {
  static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
  }
};

template <typename Tag, bool HAS_RES>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const HaloParams p) {
  constexpr int E = Tag::E;
  constexpr int KCH = 4 * E;
  constexpr int ES = 16 / E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * HBUF

  // ---- block -> (spatial tile, channel block).  All channel blocks of a tile run on ONE XCD (same blockIdx % 8
  // label, consecutive slots) so the halo they all stage is an L2 hit after the first.
  const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nblk = slot % p.NB;
  const int t = label * ((p.total_tiles + 7) >> 3) + slot / p.NB;   // an XCD owns a contiguous run of tiles: halo overlaps hit its L2
  if (t >= p.total_tiles) return;                      // whole block, before any barrier
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = t / tpi;
  const int trem = t - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * HTH, x0 = tx * HTW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int g = nblk * 4 + wave;                       // 32-channel group of this wave
  const bool wave_active = g < p.ngroups;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- staging map: thread handles pieces i = tid + 256k (k = 0..2), piece = (halo pixel hp = i>>2, q = i&3)
  uint32_t s_goff[3];      // element offset of (pixel, channel q*E) without the chunk offset; 0xFFFFFFFF = outside
  uint32_t s_lds[3];       // LDS byte offset inside a buffer
  int s_cq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + 256 * k;
    const int hp = i >> 2, pq = i & 3;
    const int py = hp / HPW, px = hp - py * HPW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = (hp < HPH * HPW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    s_goff[k] = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pq * E) : 0xFFFFFFFFu;
    s_lds[k] = (uint32_t)(pq * HPLANE + (hp < HNPIX ? hp : 0) * 16);
    s_cq[k] = pq * E;
  }
  const bool s_do[3] = {true, true, (tid + 512) < HNPIX * 4};   // pieces 720..767 pad the planes (never read)

  auto stage_load = [&](u32x4* v, int c) {
    const int c0 = c * KCH;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool ok = (s_goff[k] != 0xFFFFFFFFu) & (c0 + s_cq[k] < p.Cin);
      const uint32_t off = ok ? (s_goff[k] + (uint32_t)c0) * ES : 0x80000000u;
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };
  auto stage_write = [&](const u32x4* v, int buf) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (s_do[k]) *(u32x4*)(smem + buf * HBUF + s_lds[k]) = v[k];
  };

  f32x4 acc[HTH][2];
#pragma unroll
  for (int mt = 0; mt < HTH; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // weight fragments of (group g, chunk c, tap, nt): [g][c][tap][nt][lane][16 B]
  const uint32_t wbase = ((uint32_t)(wave_active ? g : 0) * p.nchunk * 18u * 64u + lane) * 16u;
  auto w_load = [&](u32x4* w, int c, int tap) {
    const uint32_t off = wbase + ((uint32_t)c * 18u + (uint32_t)tap * 2u) * 1024u;
    w[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
    w[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 1024u, 0, 0));
  };
  const uint32_t a_lane = (uint32_t)(q * HPLANE + x * 16);
  auto a_load = [&](u32x4* a, int buf, int tap) {          // 8 pixel-row fragments of one tap from the LDS halo
    const int r = tap / 3, s = tap - 3 * r;
    const unsigned char* base = smem + buf * HBUF + a_lane + (r * HPW + s) * 16;
#pragma unroll
    for (int mt = 0; mt < HTH; ++mt) a[mt] = *(const u32x4*)(base + mt * HPW * 16);
  };
  auto mma = [&](const u32x4* a, const u32x4* w) {
#pragma unroll
    for (int mt = 0; mt < HTH; ++mt) {
      MmaH<Tag>::run(w[0], a[mt], acc[mt][0]);
      MmaH<Tag>::run(w[1], a[mt], acc[mt][1]);
    }
  };

  // ---- prologue: chunk 0 into LDS buffer 0.  Software pipeline per tap (pinned with sched_barrier, hipcc
  // otherwise sinks every prefetch to just before its use):
  //   weights  : TWO taps ahead in three rotating register sets (9 taps = 3 x 3 -> seamless across chunks);
  //              a fragment load has ~2 x 16 MFMAs to come back from L2
  //   halo rows: ONE tap ahead in two register sets (ds_read latency hides under the current tap's MFMAs)
  u32x4 sv[3];
  stage_load(sv, 0);
  u32x4 w0[2], w1[2], w2[2], aA[HTH], aB[HTH];
  w_load(w0, 0, 0);
  w_load(w1, 0, 1);
  stage_write(sv, 0);
  __syncthreads();

#define CP_TAP(T, ACUR, ANEXT, WCUR, WNEXT, CN, TN)                      \
  w_load(WNEXT, CN, TN);                                                 \
  if (T < 8) a_load(ANEXT, buf, T + 1);                                  \
  __builtin_amdgcn_sched_barrier(0);                                     \
  mma(ACUR, WCUR);                                                       \
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < p.nchunk; ++c) {
    const int buf = c & 1;
    stage_load(sv, c + 1);               // unconditional: past the last chunk every piece is out of range -> zeros
    a_load(aA, buf, 0);
    CP_TAP(0, aA, aB, w0, w2, c, 2)
    CP_TAP(1, aB, aA, w1, w0, c, 3)
    CP_TAP(2, aA, aB, w2, w1, c, 4)
    CP_TAP(3, aB, aA, w0, w2, c, 5)
    CP_TAP(4, aA, aB, w1, w0, c, 6)
    CP_TAP(5, aB, aA, w2, w1, c, 7)
    CP_TAP(6, aA, aB, w0, w2, c, 8)
    CP_TAP(7, aB, aA, w1, w0, c + 1, 0)  // next chunk's taps 0, 1 (out of range past the end -> zeros)
    CP_TAP(8, aA, aB, w2, w1, c + 1, 1)
    stage_write(sv, buf ^ 1);
    __syncthreads();
  }
#undef CP_TAP

  if (!wave_active) return;
  // ---- epilogue: lane (x, q) holds, per tile row mt, pixel (y0+mt, x0+x) x channels g*32 + 8q + {0..7}
  const int ch = g * 32 + q * 8;
  if (ch >= p.Cout) return;
  const bool hi_ok = ch + 4 < p.Cout;                  // Cout is a multiple of 4: second half may be absent
  const int ox = x0 + x;
  float sc[8], sh[8];
  {
    const f32x4 s0 = *(const f32x4*)(p.scale + ch), t0 = *(const f32x4*)(p.shift + ch);
    const f32x4 s1 = *(const f32x4*)(p.scale + ch + 4), t1 = *(const f32x4*)(p.shift + ch + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = s0[j]; sh[j] = t0[j]; sc[4 + j] = s1[j]; sh[4 + j] = t1[j]; }
  }
  if (ox >= p.W) return;
  // all residual loads first (res may alias out: a load after a store would serialise behind it)
  float rvv[HAS_RES ? HTH : 1][8];   // only the residual instantiation pays the 64 registers
  if constexpr (HAS_RES) {
#pragma unroll
    for (int mt = 0; mt < HTH; ++mt) {
      const int oy = y0 + mt;
#pragma unroll
      for (int j = 0; j < 8; ++j) rvv[mt][j] = 0.f;
      if (oy >= p.H) continue;
      const long long o = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + ch;
      if (E == 4) {
        const f32x4 r0 = *(const f32x4*)((const float*)p.res + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) rvv[mt][j] = r0[j];
        if (hi_ok) {
          const f32x4 r1 = *(const f32x4*)((const float*)p.res + o + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) rvv[mt][4 + j] = r1[j];
        }
      } else {
        const u32x2 r0 = *(const u32x2*)((const uint16_t*)p.res + o);
        rvv[mt][0] = __uint_as_float(r0.x << 16); rvv[mt][1] = __uint_as_float(r0.x & 0xffff0000u);
        rvv[mt][2] = __uint_as_float(r0.y << 16); rvv[mt][3] = __uint_as_float(r0.y & 0xffff0000u);
        if (hi_ok) {
          const u32x2 r1 = *(const u32x2*)((const uint16_t*)p.res + o + 4);
          rvv[mt][4] = __uint_as_float(r1.x << 16); rvv[mt][5] = __uint_as_float(r1.x & 0xffff0000u);
          rvv[mt][6] = __uint_as_float(r1.y << 16); rvv[mt][7] = __uint_as_float(r1.y & 0xffff0000u);
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < HTH; ++mt) {
    const int oy = y0 + mt;
    if (oy >= p.H) continue;
    const long long o = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + ch;
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = acc[mt][0][j] * sc[j] + sh[j]; v[4 + j] = acc[mt][1][j] * sc[4 + j] + sh[4 + j]; }
    if constexpr (HAS_RES) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rvv[HAS_RES ? mt : 0][j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
    }
    if (E == 4) {
      *(f32x4*)((float*)p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
      if (hi_ok) *(f32x4*)((float*)p.out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      if (hi_ok) {
        u32x4 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
        if ((((uintptr_t)((uint16_t*)p.out + o)) & 15u) == 0) *(u32x4*)((uint16_t*)p.out + o) = pk;
        else { *(u32x2*)((uint16_t*)p.out + o) = u32x2{pk.x, pk.y}; *(u32x2*)((uint16_t*)p.out + o + 4) = u32x2{pk.z, pk.w}; }
      } else {
        u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
        *(u32x2*)((uint16_t*)p.out + o) = pk;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Wide variant of the N-split halo kernel for Cout % 64 == 0, Cout >= 256 (the decoder convs = 80 % of all FLOPs):
// each wave owns 64 channels (4 tiles) of all 128 pixels, the block covers 256 channels.  Compared with the
// 32-channel-per-wave kernel every halo row fragment read from LDS now feeds 4 MFMAs instead of 2 (half the LDS
// traffic per MFMA) and the halo is staged once per 256 instead of per 128 output channels.  The 128 accumulator
// registers leave room for only half-tap operand sets, so the halo rows are pipelined in two halves (rows 0-3 /
// 4-7): each half's ds_reads are issued under the other half's 16 MFMAs.  Rows are permuted so lane q ends with
// channels 16q..16q+15 of its 64-channel group: two 16-byte stores per pixel, 128 contiguous bytes across q.
//
// UP = true: the conv's input is bilinear_x2(align_corners)(p.in) and the upsampled tensor never exists (pipeline.py:199-200,
// up_net[1..2] conv1): per channel chunk the block loads the <= 7 x 11 SOURCE pixels under its halo (2 loads per thread
// instead of 3) two chunks ahead, parks them in a small LDS tile, and interpolates the next chunk's 10 x 18 halo from it
// LDS -> LDS between the taps of the current chunk (same arithmetic, cp_bilerp, as the stand-alone kernel: same bits).
//
// SEG = true (Cout == 256, one channel block): the 1x1 conv head that reads this conv's output (seg_block, pipeline.py:349,383:
// Conv2d(256 -> 2) + bias on the last decoder map, NCHW fp32 out) rides in the epilogue -- every lane dots the 16 activated,
// storage-rounded channels of its pixel with the head's weights, the four lane groups of a pixel meet by shuffle, the four
// waves (64 channels each) through LDS: the 537 MB re-read of the map by a separate launch disappears.
template <typename Tag, bool HAS_RES, bool UP = false, bool SEG = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo4_kernel(const HaloParams p) {
  constexpr int E = Tag::E;
  constexpr int KCH = 4 * E;
  constexpr int ES = 16 / E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * HBUF (+ 2 * USBUF)
  unsigned char* const sSrc = smem + 2 * HBUF;

  const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nblk = slot % p.NB;                       // 256-channel block
  const int t = label * ((p.total_tiles + 7) >> 3) + slot / p.NB;   // an XCD owns a contiguous run of tiles: halo overlaps hit its L2
  if (t >= p.total_tiles) return;
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = t / tpi;
  const int trem = t - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * HTH, x0 = tx * HTW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int g = nblk * 4 + wave;                      // 64-channel group of this wave (always < Cout / 64)

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  uint32_t s_goff[3], s_lds[3];
  int s_cq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + 256 * k;
    const int hp = i >> 2, pq = i & 3;
    const int py = hp / HPW, px = hp - py * HPW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = (hp < HPH * HPW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    s_goff[k] = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pq * E) : 0xFFFFFFFFu;
    s_lds[k] = (uint32_t)(pq * HPLANE + hp * 16);
    s_cq[k] = pq * E;
  }
  auto stage_load = [&](u32x4* v, int c) {
    const int c0 = c * KCH;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool ok = (s_goff[k] != 0xFFFFFFFFu) & (c0 + s_cq[k] < p.Cin);
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (s_goff[k] + (uint32_t)c0) * ES : 0x80000000u, 0, 0));
    }
  };
  auto stage_write = [&](const u32x4* v, int buf) {
#pragma unroll
    for (int k = 0; k < 3; ++k) *(u32x4*)(smem + buf * HBUF + s_lds[k]) = v[k];
  };

  // ---- UP: source-tile loads (piece i = tid + 256 k -> source pixel i >> 2, 16-byte piece i & 3) and, per halo piece this
  // thread produces, its 4 neighbours' place in the source tile + the two lerp weights
  uint32_t u_goff[2], m_off[3];
  float m_lx[3], m_ly[3];
  if constexpr (UP) {
    int sy0, sx0, st_; float fr_;                              // source tile origin = left / top neighbour of the first in-image halo pixel
    cp_up_coord(p.up_sy, y0 > 0 ? y0 - 1 : 0, p.Hs, sy0, st_, fr_);
    cp_up_coord(p.up_sx, x0 > 0 ? x0 - 1 : 0, p.Ws, sx0, st_, fr_);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + 256 * k;
      const int sp = i >> 2, pq = i & 3;
      const int ry = sp / USRW, rx = sp - ry * USRW;
      const int yy = sy0 + ry, xx = sx0 + rx;
      const bool ok = (sp < USRH * USRW) & (yy < p.Hs) & (xx < p.Ws);
      u_goff[k] = ok ? (uint32_t)(((b * p.Hs + yy) * p.Ws + xx) * p.in_cs + p.in_coff + pq * E) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = tid + 256 * k;
      const int hp = i >> 2, pq = i & 3;
      const int py = hp / HPW, px = hp - py * HPW;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = (hp < HPH * HPW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
      int iy0, ix0, my, mx;
      cp_up_coord(p.up_sy, gy, p.Hs, iy0, my, m_ly[k]);
      cp_up_coord(p.up_sx, gx, p.Ws, ix0, mx, m_lx[k]);
      m_off[k] = ok ? (uint32_t)((((iy0 - sy0) * USRW + (ix0 - sx0)) * 4 + pq) * 16) | (uint32_t)(mx | (my << 1) | 4) : 0u;
    }
  }
  auto src_load = [&](u32x4* v, int c) {
    const int c0 = c * KCH;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool ok = (u_goff[k] != 0xFFFFFFFFu) & (c0 + (tid & 3) * E < p.Cin);
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (u_goff[k] + (uint32_t)c0) * ES : 0x80000000u, 0, 0));
    }
  };
  auto src_write = [&](const u32x4* v, int sb) {
    *(u32x4*)(sSrc + sb * USBUF + tid * 16) = v[0];
    if (tid < USPIX * 4 - 256) *(u32x4*)(sSrc + sb * USBUF + (tid + 256) * 16) = v[1];
  };
  auto interp = [&](int k, int sb, int hb) {          // halo piece tid + 256 k of buffer hb from source tile sb
    const uint32_t mo = m_off[k];
    const unsigned char* s00 = sSrc + sb * USBUF + (mo & ~15u);
    const uint32_t dx = (mo & 1u) ? 64u : 0u, dy = (mo & 2u) ? (uint32_t)(USRW * 64) : 0u;
    const float lx1 = m_lx[k], lx0 = cp_one_minus(lx1), ly1 = m_ly[k], ly0 = cp_one_minus(ly1);
    const bool in_img = (mo & 4u) != 0u;              // outside the upsampled image: the conv's zero padding
    unsigned char* dst = smem + hb * HBUF + s_lds[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                      // one dword at a time: a handful of live registers beside the 128 accumulators
      const uint32_t r00 = *(const uint32_t*)(s00 + 4 * j), r01 = *(const uint32_t*)(s00 + dx + 4 * j);
      const uint32_t r10 = *(const uint32_t*)(s00 + dy + 4 * j), r11 = *(const uint32_t*)(s00 + dy + dx + 4 * j);
      uint32_t o;
      if constexpr (E == 4) {
        o = __float_as_uint(cp_bilerp(__uint_as_float(r00), __uint_as_float(r01), __uint_as_float(r10), __uint_as_float(r11), lx0, lx1, ly0, ly1));
      } else {
        const float lo = cp_bilerp(__uint_as_float(r00 << 16), __uint_as_float(r01 << 16), __uint_as_float(r10 << 16), __uint_as_float(r11 << 16), lx0, lx1, ly0, ly1);
        const float hi = cp_bilerp(__uint_as_float(r00 & 0xffff0000u), __uint_as_float(r01 & 0xffff0000u), __uint_as_float(r10 & 0xffff0000u),
                                   __uint_as_float(r11 & 0xffff0000u), lx0, lx1, ly0, ly1);
        o = pack_bf16x2(lo, hi);
      }
      *(uint32_t*)(dst + 4 * j) = in_img ? o : 0u;
    }
  };

  f32x4 acc[HTH][4];
#pragma unroll
  for (int mt = 0; mt < HTH; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weights: [g64][chunk][tap][nt 0..3][lane][16 B]
  const uint32_t wbase = ((uint32_t)g * p.nchunk * 36u * 64u + lane) * 16u;
  auto w_load = [&](u32x4* w, int c, int tap) {
    const uint32_t off = wbase + ((uint32_t)c * 36u + (uint32_t)tap * 4u) * 1024u;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w[nt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + nt * 1024u, 0, 0));
  };
  const uint32_t a_lane = (uint32_t)(q * HPLANE + x * 16);
  auto a_load = [&](u32x4* a, int buf, int tap, int half) {      // rows 4*half .. 4*half+3 of one tap
    const int r = tap / 3, s2 = tap - 3 * r;
    const unsigned char* base = smem + buf * HBUF + a_lane + ((r + 4 * half) * HPW + s2) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const u32x4*)(base + i * HPW * 16);
  };
  auto mma = [&](const u32x4* a, const u32x4* w, int half) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) MmaH<Tag>::run(w[nt], a[i], acc[4 * half + i][nt]);
  };

  u32x4 sv[3];
  u32x4 wA[4], wB[4], a0[4], a1[4];
  if constexpr (UP) {
    src_load(sv, 0);
    w_load(wA, 0, 0);
    src_write(sv, 0);
    src_load(sv, 1);
    __syncthreads();
    interp(0, 0, 0); interp(1, 0, 0); interp(2, 0, 0);
    src_write(sv, 1);
  } else {
    stage_load(sv, 0);
    w_load(wA, 0, 0);
    stage_write(sv, 0);
  }
  __syncthreads();

  // weights one tap (32 MFMAs, ~512 cycles) ahead in two sets; a third set (two taps ahead) measured no faster
#define CP_TAP4(T, WCUR, WNEXT, CN, TN)                                                 \
  a_load(a1, buf, T, 1); w_load(WNEXT, CN, TN); __builtin_amdgcn_sched_barrier(0);       \
  mma(a0, WCUR, 0); __builtin_amdgcn_sched_barrier(0);                                   \
  if (T < 8) a_load(a0, buf, T + 1, 0);                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  mma(a1, WCUR, 1); __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < p.nchunk; ++c) {
    const int buf = c & 1;
    if constexpr (UP) src_load(sv, c + 2); else stage_load(sv, c + 1);
    a_load(a0, buf, 0, 0);
    CP_TAP4(0, wA, wB, c, 1)
    CP_TAP4(1, wB, wA, c, 2)
    if constexpr (UP) { __builtin_amdgcn_sched_barrier(0); interp(0, buf ^ 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }    // chunk c+1's halo from ITS source tile
    CP_TAP4(2, wA, wB, c, 3)
    CP_TAP4(3, wB, wA, c, 4)
    CP_TAP4(4, wA, wB, c, 5)
    if constexpr (UP) { __builtin_amdgcn_sched_barrier(0); interp(1, buf ^ 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
    CP_TAP4(5, wB, wA, c, 6)
    CP_TAP4(6, wA, wB, c, 7)
    CP_TAP4(7, wB, wA, c, 8)
    if constexpr (UP) { __builtin_amdgcn_sched_barrier(0); interp(2, buf ^ 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
    CP_TAP4(8, wA, wB, c + 1, 0)          // next chunk's tap 0 (out of range past the end -> zeros)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wA[nt] = wB[nt];
    if constexpr (UP) src_write(sv, buf); else stage_write(sv, buf ^ 1);   // UP: chunk c+2's source tile (tile c's is spent)
    __syncthreads();
  }
#undef CP_TAP4

  // ---- epilogue: lane (x, q): pixel (y0+mt, x0+x), channels g*64 + 16q + 4nt + {0..3}
  const int ch = g * 64 + q * 16;
  const int ox = x0 + x;
  if constexpr (!SEG) { if (ox >= p.W) return; }
  const bool xok = SEG ? ox < p.W : true;              // SEG: every lane stays for the shuffles and the barrier
  constexpr int SMAX = 2;
  float segw[SEG ? SMAX : 1][16], segp[SEG ? 8 : 1][SEG ? SMAX : 1];
  if constexpr (SEG) {
#pragma unroll
    for (int s_ = 0; s_ < SMAX; ++s_) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 w4 = s_ < p.seg_S ? *(const f32x4*)(p.seg_w + (size_t)s_ * p.Cout + ch + 4 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) segw[s_][4 * k + j] = w4[j];
      }
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) segp[mt][s_] = 0.f;
    }
  }
  float sc[16], sh[16];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 s4 = *(const f32x4*)(p.scale + ch + 4 * nt), t4 = *(const f32x4*)(p.shift + ch + 4 * nt);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[4 * nt + j] = s4[j]; sh[4 * nt + j] = t4[j]; }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {              // residual handled 4 rows at a time (register budget)
    u32x4 rr[HAS_RES ? 4 : 1][E == 4 ? 4 : 2];
    if constexpr (HAS_RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = y0 + 4 * half + i;
        const long long o = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + ch;
#pragma unroll
        for (int k = 0; k < (E == 4 ? 4 : 2); ++k) {
          rr[i][k] = u32x4{0u, 0u, 0u, 0u};
          if (oy < p.H) rr[i][k] = E == 4 ? *(const u32x4*)((const float*)p.res + o + 4 * k) : *(const u32x4*)((const uint16_t*)p.res + o + 8 * k);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = 4 * half + i;
      const int oy = y0 + mt;
      if (oy >= p.H) continue;
      const long long o = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + ch;
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mt][nt][j] * sc[4 * nt + j] + sh[4 * nt + j];
      if constexpr (HAS_RES) {
        if (E == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[4 * k + 0] += __uint_as_float(rr[i][k].x); v[4 * k + 1] += __uint_as_float(rr[i][k].y);
            v[4 * k + 2] += __uint_as_float(rr[i][k].z); v[4 * k + 3] += __uint_as_float(rr[i][k].w);
          }
        } else {
          float r8[8];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            Vec16<BF16Tag>::unpack(rr[i][k], r8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * k + j] += r8[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
      }
      if (E == 4) {
        if (xok) {
#pragma unroll
          for (int k = 0; k < 4; ++k) *(f32x4*)((float*)p.out + o + 4 * k) = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        }
      } else {
        const u32x4 pk0 = Vec16<BF16Tag>::pack(v), pk1 = Vec16<BF16Tag>::pack(v + 8);
        if (xok) {
          *(u32x4*)((uint16_t*)p.out + o) = pk0;
          *(u32x4*)((uint16_t*)p.out + o + 8) = pk1;
        }
        if constexpr (SEG) { Vec16<BF16Tag>::unpack(pk0, v); Vec16<BF16Tag>::unpack(pk1, v + 8); }   // the head reads the STORED values
      }
      if constexpr (SEG) {
#pragma unroll
        for (int s_ = 0; s_ < SMAX; ++s_) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) a += v[j] * segw[s_][j];
          segp[mt][s_] = a;
        }
      }
    }
  }
  if constexpr (SEG) {
    float* const red = (float*)smem;                    // [wave][row][x][s]: the halo buffers are spent (last barrier of the loop)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int s_ = 0; s_ < SMAX; ++s_) {
        float a = segp[mt][s_];
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (q == 0) red[((wave * 8 + mt) * 16 + x) * SMAX + s_] = a;
      }
    __syncthreads();
    const int s_ = tid & 1, px = tid >> 1;              // 128 pixels x 2 outputs = 256 threads
    const int mt = px >> 4, xx = px & 15;
    const int oy = y0 + mt, oxx = x0 + xx;
    if (s_ < p.seg_S && oy < p.H && oxx < p.W) {
      float a = p.seg_b[s_];
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) a += red[((wv * 8 + mt) * 16 + xx) * SMAX + s_];
      p.seg_out[(((size_t)b * p.seg_S + s_) * p.H + oy) * p.W + oxx] = a;
    }
  }
}

// packing for the wide variant: [g64][chunk][tap][nt 0..3][lane][16 B]; tile row i = 4*qr + reg of tile nt is output
// channel g*64 + 16*qr + 4*nt + reg.
template <typename Tag>
__global__ void pack_halo4_weight_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int nchunk,
                                         size_t total) {
  constexpr int E = Tag::E;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % E);
  const int lane = (int)((i / E) % 64);
  size_t blk = i / (E * 64);
  const int nt = (int)(blk % 4); blk /= 4;
  const int tap = (int)(blk % 9); blk /= 9;
  const int c = (int)(blk % nchunk);
  const int g = (int)(blk / nchunk);
  const int row = lane & 15, kq = lane >> 4;
  const int n = g * 64 + (row >> 2) * 16 + nt * 4 + (row & 3);
  const int cin = c * (4 * E) + kq * E + e;
  float v = 0.f;
  if (n < Cout && cin < Cin) v = w[((size_t)n * Cin + cin) * 9 + tap];
  store_elem<Tag>(out, i, v);
}

static inline bool halo_wide(int Cout) { return Cout >= 256 && Cout % 256 == 0 && !cp_knob("CP_NO_HALO4"); }

// ------------------------------------------------------------------------------------------------------------------
// Small-Cout variant (Cout <= 80: the HRNet 18/36/72-channel body convs, 64-channel layer1/stem convs).  These
// layers are HBM/texture-path bound, not MFMA bound, and a 32-channel-per-wave split would idle most waves, so:
//   * the 4 waves split the PIXELS (wave w owns tile rows 2w, 2w+1) and every wave computes all NT = ceil(Cout/16)
//     channel tiles;
//   * both operands go through LDS: the activation halo as above and the chunk's 9*NT weight fragments (linear copy
//     of the packed image), so the texture path only sees each byte once per block;
//   * packed rows are permuted so lane q ends with channels [4*NT*q, 4*NT*(q+1)): the four q lanes of a pixel write
//     its whole channel vector contiguously and the 16 pixels of a fragment are adjacent in memory.
// KS = 2: the k = 2 / stride 1 / pad 1 conv of Index2Feat_module's patch_generator (pipeline.py:144-145,156; output (H + 1) x (W + 1)) --
// the same tile walk over the larger output grid, four taps of the same staged halo (its last row / column is input row H / column W:
// zero padding); the generic kernel read every pixel row four times through the texture path (10 % of any roof at N = 4096).
template <typename Tag, int NT, int KS = 3>
__global__ __launch_bounds__(256) void conv3x3_halo_s_kernel(const HaloParams p) {
  if (KS == 2 && p.seg_S == CP_F16) cp_f16_saturate_on();      // half output rows saturate at +-65504 (common.h)
  constexpr int E = Tag::E;
  constexpr int KCH = 4 * E;
  constexpr int ES = 16 / E;
  constexpr int NTAP = KS * KS;
  constexpr int WPIECES = NTAP * NT * 64;              // 16-byte pieces of one chunk's weights
  constexpr int WITER = (WPIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [A: HBUF][W: 9*NT KiB]
  unsigned char* const sA = smem;
  unsigned char* const sW = smem + HBUF;

  const int t = blockIdx.x;                            // one block per spatial tile (no channel blocks)
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = t / tpi;
  const int trem = t - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * HTH, x0 = tx * HTW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  uint32_t s_goff[3], s_lds[3];
  int s_cq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + 256 * k;
    const int hp = i >> 2, pq = i & 3;
    const int py = hp / HPW, px = hp - py * HPW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = (hp < HPH * HPW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    s_goff[k] = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pq * E) : 0xFFFFFFFFu;
    s_lds[k] = (uint32_t)(pq * HPLANE + hp * 16);
    s_cq[k] = pq * E;
  }

  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t a_lane = (uint32_t)(q * HPLANE + (2 * wave * HPW + x) * 16);
  // residual prefetch: issued BEFORE the main loop so its HBM latency overlaps the staging + MFMA work instead of
  // adding a second serial memory round trip in the epilogue (these layers are latency-bound).  Also satisfies the
  // "all residual loads before the first store" rule (res may alias out).
  const int OH = p.H + (3 - KS), OW = p.W + (3 - KS);             // output extent (k = 2, pad 1: one more row / column than the input)
  const int ox = x0 + x;
  const int chq = q * 4 * NT;
  long long obase[2];
  bool rok[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int oy = y0 + 2 * wave + mt;
    rok[mt] = (oy < OH) & (ox < OW);
    obase[mt] = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + chq;
  }
  f32x4 rv[2][NT];
  if (p.res) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        rv[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rok[mt] && chq + nt * 4 < p.Cout) {
          if (E == 4) rv[mt][nt] = *(const f32x4*)((const float*)p.res + obase[mt] + nt * 4);
          else {
            const u32x2 r2 = *(const u32x2*)((const uint16_t*)p.res + obase[mt] + nt * 4);
            rv[mt][nt] = f32x4{__uint_as_float(r2.x << 16), __uint_as_float(r2.x & 0xffff0000u),
                               __uint_as_float(r2.y << 16), __uint_as_float(r2.y & 0xffff0000u)};
          }
        }
      }
  }
  for (int c = 0; c < p.nchunk; ++c) {
    // ---- global -> registers (activation halo pieces + this chunk's weight image)
    const int c0 = c * KCH;
    u32x4 sv[3], wv[WITER];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool ok = (s_goff[k] != 0xFFFFFFFFu) & (c0 + s_cq[k] < p.Cin);
      const uint32_t off = ok ? (s_goff[k] + (uint32_t)c0) * ES : 0x80000000u;
      sv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < WITER; ++k) {
      const int i = tid + 256 * k;
      const uint32_t off = i < WPIECES ? ((uint32_t)c * WPIECES + (uint32_t)i) * 16u : 0x80000000u;
      wv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
    }
    if (c > 0) __syncthreads();                        // everyone is done reading the previous chunk's LDS image
#pragma unroll
    for (int k = 0; k < 3; ++k) *(u32x4*)(sA + s_lds[k]) = sv[k];
#pragma unroll
    for (int k = 0; k < WITER; ++k) {
      const int i = tid + 256 * k;
      if (i < WPIECES) *(u32x4*)(sW + i * 16) = wv[k];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int r = tap / KS, s2 = tap - KS * r;
      const unsigned char* ab = sA + a_lane + (r * HPW + s2) * 16;
      const u32x4 a0 = *(const u32x4*)(ab);
      const u32x4 a1 = *(const u32x4*)(ab + HPW * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 w = *(const u32x4*)(sW + (tap * NT + nt) * 1024 + lane * 16);
        MmaH<Tag>::run(w, a0, acc[0][nt]);
        MmaH<Tag>::run(w, a1, acc[1][nt]);
      }
    }
  }

  // ---- epilogue: lane (x, q): pixel (y0 + 2*wave + mt, x0 + x), channels 4*NT*q + 4*nt + {0..3}
  if (ox >= OW) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (!rok[mt]) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = chq + nt * 4;
      if (ch >= p.Cout) continue;
      const f32x4 sc = *(const f32x4*)(p.scale + ch), sh = *(const f32x4*)(p.shift + ch);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * sc[j] + sh[j];
      if (p.res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rv[mt][nt][j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
      }
      if (E == 4) *(f32x4*)((float*)p.out + obase[mt] + nt * 4) = f32x4{v[0], v[1], v[2], v[3]};
      else if (KS == 2 && p.seg_S == CP_F16) {       // k = 2 instance only (cp_conv2x2_halo, CpConvDesc.out_f32 = 2): IEEE-half output rows
        u32x2 pk; pk.x = cp_pack2<true>(v[0], v[1]); pk.y = cp_pack2<true>(v[2], v[3]); *(u32x2*)((uint16_t*)p.out + obase[mt] + nt * 4) = pk;
      }
      else { u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); *(u32x2*)((uint16_t*)p.out + obase[mt] + nt * 4) = pk; }
    }
  }
}



// ------------------------------------------------------------------------------------------------------------------
// Grouped small-Cout convs: SEVERAL independent 3x3 layers (the branches of an HRNet module at equal depth, forward or
// data-gradient, at the training batch of 32) in one launch.  Same algorithm as conv3x3_halo_s_kernel with the tile count NT a
// RUN-TIME value (<= 5; block-uniform branches around the unrolled tile loops), one HaloParams per layer in a device table, block b
// belongs to the layer k with prefix[k] <= b < prefix[k + 1].  At B = 32 each of these layers alone is a ~9 us launch for < 1 GFLOP.
struct HaloGroupItem {          // == CpConvGroupItem
  int32_t NT; uint32_t blocks, lds_bytes, pad;
  HaloParams p;
};
static_assert(sizeof(HaloGroupItem) == sizeof(CpConvGroupItem), "CpConvGroupItem layout");

template <typename Tag, int NTM>
__global__ __launch_bounds__(256) void conv3x3_halo_s_group_kernel(const HaloGroupItem* __restrict__ items, const uint32_t* __restrict__ prefix,
                                                                   int n) {
  constexpr int E = Tag::E;
  constexpr int KCH = 4 * E;
  constexpr int ES = 16 / E;
  constexpr int WITERM = (9 * NTM * 64 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [A: HBUF][W: 9*NT KiB]
  unsigned char* const sA = smem;
  unsigned char* const sW = smem + HBUF;

  int k = 0;
  while (k + 1 < n && blockIdx.x >= prefix[k + 1]) ++k;
  const HaloParams p = items[k].p;
  const int NT = items[k].NT;
  const int WPIECES = 9 * NT * 64;
  const int t = (int)(blockIdx.x - prefix[k]);
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = t / tpi;
  const int trem = t - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * HTH, x0 = tx * HTW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  uint32_t s_goff[3], s_lds[3];
  int s_cq[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int i = tid + 256 * kk;
    const int hp = i >> 2, pq = i & 3;
    const int py = hp / HPW, px = hp - py * HPW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = (hp < HPH * HPW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    s_goff[kk] = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pq * E) : 0xFFFFFFFFu;
    s_lds[kk] = (uint32_t)(pq * HPLANE + hp * 16);
    s_cq[kk] = pq * E;
  }

  f32x4 acc[2][NTM];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t a_lane = (uint32_t)(q * HPLANE + (2 * wave * HPW + x) * 16);
  const int ox = x0 + x;
  const int chq = q * 4 * NT;
  long long obase[2];
  bool rok[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int oy = y0 + 2 * wave + mt;
    rok[mt] = (oy < p.H) & (ox < p.W);
    obase[mt] = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + chq;
  }
  f32x4 rv[2][NTM];
  if (p.res) {          // all residual loads before the first store (res may alias out: the data-gradient accumulates in place)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) {
        rv[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nt < NT && rok[mt] && chq + nt * 4 < p.Cout) {
          if (E == 4) rv[mt][nt] = *(const f32x4*)((const float*)p.res + obase[mt] + nt * 4);
          else {
            const u32x2 r2 = *(const u32x2*)((const uint16_t*)p.res + obase[mt] + nt * 4);
            rv[mt][nt] = f32x4{__uint_as_float(r2.x << 16), __uint_as_float(r2.x & 0xffff0000u),
                               __uint_as_float(r2.y << 16), __uint_as_float(r2.y & 0xffff0000u)};
          }
        }
      }
  }
  for (int c = 0; c < p.nchunk; ++c) {
    const int c0 = c * KCH;
    u32x4 sv[3], wv[WITERM];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      const bool ok = (s_goff[kk] != 0xFFFFFFFFu) & (c0 + s_cq[kk] < p.Cin);
      const uint32_t off = ok ? (s_goff[kk] + (uint32_t)c0) * ES : 0x80000000u;
      sv[kk] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
#pragma unroll
    for (int kk = 0; kk < WITERM; ++kk) {
      const int i = tid + 256 * kk;
      const uint32_t off = i < WPIECES ? ((uint32_t)c * (uint32_t)WPIECES + (uint32_t)i) * 16u : 0x80000000u;
      wv[kk] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
    }
    if (c > 0) __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) *(u32x4*)(sA + s_lds[kk]) = sv[kk];
#pragma unroll
    for (int kk = 0; kk < WITERM; ++kk) {
      const int i = tid + 256 * kk;
      if (i < WPIECES) *(u32x4*)(sW + i * 16) = wv[kk];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s2 = tap - 3 * r;
      const unsigned char* ab = sA + a_lane + (r * HPW + s2) * 16;
      const u32x4 a0 = *(const u32x4*)(ab);
      const u32x4 a1 = *(const u32x4*)(ab + HPW * 16);
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) {
        if (nt < NT) {
          const u32x4 w = *(const u32x4*)(sW + (tap * NT + nt) * 1024 + lane * 16);
          MmaH<Tag>::run(w, a0, acc[0][nt]);
          MmaH<Tag>::run(w, a1, acc[1][nt]);
        }
      }
    }
  }

  if (ox >= p.W) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (!rok[mt]) continue;
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) {
      const int ch = chq + nt * 4;
      if (nt >= NT || ch >= p.Cout) continue;
      const f32x4 sc = *(const f32x4*)(p.scale + ch), sh = *(const f32x4*)(p.shift + ch);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * sc[j] + sh[j];
      if (p.res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rv[mt][nt][j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
      }
      if (E == 4) *(f32x4*)((float*)p.out + obase[mt] + nt * 4) = f32x4{v[0], v[1], v[2], v[3]};
      else { u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); *(u32x2*)((uint16_t*)p.out + obase[mt] + nt * 4) = pk; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused BasicBlock (timm resnet.BasicBlock inside the HRNet branches):
//     out = relu( bn2(conv3x3( relu(bn1(conv3x3(x))) )) + x )          C -> C channels, C <= 32, one 64-byte chunk
// The 18-channel branch at 64x64 is the critical path of every HRNet module (8 convs in series, each HBM-bound);
// unfused, one block moves x, y1 (write + read), the residual x again and out = 5 tensors; fused it is x (with a
// 2-pixel halo) + out.  Per 8x16 output tile:
//   phase 1: stage the (8+4)x(16+4) halo of x in LDS; conv1 is evaluated on the (8+2)x(16+2) ring conv2 needs
//            (12 fragments of 16 flattened pixels, 3 per wave), BN1+ReLU, pixels outside the image forced to 0
//            (they are conv2's zero padding), written to LDS in conv2's operand layout;
//   phase 2: conv2 exactly as conv3x3_halo_s_kernel reads its halo, residual taken from the x halo already in
//            LDS (no second global read), BN2 + add + ReLU, contiguous per-pixel stores.
// Weights of both convs travel through one LDS buffer (conv2's are prefetched into registers during phase 1).
constexpr int FXW = HTW + 4, FXH = HTH + 4, FXPLANE = 256 * 16;      // 12 x 20 = 240 halo pixels, plane padded to 256

template <typename Tag, int NT>
__global__ __launch_bounds__(256) void basicblock_fused_kernel(const HaloParams p, const void* __restrict__ w2,
                                                               const float* __restrict__ scale2, const float* __restrict__ shift2) {
  constexpr int E = Tag::E;
  constexpr int ES = 16 / E;
  constexpr int WPIECES = 9 * NT * 64;
  constexpr int WITER = (WPIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [x halo 16 KiB][y1 12 KiB][W 9*NT KiB]
  unsigned char* const sX = smem;
  unsigned char* const sY = smem + 4 * FXPLANE;
  unsigned char* const sW = sY + HBUF;

  const int t = blockIdx.x;
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = t / tpi;
  const int trem = t - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * HTH, x0 = tx * HTW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w2), 0, p.w_bytes, 0x00020000);

  // ---- stage x halo (240 px x 4 pieces = 960 pieces -> 4 per thread, 1024 slots) and conv1 weights
  u32x4 xv[4], wv[WITER];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k;
    const int hp = i >> 2, pq = i & 3;
    const int py = hp / FXW, px = hp - py * FXW;
    const int gy = y0 - 2 + py, gx = x0 - 2 + px;
    const bool ok = (hp < FXH * FXW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W) & (pq * E < p.Cin);
    const uint32_t off = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pq * E) * ES : 0x80000000u;
    xv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
  }
#pragma unroll
  for (int k = 0; k < WITER; ++k) {
    const int i = tid + 256 * k;
    wv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w1rsrc, i < WPIECES ? (uint32_t)i * 16u : 0x80000000u, 0, 0));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k;
    *(u32x4*)(sX + (i & 3) * FXPLANE + (i >> 2) * 16) = xv[k];
  }
#pragma unroll
  for (int k = 0; k < WITER; ++k) {
    const int i = tid + 256 * k;
    if (i < WPIECES) *(u32x4*)(sW + i * 16) = wv[k];
  }
  // conv2 weights: in flight during phase 1
#pragma unroll
  for (int k = 0; k < WITER; ++k) {
    const int i = tid + 256 * k;
    wv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, i < WPIECES ? (uint32_t)i * 16u : 0x80000000u, 0, 0));
  }
  __syncthreads();

  // ---- phase 1: y1 = relu(bn1(conv1(x))) on the 10x18 ring, fragments f = 3*wave + i, pixel p1 = 16 f + x
  {
    f32x4 acc1[3][NT];
    uint32_t xb[3];
    bool inimg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc1[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int p1 = (3 * wave + i) * 16 + x;
      const int pc = p1 < HPH * HPW ? p1 : 0;
      const int yy = pc / HPW, xx = pc - yy * HPW;
      xb[i] = (uint32_t)(q * FXPLANE + (yy * FXW + xx) * 16);
      const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
      inimg[i] = (p1 < HPH * HPW) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s2 = tap - 3 * r;
      u32x4 a[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = *(const u32x4*)(sX + xb[i] + (r * FXW + s2) * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 w = *(const u32x4*)(sW + (tap * NT + nt) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i) MmaH<Tag>::run(w, a[i], acc1[i][nt]);
      }
    }
    // BN1 + ReLU -> LDS in conv2's operand layout [piece][pixel][16 B]; conv1 rows are NOT permuted:
    // lane (x, q) of tile nt holds channels 16 nt + 4 q + {0..3}
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p1 = (3 * wave + i) * 16 + x;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c0 = nt * 16 + q * 4;
        const f32x4 sc = *(const f32x4*)(p.scale + c0), sh = *(const f32x4*)(p.shift + c0);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = inimg[i] ? fmaxf(acc1[i][nt][j] * sc[j] + sh[j], 0.f) : 0.f;
        if (p1 < HNPIX) {
          unsigned char* dst = sY + (c0 / E) * HPLANE + p1 * 16 + (c0 % E) * ES;
          if (E == 4) *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
          else { u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); *(u32x2*)dst = pk; }
        }
      }
    }
  }
  __syncthreads();                                   // y1 complete, everyone done with conv1's weights
#pragma unroll
  for (int k = 0; k < WITER; ++k) {
    const int i = tid + 256 * k;
    if (i < WPIECES) *(u32x4*)(sW + i * 16) = wv[k];
  }
  // y1 pieces beyond 16*NT channels were never written: zero them (conv2 weights there are zero, but 0 * NaN = NaN)
  if (16 * NT < 4 * E)
    for (int i = tid; i < (4 * E - 16 * NT) / E * HNPIX; i += 256)
      *(u32x4*)(sY + ((16 * NT) / E) * HPLANE + i * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- phase 2: conv2 on the 8x16 tile (wave w: rows 2w, 2w+1), residual from the x halo in LDS
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t a_lane = (uint32_t)(q * HPLANE + (2 * wave * HPW + x) * 16);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, s2 = tap - 3 * r;
    const unsigned char* ab = sY + a_lane + (r * HPW + s2) * 16;
    const u32x4 a0 = *(const u32x4*)(ab);
    const u32x4 a1 = *(const u32x4*)(ab + HPW * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const u32x4 w = *(const u32x4*)(sW + (tap * NT + nt) * 1024 + lane * 16);
      MmaH<Tag>::run(w, a0, acc[0][nt]);
      MmaH<Tag>::run(w, a1, acc[1][nt]);
    }
  }
  const int ox = x0 + x;
  if (ox >= p.W) return;
  const int chq = q * 4 * NT;                          // conv2 rows ARE permuted: lane q holds channels 4 NT q + 4 nt + {0..3}
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int oy = y0 + 2 * wave + mt;
    if (oy >= p.H) continue;
    const long long o = p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + chq;
    const int xp = (2 * wave + mt + 2) * FXW + x + 2;  // this output pixel inside the x halo
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = chq + nt * 4;
      if (ch >= p.Cout) continue;
      const f32x4 sc = *(const f32x4*)(scale2 + ch), sh = *(const f32x4*)(shift2 + ch);
      const unsigned char* rp = sX + (ch / E) * FXPLANE + xp * 16 + (ch % E) * ES;
      float rr[4];
      if (E == 4) { const f32x4 r4 = *(const f32x4*)rp; rr[0] = r4[0]; rr[1] = r4[1]; rr[2] = r4[2]; rr[3] = r4[3]; }
      else {
        const u32x2 r2 = *(const u32x2*)rp;
        rr[0] = __uint_as_float(r2.x << 16); rr[1] = __uint_as_float(r2.x & 0xffff0000u);
        rr[2] = __uint_as_float(r2.y << 16); rr[3] = __uint_as_float(r2.y & 0xffff0000u);
      }
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[mt][nt][j] * sc[j] + sh[j] + rr[j], 0.f);
      if (E == 4) *(f32x4*)((float*)p.out + o + nt * 4) = f32x4{v[0], v[1], v[2], v[3]};
      else { u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); *(u32x2*)((uint16_t*)p.out + o + nt * 4) = pk; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent bf16 variant of the fused BasicBlock for 16 < C <= 32 (the 18-channel HRNet branch, 32 launches per
// forward).  Knock-out timing of the one-tile-per-block kernel above (B=256: 75 us) showed neither global memory
// (no loads + no stores: 62 us) nor the MFMAs (removing them made it slower) on the critical path: it is bound by
// LDS traffic -- per tile 36 KB of weights written to and 144 KB read back from LDS next to 180 KB of activation
// fragment reads, one ds_read per two MFMAs.  Here
//   * blocks stay resident (2 x 4 waves per CU) and BOTH weight sets live in registers (36 fragments per wave),
//   * conv2 re-uses activation fragments across output rows: input row R at shift s feeds output rows R, R-1, R-2, so
//     a wave's two rows need 12 fragment reads for 36 MFMAs instead of 18,
//   * the next tile's halo loads are issued before the current tile's epilogue,
//   * the finished tile is written back over the x tile in LDS and leaves in full-line order.
// LDS reads per tile drop from 324 KB to 156 KB.  Tile order as in the fused Bottleneck (crops b == xcd (mod 8) on
// XCD blockIdx % 8).
#ifndef CP_PB_PADX
#define CP_PB_PADX 16
#endif
#ifndef CP_PB_PADT
#define CP_PB_PADT 16
#endif
#ifndef CP_PB_FXW
#define CP_PB_FXW 20
#endif
// x-halo row pitch in pixels (A/B knob).  conv1's fragments are 16 consecutive pixels of the FLATTENED 10 x 18 ring, so a
// fragment that wraps to the next ring row jumps (PFXW - 18) pixels in the plane; PFXW = 34 makes that jump one full bank
// sweep (conflict-free wrap) -- measured +-0 against 20 (2.118 vs 2.127 ms per step), so the compact 20 stays.
constexpr int PFXW = CP_PB_FXW;
constexpr int PBX_PITCH = FXH * PFXW * 16 + CP_PB_PADX, PBT_PITCH = 192 * 16 + CP_PB_PADT;      // x halo planes (240 px) / t1 planes (180 px)
constexpr int PB_LDS = 4 * PBX_PITCH + 4 * PBT_PITCH;                    // 28 800 B

__global__ __launch_bounds__(256, 2) void basicblock_persist_kernel(const HaloParams p, const void* __restrict__ w2,
                                                                    const float* __restrict__ scale2,
                                                                    const float* __restrict__ shift2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sX = smem;
  unsigned char* const sT = smem + 4 * PBX_PITCH;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
  const int x = lane & 15, q = lane >> 4;

  u32x4 W1[9][2], W2[9][2];                                    // [chunk 0][tap][nt][lane][16 B] images, NT = 2
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      W1[tap][nt] = ((const u32x4*)p.w)[(tap * 2 + nt) * 64 + lane];
      W2[tap][nt] = ((const u32x4*)w2)[(tap * 2 + nt) * 64 + lane];
    }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const int tpi = p.tiles_x * p.tiles_y;
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int npc = p.Cin >> 3;                                  // 16-byte pieces per input pixel (Cphys / 8 <= 4)

  // x halo: 240 px x 4 pieces = 960 pieces, 4 per thread (piece index tid + 256 it; >= 960 idle)
  auto issue_loads = [&](int li, u32x4* xv) {
    const int b = (li / tpi) * 8 + xcd;
    const int trem = li % tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * HTH - 2, x0 = tx * HTW - 2;
    const bool tok = b < p.B;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + 256 * it;
      const int hp = i >> 2, pq = i & 3;
      const int py = hp / FXW, px = hp - py * FXW;
      const int gy = y0 + py, gx = x0 + px;
      const bool ok = tok & (hp < FXH * FXW) & (pq < npc) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
      const uint32_t off = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pq * 8) * 2u : 0x80000000u;
      xv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };

  // per-lane geometry of conv1's three fragments (tile-independent): t1 pixel p1 = (3 wave + i) * 16 + x on the ring
  uint32_t xb[3];
  int p1y[3], p1x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p1 = (3 * wave + i) * 16 + x;
    const int pc = p1 < HPH * HPW ? p1 : 0;
    p1y[i] = p1 < HPH * HPW ? pc / HPW : -64;                  // pad pixels: never inside the image
    p1x[i] = pc - (pc / HPW) * HPW;
    xb[i] = (uint32_t)(q * PBX_PITCH + ((pc / HPW) * PFXW + p1x[i]) * 16);
  }

  u32x4 xv[4];
  int li = j0;
  issue_loads(li, xv);
  for (; (li / tpi) * 8 + xcd < p.B; li += nbx) {
    const int b = (li / tpi) * 8 + xcd;
    const int trem = li % tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * HTH, x0 = tx * HTW;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + 256 * it;
      const int hp = i >> 2, hy = hp / FXW;                            // (hy, hx) of the 12 x 20 halo -> row pitch PFXW
      if (hp < FXH * FXW) *(u32x4*)(sX + (i & 3) * PBX_PITCH + (hy * PFXW + (hp - hy * FXW)) * 16) = xv[it];
    }
    __syncthreads();

    // ---- conv1 on the 10x18 ring -> t1 (BN1 + ReLU, zero outside the image); rows of conv1 are not permuted
    {
      f32x4 acc[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int r = tap / 3, s2 = tap - 3 * r;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const u32x4 a = *(const u32x4*)(sX + xb[i] + (r * PFXW + s2) * 16);
          MmaH<BF16Tag>::run(W1[tap][0], a, acc[i][0]);
          MmaH<BF16Tag>::run(W1[tap][1], a, acc[i][1]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c1 = nt * 16 + q * 4;
        const f32x4 sc = *(const f32x4*)(p.scale + c1), sh = *(const f32x4*)(p.shift + c1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int p1 = (3 * wave + i) * 16 + x;
          const bool inimg = ((unsigned)(y0 - 1 + p1y[i]) < (unsigned)p.H) & ((unsigned)(x0 - 1 + p1x[i]) < (unsigned)p.W);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = inimg ? fmaxf(acc[i][nt][e] * sc[e] + sh[e], 0.f) : 0.f;
          u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
          *(u32x2*)(sT + (c1 >> 3) * PBT_PITCH + p1 * 16 + (c1 & 7) * 2) = pk;
        }
      }
    }
    __syncthreads();

    // ---- conv2 on rows 2 wave, 2 wave + 1: ring rows 2 wave .. 2 wave + 3, each fragment feeds both output rows
    {
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const unsigned char* ab = sT + q * PBT_PITCH + (2 * wave * HPW + x) * 16;
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          const u32x4 a = *(const u32x4*)(ab + (R * HPW + s2) * 16);
          if (R < 3) {                                          // output row 0, tap row R
            MmaH<BF16Tag>::run(W2[R * 3 + s2][0], a, acc[0][0]);
            MmaH<BF16Tag>::run(W2[R * 3 + s2][1], a, acc[0][1]);
          }
          if (R > 0) {                                          // output row 1, tap row R - 1
            MmaH<BF16Tag>::run(W2[(R - 1) * 3 + s2][0], a, acc[1][0]);
            MmaH<BF16Tag>::run(W2[(R - 1) * 3 + s2][1], a, acc[1][1]);
          }
        }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c2 = q * 8 + nt * 4;                          // conv2 rows are permuted: lane q holds channels 8 q + 4 nt + {0..3}
        const f32x4 sc = *(const f32x4*)(scale2 + c2), sh = *(const f32x4*)(shift2 + c2);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          u32x2* rp = (u32x2*)(sX + (c2 >> 3) * PBX_PITCH + ((2 * wave + mt + 2) * PFXW + x + 2) * 16 + (c2 & 7) * 2);
          const u32x2 r2 = *rp;
          float v[4];
          v[0] = fmaxf(acc[mt][nt][0] * sc[0] + sh[0] + __uint_as_float(r2.x << 16), 0.f);
          v[1] = fmaxf(acc[mt][nt][1] * sc[1] + sh[1] + __uint_as_float(r2.x & 0xffff0000u), 0.f);
          v[2] = fmaxf(acc[mt][nt][2] * sc[2] + sh[2] + __uint_as_float(r2.y << 16), 0.f);
          v[3] = fmaxf(acc[mt][nt][3] * sc[3] + sh[3] + __uint_as_float(r2.y & 0xffff0000u), 0.f);
          u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
          *rp = pk;
        }
      }
    }
    __syncthreads();

    // ---- next tile's loads take off; this tile leaves LDS: piece i -> (pixel i / opc, piece i % opc) over 128 px
    issue_loads(li + nbx, xv);
    {
      const int opc = (p.Cout + 7) >> 3;                       // output pieces per pixel (Cphys / 8)
      for (int i = tid; i < 128 * opc; i += 256) {
        const int pxl = i / opc, pc = i - pxl * opc;
        const int row = pxl >> 4, col = pxl & 15;
        const int oy = y0 + row, ox = x0 + col;
        const u32x4 v = *(const u32x4*)(sX + pc * PBX_PITCH + ((row + 2) * PFXW + col + 2) * 16);
        if (oy < p.H && ox < p.W)
          *(u32x4*)((uint16_t*)p.out + p.o_base + (long long)b * p.o_sb + (long long)oy * p.o_sy + (long long)ox * p.o_sx + pc * 8) = v;
      }
    }
    __syncthreads();
  }
}

// packing for the small-Cout variant: [chunk][tap][nt][lane][16 B]; tile row i = 4*qr + reg of tile nt is output
// channel 4*NT*qr + 4*nt + reg.
template <typename Tag>
__global__ void pack_halo_s_weight_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int NT,
                                          int perm, size_t total, int ntap = 9) {
  constexpr int E = Tag::E;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % E);
  const int lane = (int)((i / E) % 64);
  size_t blk = i / (E * 64);
  const int nt = (int)(blk % NT); blk /= NT;
  const int tap = (int)(blk % ntap);
  const int c = (int)(blk / ntap);
  const int row = lane & 15, kq = lane >> 4;
  const int n = perm ? (row >> 2) * 4 * NT + nt * 4 + (row & 3) : nt * 16 + row;
  const int cin = c * (4 * E) + kq * E + e;
  float v = 0.f;
  if (n < Cout && cin < Cin) v = w[((size_t)n * Cin + cin) * ntap + tap];
  store_elem<Tag>(out, i, v);
}

static inline bool halo_small(int Cout) { static const int mx = cp_knob("CP_HALO_S_MAX") ? atoi(cp_knob("CP_HALO_S_MAX")) : 80; return Cout <= mx; }

// ---- packing: [group g (32 ch)][chunk c][tap][nt][lane][16 B]; tile row i = 4*qr + reg of tile nt is output
// channel g*32 + 8*qr + 4*nt + reg (the permutation that makes the epilogue stores 16 bytes wide).
template <typename Tag>
__global__ void pack_halo_weight_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int nchunk,
                                        size_t total) {
  constexpr int E = Tag::E;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % E);
  const int lane = (int)((i / E) % 64);
  size_t blk = i / (E * 64);
  const int nt = (int)(blk % 2); blk /= 2;
  const int tap = (int)(blk % 9); blk /= 9;
  const int c = (int)(blk % nchunk);
  const int g = (int)(blk / nchunk);
  const int row = lane & 15, kq = lane >> 4;
  const int n = g * 32 + (row >> 2) * 8 + nt * 4 + (row & 3);
  const int cin = c * (4 * E) + kq * E + e;
  float v = 0.f;
  if (n < Cout && cin < Cin) v = w[((size_t)n * Cin + cin) * 9 + tap];
  store_elem<Tag>(out, i, v);
}

extern "C" size_t cp_packed_halo_weight_bytes(int dtype, int Cout, int cin_phys) {
  const int E = cp_chan_align(dtype), KCH = 4 * E;
  const size_t nchunk = ((size_t)cin_phys + KCH - 1) / KCH;
  if (halo_small(Cout)) return nchunk * 9 * (((size_t)Cout + 15) / 16) * 1024;   // [chunk][tap][nt][lane][16 B]
  const size_t ngroups = ((size_t)Cout + 31) / 32;
  return ngroups * nchunk * 18 * 1024;
}

extern "C" int cp_pack_conv3x3_halo_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys,
                                           void* packed) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || cin_phys < Cin) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys % E || !cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int nchunk = (cin_phys + 4 * E - 1) / (4 * E);
  const size_t total = cp_packed_halo_weight_bytes(dtype, Cout, cin_phys) / cp_elem_size(dtype);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (halo_small(Cout)) {
    const int NT = (Cout + 15) / 16;
    if (dtype == CP_F32)
      CP_LAUNCH(pack_halo_s_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, NT, 1, total);
    else
      CP_LAUNCH(pack_halo_s_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, NT, 1, total);
    return cp_check_launch();
  }
  if (halo_wide(Cout)) {
    if (dtype == CP_F32)
      CP_LAUNCH(pack_halo4_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, total);
    else
      CP_LAUNCH(pack_halo4_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, total);
    return cp_check_launch();
  }
  if (dtype == CP_F32)
    CP_LAUNCH(pack_halo_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, total);
  else
    CP_LAUNCH(pack_halo_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, nchunk, total);
  return cp_check_launch();
}

extern "C" int cp_pack_item_halo(int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed, CpPackItem* it) {
  if (!w || !packed || !it || Cout <= 0 || Cin <= 0 || cin_phys < Cin) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys % E || !cp_aligned16(packed)) return CP_ERR_ALIGN;
  for (int k = 0; k < 9; ++k) it->a[k] = 0;
  it->src = w; it->dst = packed; it->row_map = nullptr;
  it->total = cp_packed_halo_weight_bytes(dtype, Cout, cin_phys) / cp_elem_size(dtype);
  it->a[0] = Cout; it->a[1] = Cin;
  if (halo_small(Cout)) { it->kind = CP_PACK_HALO_S; it->a[2] = (Cout + 15) / 16; it->a[3] = 1; }
  else { it->kind = halo_wide(Cout) ? CP_PACK_HALO4 : CP_PACK_HALO; it->a[2] = (cin_phys + 4 * E - 1) / (4 * E); }
  return CP_OK;
}

static int build_halo_params(const CpConvDesc* d, const void* in, const void* packed_w, const float* scale, const float* shift,
                             const void* residual, void* out, HaloParams* pp, long long* tiles) {
  if (!d || !in || !packed_w || !scale || !shift || !out) return CP_ERR_INVALID;
  if ((d->dtype != CP_F32 && d->dtype != CP_BF16) || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W || d->out_f32 || d->o_sc != 1)
    return CP_ERR_INVALID;
  const int E = cp_chan_align(d->dtype), es = cp_elem_size(d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return CP_ERR_INVALID;
  if (d->Cin <= 0 || d->Cin % E || d->in_coff % E || d->in_cstride % E || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (d->Cout <= 0 || d->Cout % 4) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift)) return CP_ERR_ALIGN;
  if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4)) return CP_ERR_ALIGN;
  if (((uintptr_t)out % (4 * es)) || (residual && ((uintptr_t)residual % (4 * es)))) return CP_ERR_ALIGN;
  const long long in_bytes = (long long)d->B * d->H * d->W * d->in_cstride * es;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  HaloParams p;
  p.in = in; p.w = packed_w; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.Hs = p.Ws = 0; p.up_sy = p.up_sx = 0.f;
  p.seg_w = p.seg_b = nullptr; p.seg_out = nullptr; p.seg_S = 0;
  p.nchunk = (d->Cin + 4 * E - 1) / (4 * E);
  p.Cout = d->Cout; p.ngroups = (d->Cout + 31) / 32; p.NB = (p.ngroups + 3) / 4;
  p.tiles_x = (d->W + HTW - 1) / HTW; p.tiles_y = (d->H + HTH - 1) / HTH;
  const long long tt = (long long)d->B * p.tiles_x * p.tiles_y;
  if (tt >= (1LL << 28)) return CP_ERR_RANGE;
  p.total_tiles = (int)tt;
  p.act = d->act; p.slope = d->slope;
  p.in_bytes = (uint32_t)in_bytes;
  const size_t wb = cp_packed_halo_weight_bytes(d->dtype, d->Cout, d->Cin);
  if (wb >= (1ull << 31)) return CP_ERR_RANGE;
  p.w_bytes = (uint32_t)wb;
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  *pp = p;
  *tiles = tt;
  return CP_OK;
}

extern "C" int cp_conv3x3_halo(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                               const float* scale, const float* shift, const void* residual, void* out) {
  HaloParams p;
  long long tt;
  const int rcb = build_halo_params(d, in, packed_w, scale, shift, residual, out, &p, &tt);
  if (rcb) return rcb;
  if (halo_small(d->Cout)) {   // d->Cout is the PHYSICAL count; the pack call used the logical one -> same tile count
    const int NT = (d->Cout + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = HBUF + (size_t)9 * NT * 1024;
#define CP_HS(TAG, N) CP_LAUNCH((conv3x3_halo_s_kernel<TAG, N>), dim3((unsigned)tt), dim3(256), lds, st, p)
    if (d->dtype == CP_F32) {
      switch (NT) { case 1: CP_HS(F32Tag, 1); break; case 2: CP_HS(F32Tag, 2); break; case 3: CP_HS(F32Tag, 3); break;
                    case 4: CP_HS(F32Tag, 4); break; default: CP_HS(F32Tag, 5); break; }
    } else {
      switch (NT) { case 1: CP_HS(BF16Tag, 1); break; case 2: CP_HS(BF16Tag, 2); break; case 3: CP_HS(BF16Tag, 3); break;
                    case 4: CP_HS(BF16Tag, 4); break; default: CP_HS(BF16Tag, 5); break; }
    }
#undef CP_HS
    return cp_check_launch();
  }
  if (halo_wide(d->Cout)) {
    p.NB = d->Cout / 256 + ((d->Cout % 256) ? 1 : 0);
    if (d->Cout % 256) return CP_ERR_INVALID;          // wide variant: whole 256-channel blocks only
    const unsigned grid4 = (unsigned)(((tt + 7) / 8) * 8 * p.NB);
    hipStream_t st4 = (hipStream_t)stream;
    if (d->dtype == CP_F32) {
      if (residual) CP_LAUNCH((conv3x3_halo4_kernel<F32Tag, true, false, false>), dim3(grid4), dim3(256), 2 * HBUF, st4, p);
      else CP_LAUNCH((conv3x3_halo4_kernel<F32Tag, false, false, false>), dim3(grid4), dim3(256), 2 * HBUF, st4, p);
    } else {
      if (residual) CP_LAUNCH((conv3x3_halo4_kernel<BF16Tag, true, false, false>), dim3(grid4), dim3(256), 2 * HBUF, st4, p);
      else CP_LAUNCH((conv3x3_halo4_kernel<BF16Tag, false, false, false>), dim3(grid4), dim3(256), 2 * HBUF, st4, p);
    }
    return cp_check_launch();
  }
  const unsigned grid = (unsigned)(((tt + 7) / 8) * 8 * p.NB);
  // scale/shift are read 8 at a time at ch = g*32 + 8q < Cout: vectors must be padded to a multiple of 8 (they are: 16)
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CP_F32) {
    if (residual) CP_LAUNCH((conv3x3_halo_kernel<F32Tag, true>), dim3(grid), dim3(256), 2 * HBUF, st, p);
    else CP_LAUNCH((conv3x3_halo_kernel<F32Tag, false>), dim3(grid), dim3(256), 2 * HBUF, st, p);
  } else {
    if (residual) CP_LAUNCH((conv3x3_halo_kernel<BF16Tag, true>), dim3(grid), dim3(256), 2 * HBUF, st, p);
    else CP_LAUNCH((conv3x3_halo_kernel<BF16Tag, false>), dim3(grid), dim3(256), 2 * HBUF, st, p);
  }
  return cp_check_launch();
}

// ---- conv3x3( bilinear_x2(in) ): the decoder's upsample + first conv (pipeline.py:199-200) without the upsampled tensor
extern "C" int cp_conv3x3_halo_up2x_supported(int dtype, int Cout) {
  return (dtype == CP_F32 || dtype == CP_BF16) && halo_wide(Cout) ? 1 : 0;
}

extern "C" int cp_conv3x3_halo_up2x(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                                    const float* scale, const float* shift, void* out) {
  if (!d || !in || !packed_w || !scale || !shift || !out) return CP_ERR_INVALID;
  if (!cp_conv3x3_halo_up2x_supported(d->dtype, d->Cout) || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W || d->out_f32 || d->o_sc != 1)
    return CP_ERR_INVALID;
  if (d->B <= 0 || d->H < 2 || d->W < 2 || (d->H & 1) || (d->W & 1)) return CP_ERR_INVALID;   // d->H, d->W: the UPSAMPLED size
  const int E = cp_chan_align(d->dtype), es = cp_elem_size(d->dtype);
  if (d->Cin <= 0 || d->Cin % E || d->in_coff % E || d->in_cstride % E || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift)) return CP_ERR_ALIGN;
  if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4) || ((uintptr_t)out % (4 * es))) return CP_ERR_ALIGN;
  HaloParams p;
  p.seg_w = p.seg_b = nullptr; p.seg_out = nullptr; p.seg_S = 0;
  p.Hs = d->H / 2; p.Ws = d->W / 2;
  const long long in_bytes = (long long)d->B * p.Hs * p.Ws * d->in_cstride * es;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  p.up_sy = p.Hs > 1 ? (float)(p.Hs - 1) / (float)(d->H - 1) : 0.f;       // the stand-alone kernel's scales (elementwise.hip)
  p.up_sx = p.Ws > 1 ? (float)(p.Ws - 1) / (float)(d->W - 1) : 0.f;
  p.in = in; p.w = packed_w; p.scale = scale; p.shift = shift; p.res = nullptr; p.out = out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.nchunk = (d->Cin + 4 * E - 1) / (4 * E);
  p.Cout = d->Cout; p.ngroups = (d->Cout + 31) / 32; p.NB = d->Cout / 256;
  p.tiles_x = (d->W + HTW - 1) / HTW; p.tiles_y = (d->H + HTH - 1) / HTH;
  const long long tt = (long long)d->B * p.tiles_x * p.tiles_y;
  if (tt >= (1LL << 28)) return CP_ERR_RANGE;
  p.total_tiles = (int)tt;
  p.act = d->act; p.slope = d->slope;
  p.in_bytes = (uint32_t)in_bytes;
  const size_t wb = cp_packed_halo_weight_bytes(d->dtype, d->Cout, d->Cin);
  if (wb >= (1ull << 31)) return CP_ERR_RANGE;
  p.w_bytes = (uint32_t)wb;
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  const unsigned grid4 = (unsigned)(((tt + 7) / 8) * 8 * p.NB);
  if (d->dtype == CP_F32) CP_LAUNCH((conv3x3_halo4_kernel<F32Tag, false, true, false>), dim3(grid4), dim3(256), 2 * HBUF + 2 * USBUF, (hipStream_t)stream, p);
  else CP_LAUNCH((conv3x3_halo4_kernel<BF16Tag, false, true, false>), dim3(grid4), dim3(256), 2 * HBUF + 2 * USBUF, (hipStream_t)stream, p);
  return cp_check_launch();
}

// ---- conv3x3 + folded BN + act with the 1x1 head that reads its output fused into the epilogue (decoder's last conv + seg_block)
extern "C" int cp_conv3x3_halo_seg_supported(int dtype, int Cout, int S) {
  return (dtype == CP_F32 || dtype == CP_BF16) && Cout == 256 && halo_wide(Cout) && S >= 1 && S <= 2 ? 1 : 0;
}

extern "C" int cp_conv3x3_halo_seg(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w, const float* scale,
                                   const float* shift, void* out, const float* seg_w, const float* seg_b, int S, float* seg_out) {
  if (!d || !in || !packed_w || !scale || !shift || !out || !seg_w || !seg_b || !seg_out) return CP_ERR_INVALID;
  if (!cp_conv3x3_halo_seg_supported(d->dtype, d->Cout, S) || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W || d->out_f32 || d->o_sc != 1)
    return CP_ERR_INVALID;
  const int E = cp_chan_align(d->dtype), es = cp_elem_size(d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return CP_ERR_INVALID;
  if (d->Cin <= 0 || d->Cin % E || d->in_coff % E || d->in_cstride % E || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift) || !cp_aligned16(seg_w)) return CP_ERR_ALIGN;
  if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4) || ((uintptr_t)out % (4 * es))) return CP_ERR_ALIGN;
  const long long in_bytes = (long long)d->B * d->H * d->W * d->in_cstride * es;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  HaloParams p;
  p.in = in; p.w = packed_w; p.scale = scale; p.shift = shift; p.res = nullptr; p.out = out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.Hs = p.Ws = 0; p.up_sy = p.up_sx = 0.f;
  p.nchunk = (d->Cin + 4 * E - 1) / (4 * E);
  p.Cout = d->Cout; p.ngroups = (d->Cout + 31) / 32; p.NB = 1;
  p.tiles_x = (d->W + HTW - 1) / HTW; p.tiles_y = (d->H + HTH - 1) / HTH;
  const long long tt = (long long)d->B * p.tiles_x * p.tiles_y;
  if (tt >= (1LL << 28)) return CP_ERR_RANGE;
  p.total_tiles = (int)tt;
  p.act = d->act; p.slope = d->slope;
  p.in_bytes = (uint32_t)in_bytes;
  const size_t wb = cp_packed_halo_weight_bytes(d->dtype, d->Cout, d->Cin);
  if (wb >= (1ull << 31)) return CP_ERR_RANGE;
  p.w_bytes = (uint32_t)wb;
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  p.seg_w = seg_w; p.seg_b = seg_b; p.seg_out = seg_out; p.seg_S = S;
  const unsigned grid4 = (unsigned)(((tt + 7) / 8) * 8);
  if (d->dtype == CP_F32) CP_LAUNCH((conv3x3_halo4_kernel<F32Tag, false, false, true>), dim3(grid4), dim3(256), 2 * HBUF, (hipStream_t)stream, p);
  else CP_LAUNCH((conv3x3_halo4_kernel<BF16Tag, false, false, true>), dim3(grid4), dim3(256), 2 * HBUF, (hipStream_t)stream, p);
  return cp_check_launch();
}

// ---- fused BasicBlock API --------------------------------------------------------------------------------------
extern "C" int cp_pack_conv3x3_rows_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys,
                                           void* packed) {
  // conv1 of the fused BasicBlock: small-Cout halo image WITHOUT the row permutation (tile row i = channel 16 nt + i)
  if (!w || !packed || Cout <= 0 || Cout > 80 || Cin <= 0 || cin_phys < Cin) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys % E || !cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int NT = (Cout + 15) / 16;
  const size_t total = cp_packed_halo_weight_bytes(dtype, Cout, cin_phys) / cp_elem_size(dtype);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(pack_halo_s_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, NT, 0, total);
  else
    CP_LAUNCH(pack_halo_s_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, NT, 0, total);
  return cp_check_launch();
}

extern "C" int cp_basicblock_fused(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w1,
                                   const float* scale1, const float* shift1, const void* packed_w2, const float* scale2,
                                   const float* shift2, void* out) {
  if (!d || !in || !packed_w1 || !packed_w2 || !scale1 || !shift1 || !scale2 || !shift2 || !out) return CP_ERR_INVALID;
  if ((d->dtype != CP_F32 && d->dtype != CP_BF16) || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W || d->out_f32 || d->o_sc != 1)
    return CP_ERR_INVALID;
  const int E = cp_chan_align(d->dtype), es = cp_elem_size(d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return CP_ERR_INVALID;
  if (d->Cin != d->Cout || d->Cin > 4 * E || d->Cin > 32) return CP_ERR_INVALID;     // one 64-byte chunk, NT <= 2
  if (d->Cin % E || d->in_coff % E || d->in_cstride % E || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w1) || !cp_aligned16(packed_w2) || !cp_aligned16(scale1) ||
      !cp_aligned16(shift1) || !cp_aligned16(scale2) || !cp_aligned16(shift2)) return CP_ERR_ALIGN;
  if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4) || ((uintptr_t)out % (4 * es))) return CP_ERR_ALIGN;
  if (in == out) return CP_ERR_INVALID;               // the halo of a neighbouring tile would read half-written data
  const long long in_bytes = (long long)d->B * d->H * d->W * d->in_cstride * es;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  HaloParams p;
  p.in = in; p.w = packed_w1; p.scale = scale1; p.shift = shift1; p.res = nullptr; p.out = out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.nchunk = 1; p.Cout = d->Cout; p.ngroups = 1; p.NB = 1;
  p.tiles_x = (d->W + HTW - 1) / HTW; p.tiles_y = (d->H + HTH - 1) / HTH;
  const long long tt = (long long)d->B * p.tiles_x * p.tiles_y;
  if (tt >= (1LL << 28)) return CP_ERR_RANGE;
  p.total_tiles = (int)tt;
  p.act = CP_ACT_RELU; p.slope = 0.f;
  p.in_bytes = (uint32_t)in_bytes;
  const int NT = (d->Cout + 15) / 16;
  p.w_bytes = (uint32_t)((size_t)9 * NT * 1024);
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  const size_t lds = 4 * FXPLANE + HBUF + (size_t)9 * NT * 1024;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CP_BF16 && NT == 2 && !cp_knob("CP_NO_BB_PERSIST")) {
    const int n_cu = cp_num_cus();
    if (n_cu <= 0) return CP_ERR_HIP;
    // two resident 4-wave blocks per CU, a multiple of 8 (XCD labels), never more than the tiles one XCD owns
    const long long per_xcd = (long long)((d->B + 7) / 8) * p.tiles_x * p.tiles_y;
    long long nbx = 2 * n_cu / 8 > 0 ? 2 * n_cu / 8 : 1;
    if (nbx > per_xcd) nbx = per_xcd;
    CP_LAUNCH(basicblock_persist_kernel, dim3((unsigned)(8 * nbx)), dim3(256), PB_LDS, st, p, packed_w2, scale2, shift2);
    return cp_check_launch();
  }
  if (d->dtype == CP_F32) {
    if (NT == 1) CP_LAUNCH((basicblock_fused_kernel<F32Tag, 1>), dim3((unsigned)tt), dim3(256), lds, st, p, packed_w2, scale2, shift2);
    else CP_LAUNCH((basicblock_fused_kernel<F32Tag, 2>), dim3((unsigned)tt), dim3(256), lds, st, p, packed_w2, scale2, shift2);
  } else {
    if (NT == 1) CP_LAUNCH((basicblock_fused_kernel<BF16Tag, 1>), dim3((unsigned)tt), dim3(256), lds, st, p, packed_w2, scale2, shift2);
    else CP_LAUNCH((basicblock_fused_kernel<BF16Tag, 2>), dim3((unsigned)tt), dim3(256), lds, st, p, packed_w2, scale2, shift2);
  }
  return cp_check_launch();
}

// ---- k = 2 / stride 1 / pad 1 conv with <= 80 output channels (conv3x3_halo_s_kernel<.., KS = 2>)
extern "C" int cp_conv2x2_halo_supported(int dtype, int H, int W, int Cout_phys) {
  return ((dtype == CP_F32 || dtype == CP_BF16) && H >= 7 && W >= 15 && Cout_phys > 0 && Cout_phys % 4 == 0 && Cout_phys <= 80) ? 1 : 0;
}

extern "C" size_t cp_packed_conv2x2_halo_weight_bytes(int dtype, int Cout, int Cin_phys) {
  const int E = cp_chan_align(dtype);
  const size_t nchunk = ((size_t)Cin_phys + 4 * E - 1) / (4 * E);
  return nchunk * 4 * (((size_t)Cout + 15) / 16) * 1024;           // [chunk][tap][nt][lane][16 B]
}

extern "C" int cp_pack_conv2x2_halo_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed) {
  if (!w || !packed || Cout <= 0 || Cout > 80 || Cin <= 0 || cin_phys < Cin) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys % E || !cp_aligned16(packed)) return CP_ERR_ALIGN;
  const size_t total = cp_packed_conv2x2_halo_weight_bytes(dtype, Cout, cin_phys) / cp_elem_size(dtype);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  const int NT = (Cout + 15) / 16;
  if (dtype == CP_F32)
    CP_LAUNCH(pack_halo_s_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, NT, 1, total, 4);
  else
    CP_LAUNCH(pack_halo_s_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, NT, 1, total, 4);
  return cp_check_launch();
}

extern "C" int cp_conv2x2_halo(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w, const float* scale,
                               const float* shift, const void* residual, void* out) {
  if (!d || d->R != 2 || d->S != 2 || d->stride != 1 || d->pad != 1 || d->Ho != d->H + 1 || d->Wo != d->W + 1) return CP_ERR_INVALID;
  if (!cp_conv2x2_halo_supported(d->dtype, d->H, d->W, d->Cout)) return CP_ERR_INVALID;
  CpConvDesc d3 = *d;                      // the shared checks are the 3x3 kernel's; extents and weight size are set here
  d3.R = d3.S = 3; d3.Ho = d->H; d3.Wo = d->W;
  const bool half_out = d->out_f32 == 2;   // bf16 conv, rows written as IEEE half (the keypoint side's storage type, CP_F16)
  if (half_out && (d->dtype != CP_BF16 || residual)) return CP_ERR_INVALID;
  if (half_out) d3.out_f32 = 0;
  HaloParams p;
  long long tt;
  const int rcb = build_halo_params(&d3, in, packed_w, scale, shift, residual, out, &p, &tt);
  if (rcb) return rcb;
  p.seg_S = half_out ? CP_F16 : 0;         // (this kernel family has no seg head: the field carries the output format)
  p.tiles_x = (d->Wo + HTW - 1) / HTW; p.tiles_y = (d->Ho + HTH - 1) / HTH;
  tt = (long long)d->B * p.tiles_x * p.tiles_y;
  if (tt >= (1LL << 28)) return CP_ERR_RANGE;
  p.total_tiles = (int)tt;
  p.w_bytes = (uint32_t)cp_packed_conv2x2_halo_weight_bytes(d->dtype, d->Cout, d->Cin);
  const int NT = (d->Cout + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = HBUF + (size_t)4 * NT * 1024;
#define CP_HS2(TAG, N) CP_LAUNCH((conv3x3_halo_s_kernel<TAG, N, 2>), dim3((unsigned)tt), dim3(256), lds, st, p)
  if (d->dtype == CP_F32) {
    switch (NT) { case 1: CP_HS2(F32Tag, 1); break; case 2: CP_HS2(F32Tag, 2); break; case 3: CP_HS2(F32Tag, 3); break;
                  case 4: CP_HS2(F32Tag, 4); break; default: CP_HS2(F32Tag, 5); break; }
  } else {
    switch (NT) { case 1: CP_HS2(BF16Tag, 1); break; case 2: CP_HS2(BF16Tag, 2); break; case 3: CP_HS2(BF16Tag, 3); break;
                  case 4: CP_HS2(BF16Tag, 4); break; default: CP_HS2(BF16Tag, 5); break; }
  }
#undef CP_HS2
  return cp_check_launch();
}

// ---- grouped small-Cout convs (see conv3x3_halo_s_group_kernel)
static int build_halo_params(const CpConvDesc* d, const void* in, const void* packed_w, const float* scale, const float* shift,
                             const void* residual, void* out, HaloParams* pp, long long* tiles);

extern "C" int cp_conv3x3_halo_group_supported(int dtype, int H, int W, int Cout_phys) {
  return ((dtype == CP_F32 || dtype == CP_BF16) && H >= 8 && W >= 16 && Cout_phys > 0 && Cout_phys <= 80 && halo_small(Cout_phys)) ? 1 : 0;
}

extern "C" int cp_conv3x3_halo_item(const CpConvDesc* d, const void* in, const void* packed_w, const float* scale, const float* shift,
                                    const void* residual, void* out, CpConvGroupItem* item) {
  if (!item) return CP_ERR_INVALID;
  HaloGroupItem it;
  memset(&it, 0, sizeof(it));
  long long tt;
  const int rc = build_halo_params(d, in, packed_w, scale, shift, residual, out, &it.p, &tt);
  if (rc) return rc;
  if (!cp_conv3x3_halo_group_supported(d->dtype, d->H, d->W, d->Cout)) return CP_ERR_INVALID;
  it.NT = (d->Cout + 15) / 16;
  it.blocks = (uint32_t)tt;
  it.lds_bytes = (uint32_t)(HBUF + 9 * it.NT * 1024);
  memcpy(item, &it, sizeof(it));
  return CP_OK;
}

extern "C" int cp_conv3x3_halo_group(cp_stream_t stream, int dtype, const CpConvGroupItem* items_dev, const uint32_t* prefix_dev, int n_items,
                                     uint32_t total_blocks, uint32_t lds_bytes) {
  if (!items_dev || !prefix_dev || n_items <= 0 || n_items > 16 || total_blocks == 0 || lds_bytes < HBUF + 9 * 1024 ||
      lds_bytes > HBUF + 9 * 5 * 1024)
    return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (!cp_aligned16(items_dev)) return CP_ERR_ALIGN;
  // the kernel is built for <= 3 and <= 5 channel tiles per layer (registers and LDS per block follow the LARGEST layer of the group)
  const bool small = lds_bytes <= HBUF + 9 * 3 * 1024;
  const HaloGroupItem* it = (const HaloGroupItem*)items_dev;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CP_F32) {
    if (small) CP_LAUNCH((conv3x3_halo_s_group_kernel<F32Tag, 3>), dim3(total_blocks), dim3(256), lds_bytes, st, it, prefix_dev, n_items);
    else CP_LAUNCH((conv3x3_halo_s_group_kernel<F32Tag, 5>), dim3(total_blocks), dim3(256), lds_bytes, st, it, prefix_dev, n_items);
  } else {
    if (small) CP_LAUNCH((conv3x3_halo_s_group_kernel<BF16Tag, 3>), dim3(total_blocks), dim3(256), lds_bytes, st, it, prefix_dev, n_items);
    else CP_LAUNCH((conv3x3_halo_s_group_kernel<BF16Tag, 5>), dim3(total_blocks), dim3(256), lds_bytes, st, it, prefix_dev, n_items);
  }
  return cp_check_launch();
}
