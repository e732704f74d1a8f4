// Fused Bottleneck (timm resnet.Bottleneck inside HRNet layer1, blocks 1..3):
//     out = relu( bn3(conv1x1( relu(bn2(conv3x3( relu(bn1(conv1x1(x))) ))) )) + x )      256 -> 64 -> 64 -> 256, bf16
// Unfused, one block moves x (2 MB/crop), t1 (write+read), t2 (write+read), the residual x again and out: 8 MB per
// crop at 64x64; the three launches are bandwidth/latency-bound (~2.7 TB/s effective).  Fused it is x (with a 1-pixel
// halo, 1.4x) + out = 4.8 MB.  One persistent 8-wave block per CU, per 8x16 output tile:
//   stage : the 10x18 halo of x (180 px x 512 B = 92 KB) -> LDS as [8-channel piece][pixel][16 B] planes, loads in
//           full-line order (consecutive lanes walk one pixel's 512 bytes).  The next tile's loads are issued into
//           registers before the current tile's store phase, so their latency hides behind it.
//   conv1 : 1x1, K=256 on all 180 halo pixels (12 fragments): wave w owns channel tile w&3 and 6 fragments;
//           BN1+ReLU, pixels outside the image forced to 0 (they are conv2's zero padding) -> t1 in LDS.
//   conv2 : 3x3 on t1 -> 8x16 pixels, wave w owns channel tile w&3 and 4 rows; BN2+ReLU -> t2 in LDS.
//   conv3 : 1x1 64->256, wave w owns channel tiles 2w, 2w+1 for all 8 rows; BN3 + residual read from the x halo in
//           LDS (no second global read) + ReLU, written back IN PLACE over the x tile;
//   store : the finished 128 px x 512 B tile leaves LDS in full-line order.
// conv2's and conv3's weight fragments stay in REGISTERS for the life of the block (18 + 4 fragments = 88 VGPRs),
// conv1's 8 are re-read from L2 per tile: per tile the HBM traffic is the x halo in and the output tile out.
// Tile order: crops b == xcd (mod 8) run on XCD `xcd` (blockIdx % 8), neighbouring tiles of a crop at the same time,
// so the halo rows two tiles share are L2 hits.
#include "common.h"

namespace {

constexpr int BTH = 8, BTW = 16, BPW = BTW + 2, BPH = BTH + 2;   // 10 x 18 halo
constexpr int BNPIX = BPH * BPW;                                 // 180 halo pixels, planes padded to 192
constexpr int PITCH_X = 192 * 16 + 16;                           // +16: consecutive planes shift by one 16-byte slot
constexpr int PITCH_T = 128 * 16 + 16;
constexpr int MPLANES = 8, OPLANES = 32;                         // 64 / 8, 256 / 8 channel pieces
constexpr int ST1_BYTES = MPLANES * PITCH_X, ST2_BYTES = MPLANES * PITCH_T, SO_BYTES = OPLANES * PITCH_T;
// XCH = 64-byte chunks of the input row: 8 (256 channels, identity shortcut, output written in place over the x tile)
// or 2 (64 channels, projection shortcut `downsample`, output tile in its own LDS region).
constexpr int AFF_FLOATS = 4 * 64 + 4 * 256;                     // s1 t1 s2 t2 [64] | s3 t3 sd td [256]
constexpr int lds_main(int XCH, bool HAS_DS) { return XCH * 4 * PITCH_X + ST1_BYTES + ST2_BYTES + (HAS_DS ? SO_BYTES : 0); }   // 140 032 / 131 968
constexpr int lds_bytes(int XCH, bool HAS_DS) { return lds_main(XCH, HAS_DS) + AFF_FLOATS * 4; }

struct BottleneckParams {
  const void* in; void* out;
  const void *w1, *w2, *w3, *wd;
  const float *s1, *t1, *s2, *t2, *s3, *t3, *sd, *td;
  int B, H, W, in_cs, in_coff, tiles_x, tiles_y;
  uint32_t in_bytes;
  long long o_base, o_sb, o_sy, o_sx;
};

__device__ __forceinline__ void mma_bf16(const u32x4& w, const u32x4& a, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

// epilogue arithmetic on channel PAIRS (v_pk_fma_f32 / v_pk_add_f32, one v_cvt_pk_bf16_f32, ReLU as max(int16, 0) on the packed pair):
// the phase clock (tools/bottleneck_stamps.py) put conv3 + residual at 25 % of a tile with 288 scalar VALU instructions per wave
// beside its 32 MFMAs.  Same fma / add / rounding per value as the scalar form; a result of -0.0 is stored as +0.0.
__device__ __forceinline__ cp_f32x2 bn_pair(const f32x4& v, int h) { return h ? cp_f32x2{v[2], v[3]} : cp_f32x2{v[0], v[1]}; }
__device__ __forceinline__ cp_f32x2 bn_bf16_pair(uint32_t r) { return cp_f32x2{__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u)}; }
__device__ __forceinline__ uint32_t bn_pack_relu(cp_f32x2 v) {
  return relu_bf16x2(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, cp_bf16x2)));
}

#ifdef CP_DEBUG_KNOBS          // phase clock of workgroup 0 (tools/bottleneck_stamps.py): s_memtime sums per wave, -DCP_DEBUG_KNOBS builds only
__device__ unsigned long long b_stamps[8][10];
#define B_T0() unsigned long long b_t = __builtin_amdgcn_s_memtime(), b_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define B_MARK(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); b_acc[k] += n_ - b_t; b_t = n_; } while (0)
#define B_DUMP() do { if (blockIdx.x == 0 && lane == 0) for (int k_ = 0; k_ < 10; ++k_) b_stamps[wave][k_] = b_acc[k_]; } while (0)
#else
#define B_T0() do {} while (0)
#define B_MARK(k) do {} while (0)
#define B_DUMP() do {} while (0)
#endif

template <int XCH, bool HAS_DS>
__global__ __launch_bounds__(512) void bottleneck_fused_kernel(const BottleneckParams p) {
  constexpr int XPLANES = XCH * 4;
  constexpr int SX_BYTES = XPLANES * PITCH_X;
  constexpr int XITER = XCH == 8 ? 12 : (192 * XPLANES + 511) / 512;   // 10 halo rows + 2 extra-column loads | generic
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sX = smem;
  unsigned char* const sT1 = smem + SX_BYTES;
  unsigned char* const sT2 = sT1 + ST1_BYTES;
  unsigned char* const sO = sT2 + ST2_BYTES;                     // HAS_DS only

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int x = lane & 15, q = lane >> 4;
  const int nt = wave & 3, half = wave >> 2;
  // BN scale / shift vectors live in LDS: an epilogue that fetched them from global memory would have to wait (vmcnt is
  // in-order) for the next tile's halo loads, which are in flight during conv3
  float* const sAff = (float*)(smem + lds_main(XCH, HAS_DS));
  {
    const float* const v64[4] = {p.s1, p.t1, p.s2, p.t2};
    const float* const v256[4] = {p.s3, p.t3, p.sd, p.td};
    if (tid < 256) sAff[tid] = v64[tid >> 6][tid & 63];
#pragma unroll
    for (int k = 0; k < (HAS_DS ? 4 : 2); ++k)
      if (tid < 256) sAff[256 + k * 256 + tid] = v256[k][tid];
  }
  const float* const a_s1 = sAff, * const a_t1 = sAff + 64, * const a_s2 = sAff + 128, * const a_t2 = sAff + 192;
  const float* const a_s3 = sAff + 256, * const a_t3 = sAff + 512, * const a_sd = sAff + 768, * const a_td = sAff + 1024;

  // ---- block-resident weights (generic packing [tile][chunk][lane][16 B], see cp_pack_conv_weight)
  u32x4 W2[18], W3[2][2];
#pragma unroll
  for (int k = 0; k < 18; ++k) W2[k] = ((const u32x4*)p.w2)[(nt * 18 + k) * 64 + lane];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) W3[t][c] = ((const u32x4*)p.w3)[((2 * wave + t) * 2 + c) * 64 + lane];

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const int tpi = p.tiles_x * p.tiles_y;
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nbx = gridDim.x >> 3;

  // halo loads of local tile index li (crop (li / tpi) * 8 + xcd); a tile past the end loads zeros.
  // Thread -> (column tid>>5, piece tid&31) walks halo rows 0..9 of columns 0..15 (offsets affine in the row: no
  // per-iteration lane constants for the compiler to hoist and spill); the two right-hand columns take 2 more loads.
  const int scol = tid >> 5, spc = tid & 31;
  const int erow = tid >> 6, ecol = BTW + ((tid >> 5) & 1);     // extras: rows 0..7 (it 10), rows 8..9 (it 11, tid < 128)
  auto issue_loads = [&](int li, u32x4* xv, int lo = 0, int hi = 1 << 20) {
    const int b = (li / tpi) * 8 + xcd;
    const int trem = li % tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * BTH - 1, x0 = tx * BTW - 1;
    const bool tok = b < p.B;
    if constexpr (XCH == 8) {
      {
        const int gx = x0 + scol;
        const bool cok = tok & ((unsigned)gx < (unsigned)p.W);
        const int base = ((b * p.H + y0) * p.W + gx) * p.in_cs + p.in_coff + spc * 8;
#pragma unroll
        for (int it = 0; it < BPH; ++it) {
          if (it < lo || it >= hi) continue;
          const bool ok = cok & ((unsigned)(y0 + it) < (unsigned)p.H);
          const uint32_t off = ok ? (uint32_t)(base + it * p.W * p.in_cs) * 2u : 0x80000000u;
          xv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (BPH + e < lo || BPH + e >= hi) continue;
        const int gy = y0 + erow + 8 * e, gx = x0 + ecol;
        const bool ok = tok & (e == 0 || tid < 128) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
        const uint32_t off = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + spc * 8) * 2u : 0x80000000u;
        xv[BPH + e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
      }
    } else {                                                     // narrow input: piece i = tid + 512 it -> (pixel, piece)
#pragma unroll
      for (int it = 0; it < XITER; ++it) {
        if (it < lo || it >= hi) continue;
        const int i = tid + 512 * it;
        const int hp = i / XPLANES, pc = i % XPLANES;
        const int py = hp / BPW, px = hp - py * BPW;
        const int gy = y0 + py, gx = x0 + px;
        const bool ok = tok & (hp < BNPIX) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
        const uint32_t off = ok ? (uint32_t)(((b * p.H + gy) * p.W + gx) * p.in_cs + p.in_coff + pc * 8) * 2u : 0x80000000u;
        xv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
      }
    }
  };

  // pad pixels 180..191 of the x planes are read by conv1's last fragment (results discarded): zero them once
  if constexpr (XCH == 8) for (int i = tid; i < XPLANES * 12; i += 512) *(u32x4*)(sX + (i / 12) * PITCH_X + (BNPIX + i % 12) * 16) = u32x4{0u, 0u, 0u, 0u};

  u32x4 xv[XITER];
  int li = j0;
  issue_loads(li, xv);
  B_T0();
  for (; (li / tpi) * 8 + xcd < p.B; li += nbx) {
    const int b = (li / tpi) * 8 + xcd;
    const int trem = li % tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * BTH, x0 = tx * BTW;

    // conv1: wave w owns channel tiles 2 (w & 1), 2 (w & 1) + 1 and halo fragments 3 (w >> 1) .. + 2: every x fragment read
    // from LDS feeds two MFMAs (one tile per wave and six fragments read each fragment four times: LDS-port bound)
    const int nt1 = 2 * (wave & 1), fr1 = 3 * (wave >> 1);
    u32x4 W1[2][XCH];
    // conv1's weight fragments come from L2 per tile (16 KB per wave: resident next to conv2's 18 and conv3's 4 fragments they do not
    // fit 256 VGPRs).  The first W1_EARLY K chunks take off BEFORE the halo leaves its registers, so their L2 latency runs under the
    // staging writes (vmcnt is in-order: the halo loads are older and return first); all 8 early spill (measured: 4 early -1.6 %)
    constexpr int W1_EARLY = XCH < 4 ? XCH : 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kc = 0; kc < W1_EARLY; ++kc) W1[t][kc] = ((const u32x4*)p.w1)[((nt1 + t) * XCH + kc) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage: registers -> LDS planes
    if constexpr (XCH == 8) {
      unsigned char* dst = sX + spc * PITCH_X + scol * 16;
#pragma unroll
      for (int it = 0; it < BPH; ++it) *(u32x4*)(dst + it * BPW * 16) = xv[it];
      unsigned char* de = sX + spc * PITCH_X + (erow * BPW + ecol) * 16;
      *(u32x4*)de = xv[BPH];
      if (tid < 128) *(u32x4*)(de + 8 * BPW * 16) = xv[BPH + 1];
    } else {
#pragma unroll
      for (int it = 0; it < XITER; ++it) {
        const int i = tid + 512 * it;                            // pad pixels receive the zeros of their invalid loads
        *(u32x4*)(sX + (i % XPLANES) * PITCH_X + (i / XPLANES) * 16) = xv[it];
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kc = W1_EARLY; kc < XCH; ++kc) W1[t][kc] = ((const u32x4*)p.w1)[((nt1 + t) * XCH + kc) * 64 + lane];
    // the first HALO_EARLY of the next tile's halo loads take off right behind conv1's weights (vmcnt is in-order: conv1 waits for
    // its weights with these still in flight): as many as fit beside conv1's operands without spilling (3: -1.5 %; 4 spill)
    constexpr int HALO_EARLY = 3;
    __builtin_amdgcn_sched_barrier(0);
    issue_loads(li + nbx, xv, 0, HALO_EARLY);
    __builtin_amdgcn_sched_barrier(0);
    B_MARK(0);
    __syncthreads();
    B_MARK(1);

    // ---- conv1 (1x1, K = 256): channel tiles nt1, nt1 + 1, fragments fr1 .. fr1 + 2 of the halo
    {
      f32x4 acc[3][2];
#pragma unroll
      for (int f = 0; f < 3; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int kc = 0; kc < XCH; ++kc) {
        const unsigned char* base = sX + (kc * 4 + q) * PITCH_X + (fr1 * 16 + x) * 16;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const u32x4 a = *(const u32x4*)(base + f * 256);
          mma_bf16(W1[0][kc], a, acc[f][0]);
          mma_bf16(W1[1][kc], a, acc[f][1]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c0 = (nt1 + t) * 16 + q * 4;
        const f32x4 sc = *(const f32x4*)(a_s1 + c0), sh = *(const f32x4*)(a_t1 + c0);
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const int p1 = (fr1 + f) * 16 + x;
          const int py = p1 / BPW, px = p1 - py * BPW;
          const int gy = y0 - 1 + py, gx = x0 - 1 + px;
          const bool inimg = (p1 < BNPIX) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
          u32x2 pk;
          pk.x = bn_pack_relu(bn_pair(acc[f][t], 0) * bn_pair(sc, 0) + bn_pair(sh, 0));
          pk.y = bn_pack_relu(bn_pair(acc[f][t], 1) * bn_pair(sc, 1) + bn_pair(sh, 1));
          if (!inimg) pk = u32x2{0u, 0u};
          *(u32x2*)(sT1 + (c0 >> 3) * PITCH_X + p1 * 16 + (c0 & 7) * 2) = pk;
        }
      }
    }
    B_MARK(2);
    __syncthreads();
    B_MARK(3);
    // the rest of the next tile's halo loads take off now (conv1's weight fragments are dead) and stay in registers until this
    // tile's output has left the x planes: they are in flight under conv2, conv3 and the store phase
    issue_loads(li + nbx, xv, HALO_EARLY);

    // ---- conv2 (3x3 on t1): channel tile nt, rows 4*half .. 4*half+3
    {
      f32x4 acc[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      // per (column shift, channel half): the 6 t1 rows this wave's 4 output rows touch are read ONCE and feed the three
      // row taps (12 MFMAs per 6 fragment reads; tap-major order read every fragment up to three times -- the LDS port,
      // not the matrix pipe, bounds this phase)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const unsigned char* base = sT1 + (c2 * 4 + q) * PITCH_X + ((half * 4) * BPW + x + s3) * 16;
          u32x4 rowf[6];
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) rowf[rr] = *(const u32x4*)(base + rr * BPW * 16);
#pragma unroll
          for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
            for (int f = 0; f < 4; ++f) mma_bf16(W2[(r3 * 3 + s3) * 2 + c2], rowf[r3 + f], acc[f]);
        }
      }
      const int c0 = nt * 16 + q * 4;
      const f32x4 sc = *(const f32x4*)(a_s2 + c0), sh = *(const f32x4*)(a_t2 + c0);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        u32x2 pk;
        pk.x = bn_pack_relu(bn_pair(acc[f], 0) * bn_pair(sc, 0) + bn_pair(sh, 0));
        pk.y = bn_pack_relu(bn_pair(acc[f], 1) * bn_pair(sc, 1) + bn_pair(sh, 1));
        *(u32x2*)(sT2 + (c0 >> 3) * PITCH_T + ((half * 4 + f) * 16 + x) * 16 + (c0 & 7) * 2) = pk;
      }
    }
    B_MARK(4);
    __syncthreads();
    B_MARK(5);
    // ---- conv3 (1x1, K = 64): channel tiles 2*wave, 2*wave+1, all 8 rows (two passes of 4: bounds the live
    // accumulators + operands); + residual (x tile in LDS), written back in place
    constexpr int RP = HAS_DS ? 2 : 4;                           // rows per pass (two accumulator sets with the shortcut)
#pragma unroll 1
    for (int fh = 0; fh < 8 / RP; ++fh) {
      f32x4 acc[RP][2], accd[HAS_DS ? RP : 1][2];
#pragma unroll
      for (int f = 0; f < RP; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const unsigned char* base = sT2 + (c2 * 4 + q) * PITCH_T + (fh * RP * 16 + x) * 16;
#pragma unroll
        for (int f = 0; f < RP; ++f) {
          const u32x4 a = *(const u32x4*)(base + f * 256);
          mma_bf16(W3[0][c2], a, acc[f][0]);
          mma_bf16(W3[1][c2], a, acc[f][1]);
        }
      }
      if constexpr (HAS_DS) {                                    // projection shortcut: 1x1 conv of the tile's own x pixels
        u32x4 Wd[2][2];                                          // re-read per pass (L2): resident they would spill
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 2; ++c) Wd[t][c] = ((const u32x4*)p.wd)[((2 * wave + t) * 2 + c) * 64 + lane];
#pragma unroll
        for (int f = 0; f < RP; ++f) { accd[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accd[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c2 = 0; c2 < XCH; ++c2) {
          const unsigned char* base = sX + (c2 * 4 + q) * PITCH_X + ((fh * RP + 1) * BPW + 1 + x) * 16;
#pragma unroll
          for (int f = 0; f < RP; ++f) {
            const u32x4 a = *(const u32x4*)(base + f * BPW * 16);
            mma_bf16(Wd[0][c2], a, accd[f][0]);
            mma_bf16(Wd[1][c2], a, accd[f][1]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c0 = (2 * wave + t) * 16 + q * 4;
        const f32x4 sc = *(const f32x4*)(a_s3 + c0), sh = *(const f32x4*)(a_t3 + c0);
        if constexpr (HAS_DS) {
          const f32x4 scd = *(const f32x4*)(a_sd + c0), shd = *(const f32x4*)(a_td + c0);
          unsigned char* col = sO + (c0 >> 3) * PITCH_T + (fh * RP * 16 + x) * 16 + (c0 & 7) * 2;
#pragma unroll
          for (int f = 0; f < RP; ++f) {
            u32x2 pk;
            pk.x = bn_pack_relu(bn_pair(acc[f][t], 0) * bn_pair(sc, 0) + bn_pair(sh, 0) + (bn_pair(accd[f][t], 0) * bn_pair(scd, 0) + bn_pair(shd, 0)));
            pk.y = bn_pack_relu(bn_pair(acc[f][t], 1) * bn_pair(sc, 1) + bn_pair(sh, 1) + (bn_pair(accd[f][t], 1) * bn_pair(scd, 1) + bn_pair(shd, 1)));
            *(u32x2*)(col + f * 256) = pk;
          }
        } else {
          unsigned char* col = sX + (c0 >> 3) * PITCH_X + ((fh * RP + 1) * BPW + 1 + x) * 16 + (c0 & 7) * 2;
#pragma unroll
          for (int f = 0; f < RP; ++f) {
            u32x2* rp = (u32x2*)(col + f * BPW * 16);
            const u32x2 r2 = *rp;
            u32x2 pk;
            pk.x = bn_pack_relu(bn_pair(acc[f][t], 0) * bn_pair(sc, 0) + bn_pair(sh, 0) + bn_bf16_pair(r2.x));
            pk.y = bn_pack_relu(bn_pair(acc[f][t], 1) * bn_pair(sc, 1) + bn_pair(sh, 1) + bn_bf16_pair(r2.y));
            *rp = pk;
          }
        }
      }
    }
    B_MARK(6);
    __syncthreads();
    B_MARK(7);

    // ---- this tile's output leaves LDS (the next tile's halo loads have been in flight since conv1)
    {
      const int ox = x0 + scol;
      const unsigned char* src = HAS_DS ? sO + spc * PITCH_T + scol * 16 : sX + spc * PITCH_X + (BPW + 1 + scol) * 16;
      constexpr int RSTEP = HAS_DS ? 256 : BPW * 16;
      uint16_t* gp = (uint16_t*)p.out + p.o_base + (long long)b * p.o_sb + (long long)y0 * p.o_sy + (long long)ox * p.o_sx + spc * 8;
#pragma unroll
      for (int row = 0; row < BTH; ++row) {
        const u32x4 v = *(const u32x4*)(src + row * RSTEP);
        if (ox < p.W && y0 + row < p.H) *(u32x4*)(gp + (long long)row * p.o_sy) = v;
      }
    }
    B_MARK(8);
    __syncthreads();
    B_MARK(9);
  }
  B_DUMP();
}

}  // namespace

#ifdef CP_DEBUG_KNOBS
extern "C" int cp_debug_bottleneck_stamps(unsigned long long* out80) {
  return hipMemcpyFromSymbol(out80, HIP_SYMBOL(b_stamps), sizeof(unsigned long long) * 80) == hipSuccess ? CP_OK : CP_ERR_HIP;
}
#endif

extern "C" int cp_bottleneck_fused(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w1,
                                   const float* scale1, const float* shift1, const void* packed_w2, const float* scale2,
                                   const float* shift2, const void* packed_w3, const float* scale3, const float* shift3,
                                   const void* packed_wd, const float* scaled, const float* shiftd, void* out) {
  if (!d || !in || !out || !packed_w1 || !packed_w2 || !packed_w3 || !scale1 || !shift1 || !scale2 || !shift2 || !scale3 || !shift3)
    return CP_ERR_INVALID;
  const bool ds = packed_wd != nullptr;
  if (ds != (scaled != nullptr) || ds != (shiftd != nullptr)) return CP_ERR_INVALID;
  if (d->dtype != CP_BF16 || d->out_f32 || d->o_sc != 1) return CP_ERR_INVALID;          // bf16 storage only (LDS budget)
  if (d->Cin != (ds ? 64 : 256) || d->Cout != 256 || d->stride != 1 || d->Ho != d->H || d->Wo != d->W) return CP_ERR_INVALID;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || in == out) return CP_ERR_INVALID;
  if (d->in_coff % 8 || d->in_cstride % 8 || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(out) || !cp_aligned16(packed_w1) || !cp_aligned16(packed_w2) || !cp_aligned16(packed_w3) ||
      !cp_aligned16(scale1) || !cp_aligned16(shift1) || !cp_aligned16(scale2) || !cp_aligned16(shift2) || !cp_aligned16(scale3) ||
      !cp_aligned16(shift3) || !cp_aligned16(packed_wd) || !cp_aligned16(scaled) || !cp_aligned16(shiftd))
    return CP_ERR_ALIGN;
  if ((d->o_base % 8) || (d->o_sb % 8) || (d->o_sy % 8) || (d->o_sx % 8)) return CP_ERR_ALIGN;
  const long long in_bytes = (long long)d->B * d->H * d->W * d->in_cstride * 2;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  BottleneckParams p;
  p.in = in; p.out = out; p.w1 = packed_w1; p.w2 = packed_w2; p.w3 = packed_w3; p.wd = packed_wd;
  p.s1 = scale1; p.t1 = shift1; p.s2 = scale2; p.t2 = shift2; p.s3 = scale3; p.t3 = shift3; p.sd = scaled; p.td = shiftd;
  p.B = d->B; p.H = d->H; p.W = d->W; p.in_cs = d->in_cstride; p.in_coff = d->in_coff;
  p.tiles_x = (d->W + BTW - 1) / BTW; p.tiles_y = (d->H + BTH - 1) / BTH;
  p.in_bytes = (uint32_t)in_bytes;
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)bottleneck_fused_kernel<8, false>, lds_bytes(8, false)) &&
                                  cp_set_max_lds((const void*)bottleneck_fused_kernel<2, true>, lds_bytes(2, true)));
  const int n_cu = cp_num_cus();
  if (n_cu <= 0) return CP_ERR_HIP;
  // one persistent block per CU, a multiple of 8 (XCD labels); never more blocks than tiles of the crops one XCD owns
  const long long per_xcd = (long long)((d->B + 7) / 8) * p.tiles_x * p.tiles_y;
  long long nbx = n_cu / 8 > 0 ? n_cu / 8 : 1;
  if (nbx > per_xcd) nbx = per_xcd;
  const dim3 grid((unsigned)(8 * nbx));
  if (ds) CP_LAUNCH((bottleneck_fused_kernel<2, true>), grid, dim3(512), lds_bytes(2, true), (hipStream_t)stream, p);
  else CP_LAUNCH((bottleneck_fused_kernel<8, false>), grid, dim3(512), lds_bytes(8, false), (hipStream_t)stream, p);
  return cp_check_launch();
}
