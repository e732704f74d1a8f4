// HRNet stem in ONE launch (bf16): NCHW fp32 crop -> conv1 3x3/s2 (3 -> 64) + BN + ReLU -> conv2 3x3/s2 (64 -> 64) + BN + ReLU
// (timm HighResolutionNet.conv1/bn1/conv2/bn2 inside timm.create_model("hrnet_w18"), reference backbone.py:48-49; restated
// oracle/checkerpose_oracle.py hrnet_features).
//
// Unfused this was a layout kernel + two generic implicit-GEMM launches that moved the 128 x 128 x 64 intermediate (2 MB per
// crop) out to HBM and back: 0.8 ms per step at batch 256 for 0.36 GFLOP per crop, ~2 TB/s.  Here one persistent 8-wave
// workgroup walks a crop's 32 tiles of 8 x 16 conv2 outputs; per tile
//   image : the 35 x 67 input patch (3 planes of the NCHW fp32 image) -> bf16 [row][col][4] in LDS (prefetched into registers
//           one tile ahead);
//   conv1 : on MFMA as a 27-deep GEMM (one 32-deep chunk, k = 3 tap + channel): the im2col fragment of 16 pixels is gathered
//           from the LDS patch with 8 two-byte reads per lane; all 17 x 33 intermediate pixels conv2 needs; BN1 + ReLU, pixels
//           outside the 128 x 128 map forced to 0 (they are conv2's zero padding) -> LDS as [8-channel group][pixel][16 B]
//           with the columns DE-INTERLEAVED (even columns first): conv2's stride-2 taps then read 16 consecutive slots;
//   conv2 : wave w owns output-channel tile w & 3 (its 18 weight fragments live in registers for the whole crop) and 4 of the
//           8 tile rows; BN2 + ReLU -> HBM.
// HBM traffic: the image once (1.14x patch overlap) + the 64 x 64 x 64 output.
#include "common.h"

namespace {

constexpr int ST_TH = 8, ST_TW = 16;                  // conv2 output tile
constexpr int ST_R1 = 2 * ST_TH + 1, ST_C1 = 2 * ST_TW + 1;      // 17 x 33 conv1 pixels
constexpr int ST_IR = 2 * ST_R1 + 1, ST_IC = 2 * ST_C1 + 1;      // 35 x 67 image pixels
constexpr int ST_ICP = 68;                            // image row pitch (pixels of 8 bytes)
constexpr int ST_IMG = ST_IR * ST_ICP * 8;            // 19 040
constexpr int ST_T1PX = 576;                          // 17 * 33 = 561 pixels per plane, padded to 36 fragments
constexpr int ST_PLANE = ST_T1PX * 16;
constexpr int ST_LDS = ST_IMG + 8 * ST_PLANE;         // 92 768
constexpr int ST_NF1 = ST_T1PX / 16;                  // 36 conv1 fragments

struct StemParams {
  const float* img; const void* w1; const void* w2; const float *s1, *t1, *s2, *t2; void* out;
  int B, Hin, Win;
};

__global__ __launch_bounds__(512) void hr_stem_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sImg = smem;
  unsigned char* const sT1 = smem + ST_IMG;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  const int H1 = p.Hin >> 1, W1 = p.Win >> 1, H2 = p.Hin >> 2, W2 = p.Win >> 2;
  const int tiles_x = W2 / ST_TW, ntiles = (H2 / ST_TH) * tiles_x;

  // ---- weights: conv1 (4 fragments) and this wave's conv2 tile (18 fragments) in registers
  const int nt2 = wave & 3, half = wave >> 2;
  u32x4 W1f[4], W2f[18];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) W1f[nt] = ((const u32x4*)p.w1)[nt * 64 + lane];
#pragma unroll
  for (int kc = 0; kc < 18; ++kc) W2f[kc] = ((const u32x4*)p.w2)[(nt2 * 18 + kc) * 64 + lane];
  // conv1 epilogue: lane q holds channels 16 q + 4 nt + reg;  conv2 epilogue: channels 16 nt2 + 4 q + reg
  float s1v[16], t1v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { s1v[j] = p.s1[16 * q + j]; t1v[j] = p.t1[16 * q + j]; }
  const f32x4 s2v = *(const f32x4*)(p.s2 + 16 * nt2 + 4 * q), t2v = *(const f32x4*)(p.t2 + 16 * nt2 + 4 * q);

  // ---- im2col constants of this lane's K group: k = 8 q + e = 3 tap + c  ->  byte offset inside the image patch
  uint32_t ko[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * q + e;
    const int tap = k / 3, c = k - 3 * tap;
    const int dr = tap / 3, dc = tap - 3 * dr;
    ko[e] = k < 27 ? (uint32_t)((dr * ST_ICP + dc) * 8 + c * 2) : 0u;
  }

  // image patch staging: piece i = tid + 512 k = (patch row, ALIGNED group of 4 image columns 64 tx - 4 + 4 g): three 16-byte
  // loads (one per colour plane; a row is contiguous in NCHW) -> four [c0 c1 c2 0] bf16 pixels.  The image width is a multiple
  // of 4, so a group lies entirely inside or entirely outside the row; outside (or past the last tile) it reads through an
  // out-of-range buffer offset -> zeros: no branch, and the loop keeps counted vmcnt waits.  (One 4-byte load per pixel and
  // plane cost 144 of the kernel's 301 us.)
  constexpr int IGR = 18;                             // columns c0 - 1 .. c0 + 70 cover the patch's c0 .. c0 + 66
  constexpr int IPC = ST_IR * IGR;                    // 630 pieces
  constexpr int IIT = (IPC + 511) / 512;              // 2
  const size_t plane = (size_t)p.Hin * p.Win;
  const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.img + (size_t)b * 3 * plane), 0,
                                                                          (uint32_t)(3 * plane * 4), 0x00020000);
  u32x4 iv[IIT][3];
  auto img_load = [&](int t) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int r0 = 4 * ty * ST_TH - 3, c0 = 4 * tx * ST_TW - 3;
#pragma unroll
    for (int k = 0; k < IIT; ++k) {
      const int i = tid + 512 * k;
      const int pr = i / IGR, pg = i - pr * IGR;
      const int gy = r0 + pr, gx = c0 - 1 + 4 * pg;
      const bool ok = (i < IPC) & (t < ntiles) & ((unsigned)gy < (unsigned)p.Hin) & ((unsigned)gx < (unsigned)p.Win);
      const uint32_t off = ok ? (uint32_t)((gy * p.Win + gx) * 4) : 0x80000000u;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        iv[k][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off + (uint32_t)(c * plane * 4), 0, 0));
    }
  };
  auto img_write = [&]() {
#pragma unroll
    for (int k = 0; k < IIT; ++k) {
      const int i = tid + 512 * k;
      const int pr = i / IGR, pg = i - pr * IGR;
      if (i < IPC) {
        const uint32_t c0v[4] = {iv[k][0].x, iv[k][0].y, iv[k][0].z, iv[k][0].w};
        const uint32_t c1v[4] = {iv[k][1].x, iv[k][1].y, iv[k][1].z, iv[k][1].w};
        const uint32_t c2v[4] = {iv[k][2].x, iv[k][2].y, iv[k][2].z, iv[k][2].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pc = 4 * pg + e - 1;              // patch column of image column gx + e
          if (pc >= 0 && pc < ST_IC)
            *(u32x2*)(sImg + (pr * ST_ICP + pc) * 8) = u32x2{pack_bf16x2(__uint_as_float(c0v[e]), __uint_as_float(c1v[e])),
                                                              pack_bf16x2(__uint_as_float(c2v[e]), 0.f)};
        }
      }
    }
  };
  // the pad column (pixel 67 of every row) and nothing else is ever read uninitialised: zero the patch once
  for (int i = tid; i < ST_IMG / 8; i += 512) *(u32x2*)(sImg + i * 8) = u32x2{0u, 0u};
  img_load(0);
  __syncthreads();
  img_write();
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < ntiles; ++t) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    img_load(t + 1);                                   // in flight during this tile's MFMAs

    // ---- conv1 on the 17 x 33 ring: fragment f covers stored pixels [16 f, 16 f + 16) of the (row, de-interleaved col) order
#pragma unroll 1
    for (int f = wave; f < ST_NF1; f += 8) {
      const int i = f * 16 + x;
      const int ic = i < ST_R1 * ST_C1 ? i : 0;
      const int row = ic / ST_C1, j = ic - row * ST_C1;
      const int crel = j < 17 ? 2 * j : 2 * (j - 17) + 1;       // column inside the ring
      const unsigned char* base = sImg + ((2 * row) * ST_ICP + 2 * crel) * 8;
      uint32_t a[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const uint32_t lo = *(const uint16_t*)(base + ko[2 * e2]), hi = *(const uint16_t*)(base + ko[2 * e2 + 1]);
        a[e2] = lo | (hi << 16);
      }
      const u32x4 af = u32x4{a[0], a[1], a[2], a[3]};
      f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W1f[nt]), __builtin_bit_cast(bf16x8, af),
                                                          f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const int r1 = 2 * oy0 - 1 + row, c1 = 2 * ox0 - 1 + crel;
      const bool in1 = (i < ST_R1 * ST_C1) & ((unsigned)r1 < (unsigned)H1) & ((unsigned)c1 < (unsigned)W1);
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * nt + r] = in1 ? fmaxf(acc[nt][r] * s1v[4 * nt + r] + t1v[4 * nt + r], 0.f) : 0.f;
      *(u32x4*)(sT1 + (2 * q) * ST_PLANE + i * 16) = Vec16<BF16Tag>::pack(v);
      *(u32x4*)(sT1 + (2 * q + 1) * ST_PLANE + i * 16) = Vec16<BF16Tag>::pack(v + 8);
    }
    __syncthreads();                                   // the intermediate tile is complete; the image patch is free
    img_write();                                       // next tile's patch (visible after the barrier that ends this tile)

    // ---- conv2: output rows 4 half .. 4 half + 3 of the tile, channel tile nt2
    f32x4 acc2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments one K chunk ahead, pinned (left alone hipcc sinks each ds_read to its MFMA: one LDS latency per MFMA)
    auto c2_load = [&](u32x4* af, int kc) {
      const int tap = kc >> 1, ch = kc & 1;
      const int dr = tap / 3, dc = tap - 3 * dr;
      const int jb = dc == 1 ? 17 : (dc == 2 ? 1 : 0);           // de-interleaved column of ring column 2 x + dc
      const unsigned char* base = sT1 + (ch * 4 + q) * ST_PLANE + ((2 * 4 * half + dr) * ST_C1 + jb + x) * 16;
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = *(const u32x4*)(base + (2 * m) * ST_C1 * 16);
    };
    u32x4 af2[2][4];
    c2_load(af2[0], 0);
#pragma unroll
    for (int kc = 0; kc < 18; ++kc) {
      if (kc + 1 < 18) c2_load(af2[(kc + 1) & 1], kc + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W2f[kc]), __builtin_bit_cast(bf16x8, af2[kc & 1][m]), acc2[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      uint16_t* const ob = (uint16_t*)p.out + (((size_t)b * H2 + oy0 + 4 * half) * W2 + ox0 + x) * 64 + 16 * nt2 + 4 * q;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc2[m][r] * s2v[r] + t2v[r], 0.f);
        *(u32x2*)(ob + (size_t)m * W2 * 64) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
    __syncthreads();                                   // everyone is done with the intermediate tile
  }
}

// conv1: [nt 4][lane][8]: lane (row r, q), element e: k = 8 q + e = 3 tap + c (k < 27); row r of tile nt = channel 16 (r >> 2) + 4 nt + (r & 3)
// conv2: [nt 4][kc 18][lane][8]: kc = 2 tap + half, input channel 32 half + 8 q + e; row r of tile nt = channel 16 nt + r
__global__ void pack_stem_kernel(const float* __restrict__ w1, const float* __restrict__ w2, uint16_t* __restrict__ p1, uint16_t* __restrict__ p2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4 * 512) {
    const int e = i % 8, lane = (i / 8) % 64, nt = i / 512;
    const int r = lane & 15, q = lane >> 4;
    const int k = 8 * q + e;
    const int n = 16 * (r >> 2) + 4 * nt + (r & 3);
    p1[i] = (uint16_t)f32_to_bf16_bits(k < 27 ? w1[(n * 3 + (k % 3)) * 9 + k / 3] : 0.f);
  }
  if (i < 4 * 18 * 512) {
    const int e = i % 8, lane = (i / 8) % 64, kc = (i / 512) % 18, nt = i / (512 * 18);
    const int r = lane & 15, q = lane >> 4;
    const int tap = kc >> 1, cin = 32 * (kc & 1) + 8 * q + e;
    const int n = 16 * nt + r;
    p2[i] = (uint16_t)f32_to_bf16_bits(w2[((size_t)n * 64 + cin) * 9 + tap]);
  }
}

}  // namespace

extern "C" size_t cp_hr_stem_weight_bytes(int which) { return which == 0 ? 4 * 1024 : 4 * 18 * 1024; }

extern "C" int cp_pack_hr_stem_weights(cp_stream_t stream, const float* w1, const float* w2, void* packed1, void* packed2) {
  if (!w1 || !w2 || !packed1 || !packed2) return CP_ERR_INVALID;
  if (!cp_aligned16(packed1) || !cp_aligned16(packed2)) return CP_ERR_ALIGN;
  CP_LAUNCH(pack_stem_kernel, dim3((4 * 18 * 512 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, w2, (uint16_t*)packed1, (uint16_t*)packed2);
  return cp_check_launch();
}

extern "C" int cp_hr_stem(cp_stream_t stream, const float* img_nchw, int B, int Hin, int Win, const void* packed1, const float* scale1,
                          const float* shift1, const void* packed2, const float* scale2, const float* shift2, void* out) {
  if (!img_nchw || !packed1 || !packed2 || !scale1 || !shift1 || !scale2 || !shift2 || !out || B <= 0) return CP_ERR_INVALID;
  if (Hin <= 0 || Win <= 0 || Hin % (4 * ST_TH) || Win % (4 * ST_TW)) return CP_ERR_INVALID;     // whole 8 x 16 output tiles
  if (!cp_aligned16(packed1) || !cp_aligned16(packed2) || !cp_aligned16(scale2) || !cp_aligned16(shift2) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)hr_stem_kernel, ST_LDS));
  StemParams p;
  p.img = img_nchw; p.w1 = packed1; p.w2 = packed2; p.s1 = scale1; p.t1 = shift1; p.s2 = scale2; p.t2 = shift2; p.out = out;
  p.B = B; p.Hin = Hin; p.Win = Win;
  CP_LAUNCH(hr_stem_kernel, dim3((unsigned)B), dim3(512), ST_LDS, (hipStream_t)stream, p);
  return cp_check_launch();
}
