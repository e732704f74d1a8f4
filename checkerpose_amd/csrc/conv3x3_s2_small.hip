// 3x3 / stride 2 / pad 1 convolution with a WIDE input and few output channels (bf16): HRNet `transition1[1]`, 256 -> 36 at
// 64 x 64 -> 32 x 32 (timm HighResolutionNet.transition1 inside timm.create_model, reference backbone.py:48-49).
//
// On the generic implicit GEMM every input pixel is fetched once per tap that uses it, in fragment shape: 2.4x the input through
// the fabric (PMC: 1.3 GB for a 537 MB tensor), 226 us, and the transition phase -- this conv beside the stride-1 one reading the
// same tensor -- is bound by exactly that traffic.  This kernel: 170-178 us (tools/s2_bench.py).  Here a workgroup owns 8 input rows (4 output rows) of a crop and walks the
// input channels in chunks of 32:
//   LDS    : the chunk's 9 x (W + 1) ring pixels x 64 B as planes [8-channel group][pixel][16 B], even columns first, then the
//            odd ones (the stride-2 taps of 16 consecutive outputs read 16 consecutive slots); two buffers, the next chunk's
//            rows prefetched into registers under the current chunk's MFMAs; every input byte crosses the fabric 9/8 times;
//   GEMM   : MFMA K chunk = one tap (lane group q = channel group q of the 32): 9 chunks per channel chunk; wave (mt, half)
//            owns output-channel tile mt (of <= 3) and half of the band's pixel tiles, its 9 weight fragments of the chunk in
//            registers (126 VGPRs: two workgroups per CU, one computing while the other waits for its weights);
//   output : bf16 NHWC, 4 channels (8 bytes) per lane and pixel, folded BN + ReLU.
#include "common.h"

namespace {

constexpr int S2_BAND = 8;                                   // input rows per workgroup
constexpr int S2_MAXMT = 3;

struct S2Params {
  const void* in; const void* w; const float* scale; const float* shift; void* out;
  int B, H, W, in_cs, in_coff, nchunk, mt, out_cp, act;
  float slope;
  int plane_bytes, ntile;                                   // per 8-channel-group plane; 16-pixel output tiles per band
  uint32_t in_bytes, w_bytes;
  long long o_base, o_sb, o_sy, o_sx;
};

__device__ __forceinline__ void mma16s(const u32x4& w, const u32x4& a, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

template <int NPT>                                            // pixel tiles per wave (2 waves split the band's tiles)
__global__ __launch_bounds__(64 * 2 * S2_MAXMT, 2) void conv3x3_s2_small_kernel(const S2Params p) {
  constexpr int NTHR = 64 * 2 * S2_MAXMT;                     // 384
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 buffers x 4 planes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int mt = wave % S2_MAXMT, half = wave / S2_MAXMT;
  const bool mt_ok = mt < p.mt;
  const int nband = p.H / S2_BAND;
  const int b = blockIdx.x / nband, r0 = (blockIdx.x - b * nband) * S2_BAND;
  const int RW = p.W + 1, RH = S2_BAND + 1, Wo = p.W >> 1;
  const int HE = (p.W >> 1) + 1;                              // even ring columns 0, 2, .., W come first
  const int buf_bytes = 4 * p.plane_bytes;

  // ---- staging plan: piece i = tid + NTHR k -> (ring row ry, input column c, group g): 4 pieces of a pixel are 64 contiguous bytes
  constexpr int SIT = 6;                                      // 9 rows x 64 columns x 4 groups = 2304 pieces / 384 threads (W <= 64)
  const int per_row = p.W * 4;
  const int total = RH * per_row;
  // branch-free: pieces past the ring (or above the image) load through an out-of-range buffer offset (-> zeros) and land in
  // the plane's 16 pad bytes, so the loop keeps counted vmcnt waits (conditional loads made hipcc drain to vmcnt(0) per chunk)
  uint32_t s_goff[SIT], s_lds[SIT];
#pragma unroll
  for (int k = 0; k < SIT; ++k) {
    const int i = tid + NTHR * k;
    const int ry = i / per_row, rem = i - ry * per_row;
    const int c = rem >> 2, g = rem & 3;
    const int gy = r0 - 1 + ry;
    const bool ok = (i < total) & (gy >= 0);
    s_goff[k] = ok ? (uint32_t)(((b * p.H + gy) * p.W + c) * p.in_cs + p.in_coff + g * 8) * 2u : 0x80000000u;      // bytes
    const int rc = c + 1;
    const int slot = (rc & 1) ? HE + (rc >> 1) : (rc >> 1);
    s_lds[k] = i < total ? (uint32_t)(g * p.plane_bytes + (ry * RW + slot) * 16) : (uint32_t)(p.plane_bytes - 16);
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  auto stage_load = [&](u32x4* v, int c) {                    // chunks past the end: the offset leaves the buffer -> zeros
#pragma unroll
    for (int k = 0; k < SIT; ++k)
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, c < p.nchunk ? s_goff[k] + (uint32_t)c * 64u : 0x80000000u, 0, 0));
  };
  auto stage_write = [&](const u32x4* v, int buf) {
#pragma unroll
    for (int k = 0; k < SIT; ++k) *(u32x4*)(smem + buf * buf_bytes + s_lds[k]) = v[k];
  };
  // ring column 0 (left zero padding) of both buffers, once
  for (int i = tid; i < 2 * 4 * RH; i += NTHR) {
    const int buf = i / (4 * RH), r = i - buf * 4 * RH;
    *(u32x4*)(smem + buf * buf_bytes + (r / RH) * p.plane_bytes + ((r % RH) * RW) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- this wave's pixel tiles: output pixel n = (half * NPT + t) * 16 + x of the band (4 rows of Wo)
  uint32_t base[NPT];
#pragma unroll
  for (int t = 0; t < NPT; ++t) {
    int n = (half * NPT + t) * 16 + x;
    if (n >= (S2_BAND / 2) * Wo) n = 0;
    const int oy = n / Wo, ox = n - oy * Wo;
    base[t] = (uint32_t)(q * p.plane_bytes + ((2 * oy) * RW + ox) * 16);
  }
  // tap (r, s): ring column 2 ox + s -> even half slot ox (+1 for s = 2), odd half slot HE + ox
  uint32_t toff[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, s = tap - 3 * r;
    toff[tap] = (uint32_t)((r * RW + (s == 1 ? HE : s >> 1)) * 16);
  }

  f32x4 acc[NPT];
#pragma unroll
  for (int t = 0; t < NPT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weights: [mt][chunk][tap][lane][16 B]
  const uint32_t wbase = mt_ok ? (uint32_t)((mt * p.nchunk * 9 * 64 + lane) * 16) : 0x80000000u;
  u32x4 wcur[9];

  u32x4 sv[SIT];                                              // next chunk's rows, in flight under this chunk's MFMAs (a second set,
  stage_load(sv, 0);                                          // two chunks ahead, costs the CU's second workgroup: measured +-0)
  stage_write(sv, 0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < p.nchunk; ++c) {
    const int buf = c & 1;
    // this chunk's 9 weight fragments: one L2 latency per chunk, covered by the CU's second workgroup (168 VGPRs is the
    // limit for two of them)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      wcur[tap] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase + (uint32_t)((c * 9 + tap) * 1024), 0, 0));
    stage_load(sv, c + 1);
    __builtin_amdgcn_sched_barrier(0);                        // all 9 + 6 loads are in flight before the first wait (left alone,
    if (mt_ok) {                                              // hipcc sinks every load to its use: one L2 / LDS latency per MFMA)
      const unsigned char* const sb = smem + buf * buf_bytes;
      u32x4 a[2][NPT];
#pragma unroll
      for (int t = 0; t < NPT; ++t) a[0][t] = *(const u32x4*)(sb + base[t] + toff[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
#pragma unroll
          for (int t = 0; t < NPT; ++t) a[(tap + 1) & 1][t] = *(const u32x4*)(sb + base[t] + toff[tap + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NPT; ++t) mma16s(wcur[tap], a[tap & 1][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stage_write(sv, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane (x, q): pixel n, channels mt * 16 + 4 q .. + 3
  const int co = mt * 16 + 4 * q;
  if (mt_ok && co < p.out_cp) {
    const f32x4 sc = *(const f32x4*)(p.scale + co), sh = *(const f32x4*)(p.shift + co);
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
      const int n = (half * NPT + t) * 16 + x;
      if (n < (S2_BAND / 2) * Wo) {
        const int oy = n / Wo, ox = n - oy * Wo;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[t][j] * sc[j] + sh[j];
          v[j] = cp_act_apply(v[j], cp_act_slope(p.act, p.slope));
        }
        u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
        *(u32x2*)((uint16_t*)p.out + p.o_base + (long long)b * p.o_sb + (long long)((r0 >> 1) + oy) * p.o_sy + (long long)ox * p.o_sx + co) = pk;
      }
    }
  }
}

// [M tile][channel chunk][tap][lane][8 bf16]: lane (row = lane & 15, q = lane >> 4) element e = w[co = 16 mt + row][ci = 32 chunk + 8 q + e][tap]
__global__ void pack_s2_small_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin, int nchunk, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7);
  const int lane = (int)((i >> 3) & 63);
  size_t f = i >> 9;
  const int tap = (int)(f % 9); f /= 9;
  const int c = (int)(f % nchunk);
  const int mt = (int)(f / nchunk);
  const int co = mt * 16 + (lane & 15), ci = c * 32 + (lane >> 4) * 8 + e;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 9 + tap];
  out[i] = (uint16_t)f32_to_bf16_bits(v);
}

}  // namespace

extern "C" int cp_conv3x3_s2_small_supported(int H, int W, int cin_phys, int out_cphys) {
  if (H < S2_BAND || H % S2_BAND || W < 32 || W > 64 || W % 32 || cin_phys < 32 || cin_phys % 32) return 0;
  return (out_cphys > 0 && out_cphys % 8 == 0 && out_cphys <= 16 * S2_MAXMT) ? 1 : 0;
}

extern "C" size_t cp_conv3x3_s2_small_weight_bytes(int cin_phys, int out_cphys) {
  return (size_t)((out_cphys + 15) / 16) * (cin_phys / 32) * 9 * 1024;
}

extern "C" int cp_pack_conv3x3_s2_small_weight(cp_stream_t stream, const float* w, int Cout, int Cin, int cin_phys, int out_cphys,
                                               void* packed) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || Cin > cin_phys || Cout > out_cphys || cin_phys % 32 || out_cphys % 8) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const size_t total = cp_conv3x3_s2_small_weight_bytes(cin_phys, out_cphys) / 2;
  CP_LAUNCH(pack_s2_small_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)packed,
            Cout, Cin, cin_phys / 32, total);
  return cp_check_launch();
}

extern "C" int cp_conv3x3_s2_small(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w, const float* scale,
                                   const float* shift, void* out) {
  if (!d || !in || !packed_w || !scale || !shift || !out || !cp_act_ok(d->act, d->slope)) return CP_ERR_INVALID;
  if (d->dtype != CP_BF16 || d->out_f32 || d->o_sc != 1 || d->R != 3 || d->S != 3 || d->stride != 2 || d->pad != 1) return CP_ERR_INVALID;
  if (d->Ho != d->H / 2 || d->Wo != d->W / 2 || d->B <= 0) return CP_ERR_INVALID;
  if (!cp_conv3x3_s2_small_supported(d->H, d->W, d->Cin, d->Cout)) return CP_ERR_INVALID;
  if (d->in_coff % 8 || d->in_cstride % 8 || d->in_coff + d->Cin > d->in_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift)) return CP_ERR_ALIGN;
  if ((d->o_base % 4) || (d->o_sb % 4) || (d->o_sy % 4) || (d->o_sx % 4) || ((uintptr_t)out % 8)) return CP_ERR_ALIGN;
  if ((long long)d->B * d->H * d->W * d->in_cstride * 2 >= (1LL << 31)) return CP_ERR_RANGE;
  S2Params p;
  p.in = in; p.w = packed_w; p.scale = scale; p.shift = shift; p.out = out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.in_cs = d->in_cstride; p.in_coff = d->in_coff; p.nchunk = d->Cin / 32;
  p.mt = (d->Cout + 15) / 16; p.out_cp = d->Cout; p.act = d->act; p.slope = d->slope;
  p.in_bytes = (uint32_t)((long long)d->B * d->H * d->W * d->in_cstride * 2);
  p.w_bytes = (uint32_t)cp_conv3x3_s2_small_weight_bytes(d->Cin, d->Cout);
  p.plane_bytes = (S2_BAND + 1) * (d->W + 1) * 16 + 16;
  p.ntile = (S2_BAND / 2) * (d->W / 2) / 16;
  p.o_base = d->o_base; p.o_sb = d->o_sb; p.o_sy = d->o_sy; p.o_sx = d->o_sx;
  const size_t lds = (size_t)2 * 4 * p.plane_bytes;
  const long long grid = (long long)d->B * (d->H / S2_BAND);
  if (grid >= (1LL << 31)) return CP_ERR_RANGE;
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)conv3x3_s2_small_kernel<4>, 80 * 1024) &&
                                  cp_set_max_lds((const void*)conv3x3_s2_small_kernel<2>, 80 * 1024));
  // band = 4 output rows x W / 2 pixels: 8 tiles at W = 64 (4 per wave), 4 at W = 32 (2 per wave)
  if (p.ntile == 8) CP_LAUNCH((conv3x3_s2_small_kernel<4>), dim3((unsigned)grid), dim3(384), lds, (hipStream_t)stream, p);
  else CP_LAUNCH((conv3x3_s2_small_kernel<2>), dim3((unsigned)grid), dim3(384), lds, (hipStream_t)stream, p);
  return cp_check_launch();
}
