// Index2Feat_module (reference pipeline.py:130-164) + the RoI mask (pipeline.py:280) in ONE launch, computing the
// patch_generator conv ONLY where it is gathered (bf16, 256 -> 64 channels, k = 2, pad = 1).
//
// The reference convolves the whole (H+1) x (W+1) patch map and then picks 4 taps per keypoint: at the 64 x 64 stage that is
// 4225 positions per crop for 2048 used ones (N = 512), 0.45 - 0.56 ms at batch 256 on the tail of the forward (the last
// refinement stage can only start when the decoder is done), plus the gather launch.  Here the conv is a gathered GEMM:
//   row (b, n, tap t): patch position (py, px) = (2 y_id + k [t & 1], 2 x_id + k [t >> 1]);   K = (dy, dx, ci) = 4 x 256:
//   out[b, n, 64 t + c] = mask[b, n] * ( bias[c] + sum_{dy, dx, ci} W[c, ci, dy, dx] * f[b, py - 1 + dy, px - 1 + dx, ci] )
// One 8-wave workgroup = 128 keypoints x 4 taps of one crop; wave w owns 16 keypoints, its 4 MFMA fragments are the 4 taps.
// Activation fragments are gathered straight from the channels-last map into registers (lane = keypoint x 16-byte channel
// group: 64 contiguous bytes per keypoint per load), 4 K-chunks ahead of the MFMAs; out-of-image window pixels (the conv's
// zero padding) become out-of-range buffer offsets -> zeros.  Weights (128 KB) stream L2 -> registers -> LDS in four 32 KB
// slabs (one per window pixel), double-buffered.  Same arithmetic as conv + gather (fp32 accumulate, bias, mask, one bf16
// rounding), but the bf16 rounding of the un-masked patch value is skipped (the conv output no longer exists as a tensor).
#include "common.h"

namespace {

constexpr int PG_CIN = 256, PG_COUT = 64, PG_SLAB = 32 * 1024;       // one window pixel: 8 chunks x 4 tiles x 1 KB
constexpr int PG_LDS = 2 * PG_SLAB;

struct PatchParams {
  const void* f; const void* w; const float* bias; const int32_t* xid; const int32_t* yid; const float* mask; void* out;
  int B, N, H, W, in_cs, in_coff, out_cs, out_coff, k;
  uint32_t in_bytes;
};

template <bool HOUT>                                           // HOUT: the output rows leave as IEEE half (CP_F16: the keypoint side's storage type)
__global__ __launch_bounds__(512) void patch_gather_kernel(const PatchParams p) {
  if constexpr (HOUT) cp_f16_saturate_on();                  // half output rows saturate at +-65504 (common.h)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int groups = (p.N + 127) / 128;
  const int b = blockIdx.x / groups, grp = blockIdx.x - b * groups;
  const int n = grp * 128 + wave * 16 + x;                      // this lane's keypoint
  const bool nok = n < p.N;

  // weight slabs go global -> registers -> LDS (NOT LDS-DMA: with a DMA in flight hipcc waits vmcnt(0) for every ordinary
  // load, which would serialise the 4-deep activation prefetch below)
  const u32x4* const wg = (const u32x4*)p.w;
  u32x4 wv[4];
  auto slab_load = [&](int s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = wg[(size_t)s * (PG_SLAB / 16) + tid + 512 * k];
  };
  auto slab_write = [&](int s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *(u32x4*)(smem + (s & 1) * PG_SLAB + (tid + 512 * k) * 16) = wv[k];
  };
  slab_load(0);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.f), 0, p.in_bytes, 0x00020000);
  const int yi = nok ? p.yid[(size_t)b * p.N + n] : 0, xi = nok ? p.xid[(size_t)b * p.N + n] : 0;
  // byte offset of window pixel (dy, dx) of tap t, channel group q of chunk 0; 0x80000000 = outside the image -> zeros
  auto pix_off = [&](int t, int s) -> uint32_t {
    const int py = 2 * yi + ((t & 1) ? p.k : 0) - 1 + (s >> 1), px = 2 * xi + ((t >> 1) ? p.k : 0) - 1 + (s & 1);
    const bool ok = nok & ((unsigned)py < (unsigned)p.H) & ((unsigned)px < (unsigned)p.W);
    return ok ? (uint32_t)((((size_t)b * p.H + py) * p.W + px) * p.in_cs + p.in_coff + q * 8) * 2u : 0x80000000u;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int AHEAD = 4;                                      // activation fragments in flight: chunks kc .. kc + 3
  u32x4 af[AHEAD][4];
  uint32_t po[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) po[t] = pix_off(t, 0);
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      af[a][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, po[t] == 0x80000000u ? po[t] : po[t] + a * 64, 0, 0));
  slab_write(0);
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    if (s + 1 < 4) slab_load(s + 1);
    uint32_t pn[4];                                             // next window pixel's offsets (chunks wrap into it)
#pragma unroll
    for (int t = 0; t < 4; ++t) pn[t] = s + 1 < 4 ? pix_off(t, s + 1) : 0x80000000u;
    const unsigned char* const wb = smem + (s & 1) * PG_SLAB + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      u32x4 wf[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const u32x4*)(wb + (kc * 4 + nt) * 1024);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, af[kc % AHEAD][t]),
                                                               acc[t][nt], 0, 0, 0);
      // refill the slot just consumed with chunk kc + AHEAD (of this window pixel, or of the next one)
      const int kn = kc + AHEAD;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t base = kn < 8 ? po[t] : pn[t];
        const uint32_t off = base == 0x80000000u ? base : base + (uint32_t)((kn & 7) * 64);
        af[kc % AHEAD][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) po[t] = pn[t];
    if (s + 1 < 4) slab_write(s + 1);                            // the other buffer: its readers passed the previous barrier
    __syncthreads();
  }

  // ---- epilogue: lane (x, q): keypoint n, channels 16 q + 4 nt + reg of every tap
  if (!nok) return;
  const float mk = p.mask[(size_t)b * p.N + n];
  float bs[16];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 b4 = *(const f32x4*)(p.bias + 16 * q + 4 * nt);
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[4 * nt + j] = b4[j];
  }
  uint16_t* const ob = (uint16_t*)p.out + ((size_t)b * p.N + n) * p.out_cs + p.out_coff + 16 * q;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = (acc[t][nt][j] + bs[4 * nt + j]) * mk;
    *(u32x4*)(ob + t * PG_COUT) = cp_pack8<HOUT>(v);
    *(u32x4*)(ob + t * PG_COUT + 8) = cp_pack8<HOUT>(v + 8);
  }
}

// [window pixel s = 2 dy + dx][chunk kc 8][tile nt 4][lane][8 bf16]: lane (row r, q), element e: input channel 32 kc + 8 q + e;
// tile row r of tile nt = output channel 16 (r >> 2) + 4 nt + (r & 3)
__global__ void pack_patch_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over 4 * 8 * 4 * 512
  if (i >= 4 * 8 * 4 * 512) return;
  const int e = i % 8, lane = (i / 8) % 64, nt = (i / 512) % 4, kc = (i / 2048) % 8, s = i / 16384;
  const int r = lane & 15, q = lane >> 4;
  const int c = 16 * (r >> 2) + 4 * nt + (r & 3), ci = 32 * kc + 8 * q + e;
  out[i] = (uint16_t)f32_to_bf16_bits(w[(((size_t)c * PG_CIN + ci) * 2 + (s >> 1)) * 2 + (s & 1)]);
}

}  // namespace

extern "C" int cp_index2feat_conv_supported(int Cin, int E_ch, int k) { return (Cin == PG_CIN && E_ch == PG_COUT && k == 2) ? 1 : 0; }

extern "C" size_t cp_index2feat_conv_weight_bytes(void) { return (size_t)4 * PG_SLAB; }

extern "C" int cp_pack_index2feat_conv_weight(cp_stream_t stream, const float* w, void* packed) {
  if (!w || !packed) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  CP_LAUNCH(pack_patch_weight_kernel, dim3(4 * 8 * 4 * 512 / 256), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)packed);
  return cp_check_launch();
}

extern "C" int cp_index2feat_conv_t(cp_stream_t stream, int out_dtype, const void* f, int in_cstride, int in_coff, const void* packed_w,
                                    const float* bias, const int32_t* x_id, const int32_t* y_id, const float* mask, void* out, int B, int N,
                                    int H, int W, int k, int out_cstride, int out_coff) {
  if (out_dtype != CP_BF16 && out_dtype != CP_F16) return CP_ERR_INVALID;
  if (!f || !packed_w || !bias || !x_id || !y_id || !mask || !out || B <= 0 || N <= 0 || H <= 0 || W <= 0 || k != 2) return CP_ERR_INVALID;
  if (in_cstride % 8 || in_coff % 8 || in_coff + PG_CIN > in_cstride || out_cstride % 8 || out_coff % 8 || out_coff + 4 * PG_COUT > out_cstride)
    return CP_ERR_ALIGN;
  if (!cp_aligned16(f) || !cp_aligned16(packed_w) || !cp_aligned16(bias) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const long long in_bytes = (long long)B * H * W * in_cstride * 2;
  if (in_bytes >= (1LL << 31)) return CP_ERR_RANGE;
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)patch_gather_kernel<false>, PG_LDS) &&
                                  cp_set_max_lds((const void*)patch_gather_kernel<true>, PG_LDS));
  PatchParams p;
  p.f = f; p.w = packed_w; p.bias = bias; p.xid = x_id; p.yid = y_id; p.mask = mask; p.out = out;
  p.B = B; p.N = N; p.H = H; p.W = W; p.in_cs = in_cstride; p.in_coff = in_coff; p.out_cs = out_cstride; p.out_coff = out_coff; p.k = k;
  p.in_bytes = (uint32_t)in_bytes;
  const unsigned grid = (unsigned)(B * ((N + 127) / 128));
  if (out_dtype == CP_F16) CP_LAUNCH((patch_gather_kernel<true>), dim3(grid), dim3(512), PG_LDS, (hipStream_t)stream, p);
  else CP_LAUNCH((patch_gather_kernel<false>), dim3(grid), dim3(512), PG_LDS, (hipStream_t)stream, p);
  return cp_check_launch();
}

extern "C" int cp_index2feat_conv(cp_stream_t stream, const void* f, int in_cstride, int in_coff, const void* packed_w, const float* bias,
                                  const int32_t* x_id, const int32_t* y_id, const float* mask, void* out, int B, int N, int H, int W,
                                  int k, int out_cstride, int out_coff) {
  return cp_index2feat_conv_t(stream, CP_BF16, f, in_cstride, in_coff, packed_w, bias, x_id, y_id, mask, out, B, N, H, W, k, out_cstride,
                              out_coff);
}
