// Memory-bound channels-last helpers: bilinear x2 (align_corners), HRNet fuse sum, 3x3/s2 max pool.
// One thread per 16-byte channel group (4 f32 / 8 bf16): every access is a coalesced 16-byte vector.
#include "common.h"

// ---- nn.UpsamplingBilinear2d(scale_factor=2)  (pipeline.py:199; == F.interpolate(align_corners=True)).
// ATen (UpSample.h, align_corners): scale = (in-1)/(out-1); src = scale*dst; i0 = floor(src); i1 = i0 + (i0 < in-1);
// l1 = src - i0; l0 = 1 - l1;  out = l0h*(l0w*x00 + l1w*x01) + l1h*(l0w*x10 + l1w*x11)   -- all in fp32.
// grid.y = one output row (b, oy): the row weights / source rows are block-uniform and the only per-thread index split is
// (ox, channel group) -- a shift when the group count is a power of two (the decoder's 256 / 512 channels).
template <typename Tag>
__global__ void upsample2x_bilinear_kernel(const void* __restrict__ in, void* __restrict__ out, int H, int W, int CG, int cg_shift,
                                           int in_cs, int in_coff, int out_cs, int out_coff, float sy, float sx, unsigned rows) {
  // one thread = one 16-byte channel group of TWO horizontally adjacent output pixels (2 j, 2 j + 1): their source columns
  // overlap (scale < 1/2), so 6 loads feed 2 stores instead of 8 -- and twice the bytes are in flight per thread
  constexpr int E = Tag::E;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;           // over W * CG
  if (idx >= W * CG) return;
  const int j = cg_shift >= 0 ? idx >> cg_shift : idx / CG;
  const int g = idx - j * CG;
  const unsigned row = blockIdx.z * gridDim.y + blockIdx.y;        // (b, oy); rows beyond B*2H come from the grid round-up
  if (row >= rows) return;
  const int oy = (int)(row % (unsigned)(2 * H));
  const size_t b = row / (unsigned)(2 * H);
  int y0, ystep; float ly1;
  cp_up_coord(sy, oy, H, y0, ystep, ly1);
  const int y1 = y0 + ystep;
  const float ly0 = cp_one_minus(ly1);
  const u32x4* src = (const u32x4*)in;
  const size_t r0 = ((b * H + y0) * W) * in_cs + in_coff + (size_t)g * E, r1 = ((b * H + y1) * W) * in_cs + in_coff + (size_t)g * E;
  const int oxa = 2 * j, oxb = 2 * j + 1;
  int xa0, xb0, xas, xbs; float lxa1, lxb1;
  cp_up_coord(sx, oxa, W, xa0, xas, lxa1);
  cp_up_coord(sx, oxb, W, xb0, xbs, lxb1);
  const int xa1 = xa0 + xas, xb1 = xb0 + xbs;
  // distinct source columns: xa0 <= xb0 <= xa0 + 1, so {xa0, xa1, xb1} covers all four (xb0 is xa0 or xa1)
  u32x4 t0 = src[(r0 + (size_t)xa0 * in_cs) / E], t1 = src[(r0 + (size_t)xa1 * in_cs) / E], t2 = src[(r0 + (size_t)xb1 * in_cs) / E];
  u32x4 u0 = src[(r1 + (size_t)xa0 * in_cs) / E], u1 = src[(r1 + (size_t)xa1 * in_cs) / E], u2 = src[(r1 + (size_t)xb1 * in_cs) / E];
  float a[E], bb[E], c[E], d[E], o[E];
  {
    const float lx1 = lxa1, lx0 = cp_one_minus(lx1);
    Vec16<Tag>::unpack(t0, a); Vec16<Tag>::unpack(t1, bb); Vec16<Tag>::unpack(u0, c); Vec16<Tag>::unpack(u1, d);
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = cp_bilerp(a[e], bb[e], c[e], d[e], lx0, lx1, ly0, ly1);
    const size_t oe = ((b * 2 * H + oy) * 2 * W + oxa) * out_cs + out_coff + (size_t)g * E;
    ((u32x4*)out)[oe / E] = Vec16<Tag>::pack(o);
  }
  {
    const float lx1 = lxb1, lx0 = cp_one_minus(lx1);
    const bool same = xb0 == xa0;                                  // left column of the second pixel: xa0 or xa1
    Vec16<Tag>::unpack(same ? t0 : t1, a); Vec16<Tag>::unpack(same ? t1 : t2, bb);
    Vec16<Tag>::unpack(same ? u0 : u1, c); Vec16<Tag>::unpack(same ? u1 : u2, d);
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = cp_bilerp(a[e], bb[e], c[e], d[e], lx0, lx1, ly0, ly1);
    const size_t oe = ((b * 2 * H + oy) * 2 * W + oxb) * out_cs + out_coff + (size_t)g * E;
    ((u32x4*)out)[oe / E] = Vec16<Tag>::pack(o);
  }
}

extern "C" int cp_upsample2x_bilinear_ac(cp_stream_t stream, int dtype, const void* in, void* out, int B, int H, int W,
                                         int C, int in_cstride, int in_coff, int out_cstride, int out_coff) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || in_cstride % E || in_coff % E || out_cstride % E || out_coff % E) return CP_ERR_ALIGN;
  if (in_coff + C > in_cstride || out_coff + C > out_cstride) return CP_ERR_INVALID;
  if (!cp_aligned16(in) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const float sy = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
  const float sx = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  const int CG = C / E;
  int cg_shift = -1;
  for (int k = 0; k < 16; ++k) if ((1 << k) == CG) cg_shift = k;
  const long long rows = (long long)B * 2 * H;
  if (rows >= (1LL << 31)) return CP_ERR_RANGE;
  const unsigned gy = rows > 65535 ? 65535u : (unsigned)rows, gz = (unsigned)((rows + gy - 1) / gy);
  if (gz > 65535) return CP_ERR_RANGE;
  const dim3 grid((unsigned)((W * CG + 255) / 256), gy, gz);
  if (dtype == CP_F32)
    CP_LAUNCH(upsample2x_bilinear_kernel<F32Tag>, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W,
                       CG, cg_shift, in_cstride, in_coff, out_cstride, out_coff, sy, sx, (unsigned)rows);
  else
    CP_LAUNCH(upsample2x_bilinear_kernel<BF16Tag>, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W,
                       CG, cg_shift, in_cstride, in_coff, out_cstride, out_coff, sy, sx, (unsigned)rows);
  return cp_check_launch();
}

// ---- HRNet fuse: out = relu?( sum_t src_t[b, y>>sh_t, x>>sh_t, :] )   (timm HighResolutionModule.forward;
// nn.Upsample(mode='nearest', scale 2^sh): src index = floor(dst / 2^sh)).  Summation order t = 0..nsrc-1
// follows the reference's `y = y + fuse_outer[j](x[j])` loop.
struct FuseSrcs { const void* p[4]; int sh[4]; };

template <typename Tag>
__global__ void fuse_sum_kernel(FuseSrcs s, int nsrc, void* __restrict__ out, int H, int W, int CG, int relu, size_t total,
                                int out_sg, int out_og) {
  constexpr int E = Tag::E;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*W*CG
  if (i >= total) return;
  const int g = (int)(i % CG);
  size_t t = i / CG;
  const int x = (int)(t % W); t /= W;
  const int y = (int)(t % H);
  const size_t b = t / H;
  float acc[E], f[E];
#pragma unroll
  for (int j = 0; j < E; ++j) acc[j] = 0.f;
  for (int k = 0; k < nsrc; ++k) {
    const int sh = s.sh[k];
    const size_t v = ((b * (H >> sh) + (y >> sh)) * (W >> sh) + (x >> sh)) * CG + g;
    Vec16<Tag>::unpack(((const u32x4*)s.p[k])[v], f);
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] = (k == 0) ? f[j] : acc[j] + f[j];
  }
  if (relu) {
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] = fmaxf(acc[j], 0.f);
  }
  ((u32x4*)out)[(i / CG) * out_sg + out_og + g] = Vec16<Tag>::pack(acc);     // channel slice [out_coff, out_coff + C) of out_cstride
}

extern "C" int cp_fuse_sum_act(cp_stream_t stream, int dtype, int nsrc, const void* const* srcs, const int32_t* shifts,
                               void* out, int B, int H, int W, int C, int relu, int out_cstride, int out_coff) {
  if (!srcs || !shifts || !out || nsrc < 1 || nsrc > 4 || B <= 0 || H <= 0 || W <= 0 || C <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || !cp_aligned16(out)) return CP_ERR_ALIGN;
  if (out_cstride % E || out_coff % E || out_coff < 0 || out_coff + C > out_cstride) return CP_ERR_ALIGN;
  FuseSrcs s;
  for (int k = 0; k < 4; ++k) { s.p[k] = nullptr; s.sh[k] = 0; }
  for (int k = 0; k < nsrc; ++k) {
    if (!srcs[k] || !cp_aligned16(srcs[k]) || shifts[k] < 0 || shifts[k] > 5) return CP_ERR_INVALID;
    if ((H >> shifts[k]) << shifts[k] != H || (W >> shifts[k]) << shifts[k] != W) return CP_ERR_INVALID;
    s.p[k] = srcs[k]; s.sh[k] = shifts[k];
  }
  const int CG = C / E;
  const size_t total = (size_t)B * H * W * CG;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(fuse_sum_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, nsrc, out, H, W, CG, relu, total,
              out_cstride / E, out_coff / E);
  else
    CP_LAUNCH(fuse_sum_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, nsrc, out, H, W, CG, relu, total,
              out_cstride / E, out_coff / E);
  return cp_check_launch();
}

// ---- F.max_pool2d(x, 3, 2, 1) (resnet34 stem). Padding is -inf, i.e. out-of-range taps are skipped.
template <typename Tag>
__global__ void maxpool3x3s2_kernel(const void* __restrict__ in, void* __restrict__ out, int H, int W, int CG, size_t total) {
  constexpr int E = Tag::E;
  const int Ho = H / 2, Wo = W / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int g = (int)(i % CG);
  size_t t = i / CG;
  const int ox = (int)(t % Wo); t /= Wo;
  const int oy = (int)(t % Ho);
  const size_t b = t / Ho;
  float m[E], f[E];
#pragma unroll
  for (int j = 0; j < E; ++j) m[j] = -INFINITY;
  for (int r = 0; r < 3; ++r) {
    const int y = 2 * oy - 1 + r;
    if ((unsigned)y >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      const int x = 2 * ox - 1 + s;
      if ((unsigned)x >= (unsigned)W) continue;
      Vec16<Tag>::unpack(((const u32x4*)in)[((b * H + y) * W + x) * CG + g], f);
#pragma unroll
      for (int j = 0; j < E; ++j) m[j] = fmaxf(m[j], f[j]);
    }
  }
  ((u32x4*)out)[i] = Vec16<Tag>::pack(m);
}

extern "C" int cp_maxpool3x3s2(cp_stream_t stream, int dtype, const void* in, void* out, int B, int H, int W, int C) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || !cp_aligned16(in) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const int CG = C / E;
  const size_t total = (size_t)B * (H / 2) * (W / 2) * CG;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(maxpool3x3s2_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, H, W, CG, total);
  else
    CP_LAUNCH(maxpool3x3s2_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, H, W, CG, total);
  return cp_check_launch();
}
