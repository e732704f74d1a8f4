// Input side on the device, second half of SURVEY.md 8f row N3: the RoI crop of the reference's data loader
// (bop_dataset_pytorch.py:132-145 get_roi -> crop_square_resize :55-91 / crop_resize :94-108 -> cv2.resize) for a whole batch in ONE
// launch, from full uint8 images that are already in HBM -- a scene's image is uploaded once (0.9 MB at 640 x 480) instead of one
// 196 KB crop per detection, and the host does no pixel work.  The output is the uint8 crop itself (B, crop, crop, C), the operand of
// cp_u8hwc_to_nhwc_norm / of the model's uint8 forward.
//
// A crop is described by a window (x1, y1, x2, y2, roi_w, roi_h) computed on the host with the reference's own integer arithmetic
// (checkerpose_amd/preprocess.py): roi pixel (ry, rx) is image pixel (y1 + ry, x1 + rx) where that lies in
// [max(x1, 0), min(x2, W)) x [max(y1, 0), min(y2, H)) and zero elsewhere; the roi_h x roi_w roi is resized to crop x crop like
// cv2.resize does for 8-bit images (resize.cpp): INTER_LINEAR in fixed point (coefficients rint(c * 2048) as shorts, column index
// clamped with the fraction reset at both borders, row indices clamped with the coefficients kept, int32 horizontal pass, vertical
// pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2), or INTER_NEAREST (min(floor(d * scale), size - 1)).
// Byte gathers: HBM / L2 bound, one thread per output pixel, all channels.
#include "common.h"

struct CropParams {
  const uint8_t* images; const int32_t* win; const int32_t* img_idx; uint8_t* out;
  int n_img, H, W, C, B, crop, interp;
};

__device__ __forceinline__ void lin_coef(int d, double scale, int size, bool reset, int& i0, int& i1, int& c0, int& c1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (reset) {                       // columns
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= size - 1) { s = size - 1; f = 0.f; }
  }
  c0 = (int)rintf((1.f - f) * 2048.f);
  c1 = (int)rintf(f * 2048.f);
  i0 = min(max(s, 0), size - 1);
  i1 = min(max(s + 1, 0), size - 1);
}

__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const CropParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // over B * crop * crop
  const size_t total = (size_t)p.B * p.crop * p.crop;
  if (i >= total) return;
  const int dx = (int)(i % p.crop);
  const int dy = (int)((i / p.crop) % p.crop);
  const int b = (int)(i / ((size_t)p.crop * p.crop));
  const int32_t* w = p.win + 6 * b;
  const int x1 = w[0], y1 = w[1], x2 = w[2], y2 = w[3], rw = w[4], rh = w[5];
  uint8_t* o = p.out + i * p.C;
  const int im = p.img_idx ? p.img_idx[b] : (p.n_img == 1 ? 0 : b);
  if (rw <= 0 || rh <= 0 || im < 0 || im >= p.n_img) {
    for (int c = 0; c < p.C; ++c) o[c] = 0;
    return;
  }
  const uint8_t* img = p.images + (size_t)im * p.H * p.W * p.C;
  const int vx0 = max(x1, 0), vx1 = min(x2, p.W), vy0 = max(y1, 0), vy1 = min(y2, p.H);
  auto px = [&](int ry, int rx, int c) -> int {
    const int iy = y1 + ry, ix = x1 + rx;
    return (iy >= vy0 && iy < vy1 && ix >= vx0 && ix < vx1) ? (int)img[((size_t)iy * p.W + ix) * p.C + c] : 0;
  };
  const double sx = 1.0 / ((double)p.crop / (double)rw), sy = 1.0 / ((double)p.crop / (double)rh);     // cv2: 1. / inv_scale
  if (p.interp == 0) {
    const int cx = min((int)floor((double)dx * sx), rw - 1), cy = min((int)floor((double)dy * sy), rh - 1);
    for (int c = 0; c < p.C; ++c) o[c] = (uint8_t)px(cy, cx, c);
    return;
  }
  int xa, xb, a0, a1, ya, yb, b0, b1;
  lin_coef(dx, sx, rw, true, xa, xb, a0, a1);
  lin_coef(dy, sy, rh, false, ya, yb, b0, b1);
  for (int c = 0; c < p.C; ++c) {
    const int h0 = px(ya, xa, c) * a0 + px(ya, xb, c) * a1;
    const int h1 = px(yb, xa, c) * a0 + px(yb, xb, c) * a1;
    int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (uint8_t)min(max(v, 0), 255);
  }
}

extern "C" int cp_crop_resize_u8(cp_stream_t stream, const uint8_t* images, int n_img, int H, int W, int C, const int32_t* windows,
                                 const int32_t* img_idx, uint8_t* out, int B, int crop, int interpolation) {
  if (!images || !windows || !out || n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4 || B <= 0 || crop <= 0 ||
      (interpolation != 0 && interpolation != 1))
    return CP_ERR_INVALID;
  if (!img_idx && n_img != 1 && n_img != B) return CP_ERR_INVALID;          // which image does crop b come from?
  if ((size_t)n_img * H * W * C >= ((size_t)1 << 40) || (size_t)H * W >= ((size_t)1 << 31)) return CP_ERR_RANGE;
  CropParams p;
  p.images = images; p.win = windows; p.img_idx = img_idx; p.out = out;
  p.n_img = n_img; p.H = H; p.W = W; p.C = C; p.B = B; p.crop = crop; p.interp = interpolation;
  const size_t total = (size_t)B * crop * crop;
  CP_LAUNCH(crop_resize_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return cp_check_launch();
}
