// Weight packing into MFMA-fragment order and boundary layout conversions.  Init-time / boundary
// plumbing, not the hot loop -- written for clarity, coalesced on the write side.
#include "common.h"

extern "C" int cp_chan_align(int dtype) { return (dtype == CP_BF16 || dtype == CP_F16) ? 8 : 4; }

extern "C" size_t cp_packed_weight_bytes(int dtype, int cout_rows, int cin_phys, int R, int S) {
  const int E = cp_chan_align(dtype), KCH = 4 * E;
  const size_t KC = ((size_t)R * S * cin_phys + KCH - 1) / KCH;
  const size_t tiles = ((size_t)cout_rows + 15) / 16;
  return tiles * KC * 1024;
}

// one thread per packed ELEMENT: index -> (tile, chunk, lane, e) -> (row n, k) -> (r, s, cin) -> source
template <typename Tag>
__global__ void pack_weight_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int R, int S,
                                   int cin_phys, int transposed, int phase, const int32_t* __restrict__ row_map,
                                   int cout_rows, int KC, size_t total) {
  constexpr int E = Tag::E;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % E);
  const int lane = (int)((i / E) % 64);
  const size_t blk = i / (E * 64);
  const int kc = (int)(blk % KC);
  const int tile = (int)(blk / KC);
  const int n = tile * 16 + (lane & 15);
  const int k = kc * (4 * E) + (lane >> 4) * E + e;
  const int tap = k / cin_phys, c = k - tap * cin_phys;
  const int r = tap / S, s = tap - r * S;
  float v = 0.f;
  int src_row = n < cout_rows ? (row_map ? row_map[n] : n) : -1;
  if (src_row >= 0 && src_row < Cout && c < Cin && r < R) {
    if (!transposed) {
      v = w[(((size_t)src_row * Cin + c) * R + r) * S + s];
    } else {
      // ConvTranspose2d(k3,s2,p1,op1): out[2i+a] = sum over taps t of x[i+t] * w[kh], kh = (a ? (t==0 ? 2 : 0) : 1)
      const int a = phase >> 1, b = phase & 1;
      const int kh = a ? (r == 0 ? 2 : 0) : 1;
      const int kw = b ? (s == 0 ? 2 : 0) : 1;
      v = w[(((size_t)c * Cout + src_row) * 3 + kh) * 3 + kw];   // weight layout (Cin, Cout, 3, 3)
    }
  }
  store_elem<Tag>(out, i, v);
}

extern "C" int cp_pack_conv_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int R, int S,
                                   int cin_phys, int transposed, int phase, const int32_t* row_map, int cout_rows,
                                   void* packed) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || cout_rows <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys < Cin || cin_phys % E) return CP_ERR_ALIGN;
  if (transposed && (phase < 0 || phase > 3 || R != 1 + (phase >> 1) || S != 1 + (phase & 1))) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int KCH = 4 * E;
  const int KC = (R * S * cin_phys + KCH - 1) / KCH;
  const size_t total = cp_packed_weight_bytes(dtype, cout_rows, cin_phys, R, S) / cp_elem_size(dtype);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(pack_weight_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       R, S, cin_phys, transposed, phase, row_map, cout_rows, KC, total);
  else if (dtype == CP_F16)
    CP_LAUNCH(pack_weight_kernel<F16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       R, S, cin_phys, transposed, phase, row_map, cout_rows, KC, total);
  else
    CP_LAUNCH(pack_weight_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       R, S, cin_phys, transposed, phase, row_map, cout_rows, KC, total);
  return cp_check_launch();
}

// ---- NCHW fp32 -> channels-last (zero-padded channels).  One thread per PIXEL: the C plane reads are coalesced
// across lanes (consecutive pixels of one plane) and each lane writes its pixel's Cphys channels as 16-byte vectors.
template <typename Tag>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, void* __restrict__ out, int C, int HW, int Cphys,
                                    size_t total) {
  constexpr int E = Tag::E;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*HW
  if (pix >= total) return;
  const size_t b = pix / HW, hw = pix - b * HW;
  for (int c0 = 0; c0 < Cphys; c0 += E) {
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = (c0 + e) < C ? in[(b * C + c0 + e) * HW + hw] : 0.f;
    ((u32x4*)out)[(pix * Cphys + c0) / E] = Vec16<Tag>::pack(v);
  }
}

extern "C" int cp_nchw_to_nhwc(cp_stream_t stream, int dtype, const float* in, void* out, int B, int C, int H, int W,
                               int Cphys) {
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cphys < C) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (Cphys % cp_chan_align(dtype) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const size_t total = (size_t)B * H * W;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(nchw_to_nhwc_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, C, H * W, Cphys, total);
  else if (dtype == CP_BF16)
    CP_LAUNCH(nchw_to_nhwc_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, C, H * W, Cphys, total);
  else return CP_ERR_INVALID;
  return cp_check_launch();
}

// ---- channels-last slice -> NCHW fp32 (32x32 LDS transpose tile: coalesced on both sides)
template <typename Tag>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ in, float* __restrict__ out, int C, int HW, int cs, int coff) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int p = p0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (p < HW && c < C) ? load_elem<Tag>(in, ((size_t)b * HW + p) * cs + coff + c) : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, p = p0 + threadIdx.x;
    if (p < HW && c < C) out[((size_t)b * C + c) * HW + p] = tile[threadIdx.x][j];
  }
}

extern "C" int cp_nhwc_to_nchw_f32(cp_stream_t stream, int dtype, const void* in, float* out, int B, int C, int H,
                                   int W, int in_cstride, int in_coff) {
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || in_coff + C > in_cstride) return CP_ERR_INVALID;
  dim3 grid((H * W + 31) / 32, (C + 31) / 32, B), block(32, 8);
  if (dtype == CP_F32)
    CP_LAUNCH(nhwc_to_nchw_kernel<F32Tag>, grid, block, 0, (hipStream_t)stream, in, out, C, H * W, in_cstride, in_coff);
  else if (dtype == CP_BF16)
    CP_LAUNCH(nhwc_to_nchw_kernel<BF16Tag>, grid, block, 0, (hipStream_t)stream, in, out, C, H * W, in_cstride, in_coff);
  else return CP_ERR_INVALID;
  return cp_check_launch();
}

// ---- input side on the device: uint8 HWC crop -> ToTensor (/255) -> Normalize(mean, std)
// (reference bop_dataset_pytorch.py:385-391: transforms.ToTensor() + transforms.Normalize((0.485,0.456,0.406),
// (0.229,0.224,0.225))) -> channels-last `dtype`, zero-padded channels.  IEEE divisions, same op order as
// torchvision (x/255, then (x-mean)/std), so the fp32 result is bit-identical to the host pipeline.
template <typename Tag>
__global__ void u8_to_nhwc_norm_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, int Cphys, float m0, float m1,
                                       float m2, float s0, float s1, float s2, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*W*Cphys
  if (i >= total) return;
  const int c = (int)(i % Cphys);
  const size_t pix = i / Cphys;
  float v = 0.f;
  if (c < 3) {
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)in[pix * 3 + c], 255.0f), mean), sd);
  }
  store_elem<Tag>(out, i, v);
}

extern "C" int cp_u8hwc_to_nhwc_norm(cp_stream_t stream, int dtype, const uint8_t* in, void* out, int B, int H, int W,
                                     int Cphys, const float* mean3, const float* std3) {
  if (!in || !out || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0 || Cphys < 3) return CP_ERR_INVALID;
  const size_t total = (size_t)B * H * W * Cphys;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(u8_to_nhwc_norm_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, Cphys, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], total);
  else if (dtype == CP_BF16)
    CP_LAUNCH(u8_to_nhwc_norm_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, Cphys, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], total);
  else return CP_ERR_INVALID;
  return cp_check_launch();
}
