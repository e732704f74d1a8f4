// Per-keypoint graph kernels: EdgeConv neighbour gather + max (factored StaticGraph_module), local image
// feature gather (Index2Feat_module + RoI mask) and the binary-code -> pixel-index decode.
// All HBM/L2-bound byte movers: no MFMA here (the dense half of EdgeConv is a cp_conv2d_igemm call).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// EdgeConv aggregation.  Reference StaticGraph_module.forward (init.py:64-68 == pipeline.py:55-59) builds the
// (B,2C,N,K) edge tensor cat[x_j - x_i, x_i] (get_graph_feature, init.py:36-49), applies Conv2d(2C->C',1x1),
// BatchNorm2d, LeakyReLU and takes max over K.  With W = [W1 | W2] and BN scale s, shift t:
//     s*(W1 (x_j - x_i) + W2 x_i) + t = P'_j + Q'_i,   P' = s*W1 x,  Q' = s*(W2 - W1) x + t
// and because LeakyReLU is increasing, max_k leaky(P'_j(k) + Q'_i) = leaky(max_k P'_j(k) + Q'_i)  (s is folded
// into P', so negative BN gammas need no min/max switch).  pq rows are [P'(C) | Q'(C)].
//
// Mapping: channels across lanes -- a thread owns one 16-byte channel group of one keypoint, so the K-way max is
// a private register reduction (no cross-lane traffic) and every neighbour-row read is a coalesced 16 B/lane
// segment (a full 1 KiB row per wave at C=256 fp32).  The keypoints' neighbour lists are staged through LDS with
// one coalesced read per block.  All blocks of one crop carry the same (blockIdx % 8) label, i.e. run on one
// XCD, so the crop's P' rows (N*C*4 B = 512 KiB at N=512,C=256) are served from that XCD's L2.
template <typename Tag, int TPK>   // TPK threads per keypoint = C / E
__global__ __launch_bounds__(256) void edgeconv_gather_max_kernel(
    const void* __restrict__ pq, const int32_t* __restrict__ idx, const int32_t* __restrict__ graph_ids,
    void* __restrict__ out, int B, int N, int K, int chunks, int out_cs, int out_coff, float slope) {
  constexpr int E = Tag::E;
  if constexpr (Tag::dtype == CP_F16) cp_f16_saturate_on();
  constexpr int KPB = 256 / TPK;             // keypoints per block
  constexpr int C = TPK * E;
  extern __shared__ __attribute__((aligned(16))) int32_t s_idx[];   // KPB * K

  // XCD-aware decode: label = blockIdx % 8 ; crops b == label (mod 8) live on that label
  const int label = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int b = label + 8 * (j / chunks);
  const int chunk = j % chunks;
  if (b >= B) return;                        // whole block exits together (before any barrier)
  const int g = graph_ids ? graph_ids[b] : 0;
  const int kp0 = chunk * KPB;
  const int nkp = min(KPB, N - kp0);

  const int32_t* gidx = idx + ((size_t)g * N + kp0) * K;
  for (int t = threadIdx.x; t < nkp * K; t += 256) s_idx[t] = gidx[t];
  __syncthreads();

  const int kp_l = threadIdx.x / TPK, cg = threadIdx.x % TPK;
  if (kp_l >= nkp) return;
  const int i = kp0 + kp_l;
  const u32x4* rows = (const u32x4*)pq + (size_t)b * N * (2 * TPK);   // row stride = 2C elements = 2*TPK vectors
  float m[E], f[E];
#pragma unroll
  for (int e = 0; e < E; ++e) m[e] = -INFINITY;
  const int32_t* my = s_idx + kp_l * K;
  int k = 0;
  for (; k + 4 <= K; k += 4) {               // 4 independent 16-byte gathers in flight per lane
    const u32x4 v0 = rows[(size_t)my[k + 0] * (2 * TPK) + cg];
    const u32x4 v1 = rows[(size_t)my[k + 1] * (2 * TPK) + cg];
    const u32x4 v2 = rows[(size_t)my[k + 2] * (2 * TPK) + cg];
    const u32x4 v3 = rows[(size_t)my[k + 3] * (2 * TPK) + cg];
    Vec16<Tag>::unpack(v0, f);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], f[e]);
    Vec16<Tag>::unpack(v1, f);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], f[e]);
    Vec16<Tag>::unpack(v2, f);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], f[e]);
    Vec16<Tag>::unpack(v3, f);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], f[e]);
  }
  for (; k < K; ++k) {
    Vec16<Tag>::unpack(rows[(size_t)my[k] * (2 * TPK) + cg], f);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], f[e]);
  }
  Vec16<Tag>::unpack(rows[(size_t)i * (2 * TPK) + TPK + cg], f);          // Q'_i
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const float y = m[e] + f[e];
    m[e] = y > 0.f ? y : y * slope;
  }
  const size_t oe = ((size_t)b * N + i) * out_cs + out_coff + (size_t)cg * E;
  ((u32x4*)out)[oe / E] = Vec16<Tag>::pack(m);
  (void)C;
}

template <typename Tag, int TPK>
static int launch_edge(hipStream_t st, const void* pq, const int32_t* idx, const int32_t* gids, void* out, int B, int N,
                       int K, int out_cs, int out_coff, float slope) {
  constexpr int KPB = 256 / TPK;
  const int chunks = (N + KPB - 1) / KPB;
  const int grid = 8 * ((B + 7) / 8) * chunks;
  cp_mark_kernel("edgeconv_gather_max_kernel<%s, %d>", Tag::dtype == CP_BF16 ? "BF16Tag" : (Tag::dtype == CP_F16 ? "F16Tag" : "F32Tag"), TPK);
  hipLaunchKernelGGL((edgeconv_gather_max_kernel<Tag, TPK>), dim3(grid), dim3(256), KPB * K * sizeof(int32_t), st, pq, idx,
                     gids, out, B, N, K, chunks, out_cs, out_coff, slope);
  return cp_check_launch();
}

extern "C" int cp_edgeconv_gather_max(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx,
                                      const int32_t* graph_ids, void* out, int B, int N, int K, int C, int G,
                                      int out_cstride, int out_coff, float slope) {
  if (!pq || !idx || !out || B <= 0 || N <= 0 || K <= 0 || K > 64 || C <= 0 || G <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (C % E || out_cstride % E || out_coff % E || out_coff + C > out_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(pq) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int tpk = C / E;
#define CP_EDGE(TAG, T) case T: return launch_edge<TAG, T>(st, pq, idx, graph_ids, out, B, N, K, out_cstride, out_coff, slope);
  if (dtype == CP_F32) {
    switch (tpk) { CP_EDGE(F32Tag, 8) CP_EDGE(F32Tag, 16) CP_EDGE(F32Tag, 32) CP_EDGE(F32Tag, 64) CP_EDGE(F32Tag, 128) default: return CP_ERR_INVALID; }
  } else if (dtype == CP_F16) {
    switch (tpk) { CP_EDGE(F16Tag, 4) CP_EDGE(F16Tag, 8) CP_EDGE(F16Tag, 16) CP_EDGE(F16Tag, 32) CP_EDGE(F16Tag, 64) default: return CP_ERR_INVALID; }
  } else {
    switch (tpk) { CP_EDGE(BF16Tag, 4) CP_EDGE(BF16Tag, 8) CP_EDGE(BF16Tag, 16) CP_EDGE(BF16Tag, 32) CP_EDGE(BF16Tag, 64) default: return CP_ERR_INVALID; }
  }
#undef CP_EDGE
}

// ------------------------------------------------------------------------------------------------
// Index2Feat_module.forward gather (pipeline.py:156-163) + RoI mask (pipeline.py:280).  patches is the
// channels-last output of patch_generator, (B, Hp, Wp, E) with Hp = H+1 (kernel k, padding k-1).
// thread = one 16-byte channel group of one (keypoint, tap); taps ordered sf1..sf4 as the reference concatenates.
template <typename Tag>
__global__ void index2feat_kernel(const void* __restrict__ patches, const int32_t* __restrict__ x_id,
                                  const int32_t* __restrict__ y_id, const float* __restrict__ mask, void* __restrict__ out,
                                  int N, int Hp, int Wp, int EG, int k, int out_cs, int out_coff, size_t total) {
  constexpr int E = Tag::E;
  if constexpr (Tag::dtype == CP_F16) cp_f16_saturate_on();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*N*4*EG
  if (i >= total) return;
  const int g = (int)(i % EG);
  size_t t = i / EG;
  const int tap = (int)(t & 3);
  const size_t kp = t >> 2;                  // b*N + n
  const size_t b = kp / N;
  const int y = 2 * y_id[kp] + ((tap & 1) ? k : 0);        // sf2, sf4: row + k
  const int x = 2 * x_id[kp] + ((tap & 2) ? k : 0);        // sf3, sf4: col + k
  float f[E];
  const bool ok = (unsigned)y < (unsigned)Hp && (unsigned)x < (unsigned)Wp;   // ids are always in range; defensive
  if (ok) Vec16<Tag>::unpack(((const u32x4*)patches)[((b * Hp + y) * Wp + x) * EG + g], f);
  const float mk = ok ? mask[kp] : 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) f[e] = ok ? f[e] * mk : 0.f;
  const size_t oe = kp * out_cs + out_coff + (size_t)(tap * EG + g) * E;
  ((u32x4*)out)[oe / E] = Vec16<Tag>::pack(f);
}

extern "C" int cp_index2feat_gather(cp_stream_t stream, int dtype, const void* patches, const int32_t* x_id,
                                    const int32_t* y_id, const float* mask, void* out, int B, int N, int Hp, int Wp,
                                    int E_ch, int k, int out_cstride, int out_coff) {
  if (!patches || !x_id || !y_id || !mask || !out || B <= 0 || N <= 0 || Hp <= 0 || Wp <= 0 || E_ch <= 0 || k <= 0)
    return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (E_ch % E || out_cstride % E || out_coff % E || out_coff + 4 * E_ch > out_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(patches) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const int EG = E_ch / E;
  const size_t total = (size_t)B * N * 4 * EG;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CP_F32)
    CP_LAUNCH(index2feat_kernel<F32Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, patches, x_id, y_id, mask,
                       out, N, Hp, Wp, EG, k, out_cstride, out_coff, total);
  else if (dtype == CP_F16)
    CP_LAUNCH(index2feat_kernel<F16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, patches, x_id, y_id, mask,
                       out, N, Hp, Wp, EG, k, out_cstride, out_coff, total);
  else
    CP_LAUNCH(index2feat_kernel<BF16Tag>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, patches, x_id, y_id, mask,
                       out, N, Hp, Wp, EG, k, out_cstride, out_coff, total);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Bit decode: from_mask_prob_to_mask (pipeline.py:120-127), from_code_prob_to_id (:84-92, MSB first :72-82),
// from_bit_prob_to_id (:103-110) and the per-stage update id = 2*id + bit (:380-381).
// The reference thresholds `torch.sigmoid(z) > 0.5` in fp32, and fp32 sigmoid rounds to exactly 0.5 for every
// 0 <= z <= CP_SIGMOID_HALF_Z0 (1.5 * 2^-24, bits 0x33C00000): measured on torch 2.10 CPU (scalar, AVX2 and AVX-512 paths)
// through the reference's own from_mask_prob_to_mask (tests/golden/make_golden_r2.py sigmoid, fixture sigmoid_threshold.npz).
// Index work is bit exact, so the decision is z > Z0, not z > 0.
__device__ __forceinline__ bool cp_sigmoid_gt_half(float z) { return z > __uint_as_float(CP_SIGMOID_HALF_Z0_BITS); }
__global__ void bits_decode_kernel(const float* __restrict__ bits, int stage, float* __restrict__ mask,
                                   int32_t* __restrict__ x_id, int32_t* __restrict__ y_id, int64_t* __restrict__ x64,
                                   int64_t* __restrict__ y64, int N, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*N
  if (i >= total) return;
  const size_t b = i / N;
  const int n = (int)(i - b * N);
  const float* z = bits + b * 13 * N + n;
  int x, y;
  if (stage < 0) {
    mask[i] = cp_sigmoid_gt_half(z[0]) ? 1.f : 0.f;
    x = (cp_sigmoid_gt_half(z[1 * (size_t)N]) << 2) | (cp_sigmoid_gt_half(z[2 * (size_t)N]) << 1) | cp_sigmoid_gt_half(z[3 * (size_t)N]);
    y = (cp_sigmoid_gt_half(z[7 * (size_t)N]) << 2) | (cp_sigmoid_gt_half(z[8 * (size_t)N]) << 1) | cp_sigmoid_gt_half(z[9 * (size_t)N]);
  } else {
    x = 2 * x_id[i] + cp_sigmoid_gt_half(z[(size_t)(4 + stage) * N]);
    y = 2 * y_id[i] + cp_sigmoid_gt_half(z[(size_t)(10 + stage) * N]);
  }
  x_id[i] = x; y_id[i] = y;
  if (x64) x64[i] = x;
  if (y64) y64[i] = y;
}

extern "C" int cp_bits_decode(cp_stream_t stream, const float* bits, int stage, float* mask, int32_t* x_id,
                              int32_t* y_id, int64_t* x_id64, int64_t* y_id64, int B, int N) {
  if (!bits || !x_id || !y_id || B <= 0 || N <= 0 || stage > 2) return CP_ERR_INVALID;
  if (stage < 0 && !mask) return CP_ERR_INVALID;
  const size_t total = (size_t)B * N;
  CP_LAUNCH(bits_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bits, stage,
                     mask, x_id, y_id, x_id64, y_id64, N, total);
  return cp_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Post-forward decode on the device: what reference test.py:294-329 +
// test_network_with_test_data.py:from_id_to_pose :50-66 do on the host with six .cpu().numpy() round trips per image
// (pinned by tests/golden/n2_from_id_to_pose.npz: the lists the reference's own from_id_to_pose hands to its solver):
//   p2d[b,n]      = roi_xy_ori[b, :, y_id, x_id]                       (2-D coordinate of the predicted pixel)
//   valid[b,n,0]  = sigmoid(roi) > 0.5                                   ("all" correspondences, check_seg=False)
//   valid[b,n,1]  = valid0 && sigmoid(seg[b,1,y,x]) > 0.5                (kept by the FULL mask,  check_seg=True)
//   valid[b,n,2]  = valid0 && sigmoid(seg[b,0,y,x]) > 0.5                (kept by the VISIBLE mask)
//   discard_bd_pixel d > 0 (:60-63): every variant additionally needs d <= x < W-d and d <= y < H-d
//   count[b,k]    = number of valid correspondences per variant          (PnP needs >= 4 / 6, :69-99)
// Only B*N*(2 floats + 3 bytes) leave the GPU instead of logits, ids and two 64x64 masks.
__global__ void correspondences_kernel(const float* __restrict__ bits, const float* __restrict__ seg,
                                       const int64_t* __restrict__ x_id, const int64_t* __restrict__ y_id,
                                       const float* __restrict__ roi_xy, const int32_t* __restrict__ bbox, float* __restrict__ p2d,
                                       uint8_t* __restrict__ valid, int32_t* __restrict__ count, int N, int Hh, int Ww, int bd, size_t total) {
  const int HW = Hh * Ww;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*N
  if (i >= total) return;
  const size_t b = i / N;
  const int n = (int)(i - b * N);
  const int x = (int)x_id[i], y = (int)y_id[i];
  const size_t pix = (size_t)y * Ww + x;
  if (roi_xy) {
    p2d[2 * i + 0] = roi_xy[(b * 2 + 0) * HW + pix];
    p2d[2 * i + 1] = roi_xy[(b * 2 + 1) * HW + pix];
  } else {          // the loader's grid, built here: mapping_pixel_position_to_original_position_2d (bop_dataset_pytorch.py:223-235) in fp64,
    const int32_t* bb = bbox + 4 * b;                      // cast to fp32 as `torch.from_numpy(roi_xy_ori).type(torch.float)` does (:380)
    p2d[2 * i + 0] = (float)((double)bb[2] / (double)Ww * (double)x + (double)bb[0]);
    p2d[2 * i + 1] = (float)((double)bb[3] / (double)Hh * (double)y + (double)bb[1]);
  }
  const bool inner = (x >= bd) & (x < Ww - bd) & (y >= bd) & (y < Hh - bd);     // bd_mask[d:H-d, d:W-d] = 1  (:61-62)
  const bool v0 = inner && cp_sigmoid_gt_half(bits[b * 13 * (size_t)N + n]);
  const bool v1 = v0 && cp_sigmoid_gt_half(seg[(b * 2 + 1) * HW + pix]);
  const bool v2 = v0 && cp_sigmoid_gt_half(seg[(b * 2 + 0) * HW + pix]);
  valid[3 * i + 0] = v0; valid[3 * i + 1] = v1; valid[3 * i + 2] = v2;
  // per-image counts: integer atomics (hipcc folds a wave's adds to one address into a single atomic)
  if (v0) atomicAdd(&count[b * 3 + 0], 1);
  if (v1) atomicAdd(&count[b * 3 + 1], 1);
  if (v2) atomicAdd(&count[b * 3 + 2], 1);
}

static int correspondences_impl(cp_stream_t stream, const float* bits, const float* seg, const int64_t* x_id, const int64_t* y_id,
                                const float* roi_xy_ori, const int32_t* bbox, float* p2d, uint8_t* valid, int32_t* count, int B, int N,
                                int H, int W, int discard_bd_pixel) {
  if (!bits || !seg || !x_id || !y_id || (!roi_xy_ori && !bbox) || !p2d || !valid || !count || B <= 0 || N <= 0 || H <= 0 || W <= 0 ||
      discard_bd_pixel < 0)
    return CP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(count, 0, (size_t)B * 3 * sizeof(int32_t), st) != hipSuccess) return CP_ERR_HIP;
  const size_t total = (size_t)B * N;
  CP_LAUNCH(correspondences_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bits, seg, x_id, y_id,
                     roi_xy_ori, bbox, p2d, valid, count, N, H, W, discard_bd_pixel, total);
  return cp_check_launch();
}

extern "C" int cp_correspondences(cp_stream_t stream, const float* bits, const float* seg, const int64_t* x_id,
                                  const int64_t* y_id, const float* roi_xy_ori, float* p2d, uint8_t* valid, int32_t* count,
                                  int B, int N, int H, int W, int discard_bd_pixel) {
  if (!roi_xy_ori) return CP_ERR_INVALID;
  return correspondences_impl(stream, bits, seg, x_id, y_id, roi_xy_ori, nullptr, p2d, valid, count, B, N, H, W, discard_bd_pixel);
}

extern "C" int cp_correspondences_bbox(cp_stream_t stream, const float* bits, const float* seg, const int64_t* x_id,
                                       const int64_t* y_id, const int32_t* final_bbox, float* p2d, uint8_t* valid, int32_t* count,
                                       int B, int N, int H, int W, int discard_bd_pixel) {
  if (!final_bbox) return CP_ERR_INVALID;
  return correspondences_impl(stream, bits, seg, x_id, y_id, nullptr, final_bbox, p2d, valid, count, B, N, H, W, discard_bd_pixel);
}
