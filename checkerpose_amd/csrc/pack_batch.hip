// Batched weight preparation (training): ONE launch walks a device-resident table of pack / view / copy items instead of
// ~1100 single launches per step (each ~4 us of an otherwise idle GPU).  Block b finds its item by binary search in the
// exclusive prefix sum of the items' block counts; the per-element formulas are the ones of the single-launch kernels
// (pack_layout.hip, conv3x3_halo.hip, gemm_lds.hip, train_dense.hip) -- tests/test_gpu_train_dense.py checks the two
// paths produce bit-identical images.
#include "common.h"

template <typename Tag>
__device__ __forceinline__ void pack_item_elem(const CpPackItem& it, size_t i) {
  constexpr int E = Tag::E;
  const float* __restrict__ w = (const float*)it.src;
  const int32_t* a = it.a;
  switch (it.kind) {
    case CP_PACK_GENERIC: {
      const int Cout = a[0], Cin = a[1], R = a[2], S = a[3], cin_phys = a[4], transposed = a[5], phase = a[6], cout_rows = a[7], KC = a[8];
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
      const int src_row = n < cout_rows ? (it.row_map ? it.row_map[n] : n) : -1;
      if (src_row >= 0 && src_row < Cout && c < Cin && r < R) {
        if (!transposed) v = w[(((size_t)src_row * Cin + c) * R + r) * S + s];
        else {
          const int pa = phase >> 1, pb = phase & 1;
          const int kh = pa ? (r == 0 ? 2 : 0) : 1;
          const int kw = pb ? (s == 0 ? 2 : 0) : 1;
          v = w[(((size_t)c * Cout + src_row) * 3 + kh) * 3 + kw];
        }
      }
      store_elem<Tag>(it.dst, i, v);
      break;
    }
    case CP_PACK_HALO_S: {
      const int Cout = a[0], Cin = a[1], NT = a[2], perm = a[3];
      const int e = (int)(i % E);
      const int lane = (int)((i / E) % 64);
      size_t blk = i / (E * 64);
      const int nt = (int)(blk % NT); blk /= NT;
      const int tap = (int)(blk % 9);
      const int c = (int)(blk / 9);
      const int row = lane & 15, kq = lane >> 4;
      const int n = perm ? (row >> 2) * 4 * NT + nt * 4 + (row & 3) : nt * 16 + row;
      const int cin = c * (4 * E) + kq * E + e;
      float v = 0.f;
      if (n < Cout && cin < Cin) v = w[((size_t)n * Cin + cin) * 9 + tap];
      store_elem<Tag>(it.dst, i, v);
      break;
    }
    case CP_PACK_HALO:
    case CP_PACK_HALO4: {
      const int Cout = a[0], Cin = a[1], nchunk = a[2];
      const int NTT = it.kind == CP_PACK_HALO ? 2 : 4;
      const int e = (int)(i % E);
      const int lane = (int)((i / E) % 64);
      size_t blk = i / (E * 64);
      const int nt = (int)(blk % NTT); blk /= NTT;
      const int tap = (int)(blk % 9); blk /= 9;
      const int c = (int)(blk % nchunk);
      const int g = (int)(blk / nchunk);
      const int row = lane & 15, kq = lane >> 4;
      const int n = it.kind == CP_PACK_HALO ? g * 32 + (row >> 2) * 8 + nt * 4 + (row & 3) : g * 64 + (row >> 2) * 16 + nt * 4 + (row & 3);
      const int cin = c * (4 * E) + kq * E + e;
      float v = 0.f;
      if (n < Cout && cin < Cin) v = w[((size_t)n * Cin + cin) * 9 + tap];
      store_elem<Tag>(it.dst, i, v);
      break;
    }
    case CP_PACK_GEMM: {
      const int Cout = a[0], Cin = a[1], nchunk = a[2];
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
      store_elem<Tag>(it.dst, i, v);
      break;
    }
    case CP_PACK_DGRAD_VIEW: {
      const int Cout = a[0], Cin = a[1], R = a[2], S = a[3];
      const int s = (int)(i % S);
      size_t t = i / S;
      const int r = (int)(t % R); t /= R;
      const int co = (int)(t % Cout);
      const int ci = (int)(t / Cout);
      ((float*)it.dst)[i] = w[(((size_t)co * Cin + ci) * R + (R - 1 - r)) * S + (S - 1 - s)];
      break;
    }
    case CP_PACK_EDGE_VIEW: {
      const int Co = a[0], Ci = a[1], mode = a[2];
      float v;
      if (mode == 0) {
        const int c = (int)(i % Ci);
        const int row = (int)(i / Ci);
        const int co = row < Co ? row : row - Co;
        const float w1 = w[(size_t)co * 2 * Ci + c], w2 = w[(size_t)co * 2 * Ci + Ci + c];
        v = row < Co ? w1 : w2 - w1;
      } else {
        const int col = (int)(i % (2 * Co));
        const int c = (int)(i / (2 * Co));
        v = col < Co ? w[(size_t)col * 2 * Ci + c] : w[(size_t)(col - Co) * 2 * Ci + Ci + c];
      }
      ((float*)it.dst)[i] = v;
      break;
    }
    default:   // CP_PACK_COPY_F32
      ((float*)it.dst)[i] = w[i];
  }
}

template <typename Tag>
__global__ __launch_bounds__(256) void pack_batch_kernel(const CpPackItem* __restrict__ items, const uint32_t* __restrict__ prefix, int n) {
  // largest d with prefix[d] <= blockIdx.x  (prefix has n + 1 entries, prefix[n] = total blocks)
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const CpPackItem it = items[lo];
  const size_t i = (size_t)(blockIdx.x - prefix[lo]) * 256 + threadIdx.x;
  if (i < it.total) pack_item_elem<Tag>(it, i);
}

extern "C" int cp_pack_batch(cp_stream_t stream, int dtype, const CpPackItem* items_dev, const uint32_t* block_prefix_dev, int n_items,
                             uint32_t total_blocks) {
  if (n_items < 0) return CP_ERR_INVALID;
  if (n_items == 0 || total_blocks == 0) return CP_OK;
  if (!items_dev || !block_prefix_dev) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  if (dtype == CP_F32) CP_LAUNCH(pack_batch_kernel<F32Tag>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, block_prefix_dev, n_items);
  else CP_LAUNCH(pack_batch_kernel<BF16Tag>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, block_prefix_dev, n_items);
  return cp_check_launch();
}

static void item_clear(CpPackItem* it) {
  it->kind = 0; it->src = nullptr; it->dst = nullptr; it->row_map = nullptr; it->total = 0;
  for (int k = 0; k < 9; ++k) it->a[k] = 0;
}

extern "C" int cp_pack_item_dgrad_view(const float* w, int Cout, int Cin, int R, int S, float* wt, CpPackItem* it) {
  if (!w || !wt || !it || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return CP_ERR_INVALID;
  item_clear(it);
  it->kind = CP_PACK_DGRAD_VIEW; it->src = w; it->dst = wt; it->a[0] = Cout; it->a[1] = Cin; it->a[2] = R; it->a[3] = S;
  it->total = (uint64_t)Cout * Cin * R * S;
  return CP_OK;
}

extern "C" int cp_pack_item_edge_view(const float* w, int Cout, int Cin, int mode, float* out, CpPackItem* it) {
  if (!w || !out || !it || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return CP_ERR_INVALID;
  item_clear(it);
  it->kind = CP_PACK_EDGE_VIEW; it->src = w; it->dst = out; it->a[0] = Cout; it->a[1] = Cin; it->a[2] = mode;
  it->total = (uint64_t)2 * Cout * Cin;
  return CP_OK;
}

extern "C" int cp_pack_item_copy_f32(const float* src, float* dst, int count, CpPackItem* it) {
  if (!src || !dst || !it || count <= 0) return CP_ERR_INVALID;
  item_clear(it);
  it->kind = CP_PACK_COPY_F32; it->src = src; it->dst = dst; it->total = (uint64_t)count;
  return CP_OK;
}

extern "C" int cp_pack_item_conv(int dtype, const float* w, int Cout, int Cin, int R, int S, int cin_phys, int transposed, int phase,
                                 const int32_t* row_map, int cout_rows, void* packed, CpPackItem* it) {
  if (!w || !packed || !it || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || cout_rows <= 0) return CP_ERR_INVALID;
  if (dtype != CP_F32 && dtype != CP_BF16) return CP_ERR_INVALID;
  const int E = cp_chan_align(dtype);
  if (cin_phys < Cin || cin_phys % E) return CP_ERR_ALIGN;
  if (transposed && (phase < 0 || phase > 3 || R != 1 + (phase >> 1) || S != 1 + (phase & 1))) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int KCH = 4 * E;
  item_clear(it);
  it->kind = CP_PACK_GENERIC; it->src = w; it->dst = packed; it->row_map = row_map;
  it->a[0] = Cout; it->a[1] = Cin; it->a[2] = R; it->a[3] = S; it->a[4] = cin_phys; it->a[5] = transposed; it->a[6] = phase;
  it->a[7] = cout_rows; it->a[8] = (R * S * cin_phys + KCH - 1) / KCH;
  it->total = cp_packed_weight_bytes(dtype, cout_rows, cin_phys, R, S) / cp_elem_size(dtype);
  return CP_OK;
}
