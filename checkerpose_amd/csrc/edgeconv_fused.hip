// EdgeConv (StaticGraph_module: reference init.py:54-68 == pipeline.py:45-59, LM twin pipeline_lm.py:55-57) as ONE launch per
// layer for N = 512 keypoints: per-node GEMM to [P' | Q'] on MFMA + neighbour gather-max out of LDS.
//
// The factored form (graph_ops.hip) ran as a row GEMM that wrote the (B, N, 2C') table to HBM (134 MB per layer at batch 256)
// and a gather kernel that read it back through L2 (K = 20 neighbour rows per keypoint): 94 + 68 us per layer, both far from
// any roof.  A crop's P' table for 64 output channels is 512 x 64 x 2 B = 64 KB: it fits in LDS.  One 8-wave workgroup per
// crop, wave w owns keypoints [64 w, 64 w + 64) = 4 MFMA fragments:
//   * the crop's x rows (512 x Cin bf16) live in REGISTERS for the whole layer (the B operand of every MFMA: 128 VGPRs at
//     Cin = 256), loaded once;
//   * per 64-channel slice: the P and the Q half of the weights (32 KB each) stream L2 -> LDS by LDS-DMA, double-buffered
//     under the other half's MFMAs; P' = s * (W1 x) is written to the LDS table as packed f16 pairs
//     (so the K-way max is v_pk_maximum3_f16: two neighbours x two channels per instruction; common.h), Q' = s * ((W2 - W1) x) + t stays in the fp32
//     accumulators;
//   * gather: a lane owns 16 channels of a keypoint: 2 ds_read_b128 per neighbour row.  The table is [8-channel plane][row][16 B]:
//     the 16 lanes of a ds_read_b128 group are 16 different keypoints reading 16 different rows, bank slot = row mod 16, and the
//     host hands in every keypoint's list ORDERED so that the rows of a step have different residues (graph_sched.py: the max
//     does not care; bank-conflict share of the kernel's LDS cycles 54 % -> 27 %, LDS cycles per launch 18.8 M -> 11.8 M).  Index lists staged once per layer as int16;
//     out = leaky(max_k P'_j(k) + Q'_i) leaves as 32 B per lane.
// Two barriers per slice.  Numerics: P' rounded to f16 (3 more mantissa bits than the bf16 of the two-launch form), Q' stays fp32.
#include "common.h"

namespace {

constexpr int EF_N = 512, EF_KMAX = 20;
constexpr int EF_PLANE = EF_N * 16;                         // P' table: one plane per 8 channels, [plane][row][16 B] -> bank slot = row mod 16
constexpr int EF_TABLE = 8 * EF_PLANE;                      // 65 536
constexpr int EF_WBUF = 32 * 1024;                          // one (slice, half) of weights at Cin = 256
constexpr int EF_IDX = EF_N * EF_KMAX * 2;                  // 20 480
constexpr int EF_AFF = 2 * 512 * 4;                         // scale | shift of the 2 C' <= 512 GEMM rows
constexpr int EF_LDS = EF_TABLE + 2 * EF_WBUF + EF_IDX + EF_AFF;     // 155 648

struct EdgeFusedParams {
  const void* x; const void* w; const float* scale; const float* shift;
  const int32_t* idx; const int32_t* gids; void* out;
  int in_cs, in_coff, out_cs, out_coff, B, K, Cout;
  float slope;
};


template <int CIN, bool H = false>                            // H: x rows, weights and output rows in IEEE half (CP_F16; common.h cp_mma16)
__global__ __launch_bounds__(512) void edgeconv_fused_kernel(const EdgeFusedParams p) {
  constexpr int KC = CIN / 32;                              // 32-deep K chunks
  constexpr int HALF = KC * 4 * 1024;                       // bytes of one (slice, half) weight image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sP = smem;
  unsigned char* const sW = smem + EF_TABLE;
  int16_t* const sIdx = (int16_t*)(smem + EF_TABLE + 2 * EF_WBUF);
  float* const sScale = (float*)(smem + EF_TABLE + 2 * EF_WBUF + EF_IDX);      // [2 Cout] then shift [2 Cout] at + 512
  float* const sShift = sScale + 512;

  cp_f16_saturate_on();                                       // the f16 key pack (and the half output rows) saturate at +-65504: common.h
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  const int nslice = p.Cout / 64;

  const __amdgpu_buffer_rsrc_t wrs = cp_dma_rsrc(p.w);          // buffer-form DMA: keeps the gather's LDS waits counted (common.h)
  auto w_issue = [&](int u) {                               // u = 2 slice + half -> buffer u & 1
    constexpr int PIECES = HALF / 16;
    int ln = lane;                                          // opaque per call: hoisted out of the slice loop, the lane's piece offsets were
    asm volatile("" : "+v"(ln));                            // spilled and their scratch reloads (VMEM) put a vmcnt(0) in front of every DMA
#pragma unroll
    for (int k = 0; k < (PIECES + 511) / 512; ++k) {
      const int i0 = wave * 64 + 512 * k;
      if (i0 < PIECES)
        cp_lds_dma16(wrs, (uint32_t)(((size_t)u * PIECES + i0 + ln) * 16), sW + (u & 1) * EF_WBUF + i0 * 16);
    }
  };
  w_issue(0);

  // ---- neighbour lists of this crop's graph -> int16 in LDS
  {
    const int g = p.gids ? p.gids[b] : 0;
    const int32_t* gi = p.idx + (size_t)g * EF_N * p.K;
    for (int i = tid; i < EF_N * p.K; i += 512) sIdx[(i / p.K) * EF_KMAX + (i % p.K)] = (int16_t)gi[i];
  }
  for (int i = tid; i < 2 * p.Cout; i += 512) { sScale[i] = p.scale[i]; sShift[i] = p.shift[i]; }
  // ---- this wave's 64 x rows: registers
  u32x4 xa[4][KC];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const size_t row = (size_t)b * EF_N + wave * 64 + f * 16 + x;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      xa[f][kc] = *(const u32x4*)((const uint16_t*)p.x + row * p.in_cs + p.in_coff + kc * 32 + q * 8);
  }
  __syncthreads();

  f32x4 acc[4][4];
  auto gemm_half = [&](int u) {                              // acc = W(slice, half) x rows; weights of u in buffer u & 1
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const wb = sW + (u & 1) * EF_WBUF + lane * 16;
    u32x4 wf[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *(const u32x4*)(wb + nt * 1024);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if (kc + 1 < KC) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[(kc + 1) & 1][nt] = *(const u32x4*)(wb + ((kc + 1) * 4 + nt) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[f][nt] = cp_mma16<H>(wf[kc & 1][nt], xa[f][kc], acc[f][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int s = 0; s < nslice; ++s) {
    // ---- P' = s * (W1 x) -> packed f16 -> LDS table
    w_issue(2 * s + 1);                                       // Q half into the other buffer: its last readers passed the previous barrier
    gemm_half(2 * s);
    {
      // lane (x, q): keypoint 64 wave + 16 f + x, channels 64 s + 16 q + 4 nt + reg
      const int c0 = s * 64 + q * 16;
      float sc[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 s4 = *(const f32x4*)(sScale + c0 + 4 * nt);
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[4 * nt + j] = s4[j];
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        u32x4 lo, hi;
        lo.x = (pack_f16x2_ovfl(acc[f][0][0] * sc[0], acc[f][0][1] * sc[1]));
        lo.y = (pack_f16x2_ovfl(acc[f][0][2] * sc[2], acc[f][0][3] * sc[3]));
        lo.z = (pack_f16x2_ovfl(acc[f][1][0] * sc[4], acc[f][1][1] * sc[5]));
        lo.w = (pack_f16x2_ovfl(acc[f][1][2] * sc[6], acc[f][1][3] * sc[7]));
        hi.x = (pack_f16x2_ovfl(acc[f][2][0] * sc[8], acc[f][2][1] * sc[9]));
        hi.y = (pack_f16x2_ovfl(acc[f][2][2] * sc[10], acc[f][2][3] * sc[11]));
        hi.z = (pack_f16x2_ovfl(acc[f][3][0] * sc[12], acc[f][3][1] * sc[13]));
        hi.w = (pack_f16x2_ovfl(acc[f][3][2] * sc[14], acc[f][3][3] * sc[15]));
        unsigned char* dst = sP + (2 * q) * EF_PLANE + (wave * 64 + f * 16 + x) * 16;       // 16 consecutive rows: 16 slots
        *(u32x4*)dst = lo;
        *(u32x4*)(dst + EF_PLANE) = hi;
      }
    }
    __syncthreads();                                          // table complete; the Q half of the weights has landed
    if (s + 1 < nslice) w_issue(2 * s + 2);                   // next slice's P half: its buffer's readers (above) passed the barrier
    // ---- gather-max over the K neighbours out of the LDS table (BEFORE the Q' GEMM: no accumulators are live, so two row
    // fragments' reads -- 16 ds_read_b128 per lane -- are in flight at once; one fragment at a time left the LDS latency bare).
    // A ds_read_b128 lane group = the 16 keypoints of a fragment, each reading ITS k-th neighbour's row: slot = row mod 16; the
    // host orders every keypoint's list so that the 16 rows of a step have different residues (graph_sched.py).
    uint32_t m[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 8; ++j) m[f][j] = CP_F16X2_NEG_INF;
    {
      const unsigned char* const pq = sP + (2 * q) * EF_PLANE;
      const int16_t* const my = sIdx + (wave * 64 + x) * EF_KMAX;
#pragma unroll
      for (int fp = 0; fp < 4; fp += 2) {
        for (int k = 0; k < p.K; k += 4) {                       // K is a multiple of 4 (20)
          u32x2 i4[2];
          u32x4 a0[2], a1[2], c0_[2], c1[2], d0[2], d1[2], e0[2], e1[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) i4[h] = *(const u32x2*)(my + (fp + h) * 16 * EF_KMAX + k);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned char* b0 = pq + (i4[h].x & 0xffffu) * 16;
            const unsigned char* b1 = pq + (i4[h].x >> 16) * 16;
            const unsigned char* b2 = pq + (i4[h].y & 0xffffu) * 16;
            const unsigned char* b3 = pq + (i4[h].y >> 16) * 16;
            a0[h] = *(const u32x4*)b0; a1[h] = *(const u32x4*)(b0 + EF_PLANE); c0_[h] = *(const u32x4*)b1; c1[h] = *(const u32x4*)(b1 + EF_PLANE);
            d0[h] = *(const u32x4*)b2; d1[h] = *(const u32x4*)(b2 + EF_PLANE); e0[h] = *(const u32x4*)b3; e1[h] = *(const u32x4*)(b3 + EF_PLANE);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint32_t* mm = m[fp + h];
            pkmax5x4_f16(mm, a0[h], c0_[h], d0[h], e0[h]);
            pkmax5x4_f16(mm + 4, a1[h], c1[h], d1[h], e1[h]);
          }
        }
      }
    }
    // ---- Q' = s * ((W2 - W1) x) + t (fp32, never rounded to bf16), + max, LeakyReLU, store; two row fragments at a time (the
    // maxima of all four are live: 16 accumulator tiles beside them would spill)
    {
      const int c0 = p.Cout + s * 64 + q * 16;
      const unsigned char* const wb = sW + EF_WBUF + lane * 16;          // u = 2 s + 1 -> buffer 1
#pragma unroll
      for (int fp = 0; fp < 4; fp += 2) {
        f32x4 aq[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) aq[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 wf[2][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *(const u32x4*)(wb + nt * 1024);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          if (kc + 1 < KC) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[(kc + 1) & 1][nt] = *(const u32x4*)(wb + ((kc + 1) * 4 + nt) * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              aq[h][nt] = cp_mma16<H>(wf[kc & 1][nt], xa[fp + h][kc], aq[h][nt]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f = fp + h;
          const int n = wave * 64 + f * 16 + x;
          float v[16];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t w2 = m[f][j];
            const int nt = j >> 1, r = (j & 1) * 2;
            const float q0 = aq[h][nt][r] * sScale[c0 + 4 * nt + r] + sShift[c0 + 4 * nt + r];
            const float q1 = aq[h][nt][r + 1] * sScale[c0 + 4 * nt + r + 1] + sShift[c0 + 4 * nt + r + 1];
            const float y0 = f16_lo(w2) + q0, y1 = f16_hi(w2) + q1;
            v[2 * j] = y0 > 0.f ? y0 : y0 * p.slope;
            v[2 * j + 1] = y1 > 0.f ? y1 : y1 * p.slope;
          }
          uint16_t* dst = (uint16_t*)p.out + ((size_t)b * EF_N + n) * p.out_cs + p.out_coff + s * 64 + q * 16;
          *(u32x4*)dst = cp_pack8<H>(v);
          *(u32x4*)(dst + 8) = cp_pack8<H>(v + 8);
        }
      }
    }
    __syncthreads();                                        // every gather / Q' read of this slice is done: table and Q buffer free
  }
}

// [slice][half P/Q][chunk][tile][lane][8 bf16]; tile row r of tile nt = output channel 64 slice + (r >> 2) * 16 + 4 nt + (r & 3)
// of wpq rows [0, Cout) (P half) / [Cout, 2 Cout) (Q half); element e of lane (r, q): input channel 32 chunk + 8 q + e.
__global__ void pack_edgeconv_fused_kernel(const float* __restrict__ wpq, uint16_t* __restrict__ out, int Cin, int Cout, size_t total, int dtype) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int KC = Cin / 32;
  const int e = (int)(i % 8);
  const int lane = (int)((i / 8) % 64);
  size_t blk = i / 512;
  const int nt = (int)(blk % 4); blk /= 4;
  const int kc = (int)(blk % KC); blk /= KC;
  const int half = (int)(blk % 2);
  const int s = (int)(blk / 2);
  const int r = lane & 15, q = lane >> 4;
  const int c = s * 64 + (r >> 2) * 16 + nt * 4 + (r & 3);
  const int cin = kc * 32 + q * 8 + e;
  out[i] = (uint16_t)f32_to_half_bits(wpq[((size_t)(half * Cout + c)) * Cin + cin], dtype);
}

}  // namespace

extern "C" int cp_edgeconv_fused_supported(int N, int K, int Cin, int Cout) {
  return (N == EF_N && K > 0 && K <= EF_KMAX && K % 4 == 0 && (Cin == 64 || Cin == 256) && Cout >= 64 && Cout % 64 == 0 && Cout <= 256) ? 1 : 0;
}

extern "C" size_t cp_edgeconv_fused_weight_bytes(int Cin, int Cout) { return (size_t)2 * Cout * Cin * 2; }

extern "C" int cp_pack_edgeconv_fused_weight_t(cp_stream_t stream, int dtype, const float* wpq, int Cin, int Cout, void* packed) {
  if (!wpq || !packed || !cp_edgeconv_fused_supported(EF_N, 4, Cin, Cout) || (dtype != CP_BF16 && dtype != CP_F16)) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const size_t total = (size_t)2 * Cout * Cin;
  CP_LAUNCH(pack_edgeconv_fused_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wpq, (uint16_t*)packed, Cin, Cout, total, dtype);
  return cp_check_launch();
}

extern "C" int cp_pack_edgeconv_fused_weight(cp_stream_t stream, const float* wpq, int Cin, int Cout, void* packed) {
  return cp_pack_edgeconv_fused_weight_t(stream, CP_BF16, wpq, Cin, Cout, packed);
}

extern "C" int cp_edgeconv_fused_t(cp_stream_t stream, int dtype, const void* x, int in_cstride, int in_coff, const void* packed_w,
                                   const float* scale, const float* shift, const int32_t* idx, const int32_t* graph_ids, void* out,
                                   int out_cstride, int out_coff, int B, int N, int K, int Cin, int Cout, int G, float slope) {
  if (!x || !packed_w || !scale || !shift || !idx || !out || B <= 0 || G <= 0 || (dtype != CP_BF16 && dtype != CP_F16)) return CP_ERR_INVALID;
  if (!cp_edgeconv_fused_supported(N, K, Cin, Cout)) return CP_ERR_INVALID;
  if (in_cstride % 8 || in_coff % 8 || in_coff + Cin > in_cstride || out_cstride % 8 || out_coff % 8 || out_coff + Cout > out_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(x) || !cp_aligned16(packed_w) || !cp_aligned16(scale) || !cp_aligned16(shift) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)edgeconv_fused_kernel<64>, EF_LDS) &&
                                  cp_set_max_lds((const void*)edgeconv_fused_kernel<256>, EF_LDS) &&
                                  cp_set_max_lds((const void*)edgeconv_fused_kernel<64, true>, EF_LDS) &&
                                  cp_set_max_lds((const void*)edgeconv_fused_kernel<256, true>, EF_LDS));
  EdgeFusedParams p;
  p.x = x; p.w = packed_w; p.scale = scale; p.shift = shift; p.idx = idx; p.gids = graph_ids; p.out = out;
  p.in_cs = in_cstride; p.in_coff = in_coff; p.out_cs = out_cstride; p.out_coff = out_coff; p.B = B; p.K = K; p.Cout = Cout; p.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CP_F16) {
    if (Cin == 64) CP_LAUNCH((edgeconv_fused_kernel<64, true>), dim3((unsigned)B), dim3(512), EF_LDS, st, p);
    else CP_LAUNCH((edgeconv_fused_kernel<256, true>), dim3((unsigned)B), dim3(512), EF_LDS, st, p);
  } else if (Cin == 64) CP_LAUNCH((edgeconv_fused_kernel<64>), dim3((unsigned)B), dim3(512), EF_LDS, st, p);
  else CP_LAUNCH((edgeconv_fused_kernel<256>), dim3((unsigned)B), dim3(512), EF_LDS, st, p);
  return cp_check_launch();
}

extern "C" int cp_edgeconv_fused(cp_stream_t stream, const void* x, int in_cstride, int in_coff, const void* packed_w,
                                 const float* scale, const float* shift, const int32_t* idx, const int32_t* graph_ids, void* out,
                                 int out_cstride, int out_coff, int B, int N, int K, int Cin, int Cout, int G, float slope) {
  return cp_edgeconv_fused_t(stream, CP_BF16, x, in_cstride, in_coff, packed_w, scale, shift, idx, graph_ids, out, out_cstride, out_coff, B, N,
                             K, Cin, Cout, G, slope);
}
