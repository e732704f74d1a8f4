// MLP_QueryNet (reference pipeline.py:168-180: Linear(256 -> 256) + LeakyReLU, Linear(256 -> 64) + LeakyReLU, Linear(64 -> 2)) as ONE
// launch over the (B * N) keypoint rows (bf16 operands, fp32 accumulation).
//
// As three launches the stack moved, per row: 512 B in, 512 B out + in again, 128 B out + in again, 8 B out -- 2.3 KB for a
// 512-byte input, every launch bound by that traffic (the weight-stationary row GEMM of gemm_lds.hip reaches 2.7 TB/s; at config
// #5's 1 M rows per stage the three launches took ~0.65 ms).  Here the rows are read once and only the two logits leave the chip:
// a persistent 8-wave workgroup per CU walks tiles of 64 rows as a two-stage pipeline.
//   waves 0-3 (layer 1, weight-stationary): wave w keeps the 256 x 64 slice W1[64 w .. 64 w + 63] in registers (8 K chunks x 4
//     tiles = 128 VGPRs), multiplies the tile's rows (staged in LDS in full 512-byte row segments, double-buffered, the next
//     tile's loads in flight under the MFMAs) and writes its 64 hidden channels of the tile -- LeakyReLU, bf16 -- into an LDS tile
//     in MFMA operand order ([16-byte channel piece][row][16 B], double-buffered);
//   waves 4-7 (layers 2 + 3): wave v owns rows 16 v .. 16 v + 15 of the PREVIOUS tile: 256 -> 64 from the hidden tile with W2
//     resident in registers (128 VGPRs), LeakyReLU in fp32 (never rounded to bf16), then the 64 -> 2 head as per-lane partial dot
//     products met by two cross-lane adds; 8 bytes per row leave, in the caller's (row, channel) strides (the logit block).
// One barrier per tile.  The layer-1 image is cp_pack_gemm_weight's (gemm_lds.hip), as is the layer-2 image.
#include "common.h"

namespace {

constexpr int MQ_ROWS = 64;
constexpr int MQ_PITCH = MQ_ROWS * 16 + 16;                  // plane [row][16 B]; +16: consecutive planes shift one bank slot
constexpr int MQ_BUF = 32 * MQ_PITCH;                        // 32 pieces of 8 channels = 256 channels: 33 280 B
constexpr int MQ_LDS = 4 * MQ_BUF + 512 + 2560;              // x tile x 2, hidden tile x 2, layer-3 weights, folded-BN vectors of layers 1 / 2

struct MlpQueryParams {
  const void* in; const void* w1; const float* s1; const float* t1;
  const void* w2; const float* s2; const float* t2;
  const float* w3; const float* b3;                          // (2, 64) fp32 row-major, (2,)
  float* out;
  int M, Nrow, in_cs, in_coff, n_rt;
  uint32_t in_bytes;
  float slope1, slope2;
  long long o_base, o_sb, o_sn, o_sc;                        // logit (m = b * Nrow + n, c) at o_base + b * o_sb + n * o_sn + c * o_sc
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

template <bool H>                                              // H: rows, weights and hidden rows in IEEE half (CP_F16; common.h cp_mma16)
__global__ __launch_bounds__(512) void mlp_query_fused_kernel(const MlpQueryParams p) {
  if constexpr (H) cp_f16_saturate_on();                     // half packs saturate at +-65504 (common.h)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sX = smem;                             // 2 x MQ_BUF
  unsigned char* const sH = smem + 2 * MQ_BUF;                // 2 x MQ_BUF
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const bool l1 = wave < 4;

  // ---- resident weights: [32-channel group][chunk][nt][lane][16 B] (cp_pack_gemm_weight); layer 1: groups 2 w, 2 w + 1 of 8;
  // layer 2: groups 0, 1 of 2 (every layer-2 wave holds all 64 output channels)
  u32x4 W[8][4];
  {
    const u32x4* const wsrc = (const u32x4*)(l1 ? p.w1 : p.w2);
    const int g0 = l1 ? 2 * wave : 0;
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
#pragma unroll
      for (int t = 0; t < 4; ++t) W[kc][t] = wsrc[((size_t)((g0 + (t >> 1)) * 8 + kc) * 2 + (t & 1)) * 64 + lane];
  }
  float* const sW3 = (float*)(smem + 4 * MQ_BUF);            // [2][64]: layer-3 rows (kept out of the layer-1 waves' registers)
  // folded-BN scale / shift of layers 1 and 2 in LDS: read from global memory in the tile loop, the layer-2 waves' loads queued behind
  // the eight tile-prefetch loads they had just issued (vmcnt is in order: waiting for the vectors meant waiting for the prefetch)
  float* const sA1 = sW3 + 128;                               // s1 | t1 [256 each]
  float* const sA2 = sA1 + 512;                               // s2 | t2 [64 each]
  if (tid < 128) sW3[tid] = p.w3[tid];
  if (tid < 256) { sA1[tid] = p.s1[tid]; sA1[256 + tid] = p.t1[tid]; }
  if (tid >= 256 && tid < 320) { sA2[tid - 256] = p.s2[tid - 256]; sA2[64 + tid - 256] = p.t2[tid - 256]; }
  const float b3a = p.b3[0], b3b = p.b3[1];

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  // staging by the 256 threads of the layer-2 waves (they carry a quarter of the MFMA work; the layer-1 waves have no registers
  // to spare for the eight in-flight pieces): piece i = (tid - 256) + 256 k (k = 0..7): row = i / 32, piece-in-row = i % 32
  const int s_pr = tid & 31, s_row = (tid & 255) >> 5;        // rows s_row + 8 k
  auto stage_load = [&](u32x4* v, int rt) {
    const int m0 = rt * MQ_ROWS;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int m = m0 + s_row + 8 * k;
      const uint32_t off = (rt < p.n_rt && m < p.M) ? (uint32_t)(m * p.in_cs + p.in_coff + s_pr * 8) * 2u : 0x80000000u;
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };
  auto stage_write = [&](const u32x4* v, int buf) {
    unsigned char* dst = sX + buf * MQ_BUF + s_pr * MQ_PITCH + s_row * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) *(u32x4*)(dst + k * 8 * 16) = v[k];
  };

  const int step = gridDim.x;
  u32x4 sv[8];
  if (!l1) {
    stage_load(sv, blockIdx.x);
    stage_write(sv, 0);
  }
  __syncthreads();
  // iteration it: layer 1 works on tile rt (x in sX[it & 1] -> hidden in sH[it & 1]), layers 2 + 3 on tile rt - step (sH[(it - 1) & 1])
  int it = 0;
  for (int rt = blockIdx.x; rt - step < p.n_rt; rt += step, ++it) {
    if (l1) {
      if (rt < p.n_rt) {
#pragma unroll 1
        for (int mh = 0; mh < 2; ++mh) {                      // two passes of 32 rows: bounds the live accumulators (32 VGPRs)
          f32x4 acc[2][4];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          const unsigned char* ab = sX + (it & 1) * MQ_BUF + q * MQ_PITCH + (mh * 32 + x) * 16;
#pragma unroll
          for (int kc = 0; kc < 8; ++kc)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const u32x4 a = *(const u32x4*)(ab + kc * 4 * MQ_PITCH + mt * 256);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                acc[mt][t] = cp_mma16<H>(W[kc][t], a, acc[mt][t]);
            }
          // lane (x, q): rows 32 mh + 16 mt + x, hidden channels (2 wave + h) * 32 + 8 q + {0..7} = piece (2 wave + h) * 4 + q
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int ch = (2 * wave + h) * 32 + q * 8;
            const f32x4 s0 = *(const f32x4*)(sA1 + ch), t0 = *(const f32x4*)(sA1 + 256 + ch);
            const f32x4 s1 = *(const f32x4*)(sA1 + ch + 4), t1 = *(const f32x4*)(sA1 + 256 + ch + 4);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v[j] = leaky(acc[mt][2 * h][j] * s0[j] + t0[j], p.slope1);
                v[4 + j] = leaky(acc[mt][2 * h + 1][j] * s1[j] + t1[j], p.slope1);
              }
              *(u32x4*)(sH + (it & 1) * MQ_BUF + ((2 * wave + h) * 4 + q) * MQ_PITCH + (mh * 32 + mt * 16 + x) * 16) = cp_pack8<H>(v);
            }
          }
        }
      }
    } else {
      stage_load(sv, rt + step);                              // the tile layer 1 takes next; past the end: zeros, never consumed
      if (it > 0) {
      const int v4 = wave - 4;                                // rows 16 v4 .. 16 v4 + 15 of tile rt - step
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* ab = sH + ((it - 1) & 1) * MQ_BUF + q * MQ_PITCH + (v4 * 16 + x) * 16;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const u32x4 a = *(const u32x4*)(ab + kc * 4 * MQ_PITCH);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = cp_mma16<H>(W[kc][t], a, acc[t]);
      }
      // lane (x, q): row 16 v4 + x, hidden-2 channels 32 h + 8 q + {0..7}; layer 3 on the fp32 values
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ch = h * 32 + q * 8;
        const f32x4 s0 = *(const f32x4*)(sA2 + ch), t0 = *(const f32x4*)(sA2 + 64 + ch);
        const f32x4 s1 = *(const f32x4*)(sA2 + ch + 4), t1 = *(const f32x4*)(sA2 + 64 + ch + 4);
        const f32x4 a0 = *(const f32x4*)(sW3 + ch), a1 = *(const f32x4*)(sW3 + ch + 4);
        const f32x4 c0 = *(const f32x4*)(sW3 + 64 + ch), c1 = *(const f32x4*)(sW3 + 64 + ch + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u0 = leaky(acc[2 * h][j] * s0[j] + t0[j], p.slope2), u1 = leaky(acc[2 * h + 1][j] * s1[j] + t1[j], p.slope2);
          d0 += u0 * a0[j] + u1 * a1[j];
          d1 += u0 * c0[j] + u1 * c1[j];
        }
      }
      d0 += __shfl_xor(d0, 16); d1 += __shfl_xor(d1, 16);
      d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32);
      const int m = (rt - step) * MQ_ROWS + v4 * 16 + x;
      if (q == 0 && m < p.M) {
        const int b = m / p.Nrow, n = m - b * p.Nrow;
        float* o = p.out + p.o_base + (long long)b * p.o_sb + (long long)n * p.o_sn;
        o[0] = d0 + b3a;
        o[p.o_sc] = d1 + b3b;
      }
      }
      stage_write(sv, (it + 1) & 1);                          // sX[(it + 1) & 1]: its readers (layer 1, iteration it - 1) passed the last barrier
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int cp_mlp_query_fused_supported(int C0, int C1, int C2, int C3) { return (C0 == 256 && C1 == 256 && C2 == 64 && C3 == 2) ? 1 : 0; }

extern "C" int cp_mlp_query_fused_t(cp_stream_t stream, int dtype, const void* in, int in_cstride, int in_coff, int B, int N,
                                  const void* packed_w1, const float* scale1, const float* shift1, float slope1,
                                  const void* packed_w2, const float* scale2, const float* shift2, float slope2,
                                  const float* w3, const float* b3, float* out, long long o_base, long long o_sb, long long o_sn,
                                  long long o_sc) {
  if (!in || !packed_w1 || !scale1 || !shift1 || !packed_w2 || !scale2 || !shift2 || !w3 || !b3 || !out || B <= 0 || N <= 0 ||
      (dtype != CP_BF16 && dtype != CP_F16))
    return CP_ERR_INVALID;
  if (in_cstride % 8 || in_coff % 8 || in_coff + 256 > in_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w1) || !cp_aligned16(packed_w2) || !cp_aligned16(scale1) || !cp_aligned16(shift1) ||
      !cp_aligned16(scale2) || !cp_aligned16(shift2))
    return CP_ERR_ALIGN;
  const long long M = (long long)B * N;
  const long long in_bytes = M * in_cstride * 2;
  if (in_bytes >= (1LL << 32) || M >= (1LL << 31)) return CP_ERR_RANGE;
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)mlp_query_fused_kernel<false>, MQ_LDS) &&
                                  cp_set_max_lds((const void*)mlp_query_fused_kernel<true>, MQ_LDS));
  const int n_cu = cp_num_cus();
  if (n_cu <= 0) return CP_ERR_HIP;
  MlpQueryParams p;
  p.in = in; p.w1 = packed_w1; p.s1 = scale1; p.t1 = shift1; p.w2 = packed_w2; p.s2 = scale2; p.t2 = shift2; p.w3 = w3; p.b3 = b3; p.out = out;
  p.M = (int)M; p.Nrow = N; p.in_cs = in_cstride; p.in_coff = in_coff; p.n_rt = (int)((M + MQ_ROWS - 1) / MQ_ROWS);
  p.in_bytes = (uint32_t)in_bytes; p.slope1 = slope1; p.slope2 = slope2;
  p.o_base = o_base; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sc = o_sc;
  const int grid = p.n_rt < n_cu ? p.n_rt : n_cu;
  if (dtype == CP_F16) CP_LAUNCH((mlp_query_fused_kernel<true>), dim3((unsigned)grid), dim3(512), MQ_LDS, (hipStream_t)stream, p);
  else CP_LAUNCH((mlp_query_fused_kernel<false>), dim3((unsigned)grid), dim3(512), MQ_LDS, (hipStream_t)stream, p);
  return cp_check_launch();
}

extern "C" int cp_mlp_query_fused(cp_stream_t stream, const void* in, int in_cstride, int in_coff, int B, int N,
                                  const void* packed_w1, const float* scale1, const float* shift1, float slope1,
                                  const void* packed_w2, const float* scale2, const float* shift2, float slope2,
                                  const float* w3, const float* b3, float* out, long long o_base, long long o_sb, long long o_sn,
                                  long long o_sc) {
  return cp_mlp_query_fused_t(stream, CP_BF16, in, in_cstride, in_coff, B, N, packed_w1, scale1, shift1, slope1, packed_w2, scale2, shift2,
                              slope2, w3, b3, out, o_base, o_sb, o_sn, o_sc);
}

// ------------------------------------------------------------------------------------------------------------------------------
// pre_graph_module (reference pipeline.py:237-240, :283-286: Linear(256 + g -> 256) + LeakyReLU, Linear(256 -> 256) + LeakyReLU over
// the concatenated [local feature | previous graph feature] rows) as ONE launch: the 256-channel hidden rows never leave the chip.
// A persistent 12-wave workgroup per CU walks tiles of 32 rows as a two-stage pipeline:
//   waves 0-7 (layer 1, K <= 512): wave w keeps W1's 32-channel group w in registers (16 K chunks x 2 tiles = 128 VGPRs); the
//     tile's rows arrive by LDS-DMA (no staging registers: the budget is 168 VGPRs at three waves per SIMD), two tiles ahead, into a
//     ring of three row-major images whose 16-byte pieces are XOR-swizzled with the row number inside each 256-byte block -- a
//     fragment read (16 consecutive rows, one piece each) then covers all 64 banks, and a DMA instruction still moves whole
//     256-byte blocks of a row; LeakyReLU, bf16, into the hidden tile (same swizzle, double-buffered);
//   waves 8-11 (layer 2): wave v keeps W2's channels 64 v .. 64 v + 63 in registers (8 chunks x 4 tiles) and works on the PREVIOUS
//     tile's hidden rows; LeakyReLU, bf16, 16-byte stores (a row's 64 channels of a wave = 128 contiguous bytes).
// One barrier per tile; the layer-1 waves wait with a COUNTED vmcnt (their only vector-memory operations inside the loop are the
// DMA pieces, the same number every iteration), so the tile that streams in two iterations ahead is not waited for.
constexpr int MP_ROWS = 32;
constexpr int MP_HP = 32;                                    // pieces per hidden row (256 channels)
constexpr int MP_HBUF = MP_ROWS * MP_HP * 16;                // 16 384 B

typedef __attribute__((ext_vector_type(4))) int32_t i32x4;

struct MlpPairParams {
  const void* in; const void* w1; const float* t1;
  const void* w2; const float* t2;
  void* out;
  int M, in_cs, in_coff, nchunk1, n_rt, out_cs, out_coff;
  float slope1, slope2;
  // GATHER instances (cp_mlp_pair_fused_gather): the first 32 pieces of a row are Index2Feat_module's four 64-channel taps
  // (pipeline.py:156-163), fetched by the DMA loader straight from patch_generator's output map; `in` holds only the graph part
  const void* patches; const int32_t* x_id; const int32_t* y_id; const float* mask; const void* zeros;
  int Nrow, Hp, Wp, p_cs, p_coff, kk;                       // Nrow: log2 of the keypoints per crop
};

// slot (in 16-byte pieces) of piece `pc` of tile row `r` in an image of P pieces per row
__device__ __forceinline__ int mp_slot(int r, int pc, int P) { return r * P + (pc & ~15) + ((pc ^ r) & 15); }

#ifdef CP_DEBUG_KNOBS          // phase clock of workgroup 0 (tools/mlp_stamps.py): s_memtime sums per wave, `make KNOBS=1` builds only
__device__ unsigned long long mp_stamps[12][8];
#define MP_T0() unsigned long long mp_t = __builtin_amdgcn_s_memtime(), mp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MP_MARK(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); mp_acc[k] += n_ - mp_t; mp_t = n_; } while (0)
#define MP_DUMP() do { if (blockIdx.x == 0 && lane == 0) for (int k_ = 0; k_ < 8; ++k_) mp_stamps[wave][k_] = mp_acc[k_]; } while (0)
#else
#define MP_T0() do {} while (0)
#define MP_MARK(k) do {} while (0)
#define MP_DUMP() do {} while (0)
#endif

#ifndef MP_L2B
#define MP_L2B 2
#endif
constexpr int L2B = MP_L2B;
#ifndef MP_L2PRIO
#define MP_L2PRIO 3
#endif
// An opaque copy of the lane id per use site (keeps the dozen address instructions of a site inside the loop instead of hoisted,
// spilled and reloaded -- a scratch reload is a VMEM load and costs a vmcnt(0) on the tile DMA in flight).  The gathering instance
// has no register left even for `lane` itself (hipcc spilled IT and reloaded it in front of every site), so there the id is
// recomputed from the hardware (v_mbcnt: two VALU instructions, nothing live across the loop).
template <bool FRESH>
__device__ __forceinline__ int mp_lane(int lane) {
  int ln;
  if constexpr (FRESH) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
  else { ln = lane; asm volatile("" : "+v"(ln)); }
  return ln;
}

template <int NI, bool GATHER = false, bool H = false>       // DMA instructions per layer-1 wave and tile = P / 16: 4 / 3 / 2; H: IEEE half (CP_F16)
__global__ __launch_bounds__(768) void mlp_pair_fused_kernel(const MlpPairParams p) {
  if constexpr (H) cp_f16_saturate_on();                     // half packs saturate at +-65504 (common.h)
  static_assert(!GATHER || NI >= 3, "the gathering loader moves one row (48 or 64 pieces) per DMA instruction");
  constexpr int ND = GATHER ? 4 : NI;                         // DMA instructions per layer-1 wave and tile (gathering: its 4 rows, P lanes each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int P = 16 * NI;                                  // pieces per row image: 64 (K <= 512) / 48 (K <= 384) / 32 (K <= 256)
  constexpr int XBUF = MP_ROWS * P * 16;
  unsigned char* const sX = smem;                             // 3 x XBUF
  unsigned char* const sH = smem + 3 * XBUF;                  // 2 x MP_HBUF
  float* const sAff = (float*)(sH + 2 * MP_HBUF);             // bias 1 | bias 2, 256 floats each
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool l1 = wave < 8;

  u32x4 W[32];                                                // layer 1: [kc 0..15][t 0..1]; layer 2: [kc 0..7][t 0..3]
  if (l1) {
#pragma unroll
    for (int kc = 0; kc < 16; ++kc)
#pragma unroll
      for (int t = 0; t < 2; ++t) {                             // chunks past the layer's K: zero weights (no branches: the loop below runs 4 NI chunks)
        const bool live = kc < p.nchunk1;
        const u32x4 w_ = ((const u32x4*)p.w1)[((size_t)(wave * p.nchunk1 + (live ? kc : 0)) * 2 + t) * 64 + lane];
        W[kc * 2 + t] = live ? w_ : u32x4{0u, 0u, 0u, 0u};
      }
  } else {
    const int g0 = 2 * (wave - 8);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
#pragma unroll
      for (int t = 0; t < 4; ++t) W[kc * 4 + t] = ((const u32x4*)p.w2)[((size_t)((g0 + (t >> 1)) * 8 + kc) * 2 + (t & 1)) * 64 + lane];
  }
  for (int i = tid; i < 256; i += 768) { sAff[i] = p.t1[i]; sAff[256 + i] = p.t2[i]; }

  // DMA piece j (0 .. NI - 1) of layer-1 wave w: LDS slots (w * NI + j) * 64 + lane of the image, i.e. row = slot / P, stored piece
  // index s = slot % P holds the row's piece (s & ~15) + ((s ^ row) & 15) (the swizzle is an involution); pieces past the row's
  // real width re-read its piece 0 (their weights are zero)
  const int step = gridDim.x;
  const int npiece = p.nchunk1 * 4;
  // gathering loader: the keypoint ids / RoI bits of the wave's four rows of the tile whose DMA is issued NEXT (scalar registers)
  i32x4 gxi = {0, 0, 0, 0}, gyi = {0, 0, 0, 0}, gmk = {0, 0, 0, 0};
  long long gm0 = 0;
  // The three id arrays are read-only for the whole launch and the addresses are wave-uniform: loaded through CONSTANT-address-space
  // pointers the compiler emits s_load_dwordx4 itself and owns the wait (round 5 issued the loads in one inline-asm statement and waited in
  // another: between the two the register allocator believed the destination SGPRs defined and was free to copy or spill them before
  // the data had landed).  ids_issue() sits in front of the tile barrier, the first use behind it: the barrier's skew hides the scalar-cache
  // miss as before.
  typedef const __attribute__((address_space(4))) i32x4 c_i32x4;
  auto ids_issue = [&](int rt) {
    if constexpr (GATHER) {
      long long m0 = (long long)rt * MP_ROWS + wave * 4;
      if (m0 > p.M - 4) m0 = p.M - 4;                          // rows past the end: valid rows, never stored
      gm0 = m0;
      gxi = *(c_i32x4*)(p.x_id + m0);
      gyi = *(c_i32x4*)(p.y_id + m0);
      gmk = *(c_i32x4*)((const int32_t*)p.mask + m0);
    }
  };
  auto ids_wait = [&]() {};
  auto dma_tile = [&](int rt, int buf) {
    // The lane's piece offsets are loop invariants: hipcc hoisted them out of the tile loop as NI 64-bit pairs, spilled them (the
    // layer-1 waves hold 128 weight registers) and reloaded them in front of every DMA -- and a scratch reload is a VMEM load:
    // using it needs vmcnt(0), i.e. it waited for the tile DMA'd one iteration ago (38 % of a layer-1 wave's time by the phase
    // clock, tools/mlp_stamps.py).  An opaque copy of the lane id keeps the arithmetic (a dozen VALU instructions per piece) inside.
    int ln = mp_lane<GATHER>(lane);
    if constexpr (GATHER) {
      // One DMA instruction = one row (P = 64): the row's keypoint ids and RoI bit are wave-uniform -> scalar loads (lgkmcnt, so the
      // counted vmcnt of the tile loop still sees DMA pieces only).  Piece pc < 32: tap pc >> 3 (sf1..sf4 of Index2Feat_module.forward,
      // pipeline.py:158-161: (2v, 2u), (2v + k, 2u), (2v, 2u + k), (2v + k, 2u + k)), 16-byte channel group pc & 7 of the map pixel;
      // a row whose RoI bit is 0 reads zeros (the reference multiplies the taps by the {0, 1} mask, :280); 32 <= pc < npiece: the
      // previous graph feature from `in`; pieces past the row's width read zeros (their weights are zero too).
      // (scalar loads by hand: hipcc cannot prove that the kernel's own stores do not alias the id arrays and used per-lane
      //  global_load_dwords -- VMEM operations in the middle of the counted DMA stream, 24 vmcnt(0) in the loop.)  The wave's four rows
      //  are consecutive and N % 4 == 0, so they belong to ONE crop and their ids are one s_load_dwordx4 per array.
      //  The ids of THIS tile were loaded one call ahead (ids_issue / ids_wait below: issued before the tile barrier, waited for
      //  behind it, so the barrier's skew hides the scalar-cache miss); P < 64: the upper lanes sit out (exec mask).
      const long long m0 = gm0;
      const int b0 = (int)(m0 >> p.Nrow);                      // Nrow = log2(N) here (N is a power of two: host check)
      const unsigned char* const pbase = (const unsigned char*)p.patches + ((size_t)b0 * p.Hp * p.Wp * p.p_cs + p.p_coff) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wave * 4 + j;
        const int pc = (ln & ~15) + ((ln ^ r) & 15);
        const int tap = pc >> 3;
        const int y = min(2 * gyi[j] + ((tap & 1) ? p.kk : 0), p.Hp - 1), x = min(2 * gxi[j] + ((tap & 2) ? p.kk : 0), p.Wp - 1);
        const unsigned char* src = (const unsigned char*)p.zeros + (pc & 7) * 16;
        if (pc < 32) {
          if (gmk[j] != 0) src = pbase + ((size_t)(y * p.Wp + x) * p.p_cs) * 2 + (pc & 7) * 16;     // the {0, 1} RoI bit as fp32 bits: 0 = 0.f
        } else if (pc < npiece) {
          src = (const unsigned char*)p.in + ((size_t)(m0 + j) * p.in_cs + p.in_coff) * 2 + (size_t)(pc - 32) * 16;
        }
        if (P == 64 || ln < P)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(sX + buf * XBUF + r * (P * 16)), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int slot = (wave * NI + j) * 64 + ln;
        const int r = slot / P, s_ = slot - r * P;
        int pc = (s_ & ~15) + ((s_ ^ r) & 15);
        if (pc >= npiece) pc = 0;
        long long m = (long long)rt * MP_ROWS + r;
        if (m >= p.M) m = p.M - 1;                              // rows past the end: a valid row, never stored
        const unsigned char* src = (const unsigned char*)p.in + ((size_t)m * p.in_cs + p.in_coff) * 2 + (size_t)pc * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sX + buf * XBUF + (wave * NI + j) * 1024), 16, 0, 0);
      }
    }
  };
  __syncthreads();                                            // weights in registers (vmcnt 0), affine table written
  auto dma_target = [&](int rt) { return rt + 2 * step < p.n_rt ? rt + 2 * step : (int)blockIdx.x; };
  if (l1) {
    const int t1 = blockIdx.x + step < p.n_rt ? blockIdx.x + step : blockIdx.x;
    ids_issue(blockIdx.x); ids_wait();
    dma_tile(blockIdx.x, 0);
    ids_issue(t1); ids_wait();
    dma_tile(t1, 1);
    __builtin_amdgcn_s_waitcnt(0x0070 | ND);                  // vmcnt(ND): tile 0 has landed
    ids_issue(dma_target(blockIdx.x));                        // for the first iteration's DMA; waited for behind the barrier
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  if (l1) ids_wait();

  // iteration it: layer 1 on tile rt (sX[it % 3] -> sH[it & 1]) while tile rt + 2 step streams into sX[(it + 2) % 3]; layer 2 on
  // tile rt - step (sH[(it - 1) & 1])
  if (!l1) __builtin_amdgcn_s_setprio(MP_L2PRIO);             // the four layer-2 waves are the tile loop's critical path (phase clock: 91 % busy
  int it = 0, xb = 0;                                         // against 70 % of the layer-1 waves they share the SIMDs with): they issue first
  MP_T0();
  for (int rt = blockIdx.x; rt - step < p.n_rt; rt += step, ++it) {
    if (l1) {
      const int nb = xb >= 1 ? xb - 1 : 2;                    // (it + 2) % 3
      dma_tile(dma_target(rt), nb);                           // always ND pieces: the vmcnt arithmetic below counts them
      ids_issue(dma_target(rt + step));                       // the next iteration's DMA target: its ids travel under this tile's MFMAs
      MP_MARK(0);                                             // DMA issue
      if (rt < p.n_rt) {
        // (lane geometry from an opaque copy of the lane id, per iteration: hoisted out of the loop these few values were spilled
        //  beside the 128 weight registers, and their scratch reloads -- VMEM loads -- each cost a vmcnt(0) on the tile DMA in flight)
        int ln = mp_lane<GATHER>(lane);
        const int x = ln & 15, q = ln >> 4, qx = q ^ x;
        f32x4 acc[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // piece 4 kc + q of row 16 f + x sits at slot (kc >> 2) * 16 + ((4 (kc & 3)) ^ q ^ x) of its row (mp_slot): a lane constant
        // XOR one of four compile-time values
        const unsigned char* const xbp = sX + xb * XBUF + x * (P * 16);
        // two K chunks per step: four fragment reads, then eight MFMAs.  (Every LDS wait in this kernel is lgkmcnt(0) -- the FLAT-encoded
        // LDS-DMA makes hipcc's wait insertion conservative, common.h -- so a step pays one LDS round trip whatever it reads; the phase
        // clock, tools/mlp_stamps.py, had the chunk-at-a-time form at 16 exposed round trips per tile.)
#pragma unroll
        for (int k2 = 0; k2 < NI * 2; ++k2) {
          u32x4 a[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              const int kc = 2 * k2 + h;
              a[h][f] = *(const u32x4*)(xbp + f * (16 * P * 16) + (kc >> 2) * 256 + (((4 * (kc & 3)) ^ qx) << 4));
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              const int kc = 2 * k2 + h;
              acc[f][0] = cp_mma16<H>(W[kc * 2], a[h][f], acc[f][0]);
              acc[f][1] = cp_mma16<H>(W[kc * 2 + 1], a[h][f], acc[f][1]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        MP_MARK(1);                                           // layer-1 MFMA loop
        // lane (x, q): rows 16 f + x, hidden channels 32 wave + 8 q + {0..7} = piece 4 wave + q
        const int ch = wave * 32 + q * 8;
        const f32x4 t0 = *(const f32x4*)(sAff + ch), t1 = *(const f32x4*)(sAff + ch + 4);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = leaky(acc[f][0][j] + t0[j], p.slope1);
            v[4 + j] = leaky(acc[f][1][j] + t1[j], p.slope1);
          }
          *(u32x4*)(sH + (it & 1) * MP_HBUF + (f * 16 + x) * (MP_HP * 16) + (wave >> 2) * 256 + (((4 * (wave & 3)) ^ qx) << 4)) = cp_pack8<H>(v);
        }
      }
      MP_MARK(2);                                             // layer-1 epilogue
      __builtin_amdgcn_s_waitcnt(0x0070 | ND);                // vmcnt(ND), lgkmcnt(0): the tile of the NEXT iteration has landed
      MP_MARK(3);                                             // DMA wait
    } else {
      if (it > 0) {
        int ln = mp_lane<GATHER>(lane);
        const int x = ln & 15, q = ln >> 4, qx = q ^ x;
        const int wv = wave - 8;
        const unsigned char* const hb = sH + ((it - 1) & 1) * MP_HBUF + x * (MP_HP * 16);
        const int ch = wv * 64 + q * 8;
#pragma unroll 1
        for (int f = 0; f < 2; ++f) {
          f32x4 acc[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k4 = 0; k4 < 8 / L2B; ++k4) {                // L2B K chunks per step (see layer 1): 8 / L2B LDS round trips per fragment
            u32x4 a[L2B];
#pragma unroll
            for (int h = 0; h < L2B; ++h) {
              const int kc = L2B * k4 + h;
              a[h] = *(const u32x4*)(hb + f * (16 * MP_HP * 16) + (kc >> 2) * 256 + (((4 * (kc & 3)) ^ qx) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < L2B; ++h)
#pragma unroll
              for (int t = 0; t < 4; ++t)
                acc[t] = cp_mma16<H>(W[(L2B * k4 + h) * 4 + t], a[h], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
          }
          const long long m = (long long)(rt - step) * MP_ROWS + f * 16 + x;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = ch + 32 * h;
            const f32x4 t0 = *(const f32x4*)(sAff + 256 + c), t1 = *(const f32x4*)(sAff + 256 + c + 4);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = leaky(acc[2 * h][j] + t0[j], p.slope2);
              v[4 + j] = leaky(acc[2 * h + 1][j] + t1[j], p.slope2);
            }
            if (m < p.M) *(u32x4*)((uint16_t*)p.out + (size_t)m * p.out_cs + p.out_coff + c) = cp_pack8<H>(v);
          }
        }
      }
      MP_MARK(5);                                             // layer 2 (both fragments, stores issued)
      __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0) only: this wave's stores stay in flight
      MP_MARK(6);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    if (l1) ids_wait();
    MP_MARK(4);                                               // barrier
    xb = xb == 2 ? 0 : xb + 1;
  }
  MP_DUMP();
}

#ifdef CP_DEBUG_KNOBS
extern "C" int cp_debug_mlp_pair_stamps(unsigned long long* out96) {       // 12 waves x 8 phase sums of workgroup 0's last launch
  return hipMemcpyFromSymbol(out96, HIP_SYMBOL(mp_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? CP_OK : CP_ERR_HIP;
}
#endif

static size_t mlp_pair_lds(int P) { return (size_t)3 * MP_ROWS * P * 16 + 2 * MP_HBUF + 2 * 256 * 4; }

extern "C" int cp_mlp_pair_fused_supported(int Cin, int C1, int C2) {
  return (Cin >= 64 && Cin <= 512 && Cin % 32 == 0 && C1 == 256 && C2 == 256) ? 1 : 0;
}

static int mlp_pair_launch(cp_stream_t stream, MlpPairParams& p, int Cin, long long M, bool gather, int dtype);

extern "C" int cp_mlp_pair_fused_gather_supported(int Cg, int E_ch, int k) {
  return ((Cg == 64 || Cg == 128 || Cg == 192 || Cg == 256) && E_ch == 64 && k >= 1 && k <= 4) ? 1 : 0;      // + N a power of two >= 4 at the call
}

extern "C" int cp_mlp_pair_fused_gather_t(cp_stream_t stream, int dtype, const CpI2fGather* g, const void* gin, int gin_cstride, int gin_coff,
                                          int Cg, int B, int N, const void* packed_w1, const float* bias1, float slope1,
                                          const void* packed_w2, const float* bias2, float slope2, void* out, int out_cstride, int out_coff) {
  if (dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;
  if (!g || !g->patches || !g->x_id || !g->y_id || !g->mask || !g->zeros || !gin || !packed_w1 || !bias1 || !packed_w2 || !bias2 || !out ||
      B <= 0 || N <= 0)
    return CP_ERR_INVALID;
  if (!cp_mlp_pair_fused_gather_supported(Cg, 64, g->k) || g->Hp <= 0 || g->Wp <= 0 || N < 4 || (N & (N - 1))) return CP_ERR_INVALID;   // N = 2^n
  if (g->p_cstride % 8 || g->p_coff % 8 || g->p_coff + 64 > g->p_cstride) return CP_ERR_ALIGN;
  if (gin_cstride % 8 || gin_coff % 8 || gin_coff + Cg > gin_cstride || out_cstride % 8 || out_coff % 8 || out_coff + 256 > out_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(g->patches) || !cp_aligned16(g->zeros) || !cp_aligned16(gin) || !cp_aligned16(packed_w1) || !cp_aligned16(packed_w2) || !cp_aligned16(out))
    return CP_ERR_ALIGN;
  const long long M = (long long)B * N;
  if (M >= (1LL << 31) / MP_ROWS * MP_ROWS || (long long)B * g->Hp * g->Wp * g->p_cstride * 2 >= (1LL << 40)) return CP_ERR_RANGE;
  MlpPairParams p = {};
  p.in = gin; p.in_cs = gin_cstride; p.in_coff = gin_coff;
  p.w1 = packed_w1; p.t1 = bias1; p.w2 = packed_w2; p.t2 = bias2; p.out = out; p.out_cs = out_cstride; p.out_coff = out_coff;
  p.slope1 = slope1; p.slope2 = slope2;
  p.patches = g->patches; p.x_id = g->x_id; p.y_id = g->y_id; p.mask = g->mask; p.zeros = g->zeros;
  p.Nrow = __builtin_ctz((unsigned)N); p.Hp = g->Hp; p.Wp = g->Wp; p.p_cs = g->p_cstride; p.p_coff = g->p_coff; p.kk = g->k;
  return mlp_pair_launch(stream, p, 256 + Cg, M, true, dtype);
}

extern "C" int cp_mlp_pair_fused_gather(cp_stream_t stream, const CpI2fGather* g, const void* gin, int gin_cstride, int gin_coff, int Cg,
                                        int B, int N, const void* packed_w1, const float* bias1, float slope1, const void* packed_w2,
                                        const float* bias2, float slope2, void* out, int out_cstride, int out_coff) {
  return cp_mlp_pair_fused_gather_t(stream, CP_BF16, g, gin, gin_cstride, gin_coff, Cg, B, N, packed_w1, bias1, slope1, packed_w2, bias2, slope2,
                                    out, out_cstride, out_coff);
}

extern "C" int cp_mlp_pair_fused_t(cp_stream_t stream, int dtype, const void* in, int in_cstride, int in_coff, int Cin, int B, int N,
                                   const void* packed_w1, const float* bias1, float slope1, const void* packed_w2, const float* bias2,
                                   float slope2, void* out, int out_cstride, int out_coff) {
  if (dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;
  if (!in || !packed_w1 || !bias1 || !packed_w2 || !bias2 || !out || B <= 0 || N <= 0) return CP_ERR_INVALID;
  if (!cp_mlp_pair_fused_supported(Cin, 256, 256)) return CP_ERR_INVALID;
  if (in_cstride % 8 || in_coff % 8 || in_coff + Cin > in_cstride || out_cstride % 8 || out_coff % 8 || out_coff + 256 > out_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(in) || !cp_aligned16(packed_w1) || !cp_aligned16(packed_w2) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const long long M = (long long)B * N;
  if (M >= (1LL << 31) / MP_ROWS * MP_ROWS) return CP_ERR_RANGE;
  MlpPairParams p = {};
  p.in = in; p.w1 = packed_w1; p.t1 = bias1; p.w2 = packed_w2; p.t2 = bias2; p.out = out;
  p.in_cs = in_cstride; p.in_coff = in_coff;
  p.out_cs = out_cstride; p.out_coff = out_coff; p.slope1 = slope1; p.slope2 = slope2;
  return mlp_pair_launch(stream, p, Cin, M, false, dtype);
}

extern "C" int cp_mlp_pair_fused(cp_stream_t stream, const void* in, int in_cstride, int in_coff, int Cin, int B, int N,
                                 const void* packed_w1, const float* bias1, float slope1, const void* packed_w2, const float* bias2,
                                 float slope2, void* out, int out_cstride, int out_coff) {
  return cp_mlp_pair_fused_t(stream, CP_BF16, in, in_cstride, in_coff, Cin, B, N, packed_w1, bias1, slope1, packed_w2, bias2, slope2, out,
                             out_cstride, out_coff);
}

static int mlp_pair_launch(cp_stream_t stream, MlpPairParams& p, int Cin, long long M, bool gather, int dtype) {
  const int nchunk = Cin / 32;
  // 32 / 48 / 64 pieces per row image.  Never below 32: the smallest instance is <2> (Cin <= 128 used to give P = 16, i.e. LDS sized
  // for 16 pieces under a kernel<4> launch that addresses 64: out-of-bounds LDS, silently wrong); pieces past the row's width
  // re-read its piece 0 under zero weights, so kernel<2> serves every Cin <= 256.  The gathering loader has 48- and 64-piece instances.
  const int P = nchunk <= 8 ? (gather ? 48 : 32) : (nchunk * 4 + 15) / 16 * 16;
  const size_t lds = mlp_pair_lds(P);
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)mlp_pair_fused_kernel<2>, mlp_pair_lds(32)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<3>, mlp_pair_lds(48)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<4>, mlp_pair_lds(64)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<3, true>, mlp_pair_lds(48)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<4, true>, mlp_pair_lds(64)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<2, false, true>, mlp_pair_lds(32)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<3, false, true>, mlp_pair_lds(48)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<4, false, true>, mlp_pair_lds(64)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<3, true, true>, mlp_pair_lds(48)) &&
                                  cp_set_max_lds((const void*)mlp_pair_fused_kernel<4, true, true>, mlp_pair_lds(64)));
  const int n_cu = cp_num_cus();
  if (n_cu <= 0) return CP_ERR_HIP;
  p.M = (int)M; p.nchunk1 = nchunk; p.n_rt = (int)((M + MP_ROWS - 1) / MP_ROWS);
  const int grid = p.n_rt < n_cu ? p.n_rt : n_cu;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CP_F16) {
    if (gather && P == 48) CP_LAUNCH((mlp_pair_fused_kernel<3, true, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
    else if (gather) CP_LAUNCH((mlp_pair_fused_kernel<4, true, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
    else if (P == 32) CP_LAUNCH((mlp_pair_fused_kernel<2, false, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
    else if (P == 48) CP_LAUNCH((mlp_pair_fused_kernel<3, false, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
    else CP_LAUNCH((mlp_pair_fused_kernel<4, false, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
  } else if (gather && P == 48) CP_LAUNCH((mlp_pair_fused_kernel<3, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
  else if (gather) CP_LAUNCH((mlp_pair_fused_kernel<4, true>), dim3((unsigned)grid), dim3(768), lds, st, p);
  else if (P == 32) CP_LAUNCH((mlp_pair_fused_kernel<2>), dim3((unsigned)grid), dim3(768), lds, st, p);
  else if (P == 48) CP_LAUNCH((mlp_pair_fused_kernel<3>), dim3((unsigned)grid), dim3(768), lds, st, p);
  else CP_LAUNCH((mlp_pair_fused_kernel<4>), dim3((unsigned)grid), dim3(768), lds, st, p);
  return cp_check_launch();
}
