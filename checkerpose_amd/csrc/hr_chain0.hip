// The 64x64x18 branch of an HRNet module (branch 0: 4 BasicBlocks = 8 convs) as ONE launch per crop.  Two forms, bit-identical:
//   * the PIPELINED form (hr_chain0p_kernel, further down; the one launched): rows stream through the eight convs, a wave per conv, over rings of
//     rows in LDS; the residual never leaves the chip (117-165 MB of HBM traffic per 256-crop launch);
//   * the BAND form (hr_chain0_kernel, rounds 2-4, launched by -DCP_C0_BAND builds): the whole map in LDS, updated in place band by band; the
//     residual goes through HBM (486-570 MB per launch).  Described first, because the pipelined form reuses its K order and fragment geometry:
//
// The other branches (hr_chain.hip) keep a conv's whole output in registers and write it back over the input map.  Here the
// map alone fills the LDS (66 x 66 ring pixels x 36 B = 157 KB) and a conv's output does not fit in registers (4096 px x 32
// accumulator rows), so a conv walks the map in 8 BANDS of 8 rows (wave w owns row 8 band + w = 4 fragments of 16 pixels):
//   band: 7 K-chunks x 4 fragments x 2 tiles of MFMAs -> epilogue in registers -> barrier -> the band's rows are written
//   back over the input rows, EXCEPT the band's last row, which the next band still needs as its top neighbour: wave 7 holds
//   it in registers (16 VGPRs) for one band and writes it one barrier later (lagged in-place update, one barrier per band).
// The BasicBlock residual cannot stay on chip: every block output also goes to the output tensor in HBM (it is the final
// result for the last block) and the next block's second conv reads its residual back from there, pixel for pixel by the
// lane that then overwrites it (prefetched at the start of the band, consumed in its epilogue).
// LDS image: channel groups as planes -- [ch 0-7][px][16 B], [ch 8-15][px][16 B], [ch 16-17][px][4 B] -- so a crop costs
// 36 B per pixel.  GEMM K order: the 18 (tap, 8-channel group) slots first, then the 2-channel plane FOUR TAPS TO A GROUP (taps
// 0-3, 4-7, 8): 21 groups of 8 -> 6 chunks (round 4; one group per tap made 27 -> 7 chunks, a seventh of the MFMAs and fragment
// reads spent on zero weights).  Chunks 0-3 are pure ds_read_b128; chunk 4: lane groups q = 0, 1 read b128, q = 2, 3 four dwords
// (four taps' 2-channel pairs) into the four registers of their operand quad; chunk 5: q = 0 one dword (tap 8), the rest zero weights.
// Both tiles' weight fragments of a conv (14 KB) live in registers; the next conv's are loaded chunk by chunk during the
// last band, each behind the last use of the fragment it replaces.
#include "common.h"

namespace {

constexpr int ZH = 64, ZW = 64, ZC = 18, ZWP = 66, ZHP = 66;
constexpr int ZPLPX = (ZHP * ZWP + 15) / 16 * 16;        // 4368 ring pixels per plane (padded)
constexpr int ZPL16 = ZPLPX * 16, ZPL4 = ZPLPX * 4;
constexpr int ZAFF = 32;                                 // floats per scale / shift vector
constexpr int ZLDS = 2 * ZPL16 + ZPL4 + 2 * 2 * ZAFF * 4;
constexpr int ZKC = 6;
constexpr size_t ZCONV_W = (size_t)ZKC * 2 * 1024;
static_assert(ZLDS <= 160 * 1024, "LDS budget");

struct Chain0Params {
  const void* src[4];
  int shift[4];
  int nsrc, relu_in;
  const void* w;          // [8][7][2][64][16 B]
  const float* aff;       // [8][2][32]
  void* out;              // (B, 64, 64, 24) bf16: block outputs (residual of the next block), finally the result
  int B;
  // tail (cp_hr_branch_chain_tail): the stride-2 fuse-layer convs that read this branch's output, run off the finished map in LDS.
  // Their output channels sit side by side in 16-byte pieces (conv i owns pieces [tps[i], tps[i] + tcph[i] / 8)), 4 pieces = one slab.
  int tnp;                // pieces in all (0 = no tail)
  int tps[3], tcph[3], trelu[3];
  const void* tw;         // [slab][7][2][64][16 B], BN scale folded in, zero rows beyond a conv's channels
  const float* tshift;    // [ZTAIL_CH] folded-BN shift per combined channel
  void* tout[3];          // (B, 32, 32, tcph[i]) bf16
  uint32_t* status;       // the device's sticky status word (api.hip) or nullptr: a bounded wait that runs out ORs its bit in
  int dbg_spin_limit;     // 0 = the shipped bounds; KNOBS builds: CP_C0_FORCE_TIMEOUT=n bounds every hand-over wait to n polls (test hook)
};
constexpr int ZTAIL_CH = 96;

__device__ __forceinline__ void mma16z(const u32x4& w, const u32x4& a, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

// x0 = [relu](sum of the fuse terms) -> LDS planes AND the output tensor (it is block 0's residual)
template <int NSRC>
__device__ __forceinline__ void stage0(const Chain0Params& p, unsigned char* smem, int tid, int b) {
  constexpr int U = 4, TOTAL = ZH * ZW * 3;
  for (int i0 = tid; i0 < TOTAL; i0 += U * 512) {
    u32x4 v[U][NSRC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 512;
      const int g = i % 3, px = i / 3;
      const int y = px >> 6, xx = px & 63;
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        const int sh = p.shift[k];
        const size_t o = (((size_t)b * (ZH >> sh) + (y >> sh)) * (ZW >> sh) + (xx >> sh)) * 3 + g;
        v[u][k] = ((const u32x4*)p.src[k])[o];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 512;
      const int g = i % 3, px = i / 3;
      const int y = px >> 6, xx = px & 63;
      float acc[8], f[8];
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        Vec16<BF16Tag>::unpack(v[u][k], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (k == 0) ? f[j] : acc[j] + f[j];
      }
      if (p.relu_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
      }
      const u32x4 pk = Vec16<BF16Tag>::pack(acc);
      const int pc = (y + 1) * ZWP + xx + 1;
      if (g < 2) *(u32x4*)(smem + g * ZPL16 + pc * 16) = pk;
      else *(uint32_t*)(smem + 2 * ZPL16 + pc * 4) = pk.x;
      ((u32x4*)p.out)[((size_t)b * ZH * ZW + px) * 3 + g] = pk;
    }
  }
}

#ifdef CP_DEBUG_KNOBS          // phase clock of workgroup 0 (tools/chain0_stamps.py): s_memtime sums per wave, `make KNOBS=1` builds only
__device__ unsigned long long z_stamps[8][8];
#define Z_T0() unsigned long long z_t = __builtin_amdgcn_s_memtime(), z_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define Z_MARK(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); z_acc[k] += n_ - z_t; z_t = n_; } while (0)
#define Z_DUMP() do { if (blockIdx.x == 0 && lane == 0) for (int k_ = 0; k_ < 8; ++k_) z_stamps[wave][k_] = z_acc[k_]; } while (0)
#else
#define Z_T0() do {} while (0)
#define Z_MARK(k) do {} while (0)
#define Z_DUMP() do {} while (0)
#endif

__global__ __launch_bounds__(512) void hr_chain0_kernel(const Chain0Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const sAff = (float*)(smem + 2 * ZPL16 + ZPL4);          // [2][2][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;

  // ---- this conv's weight fragments: registers
  const u32x4* const wg = (const u32x4*)p.w;
  u32x4 Wf[ZKC][2];
#pragma unroll
  for (int kc = 0; kc < ZKC; ++kc)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) Wf[kc][nt] = wg[(kc * 2 + nt) * 64 + lane];

  // ---- zero ring + plane padding, stage x0
  for (int i = tid; i < ZPLPX; i += 512) {
    const int ry = i / ZWP, rx = i - ry * ZWP;
    if (ry == 0 || ry >= ZH + 1 || rx == 0 || rx == ZW + 1) {
      *(u32x4*)(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};
      *(u32x4*)(smem + ZPL16 + i * 16) = u32x4{0u, 0u, 0u, 0u};
      *(uint32_t*)(smem + 2 * ZPL16 + i * 4) = 0u;
    }
  }
  switch (p.nsrc) {
    case 1: stage0<1>(p, smem, tid, b); break;
    case 2: stage0<2>(p, smem, tid, b); break;
    case 3: stage0<3>(p, smem, tid, b); break;
    default: stage0<4>(p, smem, tid, b); break;
  }
  if (tid < 2 * ZAFF) sAff[tid] = p.aff[tid];
  __syncthreads();

  // ---- per-lane K-group geometry per chunk: byte offsets relative to the fragment's window-top-left pixel index pb
  //      (16-byte planes: + pb * 16; 4-byte plane: + pb * 4)
  uint32_t o16[5], o4[4], o4l;                 // chunks 0..4 (b128 lanes); chunk 4's four dword taps (q = 2: taps 0-3, q = 3: 4-7); chunk 5 (tap 8)
#pragma unroll
  for (int kc = 0; kc < 5; ++kc) {
    const int G = 4 * kc + q;
    const int tap = G >> 1, cg = G & 1;
    const int r = (tap * 11) >> 5, s = tap - 3 * r;                 // tap / 3 for tap < 9
    o16[kc] = G < 18 ? (uint32_t)(cg * ZPL16 + (r * ZWP + s) * 16) : 0u;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int tap = (q == 3 ? 4 : 0) + e;
    const int r = (tap * 11) >> 5, s = tap - 3 * r;
    o4[e] = (uint32_t)(2 * ZPL16 + (r * ZWP + s) * 4);                  // (lanes q < 2 read them too -- valid addresses -- and keep their b128)
  }
  o4l = (uint32_t)(2 * ZPL16 + (2 * ZWP + 2) * 4);                      // tap 8
  const bool small4 = q >= 2;                  // chunk 4: lanes q = 2, 3 read the 2-channel plane

  auto load_frags = [&](u32x4* a, int kc, uint32_t pb) {           // pb: window-top-left ring pixel index of fragment 0, this lane
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint32_t px = pb + f * 16;
      if (kc < 4) a[f] = *(const u32x4*)(smem + o16[kc] + px * 16);
      else if (kc == 4) {              // lanes q = 2, 3: four taps' (2-channel) dwords; lanes q = 0, 1: channel groups 16, 17 of tap 8
        const u32x4 big = *(const u32x4*)(smem + o16[4] + px * 16);
        const uint32_t s0 = *(const uint32_t*)(smem + o4[0] + px * 4), s1 = *(const uint32_t*)(smem + o4[1] + px * 4);
        const uint32_t s2 = *(const uint32_t*)(smem + o4[2] + px * 4), s3 = *(const uint32_t*)(smem + o4[3] + px * 4);
        a[f] = u32x4{small4 ? s0 : big.x, small4 ? s1 : big.y, small4 ? s2 : big.z, small4 ? s3 : big.w};
      } else {
        a[f].x = *(const uint32_t*)(smem + o4l + px * 4);             // tap 8's pair (q = 0; the other lane groups and K slots 2..7 meet zero
      }                                                              // weights: the quad keeps the finite activations of two chunks ago)
    }
  };

  auto load_frags_at = [&](u32x4* a, int kc, const uint32_t* pbs) {   // the tail's fragments: one window-top-left ring pixel per fragment
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint32_t px = pbs[f];
      if (kc < 4) a[f] = *(const u32x4*)(smem + o16[kc] + px * 16);
      else if (kc == 4) {
        const u32x4 big = *(const u32x4*)(smem + o16[4] + px * 16);
        const uint32_t s0 = *(const uint32_t*)(smem + o4[0] + px * 4), s1 = *(const uint32_t*)(smem + o4[1] + px * 4);
        const uint32_t s2 = *(const uint32_t*)(smem + o4[2] + px * 4), s3 = *(const uint32_t*)(smem + o4[3] + px * 4);
        a[f] = u32x4{small4 ? s0 : big.x, small4 ? s1 : big.y, small4 ? s2 : big.z, small4 ? s3 : big.w};
      } else {
        a[f].x = *(const uint32_t*)(smem + o4l + px * 4);             // tap 8's pair (q = 0; the other lane groups and K slots 2..7 meet zero
      }                                                              // weights: the quad keeps the finite activations of two chunks ago)
    }
  };

  Z_T0();
  u32x4 hold[4];                               // wave 7: the band's last row, written one band later
#pragma unroll
  for (int f = 0; f < 4; ++f) hold[f] = u32x4{0u, 0u, 0u, 0u};
  const size_t gpix0 = (size_t)b * ZH * ZW;

  Z_MARK(0);                                   // prologue: weights, ring, staging
#pragma unroll 1
  for (int cv = 0; cv < 8; ++cv) {
    float affv = 0.f;
    if (cv + 1 < 8 && tid < 2 * ZAFF) affv = p.aff[(cv + 1) * 2 * ZAFF + tid];
    const float* const sh = sAff + (cv & 1) * 2 * ZAFF + ZAFF + q * 8;
    const bool second = cv & 1;
#pragma unroll 1
    for (int band = 0; band < 8; ++band) {
      const int row = band * 8 + wave;
      const uint32_t pb = (uint32_t)(row * ZWP + x);               // window top-left of fragment 0 (ring coordinates)
      u32x4 rres[4];
      if (second && q < 3) {                                       // residual = previous block's output, from HBM/L2
#pragma unroll
        for (int f = 0; f < 4; ++f) rres[f] = ((const u32x4*)p.out)[(gpix0 + row * ZW + f * 16 + x) * 3 + q];
      }
      f32x4 acc[4][2];                 // start at the folded-BN shift (the scale sits in the packed weights): the first chunk's
      const f32x4 t0 = *(const f32x4*)(sh), t1 = *(const f32x4*)(sh + 4);       // MFMAs take it as their C operand (no copies)
      u32x4 af[2][4];
      load_frags(af[0], 0, pb);
#pragma unroll
      for (int kc = 0; kc < ZKC; ++kc) {
        if (kc + 1 < ZKC) load_frags(af[(kc + 1) & 1], kc + 1, pb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (kc == 0) {
            acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][0]), __builtin_bit_cast(bf16x8, af[0][f]), t0, 0, 0, 0);
            acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][1]), __builtin_bit_cast(bf16x8, af[0][f]), t1, 0, 0, 0);
          } else {
            mma16z(Wf[kc][0], af[kc & 1][f], acc[f][0]);
            mma16z(Wf[kc][1], af[kc & 1][f], acc[f][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (band == 7 && (cv + 1 < 8 || p.tnp)) {                  // next conv's (or the tail's first slab's) fragments behind the last use of this conv's
          const u32x4* const wn = cv + 1 < 8 ? wg + (size_t)(cv + 1) * ZKC * 2 * 64 : (const u32x4*)p.tw;
          Wf[kc][0] = wn[(kc * 2 + 0) * 64 + lane];
          Wf[kc][1] = wn[(kc * 2 + 1) * 64 + lane];
        }
      }
      Z_MARK(1);                       // band: fragment reads + MFMAs
      // ---- epilogue: lane (x, q) holds channels 8q .. 8q+7 of pixel (row, 16 f + x)
      u32x4 v[4];
      if (q < 3) {                     // (+ residual), round, ReLU on the packed pairs (bf16 keeps the sign bit: max(int16, 0))
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          float e[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { e[j] = acc[f][0][j]; e[4 + j] = acc[f][1][j]; }
          if (second) {
            float r8[8];
            Vec16<BF16Tag>::unpack(rres[f], r8);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] += r8[j];
          }
          u32x4 pk = Vec16<BF16Tag>::pack(e);
          pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); pk.z = relu_bf16x2(pk.z); pk.w = relu_bf16x2(pk.w);
          v[f] = pk;
          if (second) ((u32x4*)p.out)[(gpix0 + row * ZW + f * 16 + x) * 3 + q] = v[f];
        }
      }
      Z_MARK(2);                       // band epilogue
      __syncthreads();                 // every wave has read this band's input rows (8 band - 1 .. 8 band + 8)
      Z_MARK(3);                       // band barrier
      // ---- lagged write-back: rows 8 band .. 8 band + 6 now; row 8 band + 7 (wave 7) one band later
      auto put = [&](const u32x4* vv, int r) {
        if (q < 3) {
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const int pc = (r + 1) * ZWP + f * 16 + x + 1;
            if (q < 2) *(u32x4*)(smem + q * ZPL16 + pc * 16) = vv[f];
            else *(uint32_t*)(smem + 2 * ZPL16 + pc * 4) = vv[f].x;
          }
        }
      };
      if (wave != 7) put(v, row);
      else {
        if (band > 0) put(hold, row - 8);
#pragma unroll
        for (int f = 0; f < 4; ++f) hold[f] = v[f];
        if (band == 7) put(v, row);
      }
    }
    if (tid < 2 * ZAFF) sAff[((cv + 1) & 1) * 2 * ZAFF + tid] = affv;
    Z_MARK(4);                         // write-back
    __syncthreads();                   // the conv's output map is complete
    Z_MARK(5);                         // conv barrier
  }

  // ---- tail: 3x3 / stride 2 / pad 1 convs of the finished map (read-only from here on: no barriers).  Output 32 x 32: wave w owns
  //      output rows 4w .. 4w+3 = 8 fragments of 16 pixels, four at a time; a fragment's lanes read input pixels 2 apart, its window's
  //      top-left ring pixel being (2 oy, 2 ox).  Slab sl = combined channels 32 sl .. 32 sl + 31 (two tiles, Wf as in a chain conv).
  if (p.tnp) {
    const int nslab = (p.tnp + 3) >> 2;
#pragma unroll 1
    for (int sl = 0; sl < nslab; ++sl) {
      const int piece = sl * 4 + q;
      const int cvi = piece >= p.tps[2] ? 2 : (piece >= p.tps[1] ? 1 : 0);
      const bool live = piece < p.tnp;
      const f32x4 t0 = live ? *(const f32x4*)(p.tshift + piece * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 t1 = live ? *(const f32x4*)(p.tshift + piece * 8 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      const int cph = p.tcph[cvi];
      const bool relu = p.trelu[cvi] != 0;
      unsigned char* const ob = (unsigned char*)p.tout[cvi] + ((size_t)b * 1024 * cph + (size_t)(piece - p.tps[cvi]) * 8) * 2;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        uint32_t pbt[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int orow = 4 * wave + 2 * half + (f >> 1), xo = 16 * (f & 1) + x;
          pbt[f] = (uint32_t)(2 * orow * ZWP + 2 * xo);
        }
        f32x4 acc[4][2];
        u32x4 af[2][4];
        load_frags_at(af[0], 0, pbt);
#pragma unroll
        for (int kc = 0; kc < ZKC; ++kc) {
          if (kc + 1 < ZKC) load_frags_at(af[(kc + 1) & 1], kc + 1, pbt);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            if (kc == 0) {
              acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][0]), __builtin_bit_cast(bf16x8, af[0][f]), t0, 0, 0, 0);
              acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][1]), __builtin_bit_cast(bf16x8, af[0][f]), t1, 0, 0, 0);
            } else {
              mma16z(Wf[kc][0], af[kc & 1][f], acc[f][0]);
              mma16z(Wf[kc][1], af[kc & 1][f], acc[f][1]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (half == 1 && sl + 1 < nslab) {                         // next slab's fragments behind the last use of this slab's
            const u32x4* const wn = (const u32x4*)p.tw + (size_t)(sl + 1) * ZKC * 2 * 64;
            Wf[kc][0] = wn[(kc * 2 + 0) * 64 + lane];
            Wf[kc][1] = wn[(kc * 2 + 1) * 64 + lane];
          }
        }
        if (live) {
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { e[j] = acc[f][0][j]; e[4 + j] = acc[f][1][j]; }
            u32x4 pk = Vec16<BF16Tag>::pack(e);
            if (relu) { pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); pk.z = relu_bf16x2(pk.z); pk.w = relu_bf16x2(pk.w); }
            const int orow = 4 * wave + 2 * half + (f >> 1), xo = 16 * (f & 1) + x;
            *(u32x4*)(ob + (size_t)(orow * 32 + xo) * cph * 2) = pk;
          }
        }
      }
    }
  }
  Z_MARK(6);                           // tail (stride-2 fuse convs)
  Z_DUMP();
}

// ------------------------------------------------------------------------------------------------ pipelined form (the one launched)
// (-DCP_C0_BAND builds launch hr_chain0_kernel above instead: tools/chain0_pipe_check.py holds the two forms against each other bit for bit;
// measurements in HISTORY.md "Round 5" and profiles/r05_chain0_band_vs_pipelined_pmc.txt)
// The same eight convs as a PIPELINE of waves: wave s runs conv s row by row, conv s + 1 follows two rows behind; four more waves
// (one per SIMD) stage x0 (the fuse sum) a quarter row each and, when the launch has a tail, run its stride-2 fuse convs (one
// 32-channel slab per wave) behind conv 7.  Between two stages sits a ring of rows in LDS (8 deep; x0's 6, the tail's feed 4: 66 rows x 2.4 KB
// instead of the whole 157 KB map), handed over through row counters in LDS -- no workgroup barrier after the prologue: the waves drift
// apart, so one wave's epilogue and LDS traffic run beside another's MFMAs instead of all eight doing the same thing between the
// same two barriers.  The BasicBlock residual is read from the ring two stages back (it never leaves the chip: the band form moves
// 550 MB of HBM traffic per launch for 96 MB of results), and a conv's weight fragments are loaded once per crop.
// Same K order and the same arithmetic per value as the band form: bit-identical outputs (tools/chain0_pipe_check.py).
// Ring row: [plane ch 0-7][66 px][16 B] | [plane ch 8-15][66 px][16 B] | [plane ch 16-17][66 px][4 B]; pixel 0 / 65 = the zero padding.
constexpr int PD0 = 6, PDM = 8, PD8 = 4;                             // ring depths: ring 0 (x0), rings 1 .. 7, ring 8 (conv 7's output, read by the tail)
constexpr int PP1 = ZWP * 16, PP2 = 2 * ZWP * 16;                 // plane offsets inside a ring row
constexpr int PROW = (2 * ZWP * 16 + ZWP * 4 + 15) / 16 * 16;     // 2384
constexpr int PROWS = PD0 + 7 * PDM + PD8;                           // 66 ring rows
constexpr int PZERO = PROWS * PROW;                                   // a row of zeros: image rows -1 and 64
constexpr int PCNT = PZERO + PROW;                                 // 32 uint32 counters: [r] rows written to ring r (1 .. 8), [PC_CONS + s] output rows conv s has
constexpr int PC_CONS = 10, PC_PART = 20, PC_TAIL = 24;             // finished, [PC_PART + k] ring-0 rows staging part k has written, [PC_TAIL + k] tail rows slab k has finished
constexpr int PSHF = PCNT + 128;                                   // float shift[8][32]: the folded-BN shifts (read per row: 8 registers fewer per wave)
constexpr int PLDS = PSHF + 8 * ZAFF * 4;
#ifndef CP_C0_PFD
#define CP_C0_PFD 1
#endif
constexpr int PFD = CP_C0_PFD;                                      // fragment prefetch distance of the conv waves, in K chunks
constexpr int PSW = 4;                                              // staging / tail waves (one per SIMD: a single one made its SIMD the pipeline's slowest)
constexpr int PNW = 8 + PSW, PNT = 64 * PNW;
constexpr uint32_t POOR = 0x100000u;                               // a DS write this far out is discarded (tools/lds_probe/oor.py): predication by address
static_assert(PLDS <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) volatile uint32_t p_cnt_t;     // the counters are read / written with DS instructions (a generic
                                                                          // volatile pointer compiles to FLAT loads: slow, and every LDS wait becomes lgkmcnt(0))
__device__ __forceinline__ int p_peek(const p_cnt_t* c) { return (int)__builtin_amdgcn_readfirstlane(*c); }
// bounded: a lost hand-over must not hang the box -- and must not pass silently either: the wave that gives up ORs
// CP_STATUS_CHAIN0_HANDOVER into the device's status word (the launch's numbers are wrong from here on; the host raises when it looks)
__device__ __forceinline__ void p_wait(const p_cnt_t* c, int need, const Chain0Params& p) {
  const uint32_t limit = p.dbg_spin_limit > 0 ? (uint32_t)p.dbg_spin_limit : (1u << 20);
  uint32_t spin = 0;
  for (; spin < limit; ++spin) {
    if (p_peek(c) >= need) break;
#ifdef CP_C0_SPIN
    __builtin_amdgcn_s_sleep(CP_C0_SPIN);
#else
    __builtin_amdgcn_s_sleep(1);
#endif
  }
  if (spin == limit && p.status && p_peek(c) < need) {
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicOr(p.status, (uint32_t)CP_STATUS_CHAIN0_HANDOVER);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void p_post(p_cnt_t* c0, p_cnt_t* c1, uint32_t v, int lane) {
  // behind the row's LDS writes in program order: the LDS executes one wave's DS instructions in order, so whoever reads the new count
  // reads the row too -- no wait in between (a full lgkmcnt(0) here was 300 of a row's ~3 900 cycles)
  asm volatile("" ::: "memory");
  if (lane == 0) { if (c0) *c0 = v; if (c1) *c1 = v; }
}
__device__ __forceinline__ int p_depth(int r) { return r == 0 ? PD0 : (r == 8 ? PD8 : PDM); }
__device__ __forceinline__ uint32_t p_ring(int r) { return (uint32_t)((r == 0 ? 0 : PD0 + (r - 1) * PDM) * PROW); }          // byte offset of ring r
__device__ __forceinline__ uint32_t p_slot(int r, int y) { return (uint32_t)((r == 0 ? y % PD0 : (r == 8 ? (y & (PD8 - 1)) : (y & (PDM - 1)))) * PROW); }

// per-lane K-group geometry of one wave (hr_chain0_kernel's, with the tap ROW taken out -- it selects the ring row -- and the pixel
// step as a parameter: 1 for the chain convs, 2 for the stride-2 tail).  Tap rows: chunk 0 -> 0; chunk 1 -> 0 (q < 2) / 1; chunk 2 -> 1;
// chunk 3 -> 2; chunk 4: lanes q < 2 read the four dwords of their 16-byte group of tap 8 (row 2), lanes q = 2 / 3 four taps' 2-channel
// pairs (rows 0 0 0 1 / 1 1 2 2); chunk 5 -> tap 8 (row 2).
struct PGeo { uint32_t c16[4], c4[4], c4l, fs16, fs4, fs4l; };
__device__ __forceinline__ PGeo p_geo(int x, int q, int step) {
  PGeo g;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const int G = 4 * kc + q;
    const int tap = G >> 1, cg = G & 1;
    const int sft = tap - 3 * ((tap * 11) >> 5);
    g.c16[kc] = (uint32_t)(cg * PP1 + (sft + step * x) * 16);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int tap = (q == 3 ? 4 : 0) + e;
    const int sft = tap - 3 * ((tap * 11) >> 5);
    g.c4[e] = q < 2 ? (uint32_t)(q * PP1 + (2 + step * x) * 16 + 4 * e) : (uint32_t)(PP2 + (sft + step * x) * 4);
  }
  g.c4l = (uint32_t)(PP2 + (2 + step * x) * 4);
  g.fs16 = (uint32_t)(step * 256);                                   // bytes from one fragment (16 output pixels) to the next: 16-byte planes
  g.fs4l = (uint32_t)(step * 64);                                    // ... the 4-byte plane
  g.fs4 = q < 2 ? g.fs16 : g.fs4l;                                   // ... the plane this lane reads in chunk 4
  return g;
}
// the six fragment-set loads of one output row (NF fragments), ring rows rb[0..2] = the three tap rows
template <int NF>
__device__ __forceinline__ void p_load(u32x4* a, int kc, const unsigned char* smem, const PGeo& g, const uint32_t* rb, bool qlo, bool q3) {
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (kc == 0) a[f] = *(const u32x4*)(smem + rb[0] + g.c16[0] + f * g.fs16);
    else if (kc == 1) a[f] = *(const u32x4*)(smem + (qlo ? rb[0] : rb[1]) + g.c16[1] + f * g.fs16);
    else if (kc == 2) a[f] = *(const u32x4*)(smem + rb[1] + g.c16[2] + f * g.fs16);
    else if (kc == 3) a[f] = *(const u32x4*)(smem + rb[2] + g.c16[3] + f * g.fs16);
    else if (kc == 4) {
      const uint32_t d0 = (qlo ? rb[2] : q3 ? rb[1] : rb[0]) + g.c4[0], d1 = (qlo ? rb[2] : q3 ? rb[1] : rb[0]) + g.c4[1];
      const uint32_t d2 = (qlo || q3 ? rb[2] : rb[0]) + g.c4[2], d3 = (qlo || q3 ? rb[2] : rb[1]) + g.c4[3];
      a[f] = u32x4{*(const uint32_t*)(smem + d0 + f * g.fs4), *(const uint32_t*)(smem + d1 + f * g.fs4),
                   *(const uint32_t*)(smem + d2 + f * g.fs4), *(const uint32_t*)(smem + d3 + f * g.fs4)};
    } else a[f].x = *(const uint32_t*)(smem + rb[2] + g.c4l + f * g.fs4l);     // tap 8's pair (q = 0; the other K slots meet zero weights)
  }
}

// staging wave `part`: pixels [16 part, 16 part + 16) of every x0 row (48 (pixel, 16-byte piece) items on lanes 0 .. 47), and -- parts
// 0 .. nslab - 1 of a launch with a tail -- slab `part` of the stride-2 fuse convs, one output row per two rows conv 7 delivers
template <int NSRC>
__device__ __forceinline__ void p_stage_and_tail(const Chain0Params& p, unsigned char* smem, int lane, int b, int part) {
  p_cnt_t* const cnt = (p_cnt_t*)(smem + PCNT);
  const int x = lane & 15, q = lane >> 4;
  const bool qlo = q < 2, q3 = q == 3;
  const int g = lane % 3, px = (64 / PSW) * part + lane / 3;
  const bool live = lane < 3 * (64 / PSW);
  const int nslab = (p.tnp + 3) >> 2;
  const bool tail = part < nslab;
  // tail: this slab's weights (two tiles) and epilogue constants
  u32x4 Wt[ZKC][2];
  const PGeo tg = p_geo(x, q, 2);
  const int piece = part * 4 + q;
  const int cvi = piece >= p.tps[2] ? 2 : (piece >= p.tps[1] ? 1 : 0);
  const bool tlive = tail && piece < p.tnp;
  f32x4 tt0 = f32x4{0.f, 0.f, 0.f, 0.f}, tt1 = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned char* ob = nullptr;
  int cph = 8;
  bool trelu = false;
  if (tail) {
    const u32x4* const wt = (const u32x4*)p.tw + (size_t)part * ZKC * 2 * 64;
#pragma unroll
    for (int kc = 0; kc < ZKC; ++kc) { Wt[kc][0] = wt[(kc * 2 + 0) * 64 + lane]; Wt[kc][1] = wt[(kc * 2 + 1) * 64 + lane]; }
    if (tlive) { tt0 = *(const f32x4*)(p.tshift + piece * 8); tt1 = *(const f32x4*)(p.tshift + piece * 8 + 4); }
    cph = p.tcph[cvi];
    trelu = p.trelu[cvi] != 0;
    ob = (unsigned char*)p.tout[cvi] + ((size_t)b * 1024 * cph + (size_t)(piece - p.tps[cvi]) * 8) * 2;
  } else {
#pragma unroll
    for (int kc = 0; kc < ZKC; ++kc) { Wt[kc][0] = u32x4{0u, 0u, 0u, 0u}; Wt[kc][1] = u32x4{0u, 0u, 0u, 0u}; }
  }
  int ys = 0, oy = tail ? 0 : ZH / 2;
  for (uint32_t spin = 0; spin < (1u << 22) && (ys < ZH || oy < ZH / 2); ++spin) {
    bool did = false;
    if (ys < ZH && (ys < PD0 || p_peek(cnt + PC_CONS + 1) >= ys - (PD0 - 1))) {          // ring 0's slot is free: conv 1 (its residual reader, the later one) is through with row ys - PD
      u32x4 v[NSRC];
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        const int sh = p.shift[k];
        const size_t o = (((size_t)b * (ZH >> sh) + (ys >> sh)) * (ZW >> sh) + (px >> sh)) * 3 + g;
        v[k] = live ? ((const u32x4*)p.src[k])[o] : u32x4{0u, 0u, 0u, 0u};
      }
      const uint32_t ro = p_slot(0, ys);
      float acc[8], f[8];
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        Vec16<BF16Tag>::unpack(v[k], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (k == 0) ? f[j] : acc[j] + f[j];
      }
      if (p.relu_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
      }
      const u32x4 pk = Vec16<BF16Tag>::pack(acc);
      const uint32_t a16 = live && g < 2 ? ro + (uint32_t)(g * PP1 + (px + 1) * 16) : POOR;
      const uint32_t a4 = live && g == 2 ? ro + (uint32_t)(PP2 + (px + 1) * 4) : POOR;
      *(u32x4*)(smem + a16) = pk;
      *(uint32_t*)(smem + a4) = pk.x;
      p_post(cnt + PC_PART + part, (p_cnt_t*)nullptr, (uint32_t)(ys + 1), lane);
      ++ys;
      did = true;
    }
    if (oy < ZH / 2 && p_peek(cnt + 8) >= (2 * oy + 2 < ZH ? 2 * oy + 2 : ZH)) {      // conv 7's rows 2 oy - 1 .. 2 oy + 1 are in ring 8
      asm volatile("" ::: "memory");
      uint32_t rb[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yy = 2 * oy - 1 + r;
        rb[r] = yy < 0 ? (uint32_t)PZERO : p_ring(8) + p_slot(8, yy);
      }
      f32x4 acc[2][2];
      u32x4 af[2][2];
      p_load<2>(af[0], 0, smem, tg, rb, qlo, q3);
#pragma unroll
      for (int kc = 0; kc < ZKC; ++kc) {
        if (kc + 1 < ZKC) p_load<2>(af[(kc + 1) & 1], kc + 1, smem, tg, rb, qlo, q3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          if (kc == 0) {
            acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wt[0][0]), __builtin_bit_cast(bf16x8, af[0][f]), tt0, 0, 0, 0);
            acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wt[0][1]), __builtin_bit_cast(bf16x8, af[0][f]), tt1, 0, 0, 0);
          } else {
            mma16z(Wt[kc][0], af[kc & 1][f], acc[f][0]);
            mma16z(Wt[kc][1], af[kc & 1][f], acc[f][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tlive) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          float e[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { e[j] = acc[f][0][j]; e[4 + j] = acc[f][1][j]; }
          u32x4 pk = Vec16<BF16Tag>::pack(e);
          if (trelu) { pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); pk.z = relu_bf16x2(pk.z); pk.w = relu_bf16x2(pk.w); }
          *(u32x4*)(ob + (size_t)(oy * 32 + 16 * f + x) * cph * 2) = pk;
        }
      }
      p_post(cnt + PC_TAIL + part, (p_cnt_t*)nullptr, (uint32_t)(oy + 1), lane);
      ++oy;
      did = true;
    }
    if (!did) __builtin_amdgcn_s_sleep(1);
  }
  if ((ys < ZH || oy < ZH / 2) && p.status && lane == 0) atomicOr(p.status, (uint32_t)CP_STATUS_CHAIN0_STAGING);      // ran out of its bound
}

__global__ __launch_bounds__(PNT) void hr_chain0p_kernel(const Chain0Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  p_cnt_t* const cnt = (p_cnt_t*)(smem + PCNT);

  // ---- prologue: zero padding pixels of every ring row, the zero row, the counters, the shifts
  for (int i = tid; i < PROWS * 2; i += PNT) {                     // (ring row, left / right)
    const int rr = i >> 1, px = (i & 1) ? ZWP - 1 : 0;
    unsigned char* const row = smem + rr * PROW;
    *(u32x4*)(row + px * 16) = u32x4{0u, 0u, 0u, 0u};
    *(u32x4*)(row + PP1 + px * 16) = u32x4{0u, 0u, 0u, 0u};
    *(uint32_t*)(row + PP2 + px * 4) = 0u;
  }
  for (int i = tid; i < PROW / 16; i += PNT) *(u32x4*)(smem + PZERO + i * 16) = u32x4{0u, 0u, 0u, 0u};
  if (tid < 32) cnt[tid] = 0u;
  if (tid < 8 * ZAFF) ((float*)(smem + PSHF))[tid] = p.aff[(tid / ZAFF) * 2 * ZAFF + ZAFF + (tid % ZAFF)];
  __syncthreads();

  if (wave >= 8) {                                                  // staging (+ tail) waves
    switch (p.nsrc) {
      case 1: p_stage_and_tail<1>(p, smem, lane, b, wave - 8); break;
      case 2: p_stage_and_tail<2>(p, smem, lane, b, wave - 8); break;
      case 3: p_stage_and_tail<3>(p, smem, lane, b, wave - 8); break;
      default: p_stage_and_tail<4>(p, smem, lane, b, wave - 8); break;
    }
    return;
  }

  // ---- conv wave s: weights in registers for the whole crop.  Waves w and w + 4 share a SIMD: they get convs 2k and 2k + 1, which run a stage
  // latency (most of a row period) apart, so their MFMA phases mostly miss each other -- convs s and s + 4 run four stage latencies = almost
  // exactly a whole number of periods apart and would meet in the matrix pipe every row
#ifdef CP_C0_PLAIN_MAP
  const int s = wave;
#else
  const int s = 2 * (wave & 3) + (wave >> 2);
#endif
  const bool second = s & 1;
  const int nslab = (p.tnp + 3) >> 2;
  const u32x4* const wg = (const u32x4*)p.w + (size_t)s * ZKC * 2 * 64;
  u32x4 Wf[ZKC][2];
#pragma unroll
  for (int kc = 0; kc < ZKC; ++kc)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) Wf[kc][nt] = wg[(kc * 2 + nt) * 64 + lane];
  const float* const sh = (const float*)(smem + PSHF) + s * ZAFF + q * 8;
#ifndef CP_C0_PRIO_ODD
#define CP_C0_PRIO_ODD 2
#endif
#ifndef CP_C0_PRIO_EVEN
#define CP_C0_PRIO_EVEN 1
#endif
  // the second conv of a block carries the residual epilogue: the slower stage of its SIMD pair, and the pipeline runs at its slowest stage;
  // both above the staging / tail waves (priorities 2 / 1 / 0)
  if (second) __builtin_amdgcn_s_setprio(CP_C0_PRIO_ODD);
  else if (CP_C0_PRIO_EVEN) __builtin_amdgcn_s_setprio(CP_C0_PRIO_EVEN);
  const PGeo cg = p_geo(x, q, 1);
  const bool q3 = q == 3, qlo = q < 2;
  uint32_t wb16 = q < 2 ? (uint32_t)(q * PP1 + (1 + x) * 16) : POOR, wb4 = q == 2 ? (uint32_t)(PP2 + (1 + x) * 4) : POOR;
  asm volatile("" : "+v"(wb16), "+v"(wb4));
  const uint32_t rin = p_ring(s), rout = p_ring(s + 1), rrs = p_ring(second ? s - 1 : 0);
  const int dout = p_depth(s + 1);
  const size_t gpix0 = (size_t)b * ZH * ZW;

  Z_T0();
#pragma unroll 1
  for (int y = 0; y < ZH; ++y) {
    const int need = y + 2 < ZH ? y + 2 : ZH;                       // input rows y - 1 .. y + 1 are in ring s
    Z_MARK(0);                         // (loop overhead)
    if (s == 0) {                                                   // ring 0 arrives in PSW parts
#pragma unroll
      for (int k = 0; k < PSW; ++k) p_wait(cnt + PC_PART + k, need, p);
    } else p_wait(cnt + s, need, p);
    Z_MARK(1);                         // waiting for the input rows
    uint32_t rb[3];                                                 // ring rows of the three tap rows (uniform)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y - 1 + r;
      rb[r] = (yy < 0 || yy >= ZH) ? (uint32_t)PZERO : rin + p_slot(s, yy);
    }
    // residual (second conv of a block): the block's input row y from the ring two stages back
    u32x4 rbig[4];
    uint32_t rsm[4];
    if (second) {
      const uint32_t rr = rrs + p_slot(s - 1, y);
      const uint32_t rbg = (qlo ? rr : (uint32_t)PZERO) + (uint32_t)((q & 1) * PP1 + (1 + x) * 16);     // lanes q >= 2: the zero row (channels 18 .. 23 stay zero)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        rbig[f] = *(const u32x4*)(smem + rbg + f * 256);
        rsm[f] = *(const uint32_t*)(smem + rr + PP2 + (1 + x) * 4 + f * 64);
      }
    }
    f32x4 acc[4][2];
    u32x4 af[PFD + 1][4];              // fragment sets PFD chunks ahead of the MFMAs (the phase clock put an LDS round trip per chunk on the row's path)
    const f32x4 t0 = *(const f32x4*)(sh), t1 = *(const f32x4*)(sh + 4);
#pragma unroll
    for (int k = 0; k < PFD; ++k) p_load<4>(af[k], k, smem, cg, rb, qlo, q3);
#pragma unroll
    for (int kc = 0; kc < ZKC; ++kc) {
      if (kc + PFD < ZKC) p_load<4>(af[(kc + PFD) % (PFD + 1)], kc + PFD, smem, cg, rb, qlo, q3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (kc == 0) {
          acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][0]), __builtin_bit_cast(bf16x8, af[0][f]), t0, 0, 0, 0);
          acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][1]), __builtin_bit_cast(bf16x8, af[0][f]), t1, 0, 0, 0);
        } else {
          mma16z(Wf[kc][0], af[kc % (PFD + 1)][f], acc[f][0]);
          mma16z(Wf[kc][1], af[kc % (PFD + 1)][f], acc[f][1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    Z_MARK(2);                         // fragment reads + MFMAs
#ifndef CP_C0_LATE_CONS
    // this row's input rows and its residual row have been READ (the LDS returns a wave's reads in order, the last chunk's were waited for):
    // the stages upstream may have their slots back now, an epilogue earlier than at the end of the row
    p_post(cnt + PC_CONS + s, (p_cnt_t*)nullptr, (uint32_t)(y + 1), lane);
#endif
    // ---- epilogue: lane (x, q) holds channels 8q .. 8q+7 of pixel (y, 16 f + x)
    u32x4 v[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { e[j] = acc[f][0][j]; e[4 + j] = acc[f][1][j]; }
      if (second) {
        const u32x4 raw = u32x4{qlo ? rbig[f].x : rsm[f], rbig[f].y, rbig[f].z, rbig[f].w};      // (q >= 2: .yzw read from the zero row; q = 3's .x is never stored)
        float r8[8];
        Vec16<BF16Tag>::unpack(raw, r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] += r8[j];
      }
      u32x4 pk = Vec16<BF16Tag>::pack(e);
      pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); pk.z = relu_bf16x2(pk.z); pk.w = relu_bf16x2(pk.w);
      v[f] = pk;
    }
    Z_MARK(3);                         // epilogue
    if (s == 7 && q < 3) {
#pragma unroll
      for (int f = 0; f < 4; ++f) ((u32x4*)p.out)[(gpix0 + y * ZW + f * 16 + x) * 3 + q] = v[f];
    }
    if (s < 7 || nslab > 0) {
      // the slot of ring s + 1 still holds row y - (the ring's depth).  Its readers: conv s + 1 (input rows) and, for an even ring, conv s + 2's residual --
      // which trails conv s + 1, so waiting for the later one covers both; ring 8's readers are the tail waves (row j is last used by
      // tail row (j + 1) / 2)
      if (y >= dout) {
        if (s == 7) {
          for (int k = 0; k < nslab; ++k) p_wait(cnt + PC_TAIL + k, (y - dout + 1) / 2 + 1, p);
        } else if (second) p_wait(cnt + PC_CONS + s + 2, y - (dout - 1), p);
        else p_wait(cnt + PC_CONS + s + 1, y - (dout - 2), p);
      }
      Z_MARK(4);                       // waiting for the output slot
      const uint32_t ro = rout + p_slot(s + 1, y);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        *(u32x4*)(smem + ro + wb16 + f * 256) = v[f];
        *(uint32_t*)(smem + ro + wb4 + f * 64) = v[f].x;
      }
#ifdef CP_C0_LATE_CONS
      p_post(cnt + s + 1, cnt + PC_CONS + s, (uint32_t)(y + 1), lane);
    } else p_post(cnt + PC_CONS + s, (p_cnt_t*)nullptr, (uint32_t)(y + 1), lane);
#else
      p_post(cnt + s + 1, (p_cnt_t*)nullptr, (uint32_t)(y + 1), lane);
    }
#endif
    Z_MARK(5);                         // write + post
  }
  Z_DUMP();
}
// [conv][chunk][tile][lane][8 bf16]: lane (row = lane & 15, q = lane >> 4), element e, K group G = 4 kc + q:
//   G < 18: tap G >> 1, input channel 8 (G & 1) + e;   18 <= G < 21: tap 4 (G - 18) + e / 2 (< 9), input channel 16 + (e & 1);   else zero.
// tile row `row` of tile nt is output channel (row >> 2) * 8 + 4 nt + (row & 3).
__global__ void pack_chain0_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, uint16_t* __restrict__ out, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % 8);
  const int lane = (int)((i / 8) % 64);
  const int nt = (int)((i / 512) % 2);
  const int kc = (int)(i / 1024);
  const int row = lane & 15, q = lane >> 4;
  const int G = kc * 4 + q;
  int tap = -1, cin = 0;
  if (G < 18) { tap = G >> 1; cin = (G & 1) * 8 + e; }
  else if (G < 21 && (G - 18) * 4 + (e >> 1) < 9) { tap = (G - 18) * 4 + (e >> 1); cin = 16 + (e & 1); }
  const int n = (row >> 2) * 8 + nt * 4 + (row & 3);
  float v = 0.f;
  if (tap >= 0 && n < ZC && cin < ZC) v = w[((size_t)n * ZC + cin) * 9 + tap] * (scale ? scale[n] : 1.f);   // folded BN scale
  out[i] = (uint16_t)f32_to_bf16_bits(v);
}

// tail weights of ONE conv into the shared blob [slab][chunk][tile][lane][8 bf16] (K order as above): combined channel
// 32 slab + (row >> 2) * 8 + 4 nt + (row & 3) belongs to this conv if it lies in [8 ps, 8 ps + cphys); the rest of the blob is not touched
__global__ void pack_chain0_tail_kernel(const float* __restrict__ w, const float* __restrict__ scale, uint16_t* __restrict__ out, int Cout,
                                        int ps, int cphys, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % 8);
  const int lane = (int)((i / 8) % 64);
  const int nt = (int)((i / 512) % 2);
  const int kc = (int)((i / 1024) % ZKC);
  const int slab = (int)(i / (1024 * ZKC));
  const int row = lane & 15, q = lane >> 4;
  const int G = kc * 4 + q;
  int tap = -1, cin = 0;
  if (G < 18) { tap = G >> 1; cin = (G & 1) * 8 + e; }
  else if (G < 21 && (G - 18) * 4 + (e >> 1) < 9) { tap = (G - 18) * 4 + (e >> 1); cin = 16 + (e & 1); }
  const int c = slab * 32 + (row >> 2) * 8 + nt * 4 + (row & 3) - ps * 8;
  if (c < 0 || c >= cphys) return;
  float v = 0.f;
  if (tap >= 0 && c < Cout && cin < ZC) v = w[((size_t)c * ZC + cin) * 9 + tap] * (scale ? scale[c] : 1.f);
  out[i] = (uint16_t)f32_to_bf16_bits(v);
}

}  // namespace

#ifdef CP_DEBUG_KNOBS
extern "C" int cp_debug_chain0_stamps(unsigned long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(z_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? CP_OK : CP_ERR_HIP;
}
#endif


// entry points used by hr_chain.hip's dispatcher for (C, H, W) = (18, 64, 64)
size_t cp_chain0_conv_bytes() { return ZCONV_W; }
int cp_chain0_aff() { return ZAFF; }

int cp_chain0_pack(hipStream_t st, const float* w, const float* scale, int conv_index, void* blob) {
  const size_t total = ZCONV_W / 2;
  uint16_t* dst = (uint16_t*)((unsigned char*)blob + (size_t)conv_index * ZCONV_W);
  CP_LAUNCH(pack_chain0_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, scale, dst, total);
  return cp_check_launch();
}

int cp_chain0_launch(hipStream_t st, int B, int nsrc, const void* const* srcs, const int32_t* shifts, int relu_in,
                     const void* packed_w, const float* affine, void* out, const CpChainTail* tail) {
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)hr_chain0_kernel, ZLDS));
  Chain0Params p = {};
  for (int k = 0; k < 4; ++k) { p.src[k] = k < nsrc ? srcs[k] : nullptr; p.shift[k] = k < nsrc ? shifts[k] : 0; }
  p.nsrc = nsrc; p.relu_in = relu_in ? 1 : 0; p.w = packed_w; p.aff = affine; p.out = out; p.B = B;
  p.status = cp_status_word(true);      // (nullptr while a capture is under way and the word does not exist yet: the models create it first)
  p.dbg_spin_limit = cp_knob("CP_C0_FORCE_TIMEOUT") ? atoi(cp_knob("CP_C0_FORCE_TIMEOUT")) : 0;
  if (tail) {
    if (tail->nconv < 1 || tail->nconv > 3 || !tail->packed_w || !tail->shift) return CP_ERR_INVALID;
    if (!cp_aligned16(tail->packed_w) || !cp_aligned16(tail->shift)) return CP_ERR_ALIGN;
    int ps = 0;
    for (int i = 0; i < 3; ++i) { p.tps[i] = 1 << 20; p.tcph[i] = 8; p.trelu[i] = 0; p.tout[i] = nullptr; }
    for (int i = 0; i < tail->nconv; ++i) {
      const int cph = tail->out_cphys[i];
      if (cph <= 0 || cph % 8 || tail->Cout[i] <= 0 || tail->Cout[i] > cph || !tail->out[i] || tail->out[i] == out) return CP_ERR_INVALID;
      if (!cp_aligned16(tail->out[i])) return CP_ERR_ALIGN;
      p.tps[i] = ps; p.tcph[i] = cph; p.trelu[i] = tail->relu[i] ? 1 : 0; p.tout[i] = tail->out[i];
      ps += cph / 8;
    }
    if (ps * 8 > ZTAIL_CH) return CP_ERR_INVALID;
    p.tnp = ps; p.tw = tail->packed_w; p.tshift = tail->shift;
  }
#ifndef CP_C0_BAND
  {
    static CpDeviceOnce once_p;
    CP_LDS_ATTR_ONCE(once_p, dev, cp_set_max_lds((const void*)hr_chain0p_kernel, PLDS));
    CP_LAUNCH(hr_chain0p_kernel, dim3((unsigned)B), dim3(PNT), PLDS, st, p);
    return cp_check_launch();
  }
#endif
  CP_LAUNCH(hr_chain0_kernel, dim3((unsigned)B), dim3(512), ZLDS, st, p);
  return cp_check_launch();
}

size_t cp_chain0_tail_bytes() { return (size_t)(ZTAIL_CH / 32) * ZCONV_W; }
int cp_chain0_tail_channels() { return ZTAIL_CH; }

int cp_chain0_tail_pack(hipStream_t st, const float* w, const float* scale, int Cout, int first_piece, int out_cphys, void* blob) {
  if (!w || !blob || Cout <= 0 || out_cphys % 8 || Cout > out_cphys || first_piece < 0 || first_piece * 8 + out_cphys > ZTAIL_CH) return CP_ERR_INVALID;
  if (!cp_aligned16(blob)) return CP_ERR_ALIGN;
  const size_t total = cp_chain0_tail_bytes() / 2;
  CP_LAUNCH(pack_chain0_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, scale, (uint16_t*)blob, Cout, first_piece, out_cphys, total);
  return cp_check_launch();
}
