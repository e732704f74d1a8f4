// One launch per HRNet branch chain (timm HighResolutionModule.branches[j] = 4 x BasicBlock, preceded by the previous
// module's fuse sum): the crop's whole branch map lives in LDS for the 8 convolutions.
//
// Why: the 36/72/144-channel branches ran as 8 launches per module, each a 27 us wave of 128-pixel tiles that was neither
// HBM- nor MFMA- but latency/occupancy-bound (13-23 % of any roof), and every activation round-tripped HBM between the
// convs.  A 32x32x36 / 16x16x72 / 8x8x144 bf16 map is 74 / 37 / 18 KB: it fits in one CU's 160 KB of LDS.  So
//   * one 8-wave workgroup owns CPW whole crops; the map is staged ONCE (summing the up-to-4 fuse-layer terms of the
//     previous module with nearest upsampling + ReLU on the way in: the fuse_sum launch disappears), and leaves once;
//   * the map has a 1-pixel zero ring (no bounds logic in the taps).  W >= 16: one PLANE per 16-byte channel group,
//     [group][ring pixel][16 B], plane pitch = 0 mod 16 slots -- the 16 pixels of a ds_read_b128 fragment are 16
//     consecutive slots and the lane groups q / q+1 (adjacent channel groups) start on the same bank: conflict-free
//     except where a lane pair straddles a tap.  8x8 maps: [pixel][channel] with an odd number of slots per pixel;
//   * K packing ACROSS taps: the GEMM K axis is (tap, channel) flattened with channels padded to 8, not to 32 -- a
//     36-channel conv contracts over 9*40 = 360 -> 12 MFMA chunks instead of 9*64 -> 18.  Lane group q of chunk kc reads the
//     16-byte channel group G = 4 kc + q: tap G / CG at channel group G % CG, i.e. just another LDS offset;
//   * conv1 -> registers -> (barrier) -> written back OVER the input map; the block's residual is read from the map into
//     registers (packed bf16; the 36-channel config parks half of it in spare LDS) before that; conv2 + residual + ReLU
//     -> registers -> back over the map.  One map buffer;
//   * weights stream L2 -> LDS by LDS-DMA (global_load_lds, no registers) in slabs of S chunks, double-buffered under the
//     MFMAs of the previous slab (the 8-conv chain is one continuous slab sequence); every wave reads its weight
//     fragments from LDS, activations are read once per wave and feed NTW tiles.
// Numerics: bf16 storage between convs, fp32 accumulate + fp32 folded-BN epilogue, like the unfused path (the K order
// of the accumulation differs, so results agree to bf16 rounding, not bit for bit).
#include <stdlib.h>

#include "common.h"

// the 64x64x18 branch has its own kernel (hr_chain0.hip: banded in-place update, residual through HBM)
size_t cp_chain0_conv_bytes();
int cp_chain0_aff();
int cp_chain0_pack(hipStream_t st, const float* w, const float* scale, int conv_index, void* blob);
int cp_chain0_launch(hipStream_t st, int B, int nsrc, const void* const* srcs, const int32_t* shifts, int relu_in,
                     const void* packed_w, const float* affine, void* out, const CpChainTail* tail);
size_t cp_chain0_tail_bytes();
int cp_chain0_tail_channels();
int cp_chain0_tail_pack(hipStream_t st, const float* w, const float* scale, int Cout, int first_piece, int out_cphys, void* blob);

namespace {

template <int C_, int H_, int W_, int CPW_, int NWC_, int S_, int RL_>
struct ChainCfg {
  static constexpr int C = C_, H = H_, W = W_, CPW = CPW_, NWC = NWC_, S = S_, RL = RL_;
  static constexpr int NW = 8, NTHR = 512;
  static constexpr int CPHYS = (C + 7) / 8 * 8;
  static constexpr int CG = CPHYS / 8;                    // 16-byte channel groups per pixel
  static constexpr int NT_ALL = (C + 15) / 16;            // 16-channel output tiles
  static constexpr int KC = (9 * CG + 3) / 4;             // 32-deep K chunks, taps packed back to back
  static constexpr int KCP = (KC + S - 1) / S * S;        // padded to whole slabs (zero weights)
  static constexpr int NS = KCP / S;                      // slabs per conv
  static constexpr int HP = H + 2, WP = W + 2;
  static constexpr bool PLANES = W >= 16;
  // PLANES: byte address = group * PLANE + ring pixel * 16;  else: ring pixel * PITCH + group * 16
  static constexpr int PLANE = (HP * WP + 15) / 16 * 16 * 16;
  static constexpr int SLOTS = (CG % 2) ? CG : CG + 1;    // pixel-major: odd number of 16-byte slots per pixel
  static constexpr int PIX = PLANES ? 16 : SLOTS * 16;    // bytes between neighbouring pixels
  static constexpr int GRP = PLANES ? PLANE : 16;         // bytes between neighbouring channel groups
  static constexpr int MAP = PLANES ? CG * PLANE : HP * WP * SLOTS * 16;   // bytes per crop
  static constexpr int FPC = H * W / 16;                  // 16-pixel fragments per crop
  static constexpr int NFRAG = CPW * FPC;
  static constexpr int NWP = NW / NWC;                    // waves along pixels
  static constexpr int MT = NFRAG / NWP;                  // fragments per wave
  static constexpr int PM = MT < 4 ? MT : 4;              // fragments per pipeline stage
  static constexpr int NST = MT / PM;
  static constexpr int NTW = (NT_ALL + NWC - 1) / NWC;    // channel tiles per wave (the last wave group may own fewer)
  static constexpr int SLAB = S * NT_ALL * 1024;          // bytes of one weight slab
  static constexpr int WITER = (SLAB / 16 + NTHR - 1) / NTHR;
  static constexpr int AFF = NT_ALL * 16;                 // floats per scale / shift vector
  static constexpr int AFF_BYTES = 2 * 2 * AFF * 4;       // current + next conv
  static constexpr int RES_LDS = NW * RL * NTW * 64 * 8;  // residual of fragments [0, RL) parked in LDS
  static constexpr int LDS = CPW * MAP + 2 * SLAB + AFF_BYTES + RES_LDS;
  static constexpr size_t CONV_W = (size_t)KCP * NT_ALL * 1024;   // packed bytes per conv
  static_assert(NFRAG % NWP == 0 && MT % PM == 0 && RL <= MT, "fragments must split evenly over the pixel waves");
  static_assert(W >= 16 ? (W % 16 == 0) : (W == 8), "fragment = 16 pixels of a row, or two 8-pixel rows");
  static_assert((SLAB / 16) % 64 == 0, "slab pieces are copied by whole waves");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

using CfgB1 = ChainCfg<36, 32, 32, 1, 1, 3, 4>;
using CfgB2 = ChainCfg<72, 16, 16, 2, 1, 3, 0>;
using CfgB3 = ChainCfg<144, 8, 8, 4, 2, 2, 0>;      // 4 crops per workgroup: a conv's 378 KB of weights are streamed per workgroup
// mid-size batches (40 .. ~190 crops): with 2 / 4 crops per workgroup the low-resolution chains launch 10-48 workgroups on 256 CUs and
// bound the module; fewer crops per workgroup (the same arithmetic per crop, bit-identical) fill the chip instead
using CfgB2m = ChainCfg<72, 16, 16, 1, 1, 3, 0>;
using CfgB3h = ChainCfg<144, 8, 8, 2, 2, 2, 0>;
using CfgB3m = ChainCfg<144, 8, 8, 1, 2, 2, 0>;

// a fuse-layer conv that reads this branch's finished map, run in the launch's tail off the map in LDS (cp_hr_branch_chain_tails):
// kind 0 = the 1x1 conv towards a higher-resolution branch, kind 1 = the first 3x3 / stride-2 conv towards a lower-resolution one
struct ChainTailConv {
  const void* w;          // [chunk][tile][lane][16 B] (pack_chain_tail_kernel), folded-BN scale inside
  const float* shift;     // [16 nt] folded-BN shift, zero beyond Cout
  void* out;              // (B, H >> kind, W >> kind, out_cp) bf16
  int kind, nt, out_cp, relu;
};
struct ChainParams {
  const void* src[4];
  int shift[4];
  int nsrc, relu_in;
  const void* w;          // 8 convs x CONV_W bytes, [conv][chunk][tile][lane][16 B]
  const float* aff;       // [8][2][AFF] folded-BN scale, shift (zero beyond C)
  void* out;              // (B, H, W, CPHYS) bf16
  int B;
  int dbg;                // KNOBS builds only (CP_CHAIN_DBG): 1 = no source loads, 2 = no MFMA loop, 4 = no store, 8 = no weight stream
  int ntail;
  ChainTailConv tc[3];
};

#ifdef CP_DEBUG_KNOBS
#define CHAIN_DBG(bit) (p.dbg & (bit))
#else
#define CHAIN_DBG(bit) 0
#endif

__device__ __forceinline__ void mma16(const u32x4& w, const u32x4& a, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

// interior = [relu](sum of the fuse terms), same arithmetic as fuse_sum_kernel; U pieces per thread in flight per source
template <typename Cfg, int NSRC>
__device__ __forceinline__ void stage_sources(const ChainParams& p, unsigned char* sMap, int tid, int crop0) {
  constexpr int CG = Cfg::CG, H = Cfg::H, W = Cfg::W, U = 4;
  constexpr int TOTAL = Cfg::CPW * H * W * CG;
  for (int i0 = tid; i0 < TOTAL; i0 += U * Cfg::NTHR) {
    u32x4 v[U][NSRC];
    uint32_t dst[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * Cfg::NTHR;
      const int g = i % CG;
      int t = i / CG;
      const int c = t / (H * W);
      t -= c * (H * W);
      const int y = t / W, xx = t - y * W;
      const int b = crop0 + c;
      ok[u] = i < TOTAL;
      dst[u] = (uint32_t)(c * Cfg::MAP + ((y + 1) * Cfg::WP + xx + 1) * Cfg::PIX + g * Cfg::GRP);
      const bool ld = ok[u] && b < p.B && !(CHAIN_DBG(1));
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        const int sh = p.shift[k];
        const size_t o = (((size_t)b * (H >> sh) + (y >> sh)) * (W >> sh) + (xx >> sh)) * CG + g;
        v[u][k] = ld ? ((const u32x4*)p.src[k])[o] : u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
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
      if (ok[u]) *(u32x4*)(sMap + dst[u]) = Vec16<BF16Tag>::pack(acc);
    }
  }
}

// One fuse-layer conv in the chain launch's tail.  The finished map is in LDS (read-only from here on: no barriers).  The eight waves
// split the work 2-D: TG tile groups x 8 / TG fragment groups -- wave w owns channel tiles [NTW tg, NTW tg + NTW) (tg = w % TG) of the
// OUTPUT fragments F = fg + (8 / TG) f (fg = w / TG; a fragment = 16 output pixels).  Weight fragments come straight from L2, one K
// chunk ahead; a wave that owned every tile of a few fragments (the first version) fetched the whole conv's weight -- 60-190 KB per
// wave for the stride-2 convs, for 1-2 MFMAs per fragment loaded: the split is chosen so that a loaded fragment feeds >= 4 MFMAs.
// kind 1: a fragment's lanes read ring pixels two apart -- the window of output pixel (oy, ox) starts at ring (2 oy, 2 ox).
template <typename Cfg, int KIND, int NT, int TG>
__device__ __forceinline__ void chain_tail_conv(const ChainTailConv& t, const unsigned char* sMap, int lane, int wave, int crop0, int B) {
  constexpr int CG = Cfg::CG, H = Cfg::H, W = Cfg::W, WP = Cfg::WP, PIX = Cfg::PIX, GRP = Cfg::GRP, MAP = Cfg::MAP, CPW = Cfg::CPW;
  constexpr int HT = KIND ? H / 2 : H, WT = KIND ? W / 2 : W;
  static_assert(WT >= 16 ? WT % 16 == 0 : WT == 8, "a tail fragment is 16 pixels of an output row, or two 8-pixel rows");
  constexpr int FG = 8 / TG, NTW = (NT + TG - 1) / TG;
  constexpr int FPC = HT * WT / 16, NF = CPW * FPC, FW = (NF + FG - 1) / FG;
  constexpr int KC = KIND ? (9 * CG + 3) / 4 : (CG + 3) / 4;
  constexpr uint32_t CENTER = (uint32_t)((WP + 1) * PIX);
  const int x = lane & 15, q = lane >> 4;
  const int tg = wave % TG, fg = wave / TG;
  const int nt0 = tg * NTW;
  if (nt0 >= NT) return;                                        // a tile group beyond the conv's tiles: this wave has nothing to do
  const int ntn = nt0 + NTW <= NT ? NTW : NT - nt0;             // wave-uniform
  uint32_t base[FW];
  int opix[FW];                                                 // output pixel index inside the crop, -1: no fragment
  int crop[FW];
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    const int F = fg + FG * f;
    const int c = F / FPC, Fc = F - c * FPC;
    int orow, ocol;
    if (WT >= 16) { orow = Fc / (WT / 16); ocol = (Fc - orow * (WT / 16)) * 16 + x; }
    else { orow = Fc * 2 + (x >> 3); ocol = x & 7; }
    const bool live = F < NF;
    crop[f] = c;
    opix[f] = live ? orow * WT + ocol : -1;
    base[f] = live ? (uint32_t)(c * MAP + ((KIND ? 2 * orow : orow) * WP + (KIND ? 2 * ocol : ocol)) * PIX) : 0u;
  }
  auto a_off = [&](int kc) -> uint32_t {
    const int G = kc * 4 + q;
    if (KIND) {
      const int tap = G / CG, cg = G - tap * CG;
      const int r = tap / 3, s_ = tap - 3 * r;
      return G < 9 * CG ? (uint32_t)((r * WP + s_) * PIX + cg * GRP) : 0u;
    }
    return G < CG ? CENTER + (uint32_t)(G * GRP) : CENTER;       // K padding: zero weights, any finite data
  };
  const u32x4* const wt = (const u32x4*)t.w + (size_t)nt0 * 64 + lane;
  f32x4 acc[FW][NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const f32x4 t4 = nt < ntn ? *(const f32x4*)(t.shift + q * 4 * NT + (nt0 + nt) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FW; ++f) acc[f][nt] = t4;
  }
  // weight fragments PD - 1 chunks ahead in a ring of PD register sets (PD = 4 measured +-0 against 2: the tails are bound by the
  // bytes a CU draws from L2, not by the round trips)
  constexpr int PD = 2;
  u32x4 wf[PD][NTW];
#pragma unroll
  for (int pk = 0; pk < PD - 1; ++pk)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
      if (pk < KC && nt < ntn) wf[pk][nt] = wt[(pk * NT + nt) * 64];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    if (kc + PD - 1 < KC) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        if (nt < ntn) wf[(kc + PD - 1) % PD][nt] = wt[((kc + PD - 1) * NT + nt) * 64];
    }
    const uint32_t o = a_off(kc);
    u32x4 a[FW];
#pragma unroll
    for (int f = 0; f < FW; ++f) a[f] = *(const u32x4*)(sMap + base[f] + o);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
      if (nt < ntn) {
#pragma unroll
        for (int f = 0; f < FW; ++f) mma16(wf[kc % PD][nt], a[f], acc[f][nt]);
      }
  }
  // lane (x, q): channels 4 NT q + 4 nt + {0..3} of its pixel: 8-byte stores
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    const int b = crop0 + crop[f];
    if (opix[f] < 0 || b >= B) continue;
    unsigned char* const ob = (unsigned char*)t.out + ((size_t)b * HT * WT + opix[f]) * t.out_cp * 2;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int c0 = q * 4 * NT + (nt0 + nt) * 4;
      if (nt < ntn && c0 < t.out_cp) {
        u32x2 pk;
        pk.x = pack_bf16x2(acc[f][nt][0], acc[f][nt][1]);
        pk.y = pack_bf16x2(acc[f][nt][2], acc[f][nt][3]);
        if (t.relu) { pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); }
        *(u32x2*)(ob + c0 * 2) = pk;
      }
    }
  }
}

// the fuse-layer convs HRNet-W18 hangs on each branch (timm HighResolutionModule.fuse_layers; restated in oracle hr_module): per chain
// configuration the (kind, tiles) pairs that exist -- anything else is refused on the host
template <typename Cfg>
__device__ __forceinline__ void chain_tails(const ChainParams& p, const unsigned char* sMap, int lane, int wave, int crop0) {
#pragma unroll 1
  for (int i = 0; i < p.ntail; ++i) {
    const ChainTailConv t = p.tc[i];
    if constexpr (Cfg::C == 36) {
      // (kind, tiles, tile groups).  Measured (MI355X, eager, per launch at 256 crops): the tails add 4 / 11 / 16 us to the 36-channel
      // chain of a stage-2 / 3 / 4 module, 5 / 22 us to the 72-channel one (stage 3 / 4: the 72 -> 144 conv fetches 2 x 189 KB of
      // weights per workgroup at what one CU draws from L2), 5-10 us to the 144-channel one.  More tile groups (less redundant weight
      // traffic, but 16 fragments per wave) spilled and were slower; deeper weight prefetch (PD) changed nothing.
      if (t.kind == 0) chain_tail_conv<Cfg, 0, 2, 1>(t, sMap, lane, wave, crop0, p.B);
      else if (t.nt == 5) chain_tail_conv<Cfg, 1, 5, 4>(t, sMap, lane, wave, crop0, p.B);
      else chain_tail_conv<Cfg, 1, 3, 4>(t, sMap, lane, wave, crop0, p.B);
    } else if constexpr (Cfg::C == 72) {
      if (t.kind == 1) chain_tail_conv<Cfg, 1, 9, 4>(t, sMap, lane, wave, crop0, p.B);
      else if (t.nt == 3) chain_tail_conv<Cfg, 0, 3, 1>(t, sMap, lane, wave, crop0, p.B);
      else chain_tail_conv<Cfg, 0, 2, 1>(t, sMap, lane, wave, crop0, p.B);
    } else {
      if (t.nt == 5) chain_tail_conv<Cfg, 0, 5, 2>(t, sMap, lane, wave, crop0, p.B);
      else if (t.nt == 3) chain_tail_conv<Cfg, 0, 3, 1>(t, sMap, lane, wave, crop0, p.B);
      else chain_tail_conv<Cfg, 0, 2, 1>(t, sMap, lane, wave, crop0, p.B);
    }
  }
}

template <typename Cfg, bool TAIL = false>
__global__ __launch_bounds__(Cfg::NTHR) void hr_chain_kernel(const ChainParams p) {
  constexpr int CG = Cfg::CG, NT_ALL = Cfg::NT_ALL, S = Cfg::S, NS = Cfg::NS, WP = Cfg::WP, HP = Cfg::HP, H = Cfg::H, W = Cfg::W;
  constexpr int MAP = Cfg::MAP, CPW = Cfg::CPW, MT = Cfg::MT, PM = Cfg::PM, NST = Cfg::NST, NTW = Cfg::NTW, SLAB = Cfg::SLAB, WITER = Cfg::WITER;
  constexpr int CPHYS = Cfg::CPHYS, AFF = Cfg::AFF, FPC = Cfg::FPC, NTHR = Cfg::NTHR, PIX = Cfg::PIX, GRP = Cfg::GRP, RL = Cfg::RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sMap = smem;
  unsigned char* const sW = smem + CPW * MAP;
  float* const sAff = (float*)(sW + 2 * SLAB);                       // [2][2][AFF]: this conv's and the next conv's (scale, shift)
  unsigned char* const sRes = sW + 2 * SLAB + Cfg::AFF_BYTES;        // [wave][f < RL][nt][lane] u32x2

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int wc = wave % Cfg::NWC, wp = wave / Cfg::NWC;
  const int nt0 = wc * NTW;
  constexpr int NTMIN = NT_ALL - (Cfg::NWC - 1) * NTW;               // tiles of the last wave group (every group owns at least these)
  static_assert(NTMIN > 0 && NTMIN <= NTW, "every channel wave group owns at least one tile");
  const int ntn = nt0 + NTW <= NT_ALL ? NTW : NT_ALL - nt0;          // wave-uniform: this wave's tiles are [nt0, nt0 + ntn)
  auto has = [&](int nt) { return nt < NTMIN || nt < ntn; };         // folds to `true` where every group owns tile nt
  const bool active = !CHAIN_DBG(2);
  const int crop0 = blockIdx.x * CPW;

  // ---- weight slabs: L2 -> LDS by LDS-DMA (piece i of the slab lands at byte 16 i), one slab ahead of the MFMAs
  const u32x4* const wg = (const u32x4*)p.w;
  constexpr int TOTAL_SLABS = 8 * NS;
  auto slab_issue = [&](int gs, int buf) {
#pragma unroll
    for (int k = 0; k < WITER; ++k) {
      const int i0 = wave * 64 + NTHR * k;                           // wave-uniform first piece of this wave-instruction
      if (i0 < SLAB / 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)gs * (SLAB / 16) + i0 + lane),
                                         (__attribute__((address_space(3))) void*)(sW + buf * SLAB + i0 * 16), 16, 0, 0);
    }
  };
  if (!CHAIN_DBG(8)) slab_issue(0, 0);

  // ---- stage the map: zero ring (+ the planes' pad pixels), interior = relu(sum of the fuse terms)
  if (Cfg::PLANES) {
    for (int i = tid; i < CPW * CG * (Cfg::PLANE / 16); i += NTHR) {
      const int px = i % (Cfg::PLANE / 16);
      const int ry = px / WP, rx = px - ry * WP;
      if (ry == 0 || ry >= H + 1 || rx == 0 || rx == W + 1) *(u32x4*)(sMap + i * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  } else {
    for (int i = tid; i < CPW * HP * WP * Cfg::SLOTS; i += NTHR) {
      const int px = (i / Cfg::SLOTS) % (HP * WP);
      const int ry = px / WP, rx = px - ry * WP;
      if (ry == 0 || ry == H + 1 || rx == 0 || rx == W + 1) *(u32x4*)(sMap + i * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  switch (p.nsrc) {
    case 1: stage_sources<Cfg, 1>(p, sMap, tid, crop0); break;
    case 2: stage_sources<Cfg, 2>(p, sMap, tid, crop0); break;
    case 3: stage_sources<Cfg, 3>(p, sMap, tid, crop0); break;
    default: stage_sources<Cfg, 4>(p, sMap, tid, crop0); break;
  }
  for (int i = tid; i < 2 * AFF; i += NTHR) sAff[i] = p.aff[i];
  __syncthreads();

  // ---- per-lane fragment geometry: byte offset of the 3x3 window's top-left pixel (ring coordinates), group 0
  uint32_t base[MT];
#pragma unroll
  for (int f = 0; f < MT; ++f) {
    const int F = wp * MT + f;
    const int c = F / FPC, Fc = F - c * FPC;
    int row, col;
    if (W >= 16) { row = Fc / (W / 16); col = (Fc - row * (W / 16)) * 16 + x; }
    else { row = Fc * 2 + (x >> 3); col = x & 7; }
    base[f] = (uint32_t)(c * MAP + (row * WP + col) * PIX);
  }
  constexpr uint32_t CENTER = (uint32_t)((WP + 1) * PIX);
  auto a_off = [&](int kc) -> uint32_t {                            // chunk kc, this lane's 16-byte K group
    const int G = kc * 4 + q;
    const int tap = G / CG, cg = G - tap * CG;
    const int r = tap / 3, s = tap - 3 * r;
    return G < 9 * CG ? (uint32_t)((r * WP + s) * PIX + cg * GRP) : 0u;      // K padding: zero weights, any finite data
  };
  auto c_off = [&](int c0) -> uint32_t { return CENTER + (uint32_t)((c0 >> 3) * GRP + (c0 & 7) * 2); };   // own pixel, channel c0

  f32x4 acc[MT][NTW];
  u32x2 res[MT - RL > 0 ? MT - RL : 1][NTW];
  u32x2* const resL = (u32x2*)(sRes + (size_t)wave * RL * NTW * 512) + lane;      // [f][nt] -> + (f * NTW + nt) * 64
  int gs = 0;
#pragma unroll 1
  for (int cv = 0; cv < 8; ++cv) {
    if ((cv & 1) == 0) {                                              // block input = residual of this BasicBlock
#pragma unroll
      for (int f = 0; f < MT; ++f)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int c0 = q * 4 * NT_ALL + (nt0 + nt) * 4;
          uint32_t co = c_off(c0);
          asm volatile("" : "+v"(co));                                // keep the (f, nt) address table out of registers (no hoisting)
          const u32x2 r = (has(nt) && c0 < CPHYS) ? *(const u32x2*)(sMap + base[f] + co) : u32x2{0u, 0u};
          if (f < RL) resL[(f * NTW + nt) * 64] = r;
          else res[f - RL < 0 ? 0 : f - RL][nt] = r;
        }
    }
    // next conv's affine: the load is in flight during this conv's MFMAs, written to the other sAff half behind them
    static_assert(2 * AFF <= NTHR, "one affine float per thread");
    float affv = 0.f;
    if (cv + 1 < 8 && tid < 2 * AFF) affv = p.aff[(cv + 1) * 2 * AFF + tid];
    // the accumulators start at the folded-BN shift (the scale sits in the packed weights) -- plus, in a block's second
    // conv, the residual: the epilogue is ReLU + pack only (its VALU instructions, 4 issue cycles each, ran 1:1 with the MFMAs)
    {
      const float* const sh = sAff + (cv & 1) * 2 * AFF + AFF;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int c0 = q * 4 * NT_ALL + (nt0 + nt) * 4;
        f32x4 t4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has(nt) && c0 < CPHYS) t4 = *(const f32x4*)(sh + c0);
#pragma unroll
        for (int f = 0; f < MT; ++f) {
          acc[f][nt] = t4;
          if (cv & 1) {
            const u32x2 r = f < RL ? resL[(f * NTW + nt) * 64] : res[f - RL < 0 ? 0 : f - RL][nt];
            acc[f][nt][0] += __uint_as_float(r.x << 16); acc[f][nt][1] += __uint_as_float(r.x & 0xffff0000u);
            acc[f][nt][2] += __uint_as_float(r.y << 16); acc[f][nt][3] += __uint_as_float(r.y & 0xffff0000u);
          }
        }
      }
    }

#pragma unroll 1
    for (int s = 0; s < NS; ++s, ++gs) {
      if (gs + 1 < TOTAL_SLABS && !CHAIN_DBG(8)) slab_issue(gs + 1, (gs + 1) & 1);
      if (active) {
        const unsigned char* const sWb = sW + (gs & 1) * SLAB + nt0 * 1024 + lane * 16;
        const int kc0 = s * S;
        // software pipeline over stages of PM fragments: the next stage's activation fragments (and, at a chunk's first
        // stage, the next chunk's weight fragments) are in flight under the current stage's PM x NTW MFMAs
        u32x4 af[2][PM], wf[2][NTW];
        {
          const uint32_t o = a_off(kc0);
#pragma unroll
          for (int i = 0; i < PM; ++i) af[0][i] = *(const u32x4*)(sMap + base[i] + o);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            if (has(nt)) wf[0][nt] = *(const u32x4*)(sWb + nt * 1024);
        }
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
#pragma unroll
          for (int st = 0; st < NST; ++st) {
            const int cur = (cc * NST + st) & 1;
            const bool last = (cc == S - 1) && (st == NST - 1);
            if (!last) {
              const int ncc = (st == NST - 1) ? cc + 1 : cc, nst = (st == NST - 1) ? 0 : st + 1;
              const uint32_t o = a_off(kc0 + ncc);
#pragma unroll
              for (int i = 0; i < PM; ++i) af[cur ^ 1][i] = *(const u32x4*)(sMap + base[nst * PM + i] + o);
            }
            if (st == 0 && cc + 1 < S) {
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                if (has(nt)) wf[(cc + 1) & 1][nt] = *(const u32x4*)(sWb + ((cc + 1) * NT_ALL + nt) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PM; ++i)
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                if (has(nt)) mma16(wf[cc & 1][nt], af[cur][i], acc[st * PM + i][nt]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();                 // slab gs consumed by everyone; slab gs+1 has landed (the barrier drains the LDS-DMA)
    }

    if (tid < 2 * AFF) sAff[((cv + 1) & 1) * 2 * AFF + tid] = affv;   // read by conv cv+1's epilogue, many barriers from here
    // ---- epilogue in registers, then back over the map (every wave is past its last read of the old map)
    {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int c0 = q * 4 * NT_ALL + (nt0 + nt) * 4;
        if (has(nt) && c0 < CPHYS) {
          uint32_t co = c_off(c0);
          asm volatile("" : "+v"(co));
#pragma unroll
          for (int f = 0; f < MT; ++f) {
            u32x2 pk;                                                  // ReLU on the rounded pair: bf16 keeps the sign bit, so max(int16, 0)
            pk.x = relu_bf16x2(pack_bf16x2(acc[f][nt][0], acc[f][nt][1]));
            pk.y = relu_bf16x2(pack_bf16x2(acc[f][nt][2], acc[f][nt][3]));
            *(u32x2*)(sMap + base[f] + co) = pk;
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- the finished map leaves LDS in full-line order
  if (!CHAIN_DBG(4)) {
    constexpr int TOTAL = CPW * H * W * CG;
#pragma unroll 4
    for (int i = tid; i < TOTAL; i += NTHR) {
      const int g = i % CG;
      int t = i / CG;
      const int c = t / (H * W);
      t -= c * (H * W);
      const int y = t / W, xx = t - y * W;
      const int b = crop0 + c;
      if (b < p.B)
        ((u32x4*)p.out)[(((size_t)b * H + y) * W + xx) * CG + g] = *(const u32x4*)(sMap + c * MAP + ((y + 1) * WP + xx + 1) * PIX + g * GRP);
    }
  }
  // ---- tail: the module's first-level fuse convs fed by this branch, off the map that is still in LDS (the stores above are in flight)
  if constexpr (TAIL) chain_tails<Cfg>(p, sMap, lane, wave, crop0);
}

// tail-conv packing: [chunk kc][tile nt][lane][8 bf16]; K group G = 4 kc + q: kind 1 -> tap G / CG, input channel 8 (G % CG) + e;
// kind 0 -> input channel 8 G + e (G < CG); tile row `row` of tile nt = output channel (row >> 2) * 4 NT + 4 nt + (row & 3)
__global__ void pack_chain_tail_kernel(const float* __restrict__ w, const float* __restrict__ scale, uint16_t* __restrict__ out, int Cin,
                                       int CG, int Cout, int NT, int kind, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % 8);
  const int lane = (int)((i / 8) % 64);
  size_t blk = i / 512;
  const int nt = (int)(blk % NT);
  const int kc = (int)(blk / NT);
  const int row = lane & 15, q = lane >> 4;
  const int G = kc * 4 + q;
  int tap = -1, cin = 0;
  if (kind) { tap = G / CG; cin = (G % CG) * 8 + e; if (tap >= 9) tap = -1; }
  else if (G < CG) { tap = 0; cin = G * 8 + e; }
  const int n = (row >> 2) * 4 * NT + nt * 4 + (row & 3);
  const int KK = kind ? 9 : 1;
  float v = 0.f;
  if (tap >= 0 && cin < Cin && n < Cout) v = w[((size_t)n * Cin + cin) * KK + tap] * (scale ? scale[n] : 1.f);
  out[i] = (uint16_t)f32_to_bf16_bits(v);
}

// packing: [chunk kc][tile nt][lane][16 B]; lane (row = lane & 15, q = lane >> 4), element e: K group G = 4 kc + q ->
// tap G / CG, input channel 8 (G % CG) + e; tile row `row` of tile nt is output channel (row >> 2) * 4 NT_ALL + 4 nt + (row & 3)
// (lane q of the MFMA result then holds 4 NT_ALL consecutive channels of its pixel).
__global__ void pack_chain_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, uint16_t* __restrict__ out, int C,
                                         int CG, int NT_ALL, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % 8);
  const int lane = (int)((i / 8) % 64);
  size_t blk = i / 512;
  const int nt = (int)(blk % NT_ALL);
  const int kc = (int)(blk / NT_ALL);
  const int row = lane & 15, q = lane >> 4;
  const int G = kc * 4 + q;
  const int tap = G / CG, cin = (G % CG) * 8 + e;
  const int n = (row >> 2) * 4 * NT_ALL + nt * 4 + (row & 3);
  float v = 0.f;
  if (tap < 9 && cin < C && n < C) v = w[((size_t)n * C + cin) * 9 + tap] * (scale ? scale[n] : 1.f);   // folded BN scale
  out[i] = (uint16_t)f32_to_bf16_bits(v);
}

constexpr int CHAIN_CPW_FULL_B = 192;       // crops from which the low-resolution chains take their full crops-per-workgroup packing

template <typename Cfg>
int launch_chain(hipStream_t st, const ChainParams& p) {
  static CpDeviceOnce once;                  // per template instance, per device
  const int dev = cp_current_device();
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)hr_chain_kernel<Cfg, false>, Cfg::LDS) &&
                                  cp_set_max_lds((const void*)hr_chain_kernel<Cfg, true>, Cfg::LDS));
  const unsigned grid = (unsigned)((p.B + Cfg::CPW - 1) / Cfg::CPW);
  if (p.ntail > 0) CP_LAUNCH((hr_chain_kernel<Cfg, true>), dim3(grid), dim3(Cfg::NTHR), Cfg::LDS, st, p);
  else CP_LAUNCH((hr_chain_kernel<Cfg, false>), dim3(grid), dim3(Cfg::NTHR), Cfg::LDS, st, p);
  return cp_check_launch();
}

// (kind, tiles) pairs chain_tails() is instantiated for, per chain configuration; -> K chunks of the packed weight, 0 = unsupported
int tail_conv_chunks(int C, int kind, int nt) {
  const int CG = (C + 7) / 8;
  bool ok = false;
  if (C == 36) ok = (kind == 0 && nt == 2) || (kind == 1 && (nt == 5 || nt == 3));
  else if (C == 72) ok = (kind == 0 && (nt == 2 || nt == 3)) || (kind == 1 && nt == 9);
  else if (C == 144) ok = kind == 0 && (nt == 2 || nt == 3 || nt == 5);
  if (!ok) return 0;
  return kind ? (9 * CG + 3) / 4 : (CG + 3) / 4;
}

struct ChainInfo { int C, H, W, KCP, NT_ALL, CG, AFF; size_t conv_w; };
bool chain_info(int C, int H, int W, ChainInfo* o) {
#define CP_CI(CFG) if (C == CFG::C && H == CFG::H && W == CFG::W) { *o = ChainInfo{C, H, W, CFG::KCP, CFG::NT_ALL, CFG::CG, CFG::AFF, CFG::CONV_W}; return true; }
  CP_CI(CfgB1) CP_CI(CfgB2) CP_CI(CfgB3)
#undef CP_CI
  if (C == 18 && H == 64 && W == 64) { *o = ChainInfo{C, H, W, 7, 2, 3, cp_chain0_aff(), cp_chain0_conv_bytes()}; return true; }
  return false;
}

}  // namespace

extern "C" int cp_hr_chain_supported(int C, int H, int W) {
  ChainInfo ci;
  return chain_info(C, H, W, &ci) ? 1 : 0;
}

extern "C" size_t cp_hr_chain_weight_bytes(int C, int H, int W) {
  ChainInfo ci;
  return chain_info(C, H, W, &ci) ? 8 * ci.conv_w : 0;
}

extern "C" int cp_hr_chain_affine_floats(int C, int H, int W) {
  ChainInfo ci;
  return chain_info(C, H, W, &ci) ? ci.AFF : 0;
}

extern "C" int cp_pack_hr_chain_weight(cp_stream_t stream, const float* w, const float* scale, int C, int H, int W, int conv_index,
                                       void* blob) {
  ChainInfo ci;
  if (!w || !blob || conv_index < 0 || conv_index > 7 || !chain_info(C, H, W, &ci)) return CP_ERR_INVALID;
  if (!cp_aligned16(blob)) return CP_ERR_ALIGN;
  if (C == 18) return cp_chain0_pack((hipStream_t)stream, w, scale, conv_index, blob);
  const size_t total = ci.conv_w / 2;
  uint16_t* dst = (uint16_t*)((unsigned char*)blob + (size_t)conv_index * ci.conv_w);
  CP_LAUNCH(pack_chain_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, scale, dst, C, ci.CG, ci.NT_ALL, total);
  return cp_check_launch();
}

extern "C" int cp_hr_chain_tail_supported(int C, int H, int W) { return (C == 18 && H == 64 && W == 64) ? 1 : 0; }
extern "C" size_t cp_hr_chain_tail_weight_bytes(void) { return cp_chain0_tail_bytes(); }
extern "C" int cp_hr_chain_tail_channels(void) { return cp_chain0_tail_channels(); }
extern "C" int cp_pack_hr_chain_tail_weight(cp_stream_t stream, const float* w, const float* scale, int Cout, int first_piece, int out_cphys,
                                            void* blob) {
  return cp_chain0_tail_pack((hipStream_t)stream, w, scale, Cout, first_piece, out_cphys, blob);
}

static int branch_chain(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs, const int32_t* shifts, int relu_in,
                        const void* packed_w, const float* affine, void* out, const CpChainTail* tail, int ntail = 0,
                        const CpChainTailConv* tconvs = nullptr);

// ---- tails of the 36 / 72 / 144-channel chains (cp_hr_branch_chain_tails): any of the first-level fuse convs that read the branch
extern "C" int cp_hr_chain_tailconv_supported(int C, int H, int W, int kind, int Cout) {
  ChainInfo ci;
  if (C == 18 || !chain_info(C, H, W, &ci) || Cout <= 0 || (kind != 0 && kind != 1)) return 0;
  return tail_conv_chunks(C, kind, (Cout + 15) / 16) > 0 ? 1 : 0;
}
extern "C" size_t cp_hr_chain_tailconv_weight_bytes(int C, int H, int W, int kind, int Cout) {
  if (!cp_hr_chain_tailconv_supported(C, H, W, kind, Cout)) return 0;
  const int nt = (Cout + 15) / 16;
  return (size_t)tail_conv_chunks(C, kind, nt) * nt * 1024;
}
extern "C" int cp_pack_hr_chain_tailconv_weight(cp_stream_t stream, const float* w, const float* scale, int C, int H, int W, int kind, int Cout,
                                                void* packed) {
  if (!w || !packed || !cp_hr_chain_tailconv_supported(C, H, W, kind, Cout)) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int nt = (Cout + 15) / 16;
  const size_t total = cp_hr_chain_tailconv_weight_bytes(C, H, W, kind, Cout) / 2;
  CP_LAUNCH(pack_chain_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, scale, (uint16_t*)packed, C,
            (C + 7) / 8, Cout, nt, kind, total);
  return cp_check_launch();
}
extern "C" int cp_hr_branch_chain_tails(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs, const int32_t* shifts,
                                        int relu_in, const void* packed_w, const float* affine, void* out, int ntail, const CpChainTailConv* convs) {
  if (ntail < 1 || ntail > 3 || !convs || C == 18) return CP_ERR_INVALID;
  return branch_chain(stream, B, C, H, W, nsrc, srcs, shifts, relu_in, packed_w, affine, out, nullptr, ntail, convs);
}

extern "C" int cp_hr_branch_chain(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs,
                                  const int32_t* shifts, int relu_in, const void* packed_w, const float* affine, void* out) {
  return branch_chain(stream, B, C, H, W, nsrc, srcs, shifts, relu_in, packed_w, affine, out, nullptr);
}

extern "C" int cp_hr_branch_chain_tail(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs,
                                       const int32_t* shifts, int relu_in, const void* packed_w, const float* affine, void* out,
                                       const CpChainTail* tail) {
  if (!tail || !cp_hr_chain_tail_supported(C, H, W)) return CP_ERR_INVALID;
  return branch_chain(stream, B, C, H, W, nsrc, srcs, shifts, relu_in, packed_w, affine, out, tail);
}

static int branch_chain(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs, const int32_t* shifts, int relu_in,
                        const void* packed_w, const float* affine, void* out, const CpChainTail* tail, int ntail, const CpChainTailConv* tconvs) {
  ChainInfo ci;
  if (B <= 0 || !srcs || !shifts || nsrc < 1 || nsrc > 4 || !packed_w || !affine || !out) return CP_ERR_INVALID;
  if (!chain_info(C, H, W, &ci)) return CP_ERR_INVALID;
  if (!cp_aligned16(packed_w) || !cp_aligned16(affine) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  ChainParams p;
  for (int k = 0; k < 4; ++k) { p.src[k] = nullptr; p.shift[k] = 0; }
  for (int k = 0; k < nsrc; ++k) {
    if (!srcs[k] || !cp_aligned16(srcs[k]) || shifts[k] < 0 || shifts[k] > 3) return CP_ERR_INVALID;
    if (((H >> shifts[k]) << shifts[k]) != H || ((W >> shifts[k]) << shifts[k]) != W) return CP_ERR_INVALID;
    if (srcs[k] == out) return CP_ERR_INVALID;
    p.src[k] = srcs[k]; p.shift[k] = shifts[k];
  }
  if (C == 18) return cp_chain0_launch((hipStream_t)stream, B, nsrc, srcs, shifts, relu_in, packed_w, affine, out, tail);
  p.nsrc = nsrc; p.relu_in = relu_in ? 1 : 0;
  p.w = packed_w; p.aff = affine; p.out = out; p.B = B;
  p.dbg = cp_knob("CP_CHAIN_DBG") ? atoi(cp_knob("CP_CHAIN_DBG")) : 0;
  p.ntail = 0;
  for (int i = 0; i < ntail; ++i) {
    const CpChainTailConv& c = tconvs[i];
    const int nt = (c.Cout + 15) / 16;
    if (!c.packed_w || !c.shift || !c.out || c.out == out || !cp_hr_chain_tailconv_supported(C, H, W, c.kind, c.Cout)) return CP_ERR_INVALID;
    if (c.out_cphys % 8 || c.Cout > c.out_cphys || c.out_cphys > nt * 16) return CP_ERR_INVALID;
    if (!cp_aligned16(c.packed_w) || !cp_aligned16(c.shift) || !cp_aligned16(c.out)) return CP_ERR_ALIGN;
    p.tc[i].w = c.packed_w; p.tc[i].shift = c.shift; p.tc[i].out = c.out;
    p.tc[i].kind = c.kind; p.tc[i].nt = nt; p.tc[i].out_cp = c.out_cphys; p.tc[i].relu = c.relu ? 1 : 0;
  }
  p.ntail = ntail;
  hipStream_t st = (hipStream_t)stream;
  if (C == CfgB1::C) return launch_chain<CfgB1>(st, p);
  if (C == CfgB2::C) return B >= CHAIN_CPW_FULL_B ? launch_chain<CfgB2>(st, p) : launch_chain<CfgB2m>(st, p);
  if (B >= CHAIN_CPW_FULL_B) return launch_chain<CfgB3>(st, p);
  return B >= CHAIN_CPW_FULL_B / 2 ? launch_chain<CfgB3h>(st, p) : launch_chain<CfgB3m>(st, p);
}
