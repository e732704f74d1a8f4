// First-level fuse-layer convs of a timm HighResolutionModule, grouped BY SOURCE BRANCH: every conv that reads branch j's
// output y_j -- the 1x1 conv + BN towards each higher-resolution branch i < j, and the first 3x3 / stride-2 conv + BN (+ ReLU
// when the chain goes on) towards each lower-resolution branch i > j -- in ONE launch (timm HighResolutionModule.fuse_layers,
// restated in oracle/checkerpose_oracle.py hr_module).
//
// As separate implicit-GEMM launches these were 10 of a stage-4 module's 16 tiny convs: 18..144 channels, 3 GFLOP and 50 MB
// each at batch 256, 15-65 us apiece because a 48-byte pixel read at stride 2 in fragment shape wastes the texture path, and
// the module's critical path was three of them in a row.  Here a workgroup stages a band of y_j (all channels, 1-pixel top /
// left ring) in LDS with full-row coalesced loads and runs the whole list off it:
//   LDS    : planes [8-channel group][ring pixel][16 B], ring = (band + 1) x (W + 1), band = input rows of this workgroup;
//            inside a ring row the even columns come first, then the odd ones (stride-2 reads become unit-stride)
//   GEMM   : M = output channels (16-row tiles), N = 16 output pixels (linear index inside the band: rows narrower than 16
//            pixels simply wrap), K slots = (tap, channel group), 4 slots per 32-wide MFMA chunk; lane q of chunk kc reads
//            slot 4 kc + q at  plane(cg) + pixel base + tap offset  (offset table in LDS, one per conv kind)
//   weights: [M tile][chunk][lane][16 B] straight from L2 into registers, each fragment feeding up to 4 pixel tiles
//   output : bf16 NHWC, 4 channels (8 bytes) per lane and pixel; channels between Cout and the padded count come out 0
// bf16 only (the per-crop HRNet path, engine.can_chain).
#include "common.h"

namespace {

constexpr int FO_MAXCONV = 4, FO_NTB = 8, FO_MTB = 3, FO_THREADS = 256, FO_SLOTS = 192, FO_WB = 8, FO_SU = 17, FO_AFF = 160;

struct FuseConvDev {
  const void* w; const float* aff; void* out;
  int kind, relu, mt, nchunk, out_cp, aff_n;
};
struct FuseOutParams {
  const void* src;
  int B, H, W, Cg, band, nconv, plane_bytes;
  FuseConvDev c[FO_MAXCONV];
};

__device__ __forceinline__ void mma16f(const u32x4& w, const u32x4& a, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

__global__ __launch_bounds__(FO_THREADS) void hr_fuse_out_kernel(const FuseOutParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  const int nband = p.H / p.band;
  const int b = blockIdx.x / nband, r0 = (blockIdx.x - b * nband) * p.band;
  const int RW = p.W + 1, RH = p.band + 1, Cg = p.Cg;
  // a ring row keeps its even columns first, then its odd ones: the stride-2 convs' 16 lanes (columns 2 ox + s) then read 16
  // CONSECUTIVE 16-byte slots instead of every other one (a 2-way bank conflict on every fragment read)
  const int HE = (p.W >> 1) + 1;                                    // even columns 0, 2, .., W
  auto colslot = [&](int rc) { return (rc & 1) ? HE + (rc >> 1) : (rc >> 1); };
  const int planes_end = Cg * p.plane_bytes;
  uint32_t* const tab = (uint32_t*)(smem + planes_end);            // [2][192 slots]: kind 0 at 0, kind 1 at 192

  // The last wave builds the tables (its lanes have little or nothing to stage: a row of the HRNet maps is 144-192 pieces)
  // while the others already have their staging loads in flight.
  float* const saff = (float*)(tab + 2 * FO_SLOTS);                 // [conv][2][FO_AFF]
  uint32_t* const sdesc = (uint32_t*)(saff + FO_MAXCONV * 2 * FO_AFF);   // [conv][8]: w lo/hi, out lo/hi, kind|relu<<1, mt, nchunk, out_cp
  if (wave == FO_THREADS / 64 - 1) {
    // ---- slot tables: byte offset of K slot s relative to a pixel's base address
    for (int s = lane; s < 2 * FO_SLOTS; s += 64) {
      const int kind = s >= FO_SLOTS, sl = s - kind * FO_SLOTS;
      uint32_t off = 0u;
      if (kind == 0) { if (sl < Cg) off = (uint32_t)(sl * p.plane_bytes); }
      else if (sl < 9 * Cg) {
        const int tap = sl / Cg, cg = sl - tap * Cg;
        const int ts = tap % 3;                                      // column 2 ox + ts: even half at ox (+1 for ts = 2), odd half at ox
        off = (uint32_t)(cg * p.plane_bytes + ((tap / 3) * RW + (ts == 1 ? HE : ts >> 1)) * 16);
      }
      tab[s] = off;
    }
    // ---- per-conv descriptors and folded-BN vectors -> LDS.  The kernel arguments are read HERE, with compile-time indices
    // (preloaded SGPRs): indexing p.c[] by a run-time conv number inside the work loop turns every field access into a scalar
    // load from the kernarg segment -- one round trip per access.
#pragma unroll
    for (int ci = 0; ci < FO_MAXCONV; ++ci) {
      if (ci < p.nconv) {
        const float* const ga = p.c[ci].aff;
        const int an = p.c[ci].aff_n;
        for (int r = lane; r < 2 * FO_AFF; r += 64) {
          const int h = r >= FO_AFF, c = r - h * FO_AFF;
          saff[ci * 2 * FO_AFF + r] = c < an ? ga[h * an + c] : 0.f;
        }
        if (lane == 0) {
          const uint64_t wq = (uint64_t)p.c[ci].w, oq = (uint64_t)p.c[ci].out;
          uint32_t* d = sdesc + ci * 8;
          d[0] = (uint32_t)wq; d[1] = (uint32_t)(wq >> 32); d[2] = (uint32_t)oq; d[3] = (uint32_t)(oq >> 32);
          d[4] = (uint32_t)(p.c[ci].kind | (p.c[ci].relu << 1)); d[5] = (uint32_t)p.c[ci].mt; d[6] = (uint32_t)p.c[ci].nchunk;
          d[7] = (uint32_t)p.c[ci].out_cp;
        }
      }
    }
  }
  const int nconv = p.nconv;
  auto D = [&](int ci, int k) -> int { return __builtin_amdgcn_readfirstlane((int)sdesc[ci * 8 + k]); };
  // ---- stage the band: ring row 0 = input row r0 - 1 (zeros above the image), ring column 0 = zeros.  One 16-byte piece per
  // thread and row, FO_SU rows' loads in flight before the first store.
  {
    const int per_row = p.W * Cg;                                   // 16-byte pieces of one input row, contiguous in memory
    const u32x4* src = (const u32x4*)p.src + ((size_t)b * p.H * p.W) * Cg;
    for (int pc = tid; pc < per_row; pc += FO_THREADS) {
      const int c = pc / Cg, cg = pc - c * Cg;
      unsigned char* dst = smem + cg * p.plane_bytes + colslot(c + 1) * 16;
#pragma unroll 1
      for (int ry0 = 0; ry0 < RH; ry0 += FO_SU) {
        u32x4 v[FO_SU];
#pragma unroll
        for (int u = 0; u < FO_SU; ++u) {
          const int gy = r0 - 1 + ry0 + u;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          if (ry0 + u < RH && gy >= 0) v[u] = src[(size_t)gy * per_row + pc];
        }
#pragma unroll
        for (int u = 0; u < FO_SU; ++u)
          if (ry0 + u < RH) *(u32x4*)(dst + (ry0 + u) * RW * 16) = v[u];
      }
    }
    for (int i = tid; i < RH * Cg; i += FO_THREADS) {
      const int ry = i / Cg, cg = i - ry * Cg;
      *(u32x4*)(smem + cg * p.plane_bytes + (ry * RW) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  __syncthreads();

  // ---- work items of this wave: t = wave, wave + 4, .. over (conv, group of FO_MTB M tiles, group of FO_NTB pixel tiles).  A
  // pixel fragment read from LDS feeds up to FO_MTB MFMAs (the LDS port, not the matrix pipe, bounds this kernel: one MFMA per
  // 1 KB fragment read measured 2x slower); weights come FO_WB chunks x FO_MTB tiles at a time straight into registers.
  // Output pixel index inside the band -> (row, column): a shift when the output width is a power of two (every HRNet map),
  // an integer division (~40 VALU instructions, twice per pixel tile) otherwise.
  auto split = [&](int n, int Wo, int& oy, int& ox) {
    if ((Wo & (Wo - 1)) == 0) { const int lw = 31 - __builtin_clz((unsigned)Wo); oy = n >> lw; ox = n & (Wo - 1); }
    else { oy = n / Wo; ox = n - oy * Wo; }
  };
  auto geom = [&](int ci, int& Wo, int& npx, int& ntile, int& npg, int& nrg) {
    const int kind = D(ci, 4) & 1;
    Wo = kind ? p.W >> 1 : p.W;
    npx = Wo * (kind ? p.band >> 1 : p.band);
    ntile = (npx + 15) >> 4;
    npg = (ntile + FO_NTB - 1) / FO_NTB;
    nrg = (D(ci, 5) + FO_MTB - 1) / FO_MTB;
  };
#pragma unroll 1
  for (int t0 = wave, ci = 0, tbase = 0; ci < nconv; ++ci) {
    int Wo, npx, ntile, npg, nrg;
    geom(ci, Wo, npx, ntile, npg, nrg);
    const int kind = D(ci, 4) & 1, relu = D(ci, 4) >> 1, mtc = D(ci, 5), nch = D(ci, 6), out_cp = D(ci, 7);
    const u32x4* const wbase = (const u32x4*)(((uint64_t)(uint32_t)D(ci, 1) << 32) | (uint32_t)D(ci, 0)) + lane;
    uint16_t* const outp = (uint16_t*)(((uint64_t)(uint32_t)D(ci, 3) << 32) | (uint32_t)D(ci, 2));
    const uint32_t* const tb = tab + (kind ? FO_SLOTS : 0);
    const int Ho = kind ? p.H >> 1 : p.H, oy0 = kind ? r0 >> 1 : r0;
    const int nitem = nrg * npg;
#pragma unroll 1
    for (; t0 < tbase + nitem; t0 += 4) {
      const int it = t0 - tbase;
      const int rg = it / npg, pg = it - rg * npg;
      const int mt0 = rg * FO_MTB, nm = mtc - mt0 < FO_MTB ? mtc - mt0 : FO_MTB;
      const int tlim = ntile - pg * FO_NTB;                         // pixel tiles of this group that exist
      uint32_t base[FO_NTB];
      f32x4 acc[FO_MTB][FO_NTB];
#pragma unroll
      for (int i = 0; i < FO_NTB; ++i) {
        int n = (pg * FO_NTB + i) * 16 + x;
        if (n >= npx) n = 0;                                        // in-bounds address, result dropped
        int oy, ox;
        split(n, Wo, oy, ox);
        base[i] = kind ? (uint32_t)(((2 * oy) * RW + ox) * 16) : (uint32_t)(((oy + 1) * RW + colslot(ox + 1)) * 16);
#pragma unroll
        for (int m = 0; m < FO_MTB; ++m) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 1
      for (int kb = 0; kb < nch; kb += FO_WB) {
        u32x4 wf[FO_MTB][FO_WB];
#pragma unroll
        for (int m = 0; m < FO_MTB; ++m)
#pragma unroll
          for (int k = 0; k < FO_WB; ++k) {
            wf[m][k] = u32x4{0u, 0u, 0u, 0u};
            if (m < nm && kb + k < nch) wf[m][k] = wbase[(size_t)((mt0 + m) * nch + kb + k) * 64];
          }
#pragma unroll
        for (int k = 0; k < FO_WB; ++k) {
          if (kb + k < nch) {
            const uint32_t so = tb[(kb + k) * 4 + q];
#pragma unroll
            for (int i4 = 0; i4 < FO_NTB; i4 += 4) {
              if (i4 < tlim) {                                      // tiles past the end inside a block of 4 read pixel 0: dropped
                u32x4 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *(const u32x4*)(smem + base[i4 + i] + so);
#pragma unroll
                for (int m = 0; m < FO_MTB; ++m)
                  if (m < nm) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) mma16f(wf[m][k], a[i], acc[m][i4 + i]);
                  }
              }
            }
          }
        }
      }
      // ---- epilogue: lane (x, q): pixel (pg * FO_NTB + i) * 16 + x, channels (mt0 + m) * 16 + 4 q .. + 3
#pragma unroll
      for (int m = 0; m < FO_MTB; ++m) {
        const int co = (mt0 + m) * 16 + 4 * q;
        if (m < nm && co < out_cp) {
          const f32x4 sc = *(const f32x4*)(saff + ci * 2 * FO_AFF + co), sh = *(const f32x4*)(saff + ci * 2 * FO_AFF + FO_AFF + co);
#pragma unroll
          for (int i = 0; i < FO_NTB; ++i) {
            const int n = (pg * FO_NTB + i) * 16 + x;
            if (i < tlim && n < npx) {
              int oy, ox;
              split(n, Wo, oy, ox);
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v[j] = acc[m][i][j] * sc[j] + sh[j];
                if (relu) v[j] = fmaxf(v[j], 0.f);
              }
              u32x2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
              *(u32x2*)(outp + ((size_t)(b * Ho + oy0 + oy) * Wo + ox) * out_cp + co) = pk;
            }
          }
        }
      }
    }
    tbase += nitem;
  }
}

// [M tile][chunk][lane][8 bf16]: lane (row = lane & 15, q = lane >> 4) element e = w[co = 16 mt + row][ci = 8 cg + e][tap],
// K slot 4 chunk + q = tap * Cg + cg (kind 1: 9 taps; kind 0: tap 0 only); zero beyond Cout / Cin / the slot count.
__global__ void pack_fuse_out_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin, int kind,
                                            int Cg, int nchunk, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7);
  const int lane = (int)((i >> 3) & 63);
  const size_t f = i >> 9;
  const int kc = (int)(f % nchunk), mt = (int)(f / nchunk);
  const int row = lane & 15, q = lane >> 4;
  const int co = mt * 16 + row, slot = kc * 4 + q;
  const int taps = kind ? 9 : 1;
  float v = 0.f;
  if (slot < taps * Cg) {
    const int tap = slot / Cg, cg = slot - tap * Cg;
    const int ci = cg * 8 + e;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
  }
  out[i] = (uint16_t)f32_to_bf16_bits(v);
}

inline int fo_nchunk(int cin_phys, int kind) { return ((kind ? 9 : 1) * (cin_phys / 8) + 3) / 4; }
inline int fo_mt(int out_cphys) { return (out_cphys + 15) / 16; }

}  // namespace

extern "C" int cp_hr_fuse_out_supported(int H, int W, int cin_phys) {
  if (H < 2 || W < 2 || (H & 1) || (W & 1) || cin_phys <= 0 || cin_phys % 8) return 0;
  const int Cg = cin_phys / 8;
  if (9 * Cg + 3 > FO_SLOTS) return 0;                               // slot table
  return (3 * (W + 1) * Cg * 16 + 16 * Cg <= 60 * 1024) ? 1 : 0;   // the smallest band (2 input rows + ring row) must fit
}

extern "C" size_t cp_hr_fuse_out_weight_bytes(int cin_phys, int out_cphys, int kind) {
  return (size_t)fo_mt(out_cphys) * fo_nchunk(cin_phys, kind) * 1024;
}

// 0: a conv with that many (padded) output channels is not supported (more than FO_AFF = 160: e.g. hrnet_w30's 240-channel branch)
extern "C" int cp_hr_fuse_out_affine_floats(int out_cphys) {
  return (out_cphys <= 0 || out_cphys % 8 || out_cphys > FO_AFF) ? 0 : fo_mt(out_cphys) * 16;
}

extern "C" int cp_pack_hr_fuse_out_weight(cp_stream_t stream, const float* w, int Cout, int Cin, int cin_phys, int out_cphys,
                                          int kind, void* packed) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || cin_phys % 8 || Cin > cin_phys || Cout > out_cphys || (kind != 0 && kind != 1))
    return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const int nchunk = fo_nchunk(cin_phys, kind);
  const size_t total = cp_hr_fuse_out_weight_bytes(cin_phys, out_cphys, kind) / 2;
  CP_LAUNCH(pack_fuse_out_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
            (uint16_t*)packed, Cout, Cin, kind, cin_phys / 8, nchunk, total);
  return cp_check_launch();
}

extern "C" int cp_hr_fuse_out(cp_stream_t stream, const void* src, int B, int H, int W, int cin_phys, int nconv,
                              const CpFuseConv* convs) {
  if (!src || !convs || B <= 0 || nconv <= 0 || nconv > FO_MAXCONV) return CP_ERR_INVALID;
  if (!cp_hr_fuse_out_supported(H, W, cin_phys)) return CP_ERR_INVALID;
  if (!cp_aligned16(src)) return CP_ERR_ALIGN;
  FuseOutParams p;
  p.src = src; p.B = B; p.H = H; p.W = W; p.Cg = cin_phys / 8; p.nconv = nconv;
  // band = the most input rows (even, dividing H) whose ring fits ~60 KB: 2-3 workgroups per CU
  int band = 2;
  for (int cand = 2; cand <= H; cand += 2)
    if (H % cand == 0 && (size_t)(cand + 1) * (W + 1) * p.Cg * 16 + 16 * p.Cg <= 60 * 1024) band = cand;
  p.band = band;
  p.plane_bytes = (band + 1) * (W + 1) * 16 + 16;                   // + 16: consecutive planes start one slot apart
  for (int i = 0; i < nconv; ++i) {
    const CpFuseConv& c = convs[i];
    if (!c.packed_w || !c.affine || !c.out || (c.kind != 0 && c.kind != 1) || c.out_cphys <= 0 || c.out_cphys % 8 || c.out_cphys > FO_AFF) return CP_ERR_INVALID;
    if (!cp_aligned16(c.packed_w) || !cp_aligned16(c.affine) || !cp_aligned16(c.out)) return CP_ERR_ALIGN;
    if (c.out == src) return CP_ERR_INVALID;
    FuseConvDev& d = p.c[i];
    d.w = c.packed_w; d.aff = c.affine; d.out = c.out; d.kind = c.kind; d.relu = c.relu ? 1 : 0;
    d.mt = fo_mt(c.out_cphys); d.nchunk = fo_nchunk(cin_phys, c.kind); d.out_cp = c.out_cphys; d.aff_n = d.mt * 16;
  }
  const size_t lds = (size_t)p.Cg * p.plane_bytes + 2 * FO_SLOTS * 4 + (size_t)FO_MAXCONV * 2 * FO_AFF * 4 + FO_MAXCONV * 8 * 4;
  const long long grid = (long long)B * (H / band);
  if (grid >= (1LL << 31)) return CP_ERR_RANGE;
  CP_LAUNCH(hr_fuse_out_kernel, dim3((unsigned)grid), dim3(FO_THREADS), lds, (hipStream_t)stream, p);
  return cp_check_launch();
}
