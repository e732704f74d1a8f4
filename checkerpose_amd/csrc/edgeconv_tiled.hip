// EdgeConv (StaticGraph_module: reference init.py:54-68 == pipeline.py:45-59, LM twin pipeline_lm.py:55-57) for LARGE graphs
// (N = 1024 .. 4096 keypoints; BASELINE config #5 "npt=4096 dense keypoints, stress GNN gather / LDS"): the LDS-staged gather of
// edgeconv_fused.hip, tiled.  A crop's P' table no longer fits in one CU's LDS (4096 rows x 256 ch x 2 B = 2 MB), but kNN
// neighbourhoods are spatially local: the host renumbers the keypoints into compact patches of 512 (graph_sched.tile_schedule),
// and a patch's neighbour rows are its own 512 rows plus a thin rim ("halo", 170-370 rows on the LM objects).  Two launches per
// layer, both one 8-wave workgroup per (crop, patch) with the patch's 512 x rows in REGISTERS (the MFMA B operand):
//
//   edgeconv_ptable_kernel   P' = s * (W1 x) for every row, exactly once, written as packed f16 pairs (common.h: the gather's max is v_pk_maximum3_f16)
//                            in plane-major order [crop][8-channel plane][row][16 B] (the LDS table's own layout, so staging a
//                            patch's rows is a stream of full 1 KB segments and staging its halo rows 16-byte pieces);
//   edgeconv_tiled_kernel    per 32-channel slice: the table (512 own + HPAD halo rows, 4 planes) arrives by LDS-DMA with per-lane
//                            source rows (no registers, no VALU); gather-max over the K neighbours out of LDS (one ds_read_b128 per
//                            neighbour and lane, v_pk_maximum3_f16, neighbour lists as table SLOTS, scheduled against bank conflicts);
//                            Q' = s * ((W2 - W1) x) + t on the MFMA pipe from the register-resident rows while the NEXT slice's
//                            table streams in; out = leaky(max_k P'_j(k) + Q'_i), 16 B per lane.
// HBM / Infinity-Cache traffic per crop and layer at N = 4096, C = C' = 256: x 2 x 2 MB in, keys 2 MB out + 2.8 MB back in, out 2 MB
// (the round-1 path: [P'|Q'] 4 MB out and K = 20 row reads per keypoint = 42 MB through the L2).
#include "common.h"

// knock-out switches of tools/edge_tiled_bench.py (results wrong on purpose): only a `make KNOBS=1` build may define them
#if (defined(ET_NODMA) || defined(ET_NOGATHER) || defined(ET_NOQ)) && !defined(CP_DEBUG_KNOBS)
#error "ET_NODMA / ET_NOGATHER / ET_NOQ are knock-out builds: they need -DCP_DEBUG_KNOBS (make KNOBS=1)"
#endif

namespace {

constexpr int ET_BLK = 512, ET_KMAX = 20;
constexpr int ET_IDX = ET_BLK * ET_KMAX * 2;                // 20 480: neighbour slots, int16
constexpr int ET_AFF = 2 * 256 * 4;                         // scale | shift of the C' <= 256 Q rows

struct EdgeTiledParams {
  const void* x; const void* w; const float* scale; const float* shift;
  const int32_t* halo; const int16_t* nbr; const int32_t* gids; void* ptab; void* out;
  int in_cs, in_coff, out_cs, out_coff, B, N, NB, K, Cout, HPAD;
  float slope;
};


// blockIdx -> (crop, patch): all patches of a crop on ONE XCD (blockIdx % 8), so the halo pieces they share are hits in its L2
__device__ __forceinline__ void crop_patch(int NB, int B, int& b, int& t) {
  const int label = blockIdx.x & 7, jb = blockIdx.x >> 3;
  b = label + 8 * (jb / NB);
  t = jb % NB;
}

// ------------------------------------------------------------------------------------------------ launch 1: the key table
// weights: the P halves of cp_pack_edgeconv_fused_weight's image ([64-channel slice][half][32-deep chunk][tile][lane][8 bf16])
template <int CIN, bool H = false>                            // H: x rows and weights in IEEE half (CP_F16; common.h cp_mma16); the keys are halves either way
__global__ __launch_bounds__(512) void edgeconv_ptable_kernel(const EdgeTiledParams p) {
  cp_f16_saturate_on();                                       // the f16 key pack saturates at +-65504 (common.h)
  constexpr int KC = CIN / 32;
  constexpr int HALF = KC * 4 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sW = smem;                                   // 2 x HALF
  float* const sScale = (float*)(smem + 2 * HALF);                  // [Cout]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  int b, t;
  crop_patch(p.NB, p.B, b, t);
  if (b >= p.B) return;                                             // whole workgroup, before any barrier
  const int nslice = p.Cout / 64;
  const u32x4* const wg = (const u32x4*)p.w;
  auto w_issue = [&](int s) {
    constexpr int PIECES = HALF / 16;
#pragma unroll
    for (int k = 0; k < (PIECES + 511) / 512; ++k) {
      const int i0 = wave * 64 + 512 * k;
      if (i0 < PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)(2 * s) * PIECES + i0 + lane),
                                         (__attribute__((address_space(3))) void*)(sW + (s & 1) * HALF + i0 * 16), 16, 0, 0);
    }
  };
  w_issue(0);
  for (int i = tid; i < p.Cout; i += 512) sScale[i] = p.scale[i];
  u32x4 xa[4][KC];
  const size_t row0 = (size_t)b * p.N + (size_t)t * ET_BLK + wave * 64;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      xa[f][kc] = *(const u32x4*)((const uint16_t*)p.x + (row0 + f * 16 + x) * p.in_cs + p.in_coff + kc * 32 + q * 8);
  __syncthreads();
  unsigned char* const tab = (unsigned char*)p.ptab + (size_t)b * (p.Cout / 8) * p.N * 16;
  for (int s = 0; s < nslice; ++s) {
    if (s + 1 < nslice) w_issue(s + 1);
    f32x4 acc[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const wb = sW + (s & 1) * HALF + lane * 16;
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
    // lane (x, q): keypoint row0 + 16 f + x, channels 64 s + 16 q + 4 nt + reg = planes 8 s + 2 q, 8 s + 2 q + 1
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
      const size_t r = (size_t)t * ET_BLK + wave * 64 + f * 16 + x;
      unsigned char* dst = tab + ((size_t)(8 * s + 2 * q) * p.N + r) * 16;
      *(u32x4*)dst = lo;
      *(u32x4*)(dst + (size_t)p.N * 16) = hi;
    }
    __syncthreads();                                                // next slice's weights have landed, this buffer is free
  }
}

// ------------------------------------------------------------------------------------------------ launch 2: gather + Q'
// weights: Q halves in 32-channel slices, [slice][32-deep chunk][tile 0..1][lane][8 bf16]; tile row r of tile nt of slice s
// = output channel 32 s + (r >> 2) * 8 + 4 nt + (r & 3) of wpq rows [Cout, 2 Cout)  ->  lane (x, q) ends with channels
// 32 s + 8 q + 4 nt + reg
template <int CIN, bool DB, bool H = false>
__global__ __launch_bounds__(512) void edgeconv_tiled_kernel(const EdgeTiledParams p) {
  if constexpr (H) cp_f16_saturate_on();
  constexpr int KC = CIN / 32;
  constexpr int WQ = KC * 2 * 1024;                                 // bytes of one slice of Q weights
  constexpr int NTAB = DB ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = ET_BLK + p.HPAD;
  const int PLANE = T * 16;                                         // a multiple of 256 B: bank slot = row mod 16
  unsigned char* const sP = smem;                                   // NTAB x [4 planes][T rows][16 B]
  unsigned char* const sW = smem + NTAB * 4 * PLANE;                // DB: one WQ buffer; else two
  int16_t* const sIdx = (int16_t*)(sW + (DB ? 1 : 2) * WQ);
  float* const sScale = (float*)((unsigned char*)sIdx + ET_IDX);    // [Cout] then shift [Cout] at + 256
  float* const sShift = sScale + 256;
  int32_t* const sHalo = (int32_t*)((unsigned char*)sScale + ET_AFF);   // [HPAD] table rows of the halo slots (byte offsets: row * 16)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  int b, t;
  crop_patch(p.NB, p.B, b, t);
  if (b >= p.B) return;
  const int nslice = p.Cout / 32;
  const int g = p.gids ? p.gids[b] : 0;
  const int32_t* const halo = p.halo + ((size_t)g * p.NB + t) * p.HPAD;
  const unsigned char* const tab = (const unsigned char*)p.ptab + (size_t)b * (p.Cout / 8) * p.N * 16;

  // slice s holds channels 32 s + 8 q + (0..7) for lane group q: the four q lanes of a keypoint store 64 CONTIGUOUS bytes per slice
  // (two whole 32-byte sectors).  (Round 3 interleaved the slices -- 64 j + 16 q + 8 h -- and held the even slice's 16 bytes in
  // registers until the odd slice completed 32 bytes per lane; the interleaved form without the hold wrote half SECTORS: 1 016 MB of
  // HBM writes per launch for a 537 MB output at B = 256.)
  // table of slice s: plane pl <- global plane 4 s + pl; 64-row chunks: own rows are one contiguous 1 KB segment, halo rows
  // 16-byte pieces at per-lane rows.  (4 planes x (8 + HPAD / 64) chunks, dealt round-robin to the 8 waves.)
  const int nchunk = 8 + p.HPAD / 64;
  auto table_issue = [&](int s) {
    unsigned char* const dstb = sP + (DB ? (s & 1) * 4 * PLANE : 0);
    for (int c = wave; c < 4 * nchunk; c += 8) {
      const int pl = c / nchunk, ch = c - pl * nchunk;
      const int slot = ch * 64 + lane;
      // (halo rows come from the LDS copy: a global load per piece in front of its DMA put ~1 us of latency on every halo piece)
      const int row = ch < 8 ? t * ET_BLK + slot : sHalo[slot - ET_BLK];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tab + ((size_t)(4 * s + pl) * p.N + row) * 16),
                                       (__attribute__((address_space(3))) void*)(dstb + pl * PLANE + ch * 1024), 16, 0, 0);
    }
  };
  const u32x4* const wg = (const u32x4*)p.w;
  auto w_issue = [&](int s) {
    constexpr int PIECES = WQ / 16;
#pragma unroll
    for (int k = 0; k < (PIECES + 511) / 512; ++k) {
      const int i0 = wave * 64 + 512 * k;
      if (i0 < PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)s * PIECES + i0 + lane),
                                         (__attribute__((address_space(3))) void*)(sW + (DB ? 0 : (s & 1) * WQ) + i0 * 16), 16, 0, 0);
    }
  };
  for (int i = tid; i < p.HPAD; i += 512) sHalo[i] = halo[i];
  __syncthreads();
  table_issue(0);
  w_issue(0);
  {
    const int16_t* gi = p.nbr + ((size_t)g * p.NB + t) * ET_BLK * p.K;
    // staged as BYTE offsets into a table plane (slot * 16 < 65536: one shift less per neighbour in the gather)
    if (p.K == ET_KMAX) {                                           // rows already at the LDS pitch: 16-byte copies
      for (int i = tid; i < ET_BLK * ET_KMAX / 8; i += 512) {
        u32x4 v = ((const u32x4*)gi)[i];
        v.x = (v.x & 0x0fff0fffu) << 4; v.y = (v.y & 0x0fff0fffu) << 4; v.z = (v.z & 0x0fff0fffu) << 4; v.w = (v.w & 0x0fff0fffu) << 4;
        ((u32x4*)sIdx)[i] = v;
      }
    } else {
      for (int i = tid; i < ET_BLK * p.K; i += 512) sIdx[(i / p.K) * ET_KMAX + (i % p.K)] = (int16_t)(gi[i] << 4);
    }
  }
  for (int i = tid; i < p.Cout; i += 512) { sScale[i] = p.scale[p.Cout + i]; sShift[i] = p.shift[p.Cout + i]; }
  __syncthreads();                                                  // table(0), weights(0), lists, affine: all landed
  // this wave's 64 x rows -> registers; issued BEHIND the barrier: the first gather needs only the table, the 32 loads per lane
  // (64 B per row and instruction: the texture path's slowest shape) travel under it and are awaited at the mid-slice barrier
  u32x4 xa[4][KC];
  const size_t row0 = (size_t)b * p.N + (size_t)t * ET_BLK + wave * 64;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      xa[f][kc] = *(const u32x4*)((const uint16_t*)p.x + (row0 + f * 16 + x) * p.in_cs + p.in_coff + kc * 32 + q * 8);

  for (int s = 0; s < nslice; ++s) {
    // DB: the NEXT slice's table streams into the other buffer (its readers, gather(s - 1), passed the last barrier) under this
    // whole slice; the single weight buffer was refilled behind the last barrier and is awaited at the mid-slice barrier.
    // !DB (the table pair does not fit): weights double-buffered, the next table is issued behind the gather's barrier.
#ifndef ET_NODMA
    if (DB) { if (s + 1 < nslice) table_issue(s + 1); }
    else
#endif
    if (!DB && s + 1 < nslice) w_issue(s + 1);
    // ---- gather-max over the K neighbours out of the LDS table: lane (x, q) = keypoint x of fragment f, plane q.  k outer,
    // fragments inner: 4 list reads, then 16 table reads in flight per lane (one fragment at a time left the LDS latency bare)
    uint32_t m[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[f][j] = CP_F16X2_NEG_INF;
    {
      const unsigned char* const pq = sP + (DB ? (s & 1) * 4 * PLANE : 0) + q * PLANE;
      const uint16_t* const my = (const uint16_t*)sIdx + (wave * 64 + x) * ET_KMAX;
      auto rd4 = [&](const u32x2& i4, u32x4* r) {                    // the 4 neighbour rows of one list quad
        r[0] = *(const u32x4*)(pq + (i4.x & 0xffffu)); r[1] = *(const u32x4*)(pq + (i4.x >> 16));
        r[2] = *(const u32x4*)(pq + (i4.y & 0xffffu)); r[3] = *(const u32x4*)(pq + (i4.y >> 16));
      };
      auto mx4 = [&](uint32_t* mm, const u32x4* r) { pkmax5x4_f16(mm, r[0], r[1], r[2], r[3]); };
#ifdef ET_NOGATHER                                                    // knock-out builds (tools/edge_tiled_bench.py): results wrong on purpose
      if (false) {
#else
      if (p.K == ET_KMAX) {
#endif
        // software pipeline over 10 half-steps (list quad k / 4, fragment pair): the NEXT half-step's 8 table reads and the list
        // quads of the one after are issued before this half-step's 32 packed maxima -- a wave that issued, waited and only then
        // computed spent 2-3 x its VALU time per slice waiting on the LDS (knock-out: the gather was 46 of the launch's 82 us)
        u32x4 ra[8], rb[8];
        u32x2 ia[2], ib[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) ia[h] = *(const u32x2*)(my + h * 16 * ET_KMAX);
        rd4(ia[0], ra); rd4(ia[1], ra + 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) ib[h] = *(const u32x2*)(my + (2 + h) * 16 * ET_KMAX);
#pragma unroll
        for (int hs = 0; hs < 10; ++hs) {
          u32x4* const cur = (hs & 1) ? rb : ra;
          u32x4* const nxt = (hs & 1) ? ra : rb;
          u32x2* const inx = (hs & 1) ? ia : ib;                      // list quads of half-step hs + 1 (loaded one half-step ago)
          u32x2* const iaf = (hs & 1) ? ib : ia;                      // ... and the ones of hs + 2 go where hs's were
          // (LDS results return in order: the list quads go out BEFORE the table reads, so that waiting for them next half-step
          // does not wait for these 8 reads as well)
          if (hs + 2 < 10) {
            const int k2 = 4 * ((hs + 2) >> 1), f2 = 2 * ((hs + 2) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) iaf[h] = *(const u32x2*)(my + (f2 + h) * 16 * ET_KMAX + k2);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (hs + 1 < 10) { rd4(inx[0], nxt); rd4(inx[1], nxt + 4); }
          __builtin_amdgcn_sched_barrier(0);
          const int f0 = 2 * (hs & 1);
          mx4(m[f0], cur); mx4(m[f0 + 1], cur + 4);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        for (int k = 0; k < p.K; k += 4) {                           // generic K (a multiple of 4): one list quad of all 4 fragments per step
          u32x2 i4[4];
          u32x4 r[16];
#pragma unroll
          for (int f = 0; f < 4; ++f) i4[f] = *(const u32x2*)(my + f * 16 * ET_KMAX + k);
#pragma unroll
          for (int f = 0; f < 4; ++f) rd4(i4[f], r + 4 * f);
#pragma unroll
          for (int f = 0; f < 4; ++f) mx4(m[f], r + 4 * f);
        }
      }
    }
    __syncthreads();                                                // DB: weights(s) landed.  !DB: every gather done, the table is free
#ifndef ET_NODMA
    if (!DB && s + 1 < nslice) table_issue(s + 1);                  // ... and streams in under the Q' GEMM below
#endif
    // ---- Q' = s * ((W2 - W1) x) + t for this slice's 32 channels, from the register-resident rows
    f32x4 acc[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const unsigned char* const wb = sW + (DB ? 0 : (s & 1) * WQ) + lane * 16;
    u32x4 wf[2][2];
    wf[0][0] = *(const u32x4*)wb; wf[0][1] = *(const u32x4*)(wb + 1024);
#ifdef ET_NOQ
    constexpr int KCQ = 1;
#else
    constexpr int KCQ = KC;
#endif
#pragma unroll
    for (int kc = 0; kc < KCQ; ++kc) {
      if (kc + 1 < KC) {
        wf[(kc + 1) & 1][0] = *(const u32x4*)(wb + ((kc + 1) * 2) * 1024);
        wf[(kc + 1) & 1][1] = *(const u32x4*)(wb + ((kc + 1) * 2 + 1) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[f][nt] = cp_mma16<H>(wf[kc & 1][nt], xa[f][kc], acc[f][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int c0 = s * 32 + q * 8;                                   // slice s = channels 32 s .. 32 s + 31, lane group q its 8 q .. 8 q + 7
    const f32x4 s0 = *(const f32x4*)(sScale + c0), s1 = *(const f32x4*)(sScale + c0 + 4);
    const f32x4 t0 = *(const f32x4*)(sShift + c0), t1 = *(const f32x4*)(sShift + c0 + 4);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w2 = m[f][j];                                 // channels 2 j, 2 j + 1 of this lane's 8 (f16 pair)
        const int nt = j >> 1, r = (j & 1) * 2;
        const float sa = nt ? s1[r] : s0[r], sb = nt ? s1[r + 1] : s0[r + 1];
        const float ta = nt ? t1[r] : t0[r], tb = nt ? t1[r + 1] : t0[r + 1];
        const float y0 = f16_lo(w2) + (acc[f][nt][r] * sa + ta);
        const float y1 = f16_hi(w2) + (acc[f][nt][r + 1] * sb + tb);
        v[2 * j] = fmaxf(y0, y0 * p.slope);                            // LeakyReLU, 0 <= slope <= 1 (checked by the entry point)
        v[2 * j + 1] = fmaxf(y1, y1 * p.slope);
      }
      *(u32x4*)((uint16_t*)p.out + (row0 + f * 16 + x) * p.out_cs + p.out_coff + c0) = cp_pack8<H>(v);
    }
    __syncthreads();                                                // DB: table(s + 1) landed, weights(s) / table(s) free.  !DB: both landed
    if (DB && s + 1 < nslice) w_issue(s + 1);                       // awaited at the next mid-slice barrier, a whole gather away
  }
}

#ifdef CP_DEBUG_KNOBS
// phase clock of workgroup 0 (make KNOBS=1 builds only; tools/edge_tiled_stamps.py): per wave, shader-clock cycles summed over the slices
__device__ unsigned long long et_stamps[8][12];
#define ET_T0() unsigned long long et_t = __builtin_amdgcn_s_memtime(), et_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define ET_MARK(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); et_acc[k] += n_ - et_t; et_t = n_; } while (0)
#define ET_DUMP() do { if (blockIdx.x == 0 && lane == 0) for (int k_ = 0; k_ < 12; ++k_) et_stamps[wave][k_] = et_acc[k_]; } while (0)
#else
#define ET_T0() do {} while (0)
#define ET_MARK(k) do {} while (0)
#define ET_DUMP() do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------ launch 2, interleaved form
// The same slice loop with the Q' GEMM INSIDE the gather (both tables of consecutive slices resident): knock-outs of the form above
// put 46 of 82 us on the gather, 15 on the Q' MFMA loop and 15 on the table DMA, one after the other -- gather (LDS port + VALU) and
// Q' (matrix pipe) use different units and do not depend on each other, so each wave now walks its four row fragments as two PAIRS;
// a pair's pass = 10 steps (list quad k4, fragment h), and step st carries: the list read of step st + 2, the 4 table reads of step
// st + 1, the weight fragments of K chunk st + 1, the 4 MFMAs of chunk st (2 fragments x 2 channel tiles) and the 8 packed maxima
// of step st.  A pair's epilogue (scale / shift, + max, LeakyReLU, pack, 32-byte stores on odd slices) follows its pass, so only one
// pair's maxima and accumulators are live (220 VGPRs beside the 128 that hold the wave's x rows).
// Q' weights travel in K HALVES through a ring of three LDS slots (8 KB each at Cin = 256): halves 2 s and 2 s + 1 are resident
// during slice s, half 2 s + 2 streams in from the slice's start and half 2 s + 3 from the moment every wave has passed chunk
// KC / 2 - 1 of its second pass (an LDS-only barrier: the table DMA and the stores stay in flight).  One full barrier per slice.
__device__ __forceinline__ void lds_only_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int CIN, bool H = false>
__global__ __launch_bounds__(512) void edgeconv_tiled2_kernel(const EdgeTiledParams p) {
  if constexpr (H) cp_f16_saturate_on();
  constexpr int KC = CIN / 32;
  constexpr int WQ = KC * 2 * 1024;                                 // bytes of one slice of Q weights
  constexpr int WH = WQ / 2, KH = KC / 2;                           // one K half: bytes, chunks
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = ET_BLK + p.HPAD;
  const int PLANE = T * 16;                                         // a multiple of 256 B: bank slot = row mod 16
  unsigned char* const sP = smem;                                   // 2 x [4 planes][T rows][16 B]
  unsigned char* const sW = smem + 2 * 4 * PLANE;                   // ring of 3 K halves
  int16_t* const sIdx = (int16_t*)(sW + 3 * WH);
  float* const sScale = (float*)((unsigned char*)sIdx + ET_IDX);    // [Cout] then shift [Cout] at + 256
  float* const sShift = sScale + 256;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, q = lane >> 4;
  int b, t;
  crop_patch(p.NB, p.B, b, t);
  if (b >= p.B) return;
  const int nslice = p.Cout / 32;
  const int g = p.gids ? p.gids[b] : 0;
  const int32_t* const halo = p.halo + ((size_t)g * p.NB + t) * p.HPAD;
  const unsigned char* const tab = (const unsigned char*)p.ptab + (size_t)b * (p.Cout / 8) * p.N * 16;

  ET_T0();
  // this wave's 64 x rows -> registers (first: the MFMAs of slice 0 need them; they travel while the lists are staged)
  u32x4 xa[4][KC];
  const size_t row0 = (size_t)b * p.N + (size_t)t * ET_BLK + wave * 64;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      xa[f][kc] = *(const u32x4*)((const uint16_t*)p.x + (row0 + f * 16 + x) * p.in_cs + p.in_coff + kc * 32 + q * 8);

  // table of slice s -> buffer s & 1: 4 planes x (8 own + HPAD / 64 halo) chunks of 64 rows, dealt round-robin to the 8 waves
  // (piece i of this wave = chunk wave + 8 i).  Which rows a piece moves does not depend on the slice, so every lane keeps its
  // source row offsets (bytes inside a plane) in registers and a piece costs one 64-bit add + the DMA; the pieces of slice s + 1 go
  // out ONE PER STEP of slice s's first pass: issued in one burst at the slice's start they took 19 % of the launch (all eight
  // waves stalled on the texture path's queue at once, nothing else running; in-kernel clock, tools/edge_tiled_stamps.py).
  constexpr int NPC = 7;                                            // pieces per wave: ceil(4 * 14 / 8) at HPAD <= 384
  const int nchunk = 8 + p.HPAD / 64;
  uint32_t srcoff[NPC];
  auto piece_rows = [&](const int32_t* hsrc) {
    // every piece reads a halo id UNCONDITIONALLY (own-row pieces: entry 0) and selects afterwards: with the load inside the
    // `ch < 8 ? own : halo` branch each piece became a basic block of its own -- load, vmcnt(0), use -- seven global round trips
    // one after the other at the head of every workgroup (the "prologue" of the in-kernel clock)
    int hv[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int c = wave + 8 * i;
      const int ch = c % nchunk, slot = ch * 64 + lane;
      hv[i] = hsrc[(c < 4 * nchunk && ch >= 8) ? slot - ET_BLK : 0];
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int c = wave + 8 * i;
      const int ch = c % nchunk, slot = ch * 64 + lane;
      srcoff[i] = c < 4 * nchunk ? (uint32_t)(ch < 8 ? t * ET_BLK + slot : hv[i]) * 16u : 0u;
    }
  };
  auto table_piece = [&](int s, int i) {
    const int c = wave + 8 * i;
    if (c < 4 * nchunk) {
      const int pl = c / nchunk, ch = c - pl * nchunk;
      uint32_t so = srcoff[i];
      asm volatile("" : "+v"(so));                                   // (keeps `tab + srcoff[i]` from being hoisted out of the slice loop as a 64-bit pair per piece)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tab + (size_t)(4 * s + pl) * p.N * 16 + so),
                                       (__attribute__((address_space(3))) void*)(sP + (s & 1) * 4 * PLANE + pl * PLANE + ch * 1024), 16, 0, 0);
    }
  };
  const u32x4* const wg = (const u32x4*)p.w;
  auto wh_issue = [&](int u) {                                      // K half u = 2 s + kh -> ring slot u % 3
    constexpr int PIECES = WH / 16;                                 // 512 (Cin = 256) / 128 (Cin = 64)
    const int i0 = wave * 64;
    if (i0 < PIECES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)u * PIECES + i0 + lane),
                                       (__attribute__((address_space(3))) void*)(sW + (u % 3) * WH + i0 * 16), 16, 0, 0);
  };
  piece_rows(halo);                                                  // (halo row ids straight from global memory: one round trip)
#pragma unroll
  for (int i = 0; i < NPC; ++i) table_piece(0, i);
  wh_issue(0);
  wh_issue(1);
  {
    // K == ET_KMAX (checked by the entry point): 16-byte copies, staged as BYTE offsets into a table plane.  All of a thread's pieces are
    // loaded before the first is written (the copy loop compiled to load - vmcnt(0) - write per piece, and -- from an int16_t pointer --
    // to four dword loads per piece: three to six global round trips in a row at the head of every workgroup)
    const u32x4* const gi4 = (const u32x4*)__builtin_assume_aligned(p.nbr + ((size_t)g * p.NB + t) * ET_BLK * p.K, 16);
    constexpr int NLP = (ET_BLK * ET_KMAX / 8 + 511) / 512;
    u32x4 lv[NLP];
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      const int i = tid + 512 * k;
      lv[k] = gi4[i < ET_BLK * ET_KMAX / 8 ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      const int i = tid + 512 * k;
      u32x4 v = lv[k];
      v.x = (v.x & 0x0fff0fffu) << 4; v.y = (v.y & 0x0fff0fffu) << 4; v.z = (v.z & 0x0fff0fffu) << 4; v.w = (v.w & 0x0fff0fffu) << 4;
      if (i < ET_BLK * ET_KMAX / 8) ((u32x4*)sIdx)[i] = v;
    }
  }
  for (int i = tid; i < p.Cout; i += 512) { sScale[i] = p.scale[p.Cout + i]; sShift[i] = p.shift[p.Cout + i]; }
  ET_MARK(0);                                                       // prologue issue
  __syncthreads();                                                  // table(0), halves 0 and 1, lists, affine, x rows: all landed
  ET_MARK(1);                                                       // prologue wait

  const uint16_t* const my = (const uint16_t*)sIdx + (wave * 64 + x) * ET_KMAX;
  for (int s = 0; s < nslice; ++s) {
    if (s + 1 < nslice) wh_issue(2 * s + 2);
    ET_MARK(2);                                                     // DMA issue
    const unsigned char* const pq = sP + (s & 1) * 4 * PLANE + q * PLANE;
    const unsigned char* const wlo = sW + ((2 * s) % 3) * WH + lane * 16;           // chunks [0, KH)
    const unsigned char* const whi = sW + ((2 * s + 1) % 3) * WH + lane * 16;       // chunks [KH, KC)
    const int c0 = s * 32 + q * 8;                                   // slice s = channels 32 s .. 32 s + 31, lane group q its 8 q .. 8 q + 7
    auto rd4 = [&](const u32x2& i4, u32x4* r) {                      // the 4 neighbour rows of one list quad
      r[0] = *(const u32x4*)(pq + (i4.x & 0xffffu)); r[1] = *(const u32x4*)(pq + (i4.x >> 16));
      r[2] = *(const u32x4*)(pq + (i4.y & 0xffffu)); r[3] = *(const u32x4*)(pq + (i4.y >> 16));
    };
    auto wfrag = [&](int kc, u32x4* w2) {                            // the two channel tiles' fragments of K chunk kc
      const unsigned char* const wb = (kc < KH ? wlo + kc * 2048 : whi + (kc - KH) * 2048);
      w2[0] = *(const u32x4*)wb; w2[1] = *(const u32x4*)(wb + 1024);
    };
#pragma unroll
    for (int fp = 0; fp < 2; ++fp) {
      uint32_t m[2][4];
      f32x4 acc[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m[h][j] = CP_F16X2_NEG_INF;
        acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      u32x4 r[2][4], wf[2][2];
      u32x2 il[2];
      // step st = 2 k4 + h: list quad k4 of fragment 2 fp + h.  Two sets of table-read registers: step st sends out the reads of
      // step st + 1 (and the list quad of step st + 2, the weight fragments of chunk st + 1), runs its MFMAs and then folds ITS
      // four rows, which were requested a whole step earlier.
      il[0] = *(const u32x2*)(my + (2 * fp) * 16 * ET_KMAX);
      il[1] = *(const u32x2*)(my + (2 * fp + 1) * 16 * ET_KMAX);
      wfrag(0, wf[0]);
      rd4(il[0], r[0]);
#pragma unroll
      for (int st = 0; st < 10; ++st) {
        const int h = st & 1;
        // (LDS results return in order: the list quad of step st + 2 goes out BEFORE the table reads of step st + 1, the weight
        // fragments of chunk st + 1 behind them)
        u32x2 inext = il[h];
        if (st + 2 < 10) inext = *(const u32x2*)(my + (2 * fp + h) * 16 * ET_KMAX + 4 * ((st + 2) >> 1));
        if (st + 1 < 10) rd4(il[h ^ 1], r[h ^ 1]);
        if (st + 1 < KC) wfrag(st + 1, wf[(st + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (fp == 0 && st < NPC && s + 1 < nslice) table_piece(s + 1, st);
        if (st < KC) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[h2][nt] = cp_mma16<H>(wf[st & 1][nt], xa[2 * fp + h2][st], acc[h2][nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        pkmax5x4_f16(m[h], r[h][0], r[h][1], r[h][2], r[h][3]);
        il[h] = inext;
        __builtin_amdgcn_sched_barrier(0);
        if (fp == 1 && st == (KH > 0 ? KH - 1 : 0)) {
          // every wave is past its last read of K half 2 s: that ring slot takes half 2 s + 3 (LDS-only barrier)
          ET_MARK(5);                                                 // pass 1 up to the mid barrier
          lds_only_barrier();
          ET_MARK(6);                                                 // mid barrier wait
          if (s + 1 < nslice) wh_issue(2 * s + 3);
        }
      }
      if (fp == 0) ET_MARK(3); else ET_MARK(7);                       // pass 0 steps / rest of pass 1
      // ---- epilogue of this fragment pair: 16 bytes per lane and slice, 64 contiguous bytes per keypoint (whole sectors)
      const f32x4 s0 = *(const f32x4*)(sScale + c0), s1 = *(const f32x4*)(sScale + c0 + 4);
      const f32x4 t0 = *(const f32x4*)(sShift + c0), t1 = *(const f32x4*)(sShift + c0 + 4);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int f = 2 * fp + h;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w2 = m[h][j];                               // channels 2 j, 2 j + 1 of this lane's 8 (f16 pair)
          const int nt = j >> 1, rg = (j & 1) * 2;
          const float sa = nt ? s1[rg] : s0[rg], sb = nt ? s1[rg + 1] : s0[rg + 1];
          const float ta = nt ? t1[rg] : t0[rg], tb = nt ? t1[rg + 1] : t0[rg + 1];
          const float y0 = f16_lo(w2) + (acc[h][nt][rg] * sa + ta);
          const float y1 = f16_hi(w2) + (acc[h][nt][rg + 1] * sb + tb);
          v[2 * j] = fmaxf(y0, y0 * p.slope);                          // LeakyReLU, 0 <= slope <= 1 (checked by the entry point)
          v[2 * j + 1] = fmaxf(y1, y1 * p.slope);
        }
        *(u32x4*)((uint16_t*)p.out + (row0 + f * 16 + x) * p.out_cs + p.out_coff + c0) = cp_pack8<H>(v);
      }
      if (fp == 0) ET_MARK(4); else ET_MARK(8);                       // epilogues
    }
    __syncthreads();                                                // table(s + 1), halves 2 s + 2 and 2 s + 3 landed; every gather of slice s done
    ET_MARK(9);                                                     // end-of-slice barrier wait
  }
  ET_DUMP();
}

// Q halves, 32-channel slices (see the kernel's header comment)
__global__ void pack_edgeconv_tiled_q_kernel(const float* __restrict__ wpq, uint16_t* __restrict__ out, int Cin, int Cout, size_t total, int dtype) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int KC = Cin / 32;
  const int e = (int)(i % 8);
  const int lane = (int)((i / 8) % 64);
  size_t blk = i / 512;
  const int nt = (int)(blk % 2); blk /= 2;
  const int kc = (int)(blk % KC);
  const int s = (int)(blk / KC);
  const int r = lane & 15, q = lane >> 4;
  const int c = s * 32 + (r >> 2) * 8 + nt * 4 + (r & 3);
  const int cin = kc * 32 + q * 8 + e;
  out[i] = (uint16_t)f32_to_half_bits(wpq[((size_t)(Cout + c)) * Cin + cin], dtype);
}

// keypoint renumbering at the program's boundary.  rows: out[b, i, :] = in[b, perm[g, i], :] (row_bytes a multiple of 16)
__global__ __launch_bounds__(256) void permute_rows_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ gids, int B, int N, int pieces) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * N * pieces) return;
  const int pc = (int)(i % pieces);
  const size_t r = i / pieces;
  const int n = (int)(r % N), b = (int)(r / N);
  const int g = gids ? gids[b] : 0;
  out[i] = in[((size_t)b * N + perm[(size_t)g * N + n]) * pieces + pc];
}
// columns of (B, R, N) arrays of 4- or 8-byte elements: scatter = 1: out[b, r, perm[g, i]] = in[b, r, i] (internal -> original order),
// scatter = 0: out[b, r, i] = in[b, r, perm[g, i]] (original -> internal)
template <typename T>
__global__ __launch_bounds__(256) void permute_cols_kernel(const T* __restrict__ in, T* __restrict__ out, const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ gids, int B, int R, int N, int scatter) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * R * N) return;
  const int n = (int)(i % N);
  const size_t br = i / N;
  const int b = (int)(br / R);
  const int g = gids ? gids[b] : 0;
  const int pn = perm[(size_t)g * N + n];
  if (scatter) out[br * N + pn] = in[i];
  else out[i] = in[br * N + pn];
}

size_t tiled_lds(int Cin, int HPAD) { return (size_t)4 * (ET_BLK + HPAD) * 16 + (size_t)2 * (Cin / 32) * 2 * 1024 + ET_IDX + ET_AFF + (size_t)HPAD * 4; }
// both tables of consecutive slices resident (the next one streams in a whole slice ahead), one weight buffer
size_t tiled_lds_db(int Cin, int HPAD) { return (size_t)8 * (ET_BLK + HPAD) * 16 + (size_t)(Cin / 32) * 2 * 1024 + ET_IDX + ET_AFF + (size_t)HPAD * 4; }
// interleaved form: both tables + the ring of three weight K halves
size_t tiled2_lds(int Cin, int HPAD) { return (size_t)8 * (ET_BLK + HPAD) * 16 + (size_t)3 * (Cin / 32) * 1024 + ET_IDX + ET_AFF; }

}  // namespace

#ifdef CP_DEBUG_KNOBS
extern "C" int cp_debug_edge_tiled_stamps(unsigned long long* out96) {      // 8 waves x 12 phase sums of workgroup 0's last launch
  return hipMemcpyFromSymbol(out96, HIP_SYMBOL(et_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? CP_OK : CP_ERR_HIP;
}
#endif

extern "C" int cp_edgeconv_tiled_supported(int N, int K, int Cin, int Cout, int HPAD) {
  return (N > ET_BLK && N % ET_BLK == 0 && N <= 32768 && K > 0 && K <= ET_KMAX && K % 4 == 0 && (Cin == 64 || Cin == 256) && Cout >= 64 &&
          Cout % 64 == 0 && Cout <= 256 && HPAD >= 64 && HPAD % 64 == 0 && tiled_lds(Cin, HPAD) <= 160 * 1024) ? 1 : 0;
}

extern "C" size_t cp_edgeconv_tiled_weight_bytes(int Cin, int Cout) { return (size_t)Cout * Cin * 2; }
extern "C" size_t cp_edgeconv_tiled_table_bytes(int B, int N, int Cout) { return (size_t)B * N * Cout * 2; }

extern "C" int cp_pack_edgeconv_tiled_weight_t(cp_stream_t stream, int dtype, const float* wpq, int Cin, int Cout, void* packed) {
  if (!wpq || !packed || !cp_edgeconv_tiled_supported(2 * ET_BLK, 4, Cin, Cout, 64) || (dtype != CP_BF16 && dtype != CP_F16)) return CP_ERR_INVALID;
  if (!cp_aligned16(packed)) return CP_ERR_ALIGN;
  const size_t total = (size_t)Cout * Cin;
  CP_LAUNCH(pack_edgeconv_tiled_q_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wpq, (uint16_t*)packed, Cin, Cout, total, dtype);
  return cp_check_launch();
}

extern "C" int cp_pack_edgeconv_tiled_weight(cp_stream_t stream, const float* wpq, int Cin, int Cout, void* packed) {
  return cp_pack_edgeconv_tiled_weight_t(stream, CP_BF16, wpq, Cin, Cout, packed);
}

extern "C" int cp_edgeconv_tiled_t(cp_stream_t stream, int dtype, const void* x, int in_cstride, int in_coff, const void* packed_w_fused,
                                   const void* packed_w_q, const float* scale, const float* shift, const int32_t* halo,
                                   const int16_t* nbr, const int32_t* graph_ids, void* key_table, void* out, int out_cstride, int out_coff,
                                   int B, int N, int K, int Cin, int Cout, int G, int HPAD, float slope) {
  if (!x || !packed_w_fused || !packed_w_q || !scale || !shift || !halo || !nbr || !key_table || !out || B <= 0 || G <= 0) return CP_ERR_INVALID;
  if (dtype != CP_BF16 && dtype != CP_F16) return CP_ERR_INVALID;
  if (!cp_edgeconv_tiled_supported(N, K, Cin, Cout, HPAD) || !(slope >= 0.f && slope <= 1.f)) return CP_ERR_INVALID;
  if (in_cstride % 8 || in_coff % 8 || in_coff + Cin > in_cstride || out_cstride % 8 || out_coff % 8 || out_coff + Cout > out_cstride) return CP_ERR_ALIGN;
  if (!cp_aligned16(x) || !cp_aligned16(packed_w_fused) || !cp_aligned16(packed_w_q) || !cp_aligned16(scale) || !cp_aligned16(shift) ||
      !cp_aligned16(out) || !cp_aligned16(key_table) || !cp_aligned16(nbr))
    return CP_ERR_ALIGN;
  const size_t lds1[2] = {(size_t)2 * 2 * 4 * 1024 + 1024, (size_t)2 * 8 * 4 * 1024 + 1024};      // Cin = 64 / 256
  const bool il = K == ET_KMAX && HPAD <= 384 && tiled2_lds(Cin, HPAD) <= 160 * 1024 && !cp_knob("CP_NO_TILED2");      // the interleaved form
  const bool db = tiled_lds_db(Cin, HPAD) <= 160 * 1024;
  const size_t lds2 = il ? tiled2_lds(Cin, HPAD) : db ? tiled_lds_db(Cin, HPAD) : tiled_lds(Cin, HPAD);
  static CpDeviceOnce once;
  const int dev = cp_current_device();
  const size_t want = 160 * 1024;
  CP_LDS_ATTR_ONCE(once, dev, cp_set_max_lds((const void*)edgeconv_ptable_kernel<64>, lds1[0]) &&
                                  cp_set_max_lds((const void*)edgeconv_ptable_kernel<256>, lds1[1]) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<64, false>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<256, false>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<64, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<256, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled2_kernel<64>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled2_kernel<256>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_ptable_kernel<64, true>, lds1[0]) &&
                                  cp_set_max_lds((const void*)edgeconv_ptable_kernel<256, true>, lds1[1]) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<64, false, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<256, false, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<64, true, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled_kernel<256, true, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled2_kernel<64, true>, want) &&
                                  cp_set_max_lds((const void*)edgeconv_tiled2_kernel<256, true>, want));
  EdgeTiledParams p;
  p.x = x; p.scale = scale; p.shift = shift; p.halo = halo; p.nbr = nbr; p.gids = graph_ids; p.ptab = key_table; p.out = out;
  p.in_cs = in_cstride; p.in_coff = in_coff; p.out_cs = out_cstride; p.out_coff = out_coff; p.B = B; p.N = N; p.NB = N / ET_BLK; p.K = K;
  p.Cout = Cout; p.HPAD = HPAD; p.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(((B + 7) / 8) * p.NB * 8);
  const bool h = dtype == CP_F16;
  p.w = packed_w_fused;
#define CP_ET(KERNEL, LDS) CP_LAUNCH((KERNEL), dim3(grid), dim3(512), LDS, st, p)
  if (Cin == 64) { if (h) CP_ET((edgeconv_ptable_kernel<64, true>), lds1[0]); else CP_ET((edgeconv_ptable_kernel<64>), lds1[0]); }
  else { if (h) CP_ET((edgeconv_ptable_kernel<256, true>), lds1[1]); else CP_ET((edgeconv_ptable_kernel<256>), lds1[1]); }
  p.w = packed_w_q;
  if (il && Cin == 64) { if (h) CP_ET((edgeconv_tiled2_kernel<64, true>), lds2); else CP_ET((edgeconv_tiled2_kernel<64>), lds2); }
  else if (il) { if (h) CP_ET((edgeconv_tiled2_kernel<256, true>), lds2); else CP_ET((edgeconv_tiled2_kernel<256>), lds2); }
  else if (Cin == 64 && db) { if (h) CP_ET((edgeconv_tiled_kernel<64, true, true>), lds2); else CP_ET((edgeconv_tiled_kernel<64, true>), lds2); }
  else if (Cin == 64) { if (h) CP_ET((edgeconv_tiled_kernel<64, false, true>), lds2); else CP_ET((edgeconv_tiled_kernel<64, false>), lds2); }
  else if (db) { if (h) CP_ET((edgeconv_tiled_kernel<256, true, true>), lds2); else CP_ET((edgeconv_tiled_kernel<256, true>), lds2); }
  else { if (h) CP_ET((edgeconv_tiled_kernel<256, false, true>), lds2); else CP_ET((edgeconv_tiled_kernel<256, false>), lds2); }
#undef CP_ET
  return cp_check_launch();
}

extern "C" int cp_edgeconv_tiled(cp_stream_t stream, const void* x, int in_cstride, int in_coff, const void* packed_w_fused,
                                 const void* packed_w_q, const float* scale, const float* shift, const int32_t* halo,
                                 const int16_t* nbr, const int32_t* graph_ids, void* key_table, void* out, int out_cstride, int out_coff,
                                 int B, int N, int K, int Cin, int Cout, int G, int HPAD, float slope) {
  return cp_edgeconv_tiled_t(stream, CP_BF16, x, in_cstride, in_coff, packed_w_fused, packed_w_q, scale, shift, halo, nbr, graph_ids, key_table,
                             out, out_cstride, out_coff, B, N, K, Cin, Cout, G, HPAD, slope);
}

extern "C" int cp_permute_rows(cp_stream_t stream, const void* in, void* out, const int32_t* perm, const int32_t* graph_ids,
                               int B, int N, int row_bytes) {
  if (!in || !out || in == out || !perm || B <= 0 || N <= 0 || row_bytes <= 0 || row_bytes % 16) return CP_ERR_INVALID;    // not in place
  if (!cp_aligned16(in) || !cp_aligned16(out)) return CP_ERR_ALIGN;
  const size_t total = (size_t)B * N * (row_bytes / 16);
  CP_LAUNCH(permute_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)in, (u32x4*)out,
            perm, graph_ids, B, N, row_bytes / 16);
  return cp_check_launch();
}

extern "C" int cp_permute_cols(cp_stream_t stream, const void* in, void* out, const int32_t* perm, const int32_t* graph_ids,
                               int B, int R, int N, int elem_bytes, int scatter) {
  if (!in || !out || in == out || !perm || B <= 0 || R <= 0 || N <= 0 || (elem_bytes != 4 && elem_bytes != 8)) return CP_ERR_INVALID;
  const size_t total = (size_t)B * R * N;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (elem_bytes == 4)
    CP_LAUNCH(permute_cols_kernel<uint32_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t*)in, (uint32_t*)out, perm, graph_ids, B, R, N, scatter);
  else
    CP_LAUNCH(permute_cols_kernel<uint64_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint64_t*)in, (uint64_t*)out, perm, graph_ids, B, R, N, scatter);
  return cp_check_launch();
}
