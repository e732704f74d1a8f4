// Shared device/host helpers for libcheckerpose_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/checkerpose_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ---- bf16 <-> f32 on raw bits (round-to-nearest-even; NaN kept a NaN by the plain cast) --------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;                      // v_cvt_pk_bf16_f32 on gfx950
  return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
// two values in ONE v_cvt_pk_bf16_f32 (the vector conversion; two scalar casts + shift + or compiled to four instructions per pair --
// a quarter of the VALU work of every bf16 epilogue).  Same instruction, same rounding: bit-identical results.
typedef __bf16 cp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(cp_f32x2{lo, hi}, cp_bf16x2));
}

// (Leaky)ReLU / identity without per-element branches: with a run-time `act` the unrolled epilogues compiled to a scalar compare + branch
// per VALUE (25 static instructions per output in the decoder conv's epilogue).  s = 0 / slope / 1, then max(v, v * s + 0): the "+ 0"
// turns the -0 of a negative v times 0 into +0, so ReLU is exactly fmaxf(v, 0); slopes in [0, 1] only.
__device__ __forceinline__ float cp_act_slope(int act, float slope) { return act == 1 ? 0.f : (act == 2 ? slope : 1.f); }   // CP_ACT_RELU, CP_ACT_LEAKY
__device__ __forceinline__ float cp_act_apply(float v, float s) { return fmaxf(v, fmaf(v, s, 0.f)); }

// ---- LDS-DMA (global memory -> LDS without registers) in its BUFFER form, 16 bytes per lane: lane l writes LDS bytes [dst + 16 l, + 16).
// The global form (global_load_lds_dwordx4) is FLAT-encoded, and hipcc's waitcnt insertion treats a pending FLAT access that may touch
// LDS as unordered with the wave's ds_reads: from the first DMA on every LDS wait of the kernel is lgkmcnt(0).  With the buffer form the
// counted waits come back -- worth 4-8 % on edgeconv_fused (72.6 -> 69.5 us), nothing on hr_chain, and a loss on edgeconv_tiled2
// (the compiler then also guards ds_reads behind the DMA with vmcnt(0), and spills): each kernel picks its form by measurement.
// `off` = byte offset from the descriptor's base (< 4 GB); completion is tracked by vmcnt either way.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cp_dma_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void cp_lds_dma16(__amdgpu_buffer_rsrc_t rs, uint32_t off, void* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, off, 0, 0, 0);
}

// relu of two packed bf16 values: the sign bit is the int16 sign bit, so max(int16, 0) per half (-0.0 -> +0.0 as fmaxf does)
typedef short cp_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(cp_s16x2, v), cp_s16x2{0, 0}));
}

// ---- EdgeConv gather keys (edgeconv_fused.hip / edgeconv_tiled.hip): P' = s * (W1 x) is tabled as packed IEEE HALF pairs so that
// the K-way neighbour max is gfx950's three-input packed maximum (v_pk_maximum3_f16: two neighbours x two channels per
// instruction; int16 order keys of the bf16 value needed v_pk_max_i16 = one neighbour per instruction plus the key transform on
// both sides).  f16 keeps 3 more mantissa bits than the bf16 storage type, values are clamped to its finite range (+-65504).
typedef _Float16 cp_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2_sat(float a, float b) {
  cp_h2 r;
  r.x = (_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
  r.y = (_Float16)__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);          // the pair: one v_cvt_pk_f16_f32
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pkmax3_f16(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_bit_cast(cp_h2, a),
                            __builtin_elementwise_maximum(__builtin_bit_cast(cp_h2, b), __builtin_bit_cast(cp_h2, c))));
}
// m[j] = max(m[j], a[j], b[j], c[j], d[j]) for the four dwords of a 16-byte table piece, as exactly EIGHT v_pk_maximum3_f16
// (hipcc re-associates nested maxima into balanced trees: max(a, b), max(c, d), max(m, .., ..) = three instructions per dword
// instead of two, with an s_nop between dependent packed instructions).  One asm block, the two rounds interleaved over the four
// dwords: a dependent instruction is four issue slots behind its producer (gfx940+ packed / dst_sel forwarding needs one).
__device__ __forceinline__ void pkmax5x4_f16(uint32_t* m, const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d) {
  asm("v_pk_maximum3_f16 %0, %0, %4, %8\n\t"
      "v_pk_maximum3_f16 %1, %1, %5, %9\n\t"
      "v_pk_maximum3_f16 %2, %2, %6, %10\n\t"
      "v_pk_maximum3_f16 %3, %3, %7, %11\n\t"
      "v_pk_maximum3_f16 %0, %0, %12, %16\n\t"
      "v_pk_maximum3_f16 %1, %1, %13, %17\n\t"
      "v_pk_maximum3_f16 %2, %2, %14, %18\n\t"
      "v_pk_maximum3_f16 %3, %3, %15, %19"
      : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3])
      : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w),
        "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
}
__device__ __forceinline__ float f16_lo(uint32_t w) { return (float)__builtin_bit_cast(cp_h2, w).x; }
__device__ __forceinline__ float f16_hi(uint32_t w) { return (float)__builtin_bit_cast(cp_h2, w).y; }
constexpr uint32_t CP_F16X2_NEG_INF = 0xFC00FC00u;

// ---- keypoint-side (GNN) kernels in IEEE half (CP_F16, round 6): same bytes and the same MFMA rate as bf16 with 3 more mantissa
// bits.  EdgeConv differences x_j - x_i of neighbouring keypoints cancel most of |x|, so the storage rounding of x is amplified
// there; on trained-like weights the keypoint side produced 73 % of the bf16 path's logit-error variance.
// H = false: bf16 (v_mfma_f32_16x16x32_bf16, v_cvt_pk_bf16_f32); H = true: f16 (v_mfma_f32_16x16x32_f16, saturating v_cvt_pk_f16_f32).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool H>
__device__ __forceinline__ f32x4 cp_mma16(const u32x4& w, const u32x4& a, const f32x4& acc) {
  if constexpr (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
// Saturating f32 -> f16 conversions for free: MODE.FP16_OVFL (hwreg(HW_REG_MODE) bit 23) makes an f16 RESULT that overflows come out as
// +-65504 instead of +-inf (true infinities stay infinite).  Verified on gfx950 for v_cvt_pk_f16_f32 and v_cvt_f16_f32
// (tools/f16_probe/ovfl.hip: 1e6 -> 0x7bff, -1e6 -> 0xfbff, 65520 -> 0x7bff, inf -> 0x7c00).  The explicit clamps cost the fused EdgeConv
// launch 3 of 69 us (v_pk_min_f16 + v_pk_max_f16 per pair in the barrier-bound epilogue; the fmed3 form also 8 spilled VGPRs in the
// fused MLP pair).  MODE is per-wave state initialised at wave launch: a kernel that packs halves calls cp_f16_saturate_on() FIRST.
__device__ __forceinline__ void cp_f16_saturate_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory"); }
__device__ __forceinline__ uint32_t pack_f16x2_ovfl(float a, float b) {      // one v_cvt_pk_f16_f32; saturates under cp_f16_saturate_on()
  cp_h2 r;
  r.x = (_Float16)a;
  r.y = (_Float16)b;
  return __builtin_bit_cast(uint32_t, r);
}
template <bool H>
__device__ __forceinline__ uint32_t cp_pack2(float lo, float hi) {
  if constexpr (H) return pack_f16x2_ovfl(lo, hi);         // the kernel has called cp_f16_saturate_on()
  else return pack_bf16x2(lo, hi);
}
template <bool H>
__device__ __forceinline__ u32x4 cp_pack8(const float* f) {
  u32x4 v; v.x = cp_pack2<H>(f[0], f[1]); v.y = cp_pack2<H>(f[2], f[3]); v.z = cp_pack2<H>(f[4], f[5]); v.w = cp_pack2<H>(f[6], f[7]);
  return v;
}
__device__ __forceinline__ uint32_t f32_to_f16_bits_sat(float f) {      // explicit clamp: weight packers, scalar epilogues (no mode bit needed)
  return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
// storage bits of one element: dtype CP_BF16 or CP_F16 (weight packers, scalar epilogues)
__device__ __forceinline__ uint32_t f32_to_half_bits(float f, int dtype) { return dtype == CP_F16 ? f32_to_f16_bits_sat(f) : f32_to_bf16_bits(f); }

// element type tags
struct F32Tag { using elem = float; static constexpr int E = 4; static constexpr int dtype = CP_F32; };
struct BF16Tag { using elem = uint16_t; static constexpr int E = 8; static constexpr int dtype = CP_BF16; };
struct F16Tag { using elem = uint16_t; static constexpr int E = 8; static constexpr int dtype = CP_F16; };      // IEEE half (keypoint side)

// 16-byte vector <-> E floats
template <typename Tag> struct Vec16;
template <> struct Vec16<F32Tag> {
  static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) {
    u32x4 v; v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]); v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
    return v;
  }
};
template <> struct Vec16<BF16Tag> {
  static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) {
    u32x4 v; v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
  }
};

template <> struct Vec16<F16Tag> {            // pack() saturates under cp_f16_saturate_on() (the kernel sets the mode bit first)
  static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
    f[0] = f16_lo(v.x); f[1] = f16_hi(v.x); f[2] = f16_lo(v.y); f[3] = f16_hi(v.y);
    f[4] = f16_lo(v.z); f[5] = f16_hi(v.z); f[6] = f16_lo(v.w); f[7] = f16_hi(v.w);
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) { return cp_pack8<true>(f); }
};

// align_corners source coordinate of output index o (scale s = (n - 1) / (2 n - 1)): left neighbour, step to the right one, the
// right one's weight.  `contract(off)`: hipcc's default -ffp-contract=fast may fuse s * o - i0 into one fma in one kernel and
// not in another (__fmul_rn / __fsub_rn are plain * and - in HIP); with the pragma every kernel derives the same weights.
__device__ __forceinline__ void cp_up_coord(float s, int o, int n, int& i0, int& step, float& l1) {
#pragma clang fp contract(off)
  const float f = s * (float)o;
  i0 = (int)f;
  step = i0 < n - 1 ? 1 : 0;
  l1 = f - (float)i0;
}
__device__ __forceinline__ float cp_one_minus(float l1) {
#pragma clang fp contract(off)
  return 1.f - l1;
}
// bilinear sample from its four neighbours, ONE rounding sequence shared by the stand-alone x2 kernel (elementwise.hip) and
// the conv loader that interpolates while staging (conv3x3_halo.hip): both give the same bits
__device__ __forceinline__ float cp_bilerp(float a, float b, float c, float d, float lx0, float lx1, float ly0, float ly1) {
#pragma clang fp contract(off)
  const float t = __builtin_fmaf(lx1, b, lx0 * a);
  const float u = __builtin_fmaf(lx1, d, lx0 * c);
  return __builtin_fmaf(ly1, u, ly0 * t);
}

template <typename Tag> __device__ __forceinline__ float load_elem(const void* p, size_t i);
template <> __device__ __forceinline__ float load_elem<F32Tag>(const void* p, size_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float load_elem<BF16Tag>(const void* p, size_t i) {
  return bf16_bits_to_f32(((const uint16_t*)p)[i]);
}
template <> __device__ __forceinline__ float load_elem<F16Tag>(const void* p, size_t i) { return f16_bits_to_f32(((const uint16_t*)p)[i]); }
template <typename Tag> __device__ __forceinline__ void store_elem(void* p, size_t i, float v);
template <> __device__ __forceinline__ void store_elem<F16Tag>(void* p, size_t i, float v) { ((uint16_t*)p)[i] = (uint16_t)f32_to_f16_bits_sat(v); }
template <> __device__ __forceinline__ void store_elem<F32Tag>(void* p, size_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void store_elem<BF16Tag>(void* p, size_t i, float v) {
  ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16_bits(v);
}

// ---- host side -------------------------------------------------------------------------------
// Kernel-work A/B knobs (CP_NO_HALO4, CP_CONV_MT, ...): compiled OUT of the shipped library, which reads no environment
// and holds no hidden state; `make KNOBS=1` (-DCP_DEBUG_KNOBS) builds the variant that honours them.
#ifdef CP_DEBUG_KNOBS
#include <stdlib.h>
static inline const char* cp_knob(const char* name) { return getenv(name); }
#else
static inline const char* cp_knob(const char*) { return nullptr; }
#endif
// cp_set_deterministic(): the training entry points then accumulate in a fixed order (no floating-point atomics)
extern std::atomic<int> g_cp_deterministic;
static inline bool cp_deterministic() { return g_cp_deterministic.load(std::memory_order_relaxed) != 0; }
// device-side error channel (api.hip): the current device's sticky status word, or nullptr (not created yet and `create` false / a capture
// is under way); kernels that get a non-null pointer atomicOr their failure bit into it
uint32_t* cp_status_word(bool create);
// profiling aid (cp_last_kernel): every launch site records the symbol it launches, spelled as rocprofv3 prints it
void cp_mark_kernel(const char* fmt, ...);
#define CP_LAUNCH(kernel, ...) do { cp_mark_kernel("%s", #kernel); hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)
static inline int cp_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CP_OK : CP_ERR_HIP;
}
// Per-DEVICE launch prerequisites (hipFuncSetAttribute applies to the current device only; a process may drive several).
// The library keeps no other state than these idempotent caches: a bit per device ordinal, set after the attribute calls of
// that device succeeded; two threads racing here both set the same attributes (harmless) and OR the same bit.
struct CpDeviceOnce {
  std::atomic<unsigned long long> bits[4];      // device ordinals 0..255
  bool needed(int dev) const { return !((bits[(dev >> 6) & 3].load(std::memory_order_acquire) >> (dev & 63)) & 1ull); }
  void mark(int dev) { bits[(dev >> 6) & 3].fetch_or(1ull << (dev & 63), std::memory_order_release); }
};
static inline int cp_current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for `fn` on the current device, once per device; CP_OK / CP_ERR_HIP
#define CP_LDS_ATTR_ONCE(once, dev, ...)                                                     \
  do {                                                                                       \
    if ((dev) < 0) return CP_ERR_HIP;                                                        \
    if ((once).needed(dev)) {                                                                \
      if (!(__VA_ARGS__)) return CP_ERR_HIP;                                                 \
      (once).mark(dev);                                                                      \
    }                                                                                        \
  } while (0)
static inline bool cp_set_max_lds(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}
// compute units of the CURRENT device (cached per device ordinal); 0 on error
static inline int cp_num_cus() {
  static std::atomic<int> cache[256];
  const int dev = cp_current_device();
  if (dev < 0) return 0;
  int n = cache[dev & 255].load(std::memory_order_relaxed);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cache[dev & 255].store(n, std::memory_order_relaxed);
  }
  return n;
}
static inline bool cp_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
// the branch-free epilogue activation (cp_act_apply: max(v, v * s + 0)) is LeakyReLU only for slopes in [0, 1]: anything else
// (a slope > 1, a negative one, NaN) is refused at the entry point instead of computed wrongly
static inline bool cp_act_ok(int act, float slope) { return act != 2 /* CP_ACT_LEAKY */ || (slope >= 0.f && slope <= 1.f); }
static inline int cp_elem_size(int dtype) { return (dtype == CP_BF16 || dtype == CP_F16) ? 2 : 4; }
