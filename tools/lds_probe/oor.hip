// What does gfx950 do with an LDS access beyond the workgroup's LDS allocation?  (Predication by address: a lane that must not
// write gets an out-of-range address instead of a divergent branch.)  One workgroup with `lds_bytes` of dynamic LDS:
//   1. fill [0, lds_bytes) with a pattern; 2. lanes write 0xdeadbeef at lds_bytes + 16 * tid (b128) and lds_bytes + 65536 + 4 * tid (b32);
//   3. read everything back: out[0] = number of in-range dwords that changed, out[1] = a b32 read at lds_bytes + 4 * tid summed,
//   out[2] = a b128 read .x at lds_bytes + 16 * tid summed.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void oor_kernel(uint32_t* out, uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < lds_bytes / 4; i += blockDim.x) ((uint32_t*)smem)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a16 = lds_bytes + 16u * tid, a4 = lds_bytes + 65536u + 4u * tid;
  asm volatile("" : "+v"(a16), "+v"(a4));                  // keep the compiler from reasoning about the range
  *(u32x4*)(smem + a16) = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
  *(uint32_t*)(smem + a4) = 0xdeadbeefu;
  __syncthreads();
  uint32_t bad = 0;
  for (uint32_t i = tid; i < lds_bytes / 4; i += blockDim.x) bad += ((uint32_t*)smem)[i] != i * 2654435761u;
  uint32_t r4 = *(const uint32_t*)(smem + lds_bytes + 4u * tid);
  asm volatile("" : "+v"(a16));
  uint32_t r16 = (*(const u32x4*)(smem + a16)).x;
  atomicAdd(out + 0, bad);
  atomicAdd(out + 1, r4 != 0u);
  atomicAdd(out + 2, r16 != 0u);
}
extern "C" int oor_run(void* stream, uint32_t* out, uint32_t lds_bytes) {
  if (hipFuncSetAttribute((const void*)oor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
  hipLaunchKernelGGL(oor_kernel, dim3(1), dim3(512), lds_bytes, (hipStream_t)stream, out, lds_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
