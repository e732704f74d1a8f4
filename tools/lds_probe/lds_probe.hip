// Which lanes of a wave share an LDS cycle for ds_read_b128?  Each lane reads 16 bytes at its own offset `iters` times (dependent
// chain through the address); the host times patterns that are conflict-free under one candidate grouping and 2-way under another.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void lds_probe_kernel(const int* offs, int iters, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  for (int i = threadIdx.x; i < 16384; i += 256) ((uint32_t*)smem)[i] = (uint32_t)i;
  __syncthreads();
  const uint32_t off = (uint32_t)offs[threadIdx.x & 63];
  uint32_t acc = 0;
  const uint32_t base = (uint32_t)(uintptr_t)smem + off;     // LDS addresses are 32-bit offsets
  for (int i = 0; i < iters; ++i) {
    const uint32_t o = base + (uint32_t)(i & 3) * 8192u;       // throughput-bound: 8 independent reads in flight per wave
    u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                 "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                 "s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(o) : "memory");
    acc += v0.x ^ v1.x ^ v2.x ^ v3.x ^ v4.x ^ v5.x ^ v6.x ^ v7.x;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
extern "C" int lds_probe_run(void* stream, const int* offs_dev, int iters, uint32_t* out_dev, int blocks) {
  hipLaunchKernelGGL(lds_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, offs_dev, iters, out_dev);
  return (int)hipGetLastError();
}
