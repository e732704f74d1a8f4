// Does MODE.FP16_OVFL (hwreg(HW_REG_MODE) bit 23) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate at +-65504 on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 tools/f16_probe/ovfl.hip -o build/f16_ovfl_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, uint32_t* out, int set) {
  if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
  h2 r; r.x = (_Float16)a; r.y = (_Float16)b;
  out[threadIdx.x] = __builtin_bit_cast(uint32_t, r);
  _Float16 s = (_Float16)a;                      // scalar conversion
  out[64 + threadIdx.x] = (uint32_t)__builtin_bit_cast(uint16_t, s);
}
int main() {
  float h[8] = {1e6f, -1e6f, 65504.f, 65520.f, 70000.f, -70000.f, __builtin_inff(), 3.5f};
  float* d; uint32_t* o; uint32_t r[128];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int set = 0; set < 2; ++set) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(4), 0, 0, d, o, set);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d packed:", set);
    for (int i = 0; i < 4; ++i) printf(" %04x %04x", r[i] & 0xffff, r[i] >> 16);
    printf("  scalar:");
    for (int i = 0; i < 4; ++i) printf(" %04x", r[64 + i]);
    printf("\n");
  }
  return 0;
}
