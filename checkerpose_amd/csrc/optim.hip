// Optimizer step of the training loop (the reference's train.py:244-246,320: optim.Adam / optim.SGD(momentum=0.9)
// over net.parameters(), optimizer.step() once per batch) as ONE launch over all parameter tensors.
// torch's own fused Adam needs ~30 launches and ~10 ms of HOST time per step for the ~1 000 parameter tensors of PoseNet_GNNskip --
// with the forward / backward replayed as hipGraphs the step was bound by how fast the host could enqueue it (tools/train_cpu_timeline.py:
// 27 ms of host work per 32 ms step).  Here a device table holds (param, grad, exp_avg, exp_avg_sq, numel) per tensor, block b works on
// 2048 consecutive elements of the tensor k with prefix[k] <= b < prefix[k + 1]; fp32 state, the arithmetic of torch.optim.Adam
// (no amsgrad, L2 weight decay added to the gradient) / torch.optim.SGD (momentum, dampening 0, no nesterov).  HBM-bound streaming:
// 28 (Adam) / 20 (SGD with momentum) bytes per parameter.
#include "common.h"

constexpr int OPT_EPB = 2048;      // elements per block

__device__ __forceinline__ int opt_find(const uint32_t* __restrict__ prefix, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const CpOptItem* __restrict__ items, const uint32_t* __restrict__ prefix, int n, float lr,
                                                         float b1, float b2, float eps, float wd, int step) {
  const int k = opt_find(prefix, n);
  const CpOptItem it = items[k];
  const size_t base = (size_t)(blockIdx.x - prefix[k]) * OPT_EPB;
  // torch counts steps per parameter: a tensor that got its first gradient later has its own bias corrections (fp64, as torch's host code)
  const double t = (double)(step - (int)it.step0);
  const float bc1 = (float)(1.0 - pow((double)b1, t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  const float step_size = lr / bc1;
#pragma unroll
  for (int j = 0; j < OPT_EPB / 256; ++j) {
    const size_t i = base + j * 256 + threadIdx.x;
    if (i >= it.n) break;
    float p = it.p[i], g = it.g[i];
    if (wd != 0.f) g += wd * p;
    const float m = b1 * it.m[i] + (1.f - b1) * g;
    const float v = b2 * it.v[i] + (1.f - b2) * g * g;
    it.m[i] = m;
    it.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    it.p[i] = p - step_size * (m / denom);
  }
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(const CpOptItem* __restrict__ items, const uint32_t* __restrict__ prefix, int n, float lr,
                                                        float momentum, float wd, int first) {
  const int k = opt_find(prefix, n);
  const CpOptItem it = items[k];
  const size_t base = (size_t)(blockIdx.x - prefix[k]) * OPT_EPB;
#pragma unroll
  for (int j = 0; j < OPT_EPB / 256; ++j) {
    const size_t i = base + j * 256 + threadIdx.x;
    if (i >= it.n) break;
    const float p = it.p[i];
    float g = it.g[i];
    if (wd != 0.f) g += wd * p;
    if (momentum != 0.f) {
      const float buf = first ? g : momentum * it.m[i] + g;
      it.m[i] = buf;
      g = buf;
    }
    it.p[i] = p - lr * g;
  }
}

extern "C" uint32_t cp_opt_item_blocks(uint64_t numel) { return (uint32_t)((numel + OPT_EPB - 1) / OPT_EPB); }

extern "C" int cp_adam_multi(cp_stream_t stream, const CpOptItem* items_dev, const uint32_t* prefix_dev, int n_items, uint32_t total_blocks,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
  if (n_items < 0 || step < 1) return CP_ERR_INVALID;
  if (n_items == 0 || total_blocks == 0) return CP_OK;
  if (!items_dev || !prefix_dev || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(lr >= 0.f))
    return CP_ERR_INVALID;
  CP_LAUNCH(adam_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, prefix_dev, n_items, lr, beta1, beta2, eps,
            weight_decay, step);
  return cp_check_launch();
}

extern "C" int cp_sgd_multi(cp_stream_t stream, const CpOptItem* items_dev, const uint32_t* prefix_dev, int n_items, uint32_t total_blocks,
                            float lr, float momentum, float weight_decay, int first_step) {
  if (n_items < 0) return CP_ERR_INVALID;
  if (n_items == 0 || total_blocks == 0) return CP_OK;
  if (!items_dev || !prefix_dev || !(momentum >= 0.f) || !(lr >= 0.f)) return CP_ERR_INVALID;
  CP_LAUNCH(sgd_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, prefix_dev, n_items, lr, momentum, weight_decay,
            first_step ? 1 : 0);
  return cp_check_launch();
}
