// Version / error strings and hipGraph capture helpers of the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

extern "C" int cp_version(void) { return 204; }   // 0.2.4: cp_set_deterministic, cp_status_*; 0.2.3: cp_kernel_log (0.2.2: cp_graph_capture_set_deps / _tail; 0.2.1: cp_pack_hr_chain_weight takes the folded-BN scale)

// process-wide mode switch of the TRAINING entry points (include/checkerpose_hip.h): every accumulation in a fixed order
std::atomic<int> g_cp_deterministic{0};
extern "C" void cp_set_deterministic(int on) { g_cp_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); }
extern "C" int cp_get_deterministic(void) { return g_cp_deterministic.load(std::memory_order_relaxed); }

// ---- device-side error channel: ONE sticky uint32 per device (allocated on the first call of that device, outside any capture),
// OR-ed into by kernels that can detect a failure of their own (bit list: include/checkerpose_hip.h), read back by cp_device_status()
static std::atomic<uint32_t*> g_status_word[256];
uint32_t* cp_status_word(bool create) {
  const int dev = cp_current_device();
  if (dev < 0 || dev > 255) return nullptr;
  uint32_t* p = g_status_word[dev].load(std::memory_order_acquire);
  if (p || !create) return p;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(nullptr, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
  uint32_t* q = nullptr;
  if (hipMalloc((void**)&q, 256) != hipSuccess) return nullptr;
  if (hipMemset(q, 0, 256) != hipSuccess) { (void)hipFree(q); return nullptr; }
  uint32_t* expect = nullptr;
  if (!g_status_word[dev].compare_exchange_strong(expect, q, std::memory_order_acq_rel)) { (void)hipFree(q); return expect; }
  return q;
}
extern "C" int cp_device_status(uint32_t* status_out, int clear) {
  if (!status_out) return CP_ERR_INVALID;
  uint32_t* p = cp_status_word(true);
  if (!p) return CP_ERR_HIP;
  if (hipMemcpy(status_out, p, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return CP_ERR_HIP;      // synchronises with the device
  if (clear && *status_out && hipMemset(p, 0, sizeof(uint32_t)) != hipSuccess) return CP_ERR_HIP;
  return CP_OK;
}

static thread_local char g_last_kernel[128] = "";
static thread_local char g_kernel_log[1024] = "";   // every symbol since cp_kernel_log_begin(), " + " between them
void cp_mark_kernel(const char* fmt, ...) {
  char tmp[128];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tmp, sizeof(tmp), fmt, ap);
  va_end(ap);
  // "(name<args>)" from a stringified launch expression -> "name<args>"
  const char* b = tmp;
  size_t n = strlen(tmp);
  if (n >= 2 && tmp[0] == '(' && tmp[n - 1] == ')') { b = tmp + 1; n -= 2; }
  memcpy(g_last_kernel, b, n);
  g_last_kernel[n] = 0;
  const size_t have = strlen(g_kernel_log);
  if (have + n + 4 < sizeof(g_kernel_log))
    snprintf(g_kernel_log + have, sizeof(g_kernel_log) - have, "%s%s", have ? " + " : "", g_last_kernel);
}
extern "C" const char* cp_last_kernel(void) { return g_last_kernel; }
extern "C" void cp_kernel_log_begin(void) { g_kernel_log[0] = 0; }
extern "C" const char* cp_kernel_log(void) { return g_kernel_log; }

extern "C" const char* cp_strerror(int code) {
  switch (code) {
    case CP_OK: return "ok";
    case CP_ERR_INVALID: return "invalid argument or unsupported shape";
    case CP_ERR_HIP: return "HIP runtime error (kernel launch / graph call failed)";
    case CP_ERR_ALIGN: return "pointer, channel count or stride not aligned to 16 bytes / 4 channels";
    case CP_ERR_RANGE: return "buffer exceeds 2 GiB 32-bit buffer addressing";
    default: return "unknown checkerpose_hip error code";
  }
}

// hipGraph: the forward is ~350 short launches; replaying them as one graph removes the per-launch host
// cost.  Capture mode is thread-local so other threads (e.g. a data loader touching HIP) are not affected.
extern "C" int cp_graph_begin_capture(cp_stream_t stream) {
  return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? CP_OK : CP_ERR_HIP;
}

extern "C" int cp_graph_end_capture(cp_stream_t stream, void** graph_exec_out) {
  if (!graph_exec_out) return CP_ERR_INVALID;
  hipGraph_t graph = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &graph) != hipSuccess || !graph) return CP_ERR_HIP;
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return CP_ERR_HIP;
  *graph_exec_out = (void*)exec;
  return CP_OK;
}

// Dataflow capture: ONE capturing stream, and before each launch the caller names the graph nodes it depends on (the
// producers of its operands) instead of inheriting "everything so far on this stream".  The captured graph is then the
// program's true dependency DAG -- no fork/join barriers, and no cross-stream event pairs (which crash
// hipStreamEndCapture on ROCm 7.2 when two streams wait on each other).
extern "C" int cp_graph_capture_set_deps(cp_stream_t stream, void* const* nodes, int n) {
  if (n < 0 || (n > 0 && !nodes)) return CP_ERR_INVALID;
  return hipStreamUpdateCaptureDependencies((hipStream_t)stream, (hipGraphNode_t*)nodes, (size_t)n, hipStreamSetCaptureDependencies) == hipSuccess
             ? CP_OK : CP_ERR_HIP;
}

// the capturing stream's current dependency set = the node(s) the next launch would wait for (after a launch: that launch)
extern "C" int cp_graph_capture_tail(cp_stream_t stream, void** nodes_out, int cap, int* n_out) {
  if (!nodes_out || !n_out || cap < 1) return CP_ERR_INVALID;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  const hipGraphNode_t* deps = nullptr;
  size_t n = 0;
  if (hipStreamGetCaptureInfo_v2((hipStream_t)stream, &status, nullptr, nullptr, &deps, &n) != hipSuccess) return CP_ERR_HIP;
  if (status != hipStreamCaptureStatusActive) return CP_ERR_INVALID;
  if ((int)n > cap) return CP_ERR_RANGE;
  for (size_t i = 0; i < n; ++i) nodes_out[i] = (void*)deps[i];
  *n_out = (int)n;
  return CP_OK;
}

extern "C" int cp_graph_launch(void* graph_exec, cp_stream_t stream) {
  if (!graph_exec) return CP_ERR_INVALID;
  return hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) == hipSuccess ? CP_OK : CP_ERR_HIP;
}

extern "C" int cp_graph_destroy(void* graph_exec) {
  if (!graph_exec) return CP_OK;
  return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? CP_OK : CP_ERR_HIP;
}
