// Shared definitions for the StreamSpeech gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#define SS_OK 0
#define SS_ERR_HIP 1
#define SS_ERR_ARG 2
#define SS_ERR_MISSING_WEIGHT 3
#define SS_ERR_CAPACITY 4
#define SS_ERR_SCRATCH_CAP 5       // a scratch buffer would have to grow past the cap set with ss_scratch_set_cap

#define SS_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      fprintf(stderr, "[streamspeech_hip] %s failed: %s (%s:%d)\n", #expr,              \
              hipGetErrorString(_e), __FILE__, __LINE__);                               \
      return SS_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define SS_LAUNCH_CHECK() SS_HIP_CHECK(hipGetLastError())

// Raise a kernel's dynamic-LDS limit once PER DEVICE, from whichever host thread launches it first there (bench.py drives the
// library from 8 threads; hipFuncSetAttribute applies to the current device's function object only, so a process that drives
// several GPUs must repeat it on each -- ADVICE r3).  Wrap a template-id in parentheses: SS_MAX_LDS_ONCE((&k<A, B>), bytes).
// (the launch path takes no lock -- one acquire load of the per-device bit; the mutex is only taken by the first
//  launches of a kernel on a device.)
#define SS_MAX_LDS_ONCE(kernel, bytes)                                                                          \
  do {                                                                                                          \
    static std::mutex _mu;                                                                                      \
    static std::atomic<unsigned long long> _done[2];     /* devices 0..127; zero-initialised (static storage) */ \
    int _dev = 0;                                                                                               \
    SS_HIP_CHECK(hipGetDevice(&_dev));                                                                          \
    if (_dev < 0 || _dev >= 128) return SS_ERR_ARG;                                                             \
    if (!((_done[_dev >> 6].load(std::memory_order_acquire) >> (_dev & 63)) & 1ull)) {                          \
      std::lock_guard<std::mutex> _lk(_mu);                                                                     \
      if (!((_done[_dev >> 6].load(std::memory_order_relaxed) >> (_dev & 63)) & 1ull)) {                        \
        SS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
        _done[_dev >> 6].fetch_or(1ull << (_dev & 63), std::memory_order_release);                              \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

namespace ss {

constexpr int WAVE = 64;

// activation codes shared by kernels and the C ABI
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_LRELU = 3, ACT_TANH = 4 };

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace ss
