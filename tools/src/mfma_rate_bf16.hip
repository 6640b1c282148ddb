// Peak-rate probe for the bf16 MFMAs a split-bf16 (3x) contraction would use: v_mfma_f32_16x16x16_bf16 (the legacy _1k
// form, same lane layout as the f32 16x16x4 fragments) and v_mfma_f32_16x16x32_bf16 (gfx950).  Back-to-back
// independent MFMAs from registers -> TFLOP/s and cycles per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/src/mfma_rate_bf16.hip -o tools/bin/mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x4 a4, b4; bf16x8 a8, b8;
#pragma unroll
  for (int e = 0; e < 4; ++e) { a4[e] = (short)(0x3c00 + threadIdx.x % 7 + e); b4[e] = (short)(0x3c10 + threadIdx.x % 5 + e); }
#pragma unroll
  for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(1e-3f * (float)(threadIdx.x % 7 + e + 1)); b8[e] = (__bf16)(1.1e-3f * (float)(threadIdx.x % 5 + e + 1)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (MODE == 16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(int cus, float* out) {
  const int iters = 20000, NACC = 16, grid = cus;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<MODE, NACC>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 8 * NACC;                 // per wave
    const double flops = 2.0 * 16 * 16 * MODE * n_mfma * 4 * grid;
    if (rep) printf("v_mfma_f32_16x16x%d_bf16: %.2f ms  %.1f TFLOP/s  %.1f cycles per MFMA per SIMD at 2.37 GHz\n", MODE, ms,
                    flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.37e9 / n_mfma);
  }
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, 4);
  run<16>(cus, out);
  run<32>(cus, out);
  return 0;
}
