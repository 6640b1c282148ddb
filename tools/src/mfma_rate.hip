// Peak-rate probe for v_mfma_f32_16x16x4_f32 on the box it runs on (source of tools/bin/mfma_rate; replaces the
// source-less tools/mfma_rate.bin of round 1).  Back-to-back independent MFMAs from registers, no memory traffic:
//   mfma_rate [waves_per_simd=1|2] [zero|rand] -> TFLOP/s, effective clock implied by 64 FLOP/clk/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/src/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed_a, float seed_b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = seed_a * (float)(threadIdx.x % 7 + 1), b = seed_b * (float)(threadIdx.x % 5 + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;   // keep the chain alive
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const bool zero = argc > 2 && !strcmp(argv[2], "zero");
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, 4);
  const int iters = 20000, NACC = 16;
  const int grid = cus * wps;                       // 256 threads = 4 waves = one per SIMD; wps workgroups per CU
  const float sa = zero ? 0.f : 1e-3f, sb = zero ? 0.f : 1.1e-3f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, sa, sb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)iters * 8 * NACC * 4 * grid;
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (rep) printf("waves/SIMD %d, %s operands: %.2f ms  %.1f TFLOP/s  (= %.2f GHz x 64 FLOP/clk x %d SIMDs)\n", wps, zero ? "zero" : "non-zero", ms, tf,
                    tf * 1e12 / (64.0 * 4 * cus) / 1e9, 4 * cus);
  }
  return 0;
}
