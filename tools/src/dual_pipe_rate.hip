// Can the f32 VALU pipe add throughput next to the f32 matrix pipe?  (MI355X_MICROARCH.md: the two pipes are separate and
// the f32 MFMA rate equals the f32 vector rate, 64 FLOP/clk/SIMD each.)  512-thread workgroups, one per CU: waves 0-3 run
// back-to-back v_mfma_f32_16x16x4_f32 from registers, waves 4-7 (the second wave of each SIMD) run independent v_pk_fma_f32
// chains from registers.  Three launches: MFMA waves only, VALU waves only, both.  -> TFLOP/s of each pipe and the sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/src/dual_pipe_rate.hip -o tools/bin/dual_pipe_rate
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, float sa, float sb) {
  const int wave = threadIdx.x >> 6;
  const bool mfma_wave = wave < 4;
  float a = sa * (float)(threadIdx.x % 7 + 1), b = sb * (float)(threadIdx.x % 5 + 1);
  if (mfma_wave) {
    if (!(mode & 1)) return;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
  } else {
    if (!(mode & 2)) return;
    // 128 MFMAs of 32 cycles per trip on the partner = 4096 cycles; the same time of packed FMAs (4 cycles each) = 1024
    f32x2 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x2{0.f, 0.f};
    const f32x2 x = {a, b}, y = {b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_elementwise_fma(x, y, acc[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1];
    if (s == 12345.678f) out[1] = s;
  }
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, 8);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"", "MFMA waves only", "VALU (v_pk_fma_f32) waves only", "both pipes, one wave each per SIMD"};
  for (int pass = 0; pass < 2; ++pass)
    for (int mode = 1; mode <= 3; ++mode)
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, iters, mode, 1e-3f, 1.1e-3f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double f_m = (mode & 1) ? 2.0 * 1024 * 128 * (double)iters * 4 * cus : 0.0;          // 128 MFMAs x 1024 MACs per wave-trip
        const double f_v = (mode & 2) ? 2.0 * 128 * 1024 * (double)iters * 4 * cus : 0.0;           // 1024 pk_fma x 128 FMAs per wave-trip
        if (rep) printf("%-38s %7.2f ms   MFMA %6.1f TF   VALU %6.1f TF   sum %6.1f TF\n", names[mode], ms, f_m / ms / 1e9, f_v / ms / 1e9,
                        (f_m + f_v) / ms / 1e9);
      }
  return 0;
}
