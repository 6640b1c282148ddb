// Does the MFMA shape matter for the clock the chip sustains under conv_sk2-like operand traffic?  One workgroup per CU,
// 4 waves (one per SIMD), 128 accumulator registers per wave, fragments re-read from LDS (random data, conflict-free
// b128 pattern) at conv_sk2's rate of 12 ds_read_b128 per 4096 MFMA cycles:
//   variant 0: v_mfma_f32_16x16x4_f32, 128 instructions per 12 fragments (what conv_sk2 issues)
//   variant 1: v_mfma_f32_32x32x2_f32,  64 instructions per 12 fragments (same FLOPs, half the A/B operand reads per FLOP)
//   mfma_lds_rate -> TFLOP/s and the clock implied by 64 FLOP/clk/SIMD for both, with and without the LDS reads.
// Build: hipcc --offload-arch=gfx950 -O3 tools/src/mfma_lds_rate.hip -o tools/bin/mfma_lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int VAR, bool LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* init, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 48 KB: 384 rows x 32 floats
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 384 * 32; i += 256) smem[i] = init[i];
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  const int swz = (r >> 1) & 7;
  const float* base = smem + (wave * 64 + r) * 32 + ((g ^ swz) << 2);
  f32x4 fr[12];
#pragma unroll
  for (int u = 0; u < 12; ++u) fr[u] = *reinterpret_cast<const f32x4*>(base + (u % 4) * 16 * 32 + (u / 4) * 4);
  if constexpr (VAR == 0) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int u = 0; u < 12; ++u) fr[u] = *reinterpret_cast<const f32x4*>(base + ((u + it) % 4) * 16 * 32 + (u / 4) * 4 + (it & 1) * 128 * 32);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[8 + j][e], fr[i][e], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[0] = s;
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int u = 0; u < 12; ++u) fr[u] = *reinterpret_cast<const f32x4*>(base + ((u + it) % 4) * 16 * 32 + (u / 4) * 4 + (it & 1) * 128 * 32);
      }
      // 12 fragments = two 8-k slices of (4 row blocks + 2 column blocks): fr[0..3] / fr[4..7] A, fr[8..9] / fr[10..11] B
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[8 + 2 * h + j][e], fr[4 * h + i][e], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) out[0] = s;
  }
}

template <int VAR, bool LDS>
static void run(const char* name, float* out, const float* init, int cus) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VAR, LDS>), dim3(cus), dim3(256), 48 * 1024, 0, out, init, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 4096 * 32 * (double)iters * 4 * cus;   // 128 x (16x16x4) = 64 x (32x32x2) = 131072 MACs per wave per trip
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (rep) printf("%-44s %8.2f ms  %6.1f TFLOP/s  (= %.2f GHz x 64 FLOP/clk x %d SIMDs)\n", name, ms, tf, tf * 1e12 / (64.0 * 4 * cus) / 1e9, 4 * cus);
  }
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float *out, *init;
  hipMalloc(&out, 4); hipMalloc(&init, 384 * 32 * 4);
  std::vector<float> h(384 * 32);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  hipMemcpy(init, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0, false>("16x16x4, operands in registers", out, init, cus);
  run<1, false>("32x32x2, operands in registers", out, init, cus);
  run<0, true>("16x16x4, 12 ds_read_b128 per 128 MFMAs", out, init, cus);
  run<1, true>("32x32x2, 12 ds_read_b128 per  64 MFMAs", out, init, cus);
  return 0;
}
