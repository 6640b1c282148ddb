// What ONE compute unit can stream: G workgroups (256 threads, one per CU while G <= 256) each read the SAME `mbytes` MB buffer
// front to back with 16-byte loads, 8 in flight per thread -- the access pattern of a workgroup that owns whole rows of a decoder
// layer and therefore needs all of the layer's weights ("one launch per decoder layer" for the lock-step MT decode).
//   cu_stream_rate [mbytes=8] [threads=256|512|1024] -> per G in {1, 16, 64, 256}: us per pass of the buffer, GB/s per workgroup, aggregate GB/s
// Build: hipcc --offload-arch=gfx950 -O3 tools/src/cu_stream_rate.hip -o tools/bin/cu_stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void stream_all(const float4* __restrict__ w, size_t n4, int passes, float* out) {
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < passes; ++p) {
    // start each pass at a workgroup-dependent offset so that the workgroups do not all ask for the same line at the same time
    const size_t start = ((size_t)blockIdx.x * 8191u) % n4;
    for (size_t i0 = 0; i0 < n4; i0 += (size_t)blockDim.x * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        size_t i = start + i0 + (size_t)u * blockDim.x + threadIdx.x;
        if (i >= n4) i -= n4;
        v[u] = w[i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const int mb = argc > 1 ? atoi(argv[1]) : 8;
  const int nt = argc > 2 ? atoi(argv[2]) : 256;
  const size_t bytes = (size_t)mb << 20, n4 = bytes / 16;
  float4* w = nullptr;
  float* out = nullptr;
  CHECK(hipMalloc(&w, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(w, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int passes = 20;
  printf("buffer %d MB, %d passes per launch, %d-thread workgroups, 8 x 16-byte loads in flight per thread\n", mb, passes, nt);
  for (int G : {1, 16, 64, 256}) {
    hipLaunchKernelGGL(stream_all, dim3(G), dim3(nt), 0, 0, w, n4, 2, out);     // warm (TLB, caches)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream_all, dim3(G), dim3(nt), 0, 0, w, n4, passes, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us_pass = 1e3 * ms / passes;
    printf("G = %3d workgroups: %8.1f us per pass of the buffer, %7.1f GB/s per workgroup, %8.1f GB/s aggregate\n", G, us_pass,
           bytes / us_pass * 1e-3, (double)G * bytes / us_pass * 1e-3);
  }
  return 0;
}
