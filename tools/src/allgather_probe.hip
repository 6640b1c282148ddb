// What would ONE all-to-all edge of a persistent MT decode-step kernel cost on this box?  (the step is a
// chain of ~33 GEMV / attention phases, each producing a 512 ... 2048-float vector that EVERY workgroup of the next phase needs.)
// G workgroups x 256 threads, all resident; per phase every wave "computes" its share of an N-float vector while streaming its
// share of a W-byte weight matrix (the real phases stream 1-4 MB), publishes it as 8-byte {tag, value} granules (relaxed
// agent-scope 64-bit stores: the data is the flag, cdna_hip_programming.md Guideline 16 R2), then the four waves of every
// workgroup sweep a quarter of the N granules each (all loads of a pass in flight) until every tag carries the phase's epoch,
// drop the values into LDS, and the workgroup goes on.  (A first form -- wave 0 sweeping alone, 8 ... 32 loads per lane in a loop --
// measured 4.0-4.7 us per 512-float edge and 12.4-13.5 us per 2048-float edge.)
// Every spin is bounded; a time-out sets a flag and ends the kernel.
//   allgather_probe -> microseconds per phase for N in {512, 2048}, G in {64, 128, 256}, with 0 / 2 MB of weights per phase.
// Build: hipcc --offload-arch=gfx950 -O3 tools/src/allgather_probe.hip -o tools/bin/allgather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ __launch_bounds__(256) void chain(u64* gran, const float* wts, size_t w_floats, int N, int phases, unsigned epoch0,
                                             unsigned* err, float* out) {
  extern __shared__ float vec[];                       // the gathered vector
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int G = gridDim.x, gw = blockIdx.x * 4 + wave, nw = G * 4;   // global wave id
  float check = 0.f;
  for (int ph = 0; ph < phases; ++ph) {
    const unsigned epoch = epoch0 + ph;
    u64* g = gran + (size_t)(ph & 1) * N;              // two buffers alternate (a phase's buffer is rewritten two phases later)
    // ---- "compute": stream this wave's share of the weights, produce columns gw, gw + nw, ... ----
    float acc = 0.f;
    if (w_floats) {
      const size_t per = w_floats / nw;                // floats per wave per phase
      const float4* wp = reinterpret_cast<const float4*>(wts + (size_t)gw * per);
      for (size_t i = lane; i < per / 4; i += 64) { const float4 v = wp[i]; acc += (v.x + v.y) + (v.z + v.w); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    }
    for (int col = gw; col < N; col += nw) {
      const float val = (float)(col + ph) + acc * 0.f;
      if (lane == 0) __hip_atomic_store(g + col, ((u64)epoch << 32) | __float_as_uint(val), RLX_AGENT);
    }
    // ---- gather: all four waves sweep a quarter of the N granules each into LDS (every load of a pass in flight before the
    //      first tag is tested: N / 256 loads per lane) ----
    {
      constexpr int MAXL = 8;                            // N <= 2048: <= 8 granules per thread
      const int nl = N / 256;
      unsigned spins = 0;
      for (;;) {
        u64 x[MAXL];
#pragma unroll
        for (int k = 0; k < MAXL; ++k) if (k < nl) x[k] = __hip_atomic_load(g + t + k * 256, RLX_AGENT);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < MAXL; ++k) if (k < nl) { ok &= (unsigned)(x[k] >> 32) == epoch; vec[t + k * 256] = __uint_as_float((unsigned)x[k]); }
        if (__all(ok)) break;
        if (++spins > (1u << 20) || __hip_atomic_load(err, RLX_AGENT)) { if (lane == 0) atomicAdd(err, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    check += vec[(t * 7 + ph) % N];
    __syncthreads();
  }
  if (check == -1.f) out[0] = check;
  if (blockIdx.x == 0 && t == 0) out[1] = check;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  u64* gran; unsigned* err; float *out, *wts;
  const size_t WF = 512 * 1024;                         // 2 MB of weights per phase
  hipMalloc(&gran, 2 * 2048 * sizeof(u64)); hipMemset(gran, 0, 2 * 2048 * sizeof(u64));
  hipMalloc(&err, 4); hipMemset(err, 0, 4);
  hipMalloc(&out, 8); hipMalloc(&wts, WF * 4); hipMemset(wts, 0, WF * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned epoch = 1;
  const int phases = 66;
  printf("CUs %d; %d phases per launch (2 x the 33 edges of one MT decode step)\n", cus, phases);
  for (int N : {512, 2048})
    for (int G : {32, 64, 128, 256})
      for (size_t wf : {(size_t)0, WF}) {
        if (G > cus) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(chain, dim3(G), dim3(256), 2048 * 4, 0, gran, wts, wf, N, phases, epoch, err, out);
          hipEventRecord(e1); hipEventSynchronize(e1);
          epoch += phases;
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          if (rep && ms < best) best = ms;
        }
        unsigned herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("N = %4d floats, G = %3d workgroups, %s weights per phase: %7.2f us per phase (%.1f us per launch)  time-outs %u\n", N, G,
               wf ? "2 MB of" : "no    ", 1e3 * best / phases, 1e3 * best, herr);
      }
  return 0;
}
