// LDS atomic throughput on gfx950: float add vs u32 add vs u64 add, conflict-free (address = bin*32 + lane%32 style)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ unsigned long long h[160 * 32];
  for (int i = threadIdx.x; i < 160 * 32; i += 256) h[i] = 0;
  __syncthreads();
  const int rep = threadIdx.x & 31;
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    const int bin = (x >> 24) % 160;
    if (MODE == 0) atomicAdd(reinterpret_cast<float*>(h) + bin * 32 + rep, 1.0f);
    else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(h) + bin * 32 + rep, 1u);
    else atomicAdd(h + bin * 32 + rep, 1ull);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)h[5];
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4096, blocks = 2048;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) k<0><<<blocks, 256>>>(d, iters); else if (mode == 1) k<1><<<blocks, 256>>>(d, iters); else k<2><<<blocks, 256>>>(d, iters);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d (%s): %.3f ms  -> %.1f G lane-atomics/s\n", mode, mode == 0 ? "f32" : mode == 1 ? "u32" : "u64", ms,
           (double)blocks * 256 * iters / ms / 1e6);
  }
  return 0;
}
