// Calibrates the MFMA issue rate on this pool: dependent chains vs independent accumulators, waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int threads, int blocks, const char* name) {
  float* out; hipMalloc(&out, sizeof(float) * threads * blocks);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfmas = (double)blocks * (threads / 64) * iters * 24;
  printf("%-28s threads=%3d blocks=%4d: %8.3f ms  %8.1f TF/s\n", name, threads, blocks, ms, mfmas * 32768.0 / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(256, 256, "1 acc, 1 wave/SIMD");
  run<1>(512, 256, "1 acc, 2 waves/SIMD");
  run<1>(256, 512, "1 acc, 2 WG x 4 waves");
  run<2>(512, 256, "2 acc, 2 waves/SIMD");
  run<4>(256, 256, "4 acc, 1 wave/SIMD");
  run<4>(512, 256, "4 acc, 2 waves/SIMD");
  run<1>(1024, 256, "1 acc, 4 waves/SIMD");
  run<2>(1024, 256, "2 acc, 4 waves/SIMD");
  return 0;
}
