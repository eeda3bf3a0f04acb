// Does the B-operand register variety / accumulator reuse pattern of the pair-heads loop cost MFMA rate?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const uint4* src, int iters) {
  bf16x8 b[24];
  for (int i = 0; i < 24; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x + i * 64) & 1023]);
  bf16x8 a = __builtin_bit_cast(bf16x8, src[threadIdx.x & 1023]);
  f32x16 lg;
  for (int i = 0; i < 16; ++i) lg[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, MODE == 0 ? b[0] : b[ks], z, 0, 0, 0);
    if (MODE == 2) {   // like the kernel: read the accumulator, convert, feed 2 more MFMAs
      bf16x8 y0, y1;
      for (int i = 0; i < 8; ++i) { y0[i] = (__bf16)z[i]; y1[i] = (__bf16)z[8 + i]; }
      lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, y0, lg, 0, 0, 0);
      lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, y1, lg, 0, 0, 0);
    } else {
      for (int i = 0; i < 16; ++i) lg[i] += z[i];
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += lg[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, const uint4* src) {
  const int threads = 512, blocks = 256 * 4;
  float* out; hipMalloc(&out, sizeof(float) * threads * blocks);
  int iters = 600;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 100 * 1024, 0, out, src, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 100 * 1024, 0, out, src, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfmas = (double)blocks * (threads / 64) * iters * (MODE == 2 ? 26 : 24);
  printf("%-40s: %8.3f ms  %8.1f TF/s  (%s)\n", name, ms, mfmas * 32768.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
  hipFree(out);
}
#include <cstring>
int main(int argc, char** argv) {
  uint4* src; hipMalloc(&src, 16 * 1024);
  {
    unsigned short h[8192];
    unsigned x = 12345u;
    for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; float f = ((x >> 8) * (1.0f / 8388608.0f)) - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    if (argc > 1) hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice); else hipMemset(src, 0x3c, 16 * 1024);
    printf("data: %s\n", argc > 1 ? "uniform random [-1,1) bf16" : "constant");
  }
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  run<0>("same B, 1 WG/CU (100 KB LDS)", src);
  run<1>("24 different B regs", src);
  run<2>("24 different B + cvt + 2 MFMA (kernel-like)", src);
  return 0;
}
