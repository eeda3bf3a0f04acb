// Does a wave's own VALU work issue in the shadow of its MFMA, and do two such waves on one SIMD fill each other's gaps?  (gfx950)
// Every wave runs [MFMA, NV x VALU, MFMA, NV x VALU] per iteration (the VALU instructions independent of each other and of the MFMAs);
// cycles per iteration by s_memtime (= shader clock), at one and at two waves per SIMD, with and without the MFMAs.
//    hipcc --offload-arch=gfx950 -O3 valu_mfma_overlap.hip -o valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NV, int KIND> __device__ __forceinline__ void valu(float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k & 7]));
    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k & 7]));
    else asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(v[k & 7]) : "v"(v[(k + 1) & 7]), "v"(v[(k + 3) & 7]), "v"(v[(k + 5) & 7]));   // three distinct source registers
  }
}
template <int NV, int KIND, bool MF, bool AG = false>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int iters) {
  f32x16 a0 = {}, a1 = {};
  u32x4 x = {threadIdx.x, 1, 2, 3}, y = {5, 6, 7, threadIdx.x};
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, (float)threadIdx.x};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MF) { if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a0) : "v"(x), "v"(y)); else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); }
    valu<NV, KIND>(v);
    if (MF) { if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a1) : "v"(x), "v"(y)); else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y)); }
    valu<NV, KIND>(v);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  float s = a0[0] + a1[1];
  for (int q = 0; q < 8; ++q) s += v[q];
  if (s == 0.123f) out[1] = 1;
}
template <int NV, int KIND, bool MF, bool AG = false> static double run(unsigned long long* d, int waves) {
  const int it = 4000;
  hipLaunchKernelGGL((k<NV, KIND, MF, AG>), dim3(256), dim3(waves * 64), 0, 0, d, it); hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NV, KIND, MF, AG>), dim3(256), dim3(waves * 64), 0, 0, d, it); hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  return (double)h / it;
}
template <int NV, int KIND> static void row(unsigned long long* d) {
  const double a = run<NV, KIND, true>(d, 4), b = run<NV, KIND, false>(d, 4), c = run<NV, KIND, true>(d, 8), e = run<NV, KIND, false>(d, 8);
  if (KIND == 2) {
    printf("v_fma_f32 (3 distinct sources) x %2d:  2 waves/SIMD: both %6.1f  VALU only %6.1f   | with AGPR accumulators: both %6.1f\n", NV, c, e, run<NV, KIND, true, true>(d, 8));
    return;
  }
  printf("%s x %2d behind each MFMA:  1 wave/SIMD: both %6.1f  VALU only %6.1f   |   2 waves/SIMD: both %6.1f  VALU only %6.1f   cycles per iteration (2 MFMA + %d VALU per wave)\n",
         KIND == 0 ? "v_fma_f32" : "v_exp_f32", NV, a, b, c, e, 2 * NV);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<16, 0, true>), dim3(256), dim3(512), 0, 0, d, 4000);   // warm the clocks
  hipDeviceSynchronize();
  row<0, 0>(d); row<2, 0>(d); row<4, 0>(d); row<6, 0>(d); row<8, 0>(d); row<12, 0>(d); row<16, 0>(d);
  row<1, 1>(d); row<2, 1>(d); row<4, 1>(d);
  row<4, 2>(d); row<8, 2>(d); row<12, 2>(d); row<16, 2>(d);
  return 0;
}
