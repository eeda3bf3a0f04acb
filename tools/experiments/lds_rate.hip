// LDS read throughput per CU on gfx950: ds_read_b128 against ds_read_b64_tr_b16 (and plain ds_read_b64), 4 / 8 waves per CU,
// every lane a distinct 16-byte (8-byte) slot, no bank conflicts.  hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(unsigned long long* out, int iters) {
  extern __shared__ char smem[];
  const unsigned base = (unsigned)(size_t)smem;   // LDS addresses start at 0 for the dynamic segment in practice
  const unsigned a = (threadIdx.x & 63) * (MODE == 0 ? 16 : 8) + (threadIdx.x >> 6) * 4096;
  u32x4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
  u32x2 q0 = {0, 0}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0, q7 = q0;
  (void)base;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a));
    } else if (MODE == 1) {
      asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\tds_read_b64_tr_b16 %2, %8 offset:1024\n\tds_read_b64_tr_b16 %3, %8 offset:1536\n\t"
                   "ds_read_b64_tr_b16 %4, %8 offset:2048\n\tds_read_b64_tr_b16 %5, %8 offset:2560\n\tds_read_b64_tr_b16 %6, %8 offset:3072\n\tds_read_b64_tr_b16 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(a));
    } else {
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                   "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(a));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (r0.x + r1.x + r2.x + r3.x + q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x == 0x12345) out[1] = 1;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  const int iters = 20000;
  for (int waves : {4, 8}) for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 65536, 0, d, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), 65536, 0, d, iters);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(waves * 64), 65536, 0, d, iters);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double bytes = (double)iters * 4096 * waves;   // per CU
    printf("%d waves/CU  %-22s %8.3f ms  %7.1f B/clk/CU (s_memtime ticks %llu -> %.1f bytes / tick)  %6.1f GB/s/CU\n", waves,
           mode == 0 ? "ds_read_b128" : mode == 1 ? "ds_read_b64_tr_b16" : "ds_read_b64", ms, bytes / (ms * 1e-3) / 2.4e9, h[0], bytes / h[0], bytes / ms * 1e-6);
  }
  return 0;
}
