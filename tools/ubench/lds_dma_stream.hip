// LDS-DMA streaming in the access pattern of the GEMM kernel, without the MFMAs: every workgroup (256 threads) streams the
// k-tiles of a [128 rows x K] k-major A panel of its own and of a shared B panel (32 KiB per k-tile, 8 pieces of 1 KiB per
// wave) through a ring of DEPTH stages, with one counted s_waitcnt + s_barrier per k-tile and `work` dummy MFMAs per wave and
// k-tile in between.  Prints the time per k-tile per workgroup and the aggregate GB/s for DEPTH 2 / 3 / 4 and several grids:
// tells whether the GEMM's ~1 us per k-tile of a lone workgroup is the latency of a too shallow ring or the throughput of the
// global -> LDS path.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_stream lds_dma_stream.hip && ./lds_dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OFF>
__device__ __forceinline__ void dma1k(const char* g, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* A, const char* B, int K2 /* bytes per row */, int ktiles, int work,
                                                     float* sink, int gx /* n-tiles: 0 = every workgroup its own A panel */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // piece p of a tile = rows 8p..8p+7, 128 bytes each; this wave moves pieces 4w..4w+3 of A and of B
  // the GEMM's tile order: workgroups of one XCD (blockIdx & 7) take consecutive tiles, tile -> (m-tile, n-tile)
  int mt = blockIdx.x, nt = 0;
  if (gx > 0) {
    const int total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * q8 + min(xcd, r8) + slot;
    mt = tile / gx; nt = tile % gx;
  }
  const char* a = A + ((size_t)mt * 128 + wave * 32 + (lane >> 3)) * K2 + (lane & 7) * 16;
  const char* b = B + ((size_t)nt * 128 + (wave * 32 + (lane >> 3))) * K2 + (lane & 7) * 16;
  const size_t rstep = (size_t)8 * K2;
  auto issue = [&](int kt) {
    const uint32_t d = lds0 + (kt % DEPTH) * 32768 + wave * 4096;
    const char* pa = a + (size_t)kt * 128;
    const char* pb = b + (size_t)kt * 128;
    dma1k<0>(pa, d); dma1k<0>(pa + rstep, d + 1024); dma1k<0>(pa + 2 * rstep, d + 2048); dma1k<0>(pa + 3 * rstep, d + 3072);
    dma1k<0>(pb, d + 16384); dma1k<0>(pb + rstep, d + 16384 + 1024); dma1k<0>(pb + 2 * rstep, d + 16384 + 2048);
    dma1k<0>(pb + 3 * rstep, d + 16384 + 3072);
  };
  f32x16 acc = {0};
  bf16x8 fa = {0}, fb = {0};
  for (int kt = 0; kt < DEPTH - 1 && kt < ktiles; ++kt) issue(kt);
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + DEPTH - 1 < ktiles) issue(kt + DEPTH - 1);
    // tile kt has landed when at most (DEPTH - 1) younger tiles (8 pieces each) are outstanding
    const int younger = min(DEPTH - 1, ktiles - 1 - kt);
    if (younger >= 3) wait_vm<24>(); else if (younger == 2) wait_vm<16>(); else if (younger == 1) wait_vm<8>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const uint4 v = *reinterpret_cast<const uint4*>(smem + (kt % DEPTH) * 32768 + tid * 16);
    fa = __builtin_bit_cast(bf16x8, v);
    for (int w = 0; w < work; ++w) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    __builtin_amdgcn_s_barrier();
  }
  if (acc[0] == 123.f) sink[0] = acc[1];
}

// Hybrid fill: the A half of every k-tile by LDS-DMA (4 pieces per wave), the B half through registers (4 global_load_dwordx4 per
// lane issued one k-tile ahead from inline asm, ds_write_b128 after the barrier).  MODE 1 = hybrid, MODE 2 = both halves through
// registers.  Answers whether the ~20 B/clk of the all-DMA stream is a limit of the DMA path or of the CU's vector-memory path.
__device__ __forceinline__ void gload16(uint4& r, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
template <int MODE>
__global__ __launch_bounds__(256) void stream_hybrid_kernel(const char* A, const char* B, int K2, int ktiles, int work, float* sink, int gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  int mt = blockIdx.x, nt = 0;
  if (gx > 0) {
    const int total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * q8 + min(xcd, r8) + slot;
    mt = tile / gx; nt = tile % gx;
  }
  const char* a = A + ((size_t)mt * 128 + wave * 32 + (lane >> 3)) * K2 + (lane & 7) * 16;
  const char* b = B + ((size_t)nt * 128 + (wave * 32 + (lane >> 3))) * K2 + (lane & 7) * 16;
  const size_t rstep = (size_t)8 * K2;
  uint4 rb[4], ra[4];
  auto issue = [&](int kt) {
    const uint32_t d = lds0 + (kt & 1) * 32768 + wave * 4096;
    const char* pa = a + (size_t)kt * 128;
    const char* pb = b + (size_t)kt * 128;
    if (MODE == 1) { dma1k<0>(pa, d); dma1k<0>(pa + rstep, d + 1024); dma1k<0>(pa + 2 * rstep, d + 2048); dma1k<0>(pa + 3 * rstep, d + 3072); }
    else { gload16(ra[0], pa); gload16(ra[1], pa + rstep); gload16(ra[2], pa + 2 * rstep); gload16(ra[3], pa + 3 * rstep); }
    gload16(rb[0], pb); gload16(rb[1], pb + rstep); gload16(rb[2], pb + 2 * rstep); gload16(rb[3], pb + 3 * rstep);
  };
  f32x16 acc = {0};
  bf16x8 fa = {0}, fb = {0};
  issue(0);
  for (int kt = 0; kt < ktiles; ++kt) {
    wait_vm<0>();                               // tile kt: DMA half landed, register half arrived
    char* st = smem + (kt & 1) * 32768 + wave * 4096 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<uint4*>(st + 16384 + j * 1024) = rb[j];
      if (MODE == 2) *reinterpret_cast<uint4*>(st + j * 1024) = ra[j];
    }
    if (kt + 1 < ktiles) issue(kt + 1);         // next tile's loads fly during this tile's compute
    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the ds_writes
    __builtin_amdgcn_s_barrier();
    const uint4 v = *reinterpret_cast<const uint4*>(smem + (kt & 1) * 32768 + tid * 16);
    fa = __builtin_bit_cast(bf16x8, v);
    for (int w = 0; w < work; ++w) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    __builtin_amdgcn_s_barrier();
  }
  if (acc[0] == 123.f) sink[0] = acc[1];
}

// the same 32 KiB k-tiles issued by EIGHT waves (4 pieces each) instead of four (8 each): does the fill rate of a CU depend on how many
// waves issue?
__global__ __launch_bounds__(512) void stream_w8_kernel(const char* A, const char* B, int K2, int ktiles, int work, float* sink, int gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  int mt = blockIdx.x, nt = 0;
  if (gx > 0) {
    const int total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * q8 + min(xcd, r8) + slot;
    mt = tile / gx; nt = tile % gx;
  }
  const bool isb = wave >= 4;                       // waves 0-3 move the A tile, 4-7 the B tile, 4 pieces each
  const int w4 = wave & 3;
  const char* src = (isb ? B + (size_t)nt * 128 * K2 : A + (size_t)mt * 128 * K2) + ((size_t)w4 * 32 + (lane >> 3)) * K2 + (lane & 7) * 16;
  const size_t rstep = (size_t)8 * K2;
  auto issue = [&](int kt) {
    const uint32_t d = lds0 + (kt & 1) * 32768 + (isb ? 16384 : 0) + w4 * 4096;
    const char* ps = src + (size_t)kt * 128;
    dma1k<0>(ps, d); dma1k<0>(ps + rstep, d + 1024); dma1k<0>(ps + 2 * rstep, d + 2048); dma1k<0>(ps + 3 * rstep, d + 3072);
  };
  f32x16 acc = {0};
  bf16x8 fa = {0}, fb = {0};
  issue(0);
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + 1 < ktiles) { issue(kt + 1); wait_vm<4>(); } else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const uint4 v = *reinterpret_cast<const uint4*>(smem + (kt & 1) * 32768 + (tid & 255) * 16);
    fa = __builtin_bit_cast(bf16x8, v);
    for (int w = 0; w < work; ++w) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    __builtin_amdgcn_s_barrier();
  }
  if (acc[0] == 123.f) sink[0] = acc[1];
}

// k-tiles of 32 elements (64-byte rows, 16 KiB per k-tile, two stages = 32 KiB): up to four workgroups per CU.  Does the aggregate
// fill rate go up with the occupancy?
__global__ __launch_bounds__(256) void stream_k32_kernel(const char* A, const char* B, int K2, int ktiles32, int work, float* sink, int gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  int mt = blockIdx.x, nt = 0;
  if (gx > 0) {
    const int total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * q8 + min(xcd, r8) + slot;
    mt = tile / gx; nt = tile % gx;
  }
  // piece = 16 rows x 64 bytes; this wave moves pieces 2w, 2w+1 of A and of B (32 rows each)
  const char* a = A + ((size_t)mt * 128 + wave * 32 + (lane >> 2)) * K2 + (lane & 3) * 16;
  const char* b = B + ((size_t)nt * 128 + wave * 32 + (lane >> 2)) * K2 + (lane & 3) * 16;
  const size_t rstep = (size_t)16 * K2;
  auto issue = [&](int kt) {
    const uint32_t d = lds0 + (kt & 1) * 16384 + wave * 2048;
    const char* pa = a + (size_t)kt * 64;
    const char* pb = b + (size_t)kt * 64;
    dma1k<0>(pa, d); dma1k<0>(pa + rstep, d + 1024);
    dma1k<0>(pb, d + 8192); dma1k<0>(pb + rstep, d + 8192 + 1024);
  };
  f32x16 acc = {0};
  bf16x8 fa = {0}, fb = {0};
  issue(0);
  for (int kt = 0; kt < ktiles32; ++kt) {
    if (kt + 1 < ktiles32) { issue(kt + 1); wait_vm<4>(); } else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const uint4 v = *reinterpret_cast<const uint4*>(smem + (kt & 1) * 16384 + tid * 16);
    fa = __builtin_bit_cast(bf16x8, v);
    for (int w = 0; w < work; ++w) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    __builtin_amdgcn_s_barrier();
  }
  if (acc[0] == 123.f) sink[0] = acc[1];
}

int main(int argc, char** argv) {
  const int K = argc > 2 ? atoi(argv[2]) : 3072, pad = argc > 1 ? atoi(argv[1]) : 0, K2 = K * 2 + pad, ktiles = K / 64, maxg = 1080;
  printf("K = %d, row stride %d bytes (%d lines of 128 B)\n", K, K2, K2 / 128);
  char *A, *B; float* sink;
  hipMalloc(&A, (size_t)maxg * 128 * K2); hipMalloc(&B, (size_t)32 * 128 * K2); hipMalloc(&sink, 64);
  hipMemset(A, 0, (size_t)maxg * 128 * K2); hipMemset(B, 0, (size_t)32 * 128 * K2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int work : {0, 16}) {
    for (int depth = 2; depth <= 2; ++depth) {
      for (int cfg = 0; cfg < 9; ++cfg) {
        const int grids[9] = {256, 270, 512, 1024, 270, 540, 810, 1080, 1024}, gxs[9] = {0, 0, 0, 0, 6, 6, 18, 24, 32};
        const int grid = grids[cfg], gx = gxs[cfg];
        auto launch = [&]() {
          size_t sh = (size_t)depth * 32768;
          if (depth == 2) { hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), sh, 0, A, B, K2, ktiles, work, sink, gx); }
          if (depth == 3) { hipFuncSetAttribute((const void*)stream_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                            hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(256), sh, 0, A, B, K2, ktiles, work, sink, gx); }
          if (depth == 4) { hipFuncSetAttribute((const void*)stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                            hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(256), sh, 0, A, B, K2, ktiles, work, sink, gx); }
        };
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 20;
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / n;
        printf("mfma/wave/k-tile %2d  depth %d  grid %4d n-tiles %2d : %7.1f us  = %5.2f us per k-tile per workgroup, %6.0f GB/s moved\n", work,
               depth, grid, gx, us, us / ktiles, (double)grid * ktiles * 32768 / us / 1e3);
      }
    }
  }
  for (int work : {0, 8})
    for (int cfg = 0; cfg < 4; ++cfg) {
      const int grids[4] = {270, 540, 810, 1080}, gxs[4] = {6, 6, 18, 24};
      const int grid = grids[cfg], gx = gxs[cfg];
      auto launch = [&]() { hipLaunchKernelGGL(stream_w8_kernel, dim3(grid), dim3(512), 65536, 0, A, B, K2, ktiles, work, sink, gx); };
      for (int i = 0; i < 3; ++i) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int n = 20;
      for (int i = 0; i < n; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / n;
      printf("mfma/wave/k-tile %2d  EIGHT WAVES per 32 KiB k-tile  grid %4d n-tiles %2d : %7.1f us  = %5.2f us per k-tile per workgroup, %6.0f GB/s moved\n",
             work, grid, gx, us, us / ktiles, (double)grid * ktiles * 32768 / us / 1e3);
    }
  for (int work : {0, 8})
    for (int cfg = 0; cfg < 4; ++cfg) {
      const int grids[4] = {270, 540, 810, 1080}, gxs[4] = {6, 6, 18, 24};
      const int grid = grids[cfg], gx = gxs[cfg];
      auto launch = [&]() { hipLaunchKernelGGL(stream_k32_kernel, dim3(grid), dim3(256), 32768, 0, A, B, K2, 2 * ktiles, work, sink, gx); };
      for (int i = 0; i < 3; ++i) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int n = 20;
      for (int i = 0; i < n; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / n;
      printf("mfma/wave/k-tile %2d  K-TILE 32, 32 KiB per workgroup  grid %4d n-tiles %2d : %7.1f us  = %5.2f us per 64 k per workgroup, %6.0f GB/s moved\n",
             work, grid, gx, us, us / ktiles, (double)grid * ktiles * 32768 / us / 1e3);
    }
  for (int mode = 1; mode <= 2; ++mode)
    for (int work : {0, 16})
      for (int cfg = 0; cfg < 4; ++cfg) {
        const int grids[4] = {270, 540, 810, 1080}, gxs[4] = {6, 6, 18, 24};
        const int grid = grids[cfg], gx = gxs[cfg];
        auto launch = [&]() {
          if (mode == 1) hipLaunchKernelGGL(stream_hybrid_kernel<1>, dim3(grid), dim3(256), 65536, 0, A, B, K2, ktiles, work, sink, gx);
          else hipLaunchKernelGGL(stream_hybrid_kernel<2>, dim3(grid), dim3(256), 65536, 0, A, B, K2, ktiles, work, sink, gx);
        };
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 20;
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / n;
        printf("mfma/wave/k-tile %2d  %s  grid %4d n-tiles %2d : %7.1f us  = %5.2f us per k-tile per workgroup, %6.0f GB/s moved\n", work,
               mode == 1 ? "HYBRID (A dma, B regs)" : "REGISTERS (A and B)   ", grid, gx, us, us / ktiles, (double)grid * ktiles * 32768 / us / 1e3);
      }
  return 0;
}
