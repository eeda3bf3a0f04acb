// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 values == their own element index; every lane reads 8 bytes at
// lane*8 (+ an optional stride pattern) and prints the 4 u16 it received.  hipcc --offload-arch=gfx950 -O2 -o tr_probe tr_b16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int pitch_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // [4 keys][16 cols] block per 16-lane group: lane i -> key i>>2, cols 4*(i&3)..+3 ; groups side by side in columns
  const int key = (l & 15) >> 2, col = 16 * (l >> 4) + 4 * (l & 3);
  uint32_t addr = (uint32_t)(uintptr_t)lds + key * pitch_bytes + col * 2;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  const int pitch = 256;  // bytes per key row = 128 u16
  probe<<<1, 64>>>(d, pitch);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf("  (key %d, col %3d)", h[l * 4 + e] / 128, h[l * 4 + e] % 128);
    printf("\n");
  }
  return 0;
}
