// K4: relative-position buckets and the summed 1-D + 2-D attention bias, computed once per forward
// and shared by all layers (reference: modeling_layoutlmv3.py:586-676 builds one-hot [B,T,T,bins]
// tensors and multiplies them by Linear(bins -> heads); algorithmically it is a three-table lookup).
#include "common.h"

namespace peneo {

__device__ __forceinline__ uint8_t bucket_of(int d, const uint8_t* lut, int lut_len, int half) {
  int n = d < 0 ? -d : d;
  if (n > lut_len - 1) n = lut_len - 1;
  return (uint8_t)(lut[n] + (d > 0 ? half : 0));
}

__global__ __launch_bounds__(256) void relpos_buckets_kernel(const int32_t* pos, const int32_t* xs, const int32_t* ys, int T,
                                                             const uint8_t* lut1, int n1, int half1, const uint8_t* lut2,
                                                             int n2, int half2, uint8_t* bk1, uint8_t* bkx, uint8_t* bky) {
  const int64_t row = blockIdx.x;  // b * T + i
  const int64_t b = row / T;
  const int32_t pi = pos ? pos[row] : 0, xi = xs ? xs[row] : 0, yi = ys ? ys[row] : 0;
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    const int64_t o = row * T + j, cj = b * T + j;
    if (bk1) bk1[o] = bucket_of(pos[cj] - pi, lut1, n1, half1);
    if (bkx) bkx[o] = bucket_of(xs[cj] - xi, lut2, n2, half2);
    if (bky) bky[o] = bucket_of(ys[cj] - yi, lut2, n2, half2);
  }
}

// bias rows are padded to Tp columns; padding and keys with key_mask == 0 hold MASKED_BIAS so the attention
// kernels need no mask logic of their own (the reference adds finfo.min: modeling_layoutlmv3.py:1126-1128)
constexpr float MASKED_BIAS = -1.0e30f;
template <typename T>
__global__ __launch_bounds__(256) void relpos_bias_fwd_kernel(const uint8_t* bk1, const uint8_t* bkx, const uint8_t* bky,
                                                              const float* w1, int bins1, const float* wx, const float* wy,
                                                              int bins2, float scale, int nh, int Tn, int Tp,
                                                              const int32_t* key_mask, T* bias) {
  extern __shared__ float tab[];  // [nh][bins1] [nh][bins2] [nh][bins2]
  float* t1 = tab;
  float* tx = t1 + nh * bins1;
  float* ty = tx + nh * bins2;
  for (int i = threadIdx.x; i < nh * bins1; i += blockDim.x) t1[i] = w1 ? w1[i] * scale : 0.f;
  for (int i = threadIdx.x; i < nh * bins2; i += blockDim.x) { tx[i] = wx ? wx[i] * scale : 0.f; ty[i] = wy ? wy[i] * scale : 0.f; }
  __syncthreads();
  const int64_t row = blockIdx.x;  // b * T + i
  const int64_t b = row / Tn, i = row % Tn;
  for (int j = threadIdx.x; j < Tp; j += blockDim.x) {
    const bool valid = j < Tn && (!key_mask || key_mask[b * Tn + j] != 0);
    const int64_t o = row * Tn + j;
    const int a = (valid && bk1) ? bk1[o] : 0, bx = (valid && bkx) ? bkx[o] : 0, by = (valid && bky) ? bky[o] : 0;
    for (int h = 0; h < nh; ++h) {
      float v = 0.f;
      if (bk1) v += t1[h * bins1 + a];
      if (bkx) v += tx[h * bins2 + bx] + ty[h * bins2 + by];
      Elem<T>::store(bias + (((b * nh + h) * Tn + i) * (int64_t)Tp + j), valid ? v : MASKED_BIAS);
    }
  }
}

// one block per (b, h, 32-row slab): LDS histograms replicated 32x (slot = bin*32 + lane%32: every lane of a half-wave
// hits its own bank and its own address, so the LDS atomics neither bank-conflict nor serialise on popular buckets),
// folded at the end into one global atomic per bin.
constexpr int RB_REP = 32;
// The LDS histograms accumulate in 64-bit fixed point (2^-32 units): on gfx950 an LDS float atomic add runs at ~0.2 T
// lane-ops/s, the integer one (ds_add_u64) at ~4.6 T (tools/ubench/lds_atomic.hip), and integer sums are order-independent.
__device__ __forceinline__ unsigned long long rb_fix(float v) { return (unsigned long long)__float2ll_rn(v * 4294967296.0f); }
__device__ __forceinline__ float rb_unfix(unsigned long long q) { return (float)((double)(long long)q * (1.0 / 4294967296.0)); }
__global__ __launch_bounds__(256) void relpos_bias_bwd_kernel(const float* g, const uint8_t* bk1, const uint8_t* bkx,
                                                              const uint8_t* bky, float* dw1, int bins1, float* dwx, float* dwy,
                                                              int bins2, float scale, int nh, int Tn, int64_t ldg) {
  extern __shared__ unsigned long long hist[];  // [bins1 | bins2 | bins2][RB_REP], fixed point
  unsigned long long* h1 = hist;
  unsigned long long* hx = h1 + bins1 * RB_REP;
  unsigned long long* hy = hx + bins2 * RB_REP;
  const int nb = bins1 + 2 * bins2;
  for (int i = threadIdx.x; i < nb * RB_REP; i += blockDim.x) hist[i] = 0ull;
  __syncthreads();
  const int rep = threadIdx.x & (RB_REP - 1);
  const int slabs = (Tn + 31) / 32;
  const int slab = blockIdx.x % slabs;
  const int h = (blockIdx.x / slabs) % nh;
  const int64_t b = blockIdx.x / ((int64_t)slabs * nh);
  const int i0 = slab * 32, i1 = min(Tn, i0 + 32);
  // 4 rows per step with every load issued before the first LDS atomic: the loop is latency-bound otherwise
  for (int i = i0; i < i1; i += 4) {
    for (int j = threadIdx.x; j < Tn; j += blockDim.x) {
      float v[4]; int c1[4], cx[4], cy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ii = min(i + u, i1 - 1);
        const int64_t brow = (b * Tn + ii) * (int64_t)Tn + j;
        v[u] = (i + u < i1) ? g[(((b * nh + h) * Tn + ii) * ldg) + j] : 0.f;
        c1[u] = bk1 ? bk1[brow] : 0;
        cx[u] = bkx ? bkx[brow] : 0;
        cy[u] = bkx ? bky[brow] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long q = rb_fix(v[u]);
        if (bk1) atomicAdd(h1 + c1[u] * RB_REP + rep, q);
        if (bkx) { atomicAdd(hx + cx[u] * RB_REP + rep, q); atomicAdd(hy + cy[u] * RB_REP + rep, q); }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    unsigned long long acc = 0ull;
#pragma unroll 8
    for (int r = 0; r < RB_REP; ++r) acc += hist[i * RB_REP + ((r + i) & (RB_REP - 1))];
    const float sum = rb_unfix(acc) * scale;
    if (i < bins1) { if (dw1) atomicAdd(dw1 + h * bins1 + i, sum); }
    else if (i < bins1 + bins2) { if (dwx) atomicAdd(dwx + h * bins2 + (i - bins1), sum); }
    else if (dwy) atomicAdd(dwy + h * bins2 + (i - bins1 - bins2), sum);
  }
}

// same reduction from L per-layer bf16 dS^T buffers [B, nh, T keys, Tp queries]: block = (b, h, 32-key slab).
// Work item = 8 consecutive queries of one key row: L 16-byte loads (one per layer) summed in fp32, three 8-byte bucket
// loads (the transposed maps are padded to row stride Tp, so both are aligned), then the LDS integer atomics.
__global__ __launch_bounds__(256) void relpos_bias_bwd_layers_kernel(const bf16_t* ds, int L, int64_t lstride, const uint8_t* bk1,
                                                                     const uint8_t* bkx, const uint8_t* bky, float* dw1, int bins1,
                                                                     float* dwx, float* dwy, int bins2, float scale, int nh, int Tn,
                                                                     int Tp) {
  extern __shared__ unsigned long long hist[];  // [bins1 | bins2 | bins2][RB_REP], fixed point
  unsigned long long* h1 = hist;
  unsigned long long* hx = h1 + bins1 * RB_REP;
  unsigned long long* hy = hx + bins2 * RB_REP;
  const int nb = bins1 + 2 * bins2;
  for (int i = threadIdx.x; i < nb * RB_REP; i += blockDim.x) hist[i] = 0ull;
  __syncthreads();
  const int rep = threadIdx.x & (RB_REP - 1);
  const int slabs = (Tn + 31) / 32;
  const int slab = blockIdx.x % slabs;
  const int h = (blockIdx.x / slabs) % nh;
  const int64_t b = blockIdx.x / ((int64_t)slabs * nh);
  const int j0 = slab * 32, nrow = min(Tn, j0 + 32) - j0;
  const int vpr = Tp / 8;
  for (int item = threadIdx.x; item < nrow * vpr; item += blockDim.x) {
    const int j = j0 + item / vpr, i0 = (item % vpr) * 8;
    if (i0 >= Tn) continue;
    const bf16_t* row = ds + (((b * nh + h) * Tn + j) * (int64_t)Tp) + i0;
    const int64_t brow = (b * Tn + j) * (int64_t)Tp + i0;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    int l = 0;
    for (; l + 4 <= L; l += 4) {   // four layers in flight
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(row + (int64_t)(l + u) * lstride);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack16<bf16_t>(r[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += f[e];
      }
    }
    for (; l < L; ++l) {
      float f[8];
      unpack16<bf16_t>(*reinterpret_cast<const uint4*>(row + (int64_t)l * lstride), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
    const uint2 z2 = make_uint2(0u, 0u);
    const uint2 c1 = bk1 ? *reinterpret_cast<const uint2*>(bk1 + brow) : z2;
    const uint2 cx = bkx ? *reinterpret_cast<const uint2*>(bkx + brow) : z2;
    const uint2 cy = bkx ? *reinterpret_cast<const uint2*>(bky + brow) : z2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (i0 + e < Tn) {
        const unsigned long long q = rb_fix(v[e]);
        const int sh = (e & 3) * 8;
        if (bk1) atomicAdd(h1 + (((e < 4 ? c1.x : c1.y) >> sh) & 0xff) * RB_REP + rep, q);
        if (bkx) {
          atomicAdd(hx + (((e < 4 ? cx.x : cx.y) >> sh) & 0xff) * RB_REP + rep, q);
          atomicAdd(hy + (((e < 4 ? cy.x : cy.y) >> sh) & 0xff) * RB_REP + rep, q);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    unsigned long long acc = 0ull;
#pragma unroll 8
    for (int r = 0; r < RB_REP; ++r) acc += hist[i * RB_REP + ((r + i) & (RB_REP - 1))];
    const float sum = rb_unfix(acc) * scale;
    if (i < bins1) { if (dw1) atomicAdd(dw1 + h * bins1 + i, sum); }
    else if (i < bins1 + bins2) { if (dwx) atomicAdd(dwx + h * bins2 + (i - bins1), sum); }
    else if (dwy) atomicAdd(dwy + h * bins2 + (i - bins1 - bins2), sum);
  }
}

}  // namespace peneo
using namespace peneo;

extern "C" int peneo_relpos_bias_bwd_layers(const void* ds, int L, int64_t layer_stride, const uint8_t* bk1_t,
                                            const uint8_t* bkx_t, const uint8_t* bky_t, float* dw1, int bins1, float* dwx,
                                            float* dwy, int bins2, float scale, int B, int nh, int T, int Tp,
                                            peneo_stream_t stream) {
  PENEO_REQUIRE(ds && L > 0 && B > 0 && nh > 0 && T > 0 && Tp >= T, "peneo_relpos_bias_bwd_layers: bad arguments");
  PENEO_REQUIRE((bkx_t != nullptr) == (bky_t != nullptr), "peneo_relpos_bias_bwd_layers: 2-D inputs mismatch");
  PENEO_REQUIRE(L == 1 || layer_stride >= (int64_t)B * nh * T * Tp, "peneo_relpos_bias_bwd_layers: layer stride too small");
  PENEO_REQUIRE(Tp % 8 == 0 && (reinterpret_cast<uintptr_t>(ds) & 15) == 0 && (layer_stride % 8) == 0,
                "peneo_relpos_bias_bwd_layers: dS^T rows must be 16-byte aligned (Tp multiple of 8)");
  PENEO_REQUIRE(((reinterpret_cast<uintptr_t>(bk1_t) | reinterpret_cast<uintptr_t>(bkx_t) | reinterpret_cast<uintptr_t>(bky_t)) & 7) == 0,
                "peneo_relpos_bias_bwd_layers: bucket maps must be 8-byte aligned");
  size_t sh = sizeof(unsigned long long) * (size_t)(bins1 + 2 * bins2) * RB_REP;
  PENEO_REQUIRE(sh <= 64 * 1024, "peneo_relpos_bias_bwd_layers: tables too large for LDS");
  int slabs = (T + 31) / 32;
  dim3 grid((unsigned)((int64_t)B * nh * slabs));
  hipLaunchKernelGGL(relpos_bias_bwd_layers_kernel, grid, dim3(256), sh, (hipStream_t)stream, (const bf16_t*)ds, L, layer_stride,
                     bk1_t, bkx_t, bky_t, dw1, bins1, dwx, dwy, bins2, scale, nh, T, Tp);
  return check_launch("peneo_relpos_bias_bwd_layers");
}

// Inputs of the bucket kernel and the key mask over text + visual tokens in one launch (they used to be a dozen torch
// element-wise kernels - arange, cat, casts, slice copies - in front of every forward): token t < S is text token t
// (position t, x = bbox[..., 0], y = bbox[..., 3], attended iff attention_mask), t >= S visual token t - S (position t - S,
// the patch's grid coordinates, always attended: modeling_layoutlmv3.py:1052-1080).
__global__ __launch_bounds__(256) void relpos_inputs_kernel(const int64_t* attention_mask, const int64_t* bbox, const int32_t* vx,
                                                            const int32_t* vy, int B, int S, int nv, int32_t* key_mask, int32_t* pos,
                                                            int32_t* xs, int32_t* ys) {
  const int T = S + nv;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T) return;
  const int b = (int)(i / T), t = (int)(i % T);
  const bool text = t < S;
  const int64_t src = (int64_t)b * S + t;
  if (key_mask) key_mask[i] = text ? (attention_mask ? (int32_t)(attention_mask[src] != 0) : 1) : 1;
  if (pos) pos[i] = text ? t : t - S;
  if (xs) xs[i] = text ? (int32_t)bbox[src * 4 + 0] : vx[t - S];
  if (ys) ys[i] = text ? (int32_t)bbox[src * 4 + 3] : vy[t - S];
}

extern "C" int peneo_relpos_inputs(const int64_t* attention_mask, const int64_t* bbox, const int32_t* vx, const int32_t* vy, int B, int S,
                                   int nv, int32_t* key_mask, int32_t* pos, int32_t* xs, int32_t* ys, peneo_stream_t stream) {
  PENEO_REQUIRE(B > 0 && S > 0 && nv >= 0, "peneo_relpos_inputs: bad sizes");
  PENEO_REQUIRE(!(xs || ys) || (xs && ys && bbox && (nv == 0 || (vx && vy))), "peneo_relpos_inputs: 2-D inputs missing");
  const int64_t n = (int64_t)B * (S + nv);
  hipLaunchKernelGGL(relpos_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, attention_mask, bbox, vx,
                     vy, B, S, nv, key_mask, pos, xs, ys);
  return check_launch("peneo_relpos_inputs");
}

extern "C" int peneo_relpos_buckets(const int32_t* pos, const int32_t* xs, const int32_t* ys, int B, int T,
                                    const uint8_t* lut1, int lut1_len, int half1, const uint8_t* lut2, int lut2_len,
                                    int half2, uint8_t* bk1, uint8_t* bkx, uint8_t* bky, peneo_stream_t stream) {
  PENEO_REQUIRE(B > 0 && T > 0, "peneo_relpos_buckets: empty problem");
  if (bk1) PENEO_REQUIRE(pos && lut1 && lut1_len > 0, "peneo_relpos_buckets: 1-D inputs missing");
  if (bkx || bky) PENEO_REQUIRE(bkx && bky && xs && ys && lut2 && lut2_len > 0, "peneo_relpos_buckets: 2-D inputs missing");
  hipLaunchKernelGGL(relpos_buckets_kernel, dim3((unsigned)((int64_t)B * T)), dim3(256), 0, (hipStream_t)stream, pos, xs, ys, T,
                     lut1, lut1_len, half1, lut2, lut2_len, half2, bk1, bkx, bky);
  return check_launch("peneo_relpos_buckets");
}

extern "C" int peneo_relpos_bias_fwd(int dtype, const uint8_t* bk1, const uint8_t* bkx, const uint8_t* bky, const float* w1,
                                     int bins1, const float* wx, const float* wy, int bins2, float scale, int B, int nh, int T,
                                     int Tp, const int32_t* key_mask, void* bias, peneo_stream_t stream) {
  PENEO_REQUIRE(Tp >= T, "peneo_relpos_bias_fwd: Tp < T");
  PENEO_REQUIRE((dtype == PENEO_F32 || dtype == PENEO_BF16) && bias && B > 0 && nh > 0 && T > 0, "peneo_relpos_bias_fwd: bad arguments");
  PENEO_REQUIRE((bk1 != nullptr) == (w1 != nullptr), "peneo_relpos_bias_fwd: bk1/w1 mismatch");
  PENEO_REQUIRE((bkx != nullptr) == (wx != nullptr) && (bky != nullptr) == (wy != nullptr) && (bkx != nullptr) == (bky != nullptr),
                "peneo_relpos_bias_fwd: 2-D inputs mismatch");
  size_t sh = sizeof(float) * (size_t)nh * (bins1 + 2 * bins2);
  PENEO_REQUIRE(sh <= 64 * 1024, "peneo_relpos_bias_fwd: tables too large for LDS");
  dim3 grid((unsigned)((int64_t)B * T));
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(relpos_bias_fwd_kernel<bf16_t>, grid, dim3(256), sh, (hipStream_t)stream, bk1, bkx, bky, w1, bins1, wx, wy,
                       bins2, scale, nh, T, Tp, key_mask, (bf16_t*)bias);
  else
    hipLaunchKernelGGL(relpos_bias_fwd_kernel<float>, grid, dim3(256), sh, (hipStream_t)stream, bk1, bkx, bky, w1, bins1, wx, wy,
                       bins2, scale, nh, T, Tp, key_mask, (float*)bias);
  return check_launch("peneo_relpos_bias_fwd");
}

extern "C" int peneo_relpos_bias_bwd(const float* g, int64_t ldg, const uint8_t* bk1, const uint8_t* bkx, const uint8_t* bky,
                                     float* dw1, int bins1, float* dwx, float* dwy, int bins2, float scale, int B, int nh, int T,
                                     peneo_stream_t stream) {
  PENEO_REQUIRE(ldg >= T, "peneo_relpos_bias_bwd: ldg < T");
  PENEO_REQUIRE(g && B > 0 && nh > 0 && T > 0, "peneo_relpos_bias_bwd: bad arguments");
  PENEO_REQUIRE((bkx != nullptr) == (bky != nullptr), "peneo_relpos_bias_bwd: 2-D inputs mismatch");
  size_t sh = sizeof(unsigned long long) * (size_t)(bins1 + 2 * bins2) * RB_REP;
  PENEO_REQUIRE(sh <= 64 * 1024, "peneo_relpos_bias_bwd: tables too large for LDS");
  int slabs = (T + 31) / 32;
  dim3 grid((unsigned)((int64_t)B * nh * slabs));
  hipLaunchKernelGGL(relpos_bias_bwd_kernel, grid, dim3(256), sh, (hipStream_t)stream, g, bk1, bkx, bky, dw1, bins1, dwx, dwy,
                     bins2, scale, nh, T, ldg);
  return check_launch("peneo_relpos_bias_bwd");
}
