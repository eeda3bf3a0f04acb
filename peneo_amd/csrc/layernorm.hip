// LayerNorm forward / backward: one 64-lane wave per row, the row held in registers as 16-byte
// vectors (coalesced 1 KiB per wave-instruction), wavefront shuffles for the mean/variance and for
// the two backward row-reductions.  fp32 statistics regardless of the storage dtype.
#include <type_traits>
#include <stdlib.h>
#include "common.h"

namespace peneo {

constexpr int LN_MAXNV = 4;  // 16-byte vectors per lane: H <= 64 * VEC * 4

struct LnMap { int64_t rpb, bstride; };
__device__ __forceinline__ int64_t ln_row_off(int64_t r, LnMap m, int H) {
  return m.rpb > 0 ? (r / m.rpb) * m.bstride + (r % m.rpb) * (int64_t)H : r * (int64_t)H;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, LnMap xm, T* y, LnMap ym, const float* gamma,
                                                     const float* beta, float eps, float* mean, float* rstd,
                                                     int64_t rows, int H, float drop_p, uint32_t seed) {
  constexpr int VEC = Elem<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int nvec = H / VEC;
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const T* xr = x + ln_row_off(r, xm, H);
    T* yr = y + ln_row_off(r, ym, H);
    float v[LN_MAXNV][VEC];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      int vi = lane + 64 * k;
      if (vi < nvec) {
        unpack16<T>(*reinterpret_cast<const uint4*>(xr + vi * VEC), v[k]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[k][e];
      }
    }
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      int vi = lane + 64 * k;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { float d = v[k][e] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) {
      if (mean) mean[r] = mu;
      if (rstd) rstd[r] = rs;
    }
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      int vi = lane + 64 * k;
      if (vi < nvec) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          int c = vi * VEC + e;
          float t = (v[k][e] - mu) * rs * gamma[c] + beta[c];
          if (drop_p > 0.f) t = dropout_keep(seed, (uint64_t)r * H + c, thresh) ? t * keep_scale : 0.f;
          o[e] = t;
        }
        *reinterpret_cast<uint4*>(yr + vi * VEC) = pack16<T>(o);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* dy, LnMap dym, const T* x, LnMap xm, T* dx, LnMap dxm,
                                                     const float* gamma, const float* mean, const float* rstd,
                                                     float* dgamma, float* dbeta, int64_t rows, int H, float drop_p,
                                                     uint32_t seed, T* dx2, float drop2_p, uint32_t seed2, float* dcol) {
  constexpr int VEC = Elem<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int nvec = H / VEC;
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint32_t thresh2 = (uint32_t)fminf(drop2_p * 4294967296.0f, 4294967040.0f);
  const float keep2 = drop2_p > 0.f ? 1.0f / (1.0f - drop2_p) : 1.0f;
  float ag[LN_MAXNV][VEC], ab[LN_MAXNV][VEC], ac[LN_MAXNV][VEC];   // ac: column sums of the (dropped) dx = the feeding Linear's bias gradient
#pragma unroll
  for (int k = 0; k < LN_MAXNV; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; ac[k][e] = 0.f; }

  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const T* dyr = dy + ln_row_off(r, dym, H);
    const T* xr = x + ln_row_off(r, xm, H);
    T* dxr = dx + ln_row_off(r, dxm, H);
    const float mu = mean[r], rs = rstd[r];
    float g[LN_MAXNV][VEC], xh[LN_MAXNV][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      int vi = lane + 64 * k;
      if (vi < nvec) {
        float d[VEC], xv[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(dyr + vi * VEC), d);
        unpack16<T>(*reinterpret_cast<const uint4*>(xr + vi * VEC), xv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          int c = vi * VEC + e;
          float dd = d[e];
          if (drop_p > 0.f) dd = dropout_keep(seed, (uint64_t)r * H + c, thresh) ? dd * keep_scale : 0.f;
          float h = (xv[e] - mu) * rs;
          xh[k][e] = h;
          ag[k][e] += dd * h;
          ab[k][e] += dd;
          float gg = dd * gamma[c];
          g[k][e] = gg;
          s1 += gg;
          s2 += gg * h;
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      int vi = lane + 64 * k;
      if (vi < nvec) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rs * (g[k][e] - s1 - xh[k][e] * s2);
        *reinterpret_cast<uint4*>(dxr + vi * VEC) = pack16<T>(o);
        if (dx2) {   // second output: dx through the dropout mask of the producer GEMM (contiguous [rows, H])
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            o[e] = (drop2_p > 0.f && !dropout_keep(seed2, (uint64_t)r * H + vi * VEC + e, thresh2)) ? 0.f : o[e] * keep2;
          *reinterpret_cast<uint4*>(dx2 + r * (int64_t)H + vi * VEC) = pack16<T>(o);
        }
        if (dcol) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) ac[k][e] += o[e];
        }
      }
    }
  }
  if (dcol) {   // (this generic form: one atomic per column and wave)
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      const int vi = lane + 64 * k;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) atomicAdd(dcol + vi * VEC + e, ac[k][e]);
      }
    }
  }
  // parameter gradients: reduce the 4 waves of the block through LDS, then one atomic per column per block
  __shared__ float red[2][3][LN_MAXNV * 64 * 8];   // [gamma|beta][waves 1..3][column slot]
  const int wave = threadIdx.x >> 6;
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        red[0][wave - 1][(k * VEC + e) * 64 + lane] = ag[k][e];
        red[1][wave - 1][(k * VEC + e) * 64 + lane] = ab[k][e];
      }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < LN_MAXNV; ++k) {
      int vi = lane + 64 * k;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          int c = vi * VEC + e, sl = (k * VEC + e) * 64 + lane;
          float a = ag[k][e] + red[0][0][sl] + red[0][1][sl] + red[0][2][sl];
          float b = ab[k][e] + red[1][0][sl] + red[1][1][sl] + red[1][2][sl];
          if (dgamma) atomicAdd(dgamma + c, a);
          if (dbeta) atomicAdd(dbeta + c, b);
        }
      }
    }
  }
}

// ---- fast path: rows of 32*NV 16-byte vectors; a half-wave (32 lanes) owns a row, so no lane idles at H = 768 and
// two rows are in flight per wave-instruction; everything is compile-time sized (no predicated register arrays).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LW = lanes of the half-wave that carry data: 32, or 24 for row lengths like 192 = 24 x 8 (LiLT's layout stream)
template <typename T, int NV, bool DROP, int LW = 32>
__global__ __launch_bounds__(256) void ln_fwd32_kernel(const T* x, LnMap xm, T* y, LnMap ym, const float* gamma,
                                                       const float* beta, float eps, float* mean, float* rstd,
                                                       int64_t rows, float drop_p, uint32_t seed) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int H = LW * NV * VEC;
  const int hl = threadIdx.x & 31;
  const bool act = LW == 32 || hl < LW;
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = DROP ? 1.0f / (1.0f - drop_p) : 1.0f;
  const int64_t r = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (r >= rows) return;
  const T* xr = x + ln_row_off(r, xm, H);
  T* yr = y + ln_row_off(r, ym, H);
  float v[NV][VEC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (act) unpack16<T>(*reinterpret_cast<const uint4*>(xr + (hl + LW * k) * VEC), v[k]);
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[k][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += v[k][e];
  }
  const float mu = half_sum(s) * (1.0f / (float)H);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) { float d = act ? v[k][e] - mu : 0.f; q += d * d; }
  const float rs = rsqrtf(half_sum(q) * (1.0f / (float)H) + eps);
  if (hl == 0) {
    if (mean) mean[r] = mu;
    if (rstd) rstd[r] = rs;
  }
  if (!act) return;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c0 = (hl + LW * k) * VEC;
    float gm[VEC], bt[VEC], o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      *reinterpret_cast<float4*>(gm + e) = *reinterpret_cast<const float4*>(gamma + c0 + e);
      *reinterpret_cast<float4*>(bt + e) = *reinterpret_cast<const float4*>(beta + c0 + e);
    }
    const uint32_t dbase = DROP ? dropout_base(seed, (uint64_t)r * H + c0) : 0u;   // a 16-byte vector never straddles 2^32
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = (v[k][e] - mu) * rs * gm[e] + bt[e];
      if (DROP) t = dropout_keep_b(dbase, (uint32_t)((uint64_t)r * H + c0 + e), thresh) ? t * keep_scale : 0.f;
      o[e] = t;
    }
    *reinterpret_cast<uint4*>(yr + c0) = pack16<T>(o);
  }
}

template <typename T, int NV, bool DROP, int LW = 32, bool COL = false>
__global__ __launch_bounds__(256) void ln_bwd32_kernel(const T* dy, LnMap dym, const T* x, LnMap xm, T* dx, LnMap dxm,
                                                       const float* gamma, const float* mean, const float* rstd,
                                                       float* dgamma, float* dbeta, int64_t rows, float drop_p,
                                                       uint32_t seed, T* dx2, float drop2_p, uint32_t seed2, float* partial, float* dcol) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int H = LW * NV * VEC;
  const int hl = threadIdx.x & 31;
  const bool act = LW == 32 || hl < LW;
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = DROP ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint32_t thresh2 = (uint32_t)fminf(drop2_p * 4294967296.0f, 4294967040.0f);
  const float keep2 = drop2_p > 0.f ? 1.0f / (1.0f - drop2_p) : 1.0f;
  float gm[NV][VEC], ag[NV][VEC], ab[NV][VEC], ac[NV][VEC];   // ac: column sums of the (dropped) dx
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      gm[k][e] = act ? gamma[(hl + LW * k) * VEC + e] : 0.f;
      ag[k][e] = 0.f; ab[k][e] = 0.f; ac[k][e] = 0.f;
    }
  // the rows of a half-wave are software-pipelined: the loads of its next row are in flight while this one is reduced
  const int64_t rstep = (int64_t)gridDim.x * 8;
  int64_t r = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  uint4 nd[NV], nx[NV];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int64_t rr) {
    const T* dyr = dy + ln_row_off(rr, dym, H);
    const T* xr = x + ln_row_off(rr, xm, H);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      nd[k] = act ? *reinterpret_cast<const uint4*>(dyr + (hl + LW * k) * VEC) : make_uint4(0, 0, 0, 0);
      nx[k] = act ? *reinterpret_cast<const uint4*>(xr + (hl + LW * k) * VEC) : make_uint4(0, 0, 0, 0);
    }
    nmu = mean[rr]; nrs = rstd[rr];
  };
  if (r < rows) fetch(r);
  for (; r < rows; r += rstep) {
    T* dxr = dx + ln_row_off(r, dxm, H);
    uint4 rd[NV], rx[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { rd[k] = nd[k]; rx[k] = nx[k]; }
    const float mu = nmu, rs = nrs;
    if (r + rstep < rows) fetch(r + rstep);
    float g[NV][VEC], xh[NV][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float d[VEC], xv[VEC];
      unpack16<T>(rd[k], d);
      unpack16<T>(rx[k], xv);
      const uint32_t dbase = DROP ? dropout_base(seed, (uint64_t)r * H + (hl + LW * k) * VEC) : 0u;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float dd = d[e];
        if (DROP) dd = dropout_keep_b(dbase, (uint32_t)((uint64_t)r * H + (hl + LW * k) * VEC + e), thresh) ? dd * keep_scale : 0.f;
        const float h = act ? (xv[e] - mu) * rs : 0.f;
        xh[k][e] = h;
        ag[k][e] += dd * h;
        ab[k][e] += dd;
        const float gg = dd * gm[k][e];
        g[k][e] = gg;
        s1 += gg;
        s2 += gg * h;
      }
    }
    s1 = half_sum(s1) * (1.0f / (float)H);
    s2 = half_sum(s2) * (1.0f / (float)H);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (!act) continue;
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = rs * (g[k][e] - s1 - xh[k][e] * s2);
      *reinterpret_cast<uint4*>(dxr + (hl + LW * k) * VEC) = pack16<T>(o);
      if (dx2) {   // second output: dx through the dropout mask of the producer GEMM (contiguous [rows, H])
        const uint32_t b2 = dropout_base(seed2, (uint64_t)r * H + (hl + LW * k) * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          o[e] = (drop2_p > 0.f && !dropout_keep_b(b2, (uint32_t)((uint64_t)r * H + (hl + LW * k) * VEC + e), thresh2)) ? 0.f : o[e] * keep2;
        *reinterpret_cast<uint4*>(dx2 + r * (int64_t)H + (hl + LW * k) * VEC) = pack16<T>(o);
      }
      if constexpr (COL) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) ac[k][e] += o[e];
      }
    }
  }
  // parameter gradients: 8 half-waves -> LDS -> one atomic per column per block
  __shared__ float red[2][8][LW * NV * VEC + 1];
  const int hw = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (act) {
        red[0][hw][(hl + LW * k) * VEC + e] = ag[k][e];
        red[1][hw][(hl + LW * k) * VEC + e] = ab[k][e];
      }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
    if (partial) {   // [gridDim.x][2][H]: summed by a column-sum launch off the critical path (no same-address atomics)
      partial[((int64_t)blockIdx.x * 2 + 0) * H + c] = a;
      partial[((int64_t)blockIdx.x * 2 + 1) * H + c] = b;
    } else {
      if (dgamma) atomicAdd(dgamma + c, a);
      if (dbeta) atomicAdd(dbeta + c, b);
    }
  }
  if constexpr (COL) {   // the bias gradient of the Linear that fed this LayerNorm: same reduction, the LDS rows reused
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (act) red[0][hw][(hl + LW * k) * VEC + e] = ac[k][e];
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[0][w][c];
      atomicAdd(dcol + c, a);
    }
  }
}

template <typename T, int NV, int LW = 32>
static void launch_ln_fwd32(hipStream_t st, const void* x, LnMap xm, void* y, LnMap ym, const float* gamma, const float* beta,
                            float eps, float* mean, float* rstd, int64_t rows, float drop_p, uint32_t seed) {
  dim3 grid((unsigned)((rows + 7) / 8));
  if (drop_p > 0.f)
    hipLaunchKernelGGL((ln_fwd32_kernel<T, NV, true, LW>), grid, dim3(256), 0, st, (const T*)x, xm, (T*)y, ym, gamma, beta, eps, mean,
                       rstd, rows, drop_p, seed);
  else
    hipLaunchKernelGGL((ln_fwd32_kernel<T, NV, false, LW>), grid, dim3(256), 0, st, (const T*)x, xm, (T*)y, ym, gamma, beta, eps, mean,
                       rstd, rows, drop_p, seed);
}
template <typename T, int NV, int LW = 32>
static void launch_ln_bwd32(hipStream_t st, const void* dy, LnMap dym, const void* x, LnMap xm, void* dx, LnMap dxm,
                            const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                            int64_t rows, float drop_p, uint32_t seed, void* dx2, float drop2_p, uint32_t seed2,
                            float* partial = nullptr, int64_t partial_rows = 0, float* dcol = nullptr) {
  constexpr int64_t cap = 256;   // fewer same-address atomics on dgamma / dbeta (measured 64..1024 alone; 128..1024 in the step, round 5: +-0.1 ms)
  int64_t blocks = (rows + 7) / 8;
  if (partial) blocks = partial_rows;          // one partial row per block: the caller sized the buffer (ln_partial_rows)
  else if (blocks > cap) blocks = cap;
  dim3 grid((unsigned)blocks);
#define PENEO_LN_BWD32(DROP_, COL_)                                                                                              \
  hipLaunchKernelGGL((ln_bwd32_kernel<T, NV, DROP_, LW, COL_>), grid, dim3(256), 0, st, (const T*)dy, dym, (const T*)x, xm, (T*)dx, dxm, \
                     gamma, mean, rstd, dgamma, dbeta, rows, drop_p, seed, (T*)dx2, drop2_p, seed2, partial, dcol)
  if (dcol) { if (drop_p > 0.f) PENEO_LN_BWD32(true, true); else PENEO_LN_BWD32(false, true); }
  else { if (drop_p > 0.f) PENEO_LN_BWD32(true, false); else PENEO_LN_BWD32(false, false); }
#undef PENEO_LN_BWD32
}
// blocks of the partial-sum form: one row per half-wave up to 1024 blocks (4 per CU), then a grid-stride loop
static int64_t ln_partial_rows(int64_t rows) {
  constexpr int64_t cap = 1024;
  int64_t blocks = (rows + 7) / 8;
  return blocks > cap ? cap : blocks;
}

// true when a fast instantiation exists for this row length: f(NV, LW) with H = LW * NV * VEC, LW = 32 data lanes per half-wave
// or 24 (H = 192 / 384-fp32 ...: LiLT's layout stream)
template <typename T, typename F> static bool ln32_dispatch(int H, F&& f) {
  constexpr int VEC = sizeof(T) == 2 ? 8 : 4;
  using L32 = std::integral_constant<int, 32>;
  using L24 = std::integral_constant<int, 24>;
  if (H % (32 * VEC) == 0) {
    switch (H / (32 * VEC)) {
      case 1: f(std::integral_constant<int, 1>{}, L32{}); return true;
      case 2: f(std::integral_constant<int, 2>{}, L32{}); return true;
      case 3: f(std::integral_constant<int, 3>{}, L32{}); return true;
      case 4: f(std::integral_constant<int, 4>{}, L32{}); return true;
      case 6: if (sizeof(T) == 4) { f(std::integral_constant<int, 6>{}, L32{}); return true; } return false;
      case 8: if (sizeof(T) == 4) { f(std::integral_constant<int, 8>{}, L32{}); return true; } return false;
      default: return false;
    }
  }
  if (H % (24 * VEC) == 0) {
    switch (H / (24 * VEC)) {
      case 1: f(std::integral_constant<int, 1>{}, L24{}); return true;
      case 2: f(std::integral_constant<int, 2>{}, L24{}); return true;
      default: return false;
    }
  }
  return false;
}

}  // namespace peneo
using namespace peneo;

// 16-byte alignment of a row base: base pointer and (for sliced [B, R, H] maps) the batch stride
static bool ln_aligned(const void* p, int64_t bstride, int dtype) {
  const int esz = dtype == PENEO_BF16 ? 2 : 4;
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ((bstride * esz) % 16) == 0;
}

static int ln_check(const char* who, int dtype, int64_t rows, int H) {
  PENEO_REQUIRE(dtype == PENEO_F32 || dtype == PENEO_BF16, "%s: bad dtype", who);
  const int vec = dtype == PENEO_BF16 ? 8 : 4;
  PENEO_REQUIRE(rows > 0 && H > 0, "%s: empty problem", who);
  PENEO_REQUIRE(H % vec == 0 && H <= 64 * vec * LN_MAXNV, "%s: H=%d must be a multiple of %d and <= %d", who, H, vec,
                64 * vec * LN_MAXNV);
  return PENEO_OK;
}

static unsigned ln_grid(int64_t rows, int waves_per_row_target) {
  int64_t blocks = (rows + 3) / 4;
  int64_t cap = 256 * 8 / (waves_per_row_target > 0 ? waves_per_row_target : 1);
  if (blocks > cap) blocks = cap;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int peneo_layernorm_fwd(int dtype, const void* x, int64_t x_rpb, int64_t x_bstride, void* y, int64_t y_rpb,
                                   int64_t y_bstride, const float* gamma, const float* beta, float eps, float* mean,
                                   float* rstd, int64_t rows, int H, float drop_p, uint32_t drop_seed,
                                   peneo_stream_t stream) {
  int rc = ln_check("peneo_layernorm_fwd", dtype, rows, H);
  if (rc) return rc;
  PENEO_REQUIRE(x && y && gamma && beta, "peneo_layernorm_fwd: null pointer");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "peneo_layernorm_fwd: drop_p out of range");
  LnMap xm{x_rpb, x_bstride}, ym{y_rpb, y_bstride};
  hipStream_t st = (hipStream_t)stream;
  const bool al = ln_aligned(x, x_bstride, dtype) && ln_aligned(y, y_bstride, dtype) && ln_aligned(gamma, 0, PENEO_F32) &&
                  ln_aligned(beta, 0, PENEO_F32);
  if (al) {
    bool done = dtype == PENEO_BF16
        ? ln32_dispatch<bf16_t>(H, [&](auto nv, auto lw) { launch_ln_fwd32<bf16_t, decltype(nv)::value, decltype(lw)::value>(st, x, xm, y, ym, gamma, beta, eps, mean, rstd, rows, drop_p, drop_seed); })
        : ln32_dispatch<float>(H, [&](auto nv, auto lw) { launch_ln_fwd32<float, decltype(nv)::value, decltype(lw)::value>(st, x, xm, y, ym, gamma, beta, eps, mean, rstd, rows, drop_p, drop_seed); });
    if (done) return check_launch("peneo_layernorm_fwd");
  }
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, xm, (bf16_t*)y, ym,
                       gamma, beta, eps, mean, rstd, rows, H, drop_p, drop_seed);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, xm, (float*)y, ym,
                       gamma, beta, eps, mean, rstd, rows, H, drop_p, drop_seed);
  return check_launch("peneo_layernorm_fwd");
}

extern "C" int peneo_layernorm_bwd(int dtype, const void* dy, int64_t dy_rpb, int64_t dy_bstride, const void* x,
                                   int64_t x_rpb, int64_t x_bstride, void* dx, int64_t dx_rpb, int64_t dx_bstride,
                                   const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                   int64_t rows, int H, float drop_p, uint32_t drop_seed, void* dx_dropped, float drop2_p,
                                   uint32_t drop2_seed, float* dx_colsum, peneo_stream_t stream) {
  int rc = ln_check("peneo_layernorm_bwd", dtype, rows, H);
  if (rc) return rc;
  PENEO_REQUIRE(dy && x && dx && gamma && mean && rstd, "peneo_layernorm_bwd: null pointer");
  LnMap dym{dy_rpb, dy_bstride}, xm{x_rpb, x_bstride}, dxm{dx_rpb, dx_bstride};
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f && drop2_p >= 0.f && drop2_p < 1.f, "peneo_layernorm_bwd: drop_p out of range");
  PENEO_REQUIRE(!dx_dropped || (reinterpret_cast<uintptr_t>(dx_dropped) & 15) == 0, "peneo_layernorm_bwd: dx_dropped must be 16-byte aligned");
  if (ln_aligned(dy, dy_bstride, dtype) && ln_aligned(x, x_bstride, dtype) && ln_aligned(dx, dx_bstride, dtype)) {
    hipStream_t st = (hipStream_t)stream;
    bool done = dtype == PENEO_BF16
        ? ln32_dispatch<bf16_t>(H, [&](auto nv, auto lw) { launch_ln_bwd32<bf16_t, decltype(nv)::value, decltype(lw)::value>(st, dy, dym, x, xm, dx, dxm, gamma, mean, rstd, dgamma, dbeta, rows, drop_p, drop_seed, dx_dropped, drop2_p, drop2_seed, nullptr, 0, dx_colsum); })
        : ln32_dispatch<float>(H, [&](auto nv, auto lw) { launch_ln_bwd32<float, decltype(nv)::value, decltype(lw)::value>(st, dy, dym, x, xm, dx, dxm, gamma, mean, rstd, dgamma, dbeta, rows, drop_p, drop_seed, dx_dropped, drop2_p, drop2_seed, nullptr, 0, dx_colsum); });
    if (done) return check_launch("peneo_layernorm_bwd");
  }
  dim3 grid(ln_grid(rows, 8));  // <= 256 blocks: each wave reduces several rows; one atomic per column per block
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, dym,
                       (const bf16_t*)x, xm, (bf16_t*)dx, dxm, gamma, mean, rstd, dgamma, dbeta, rows, H, drop_p, drop_seed,
                       (bf16_t*)dx_dropped, drop2_p, drop2_seed, dx_colsum);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, dym,
                       (const float*)x, xm, (float*)dx, dxm, gamma, mean, rstd, dgamma, dbeta, rows, H, drop_p, drop_seed,
                       (float*)dx_dropped, drop2_p, drop2_seed, dx_colsum);
  return check_launch("peneo_layernorm_bwd");
}

// Same backward with the LayerNorm parameter gradients left as per-block partial sums: partials[P][2][H] (gamma row, beta
// row per block), P = peneo_layernorm_bwd_partial_rows(dtype, rows, H) (0: this row length / dtype has no such form, use
// peneo_layernorm_bwd).  The caller column-sums the partials whenever it likes (peneo_colsum over [P, 2H]); the kernel then
// has no same-address atomics at its end and can run one row per half-wave.
extern "C" int64_t peneo_layernorm_bwd_partial_rows(int dtype, int64_t rows, int H) {
  if (rows <= 0 || (dtype != PENEO_F32 && dtype != PENEO_BF16)) return 0;
  const bool ok = dtype == PENEO_BF16 ? ln32_dispatch<bf16_t>(H, [](auto, auto) {}) : ln32_dispatch<float>(H, [](auto, auto) {});
  return ok ? ln_partial_rows(rows) : 0;
}

extern "C" int peneo_layernorm_bwd_partial(int dtype, const void* dy, int64_t dy_rpb, int64_t dy_bstride, const void* x,
                                           int64_t x_rpb, int64_t x_bstride, void* dx, int64_t dx_rpb, int64_t dx_bstride,
                                           const float* gamma, const float* mean, const float* rstd, float* partials,
                                           int64_t partial_rows, int64_t rows, int H, float drop_p, uint32_t drop_seed,
                                           void* dx_dropped, float drop2_p, uint32_t drop2_seed, peneo_stream_t stream) {
  int rc = ln_check("peneo_layernorm_bwd_partial", dtype, rows, H);
  if (rc) return rc;
  PENEO_REQUIRE(dy && x && dx && gamma && mean && rstd && partials, "peneo_layernorm_bwd_partial: null pointer");
  PENEO_REQUIRE(partial_rows > 0 && partial_rows == peneo_layernorm_bwd_partial_rows(dtype, rows, H),
                "peneo_layernorm_bwd_partial: partial_rows must come from peneo_layernorm_bwd_partial_rows");
  LnMap dym{dy_rpb, dy_bstride}, xm{x_rpb, x_bstride}, dxm{dx_rpb, dx_bstride};
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f && drop2_p >= 0.f && drop2_p < 1.f, "peneo_layernorm_bwd_partial: drop_p out of range");
  PENEO_REQUIRE(ln_aligned(dy, dy_bstride, dtype) && ln_aligned(x, x_bstride, dtype) && ln_aligned(dx, dx_bstride, dtype) &&
                (!dx_dropped || (reinterpret_cast<uintptr_t>(dx_dropped) & 15) == 0),
                "peneo_layernorm_bwd_partial: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PENEO_BF16)
    ln32_dispatch<bf16_t>(H, [&](auto nv, auto lw) { launch_ln_bwd32<bf16_t, decltype(nv)::value, decltype(lw)::value>(st, dy, dym, x, xm, dx, dxm, gamma, mean, rstd, nullptr, nullptr, rows, drop_p, drop_seed, dx_dropped, drop2_p, drop2_seed, partials, partial_rows); });
  else
    ln32_dispatch<float>(H, [&](auto nv, auto lw) { launch_ln_bwd32<float, decltype(nv)::value, decltype(lw)::value>(st, dy, dym, x, xm, dx, dxm, gamma, mean, rstd, nullptr, nullptr, rows, drop_p, drop_seed, dx_dropped, drop2_p, drop2_seed, partials, partial_rows); });
  return check_launch("peneo_layernorm_bwd_partial");
}
