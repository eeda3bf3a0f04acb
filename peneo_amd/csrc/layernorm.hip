// LayerNorm forward / backward: one 64-lane wave per row, the row held in registers as 16-byte
// vectors (coalesced 1 KiB per wave-instruction), wavefront shuffles for the mean/variance and for
// the two backward row-reductions.  fp32 statistics regardless of the storage dtype.
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
                                                     uint32_t seed) {
  constexpr int VEC = Elem<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int nvec = H / VEC;
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float ag[LN_MAXNV][VEC], ab[LN_MAXNV][VEC];
#pragma unroll
  for (int k = 0; k < LN_MAXNV; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; }

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

}  // namespace peneo
using namespace peneo;

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
                                   int64_t rows, int H, float drop_p, uint32_t drop_seed, peneo_stream_t stream) {
  int rc = ln_check("peneo_layernorm_bwd", dtype, rows, H);
  if (rc) return rc;
  PENEO_REQUIRE(dy && x && dx && gamma && mean && rstd, "peneo_layernorm_bwd: null pointer");
  LnMap dym{dy_rpb, dy_bstride}, xm{x_rpb, x_bstride}, dxm{dx_rpb, dx_bstride};
  dim3 grid(ln_grid(rows, 8));  // <= 256 blocks: each wave reduces several rows; one atomic per column per block
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, dym,
                       (const bf16_t*)x, xm, (bf16_t*)dx, dxm, gamma, mean, rstd, dgamma, dbeta, rows, H, drop_p, drop_seed);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, dym,
                       (const float*)x, xm, (float*)dx, dxm, gamma, mean, rstd, dgamma, dbeta, rows, H, drop_p, drop_seed);
  return check_launch("peneo_layernorm_bwd");
}
