// Shared by gemm.hip (128 x 128 tiles) and gemm_big.hip (one 8-wave workgroup per CU): launch parameters and the fused
// epilogue of peneo_gemm (bias, activation, pre-activation store, x act'(src), dropout, residual, fp32 accumulate) on 8
// consecutive columns of one output row.
#pragma once
#include "common.h"

namespace peneo {

constexpr int GB = 128;        // tile edge (both M and N)
constexpr int ROWB = 128;      // bytes per LDS row (8 slots of 16 B)
constexpr int TILE_BYTES = GB * ROWB;

struct GemmParams {
  const void* A; const void* B; void* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int c_dtype;
  peneo_gemm_epilogue ep;
  int split_k;        // >1: write raw fp32 partials to `ws` [split][M][N]
  float* ws;
  int kt_per_split;   // k-tiles per split
  int dz_on;          // pair-head backward epilogue (z -> dz, dW2 / db1 partials)
  peneo_pair_dz_args dz;
  float* dz_ws;
};
constexpr int GEMM_DZ_SLOTS = 256;   // == DZ_SLOTS of pair_heads.hip (rows of the partial-sum workspace)


// ---- epilogue --------------------------------------------------------------------------------
__device__ __forceinline__ float load_any(const void* p, int dtype, int64_t idx) {
  return dtype == PENEO_F32 ? reinterpret_cast<const float*>(p)[idx] : bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[idx]);
}
__device__ __forceinline__ void store_any(void* p, int dtype, int64_t idx, float v) {
  if (dtype == PENEO_F32) reinterpret_cast<float*>(p)[idx] = v;
  else reinterpret_cast<bf16_t*>(p)[idx] = f32_to_bf16(v);
}

__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, float acc) {
  const peneo_gemm_epilogue& e = p.ep;
  float v = acc * e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.preact) store_any(e.preact, p.c_dtype, (int64_t)m * e.ld_preact + n, v);
  // the same GELU as the vectorised columns of this tile (epilogue_store8): polynomial erf for bf16 tiles, erff for fp32
  const bool fast = p.c_dtype == PENEO_BF16;
  v = (fast && e.act == PENEO_ACT_GELU) ? gelu_fast_f(v) : act_f(e.act, v);
  if (e.grad_src) {
    const float g = load_any(e.grad_src, p.c_dtype, (int64_t)m * e.ld_grad + n);
    v *= (fast && e.grad_act == PENEO_ACT_GELU) ? gelu_grad_fast_f(g) : act_grad_f(e.grad_act, g);
  }
  if (e.drop_p > 0.f) {
    uint32_t thresh = (uint32_t)fminf(e.drop_p * 4294967296.0f, 4294967040.0f);
    v = dropout_keep(e.drop_seed, (uint64_t)m * (uint64_t)p.N + n, thresh) ? v * (1.0f / (1.0f - e.drop_p)) : 0.f;
  }
  if (e.residual) v += load_any(e.residual, p.c_dtype, (int64_t)m * e.ld_res + n);
  int64_t ci = (int64_t)m * p.ldc + n;
  if (e.accumulate) v += reinterpret_cast<float*>(p.C)[ci];
  store_any(p.C, p.c_dtype, ci, v);
}

// 8 consecutive columns of one row; requires n + 8 <= N and 16-byte aligned rows of every operand involved
__device__ __forceinline__ void load8_any(const void* p, int dtype, int64_t idx, float* f) {
  if (dtype == PENEO_F32) {
    const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(p) + idx);
    unpack16<float>(q[0], f); unpack16<float>(q[1], f + 4);
  } else {
    unpack16<bf16_t>(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p) + idx), f);
  }
}
__device__ __forceinline__ void store8_any(void* p, int dtype, int64_t idx, const float* f) {
  if (dtype == PENEO_F32) {
    uint4* q = reinterpret_cast<uint4*>(reinterpret_cast<float*>(p) + idx);
    q[0] = pack16<float>(f); q[1] = pack16<float>(f + 4);
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p) + idx) = pack16<bf16_t>(f);
  }
}
__device__ __forceinline__ void epilogue_store8(const GemmParams& p, int m, int n, float* v) {
  const peneo_gemm_epilogue& e = p.ep;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
  if (e.bias) {
    float b[8];
    load8_any(e.bias, PENEO_F32, n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (e.preact) store8_any(e.preact, p.c_dtype, (int64_t)m * e.ld_preact + n, v);
  const bool fast = p.c_dtype == PENEO_BF16;   // bf16 tiles: polynomial erf (2e-5) instead of the library erff
  // one uniform branch per activation kind AROUND the element loop: with the kind tested per element the compiler turned the
  // nested conditional into selects and evaluated GELU and SiLU for every element (FFN1 forward: +15 us for either)
  if (e.act == PENEO_ACT_GELU) {
    if (fast) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = gelu_fast_f(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = gelu_f(v[i]);
    }
  } else if (e.act == PENEO_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = silu_f(v[i]);
  }
  if (e.grad_src) {
    float g[8];
    load8_any(e.grad_src, p.c_dtype, (int64_t)m * e.ld_grad + n, g);
    if (e.grad_act == PENEO_ACT_GELU) {
      if (fast) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_fast_f(g[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_f(g[i]);
      }
    } else if (e.grad_act == PENEO_ACT_SILU) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= silu_grad_f(g[i]);
    }
  }
  if (e.drop_p > 0.f) {
    const uint32_t thresh = (uint32_t)fminf(e.drop_p * 4294967296.0f, 4294967040.0f);
    const float ks = 1.0f / (1.0f - e.drop_p);
    const uint32_t keep = dropout_keep8(e.drop_seed, (uint64_t)m * (uint64_t)p.N + n, thresh);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ((keep >> i) & 1u) ? v[i] * ks : 0.f;
  }
  if (e.residual) {
    float r[8];
    load8_any(e.residual, p.c_dtype, (int64_t)m * e.ld_res + n, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  const int64_t ci = (int64_t)m * p.ldc + n;
  if (e.accumulate) {
    float c[8];
    load8_any(p.C, PENEO_F32, ci, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += c[i];
  }
  store8_any(p.C, p.c_dtype, ci, v);
}


// gemm_big.hip: 0 = the shape / options are not covered (the caller runs the 128 x 128 kernel), 1 = launched
int launch_gemm_big(const GemmParams& p, bool b_kmajor, hipStream_t st);

}  // namespace peneo
