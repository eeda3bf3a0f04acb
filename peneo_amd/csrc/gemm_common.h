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
// The epilogue of 8 consecutive columns in two halves, so that a kernel can issue the loads of a BATCH of granules before the
// arithmetic and stores of any of them (every wait for a load issued behind a store is a full memory round trip on the one vm
// counter: gemm_sk.hip measured 0.56 us per granule with load and store alternating).  The "primary" matrix-shaped input of
// the granule -- the gradient source if there is one, else the residual, else the old fp32 C of an accumulating call -- travels as
// raw 16-byte words; any further matrix input (combinations the model does not use) is loaded in place by the arithmetic half.
enum { EP_PRIM_NONE = 0, EP_PRIM_GRAD = 1, EP_PRIM_RES = 2, EP_PRIM_ACC = 3 };
struct EpIn8 { uint4 x0, x1; };
__device__ __forceinline__ int ep_primary(const GemmParams& p) {
  return p.ep.grad_src ? EP_PRIM_GRAD : (p.ep.residual ? EP_PRIM_RES : (p.ep.accumulate ? EP_PRIM_ACC : EP_PRIM_NONE));
}
__device__ __forceinline__ void ep_load_primary(const GemmParams& p, int prim, int m, int n, EpIn8& in) {
  if (prim == EP_PRIM_NONE) return;
  const peneo_gemm_epilogue& e = p.ep;
  const void* base = prim == EP_PRIM_GRAD ? e.grad_src : (prim == EP_PRIM_RES ? e.residual : p.C);
  const int64_t ld = prim == EP_PRIM_GRAD ? e.ld_grad : (prim == EP_PRIM_RES ? e.ld_res : p.ldc);
  const int64_t idx = (int64_t)m * ld + n;
  if (prim == EP_PRIM_ACC || p.c_dtype == PENEO_F32) {
    const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(base) + idx);
    in.x0 = q[0]; in.x1 = q[1];
  } else {
    in.x0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + idx);
  }
}
__device__ __forceinline__ void ep_unpack_primary(const GemmParams& p, int prim, const EpIn8& in, float* f) {
  if (prim == EP_PRIM_ACC || p.c_dtype == PENEO_F32) { unpack16<float>(in.x0, f); unpack16<float>(in.x1, f + 4); }
  else unpack16<bf16_t>(in.x0, f);
}
// bias8: the 8 bias values of these columns (already loaded), or null when the call has no bias
__device__ __forceinline__ void epilogue_apply8(const GemmParams& p, int m, int n, float* v, const float* bias8, int prim, const EpIn8& in) {
  const peneo_gemm_epilogue& e = p.ep;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
  if (bias8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bias8[i];
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
    if (prim == EP_PRIM_GRAD) ep_unpack_primary(p, prim, in, g);
    else load8_any(e.grad_src, p.c_dtype, (int64_t)m * e.ld_grad + n, g);
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
    if (prim == EP_PRIM_RES) ep_unpack_primary(p, prim, in, r);
    else load8_any(e.residual, p.c_dtype, (int64_t)m * e.ld_res + n, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  const int64_t ci = (int64_t)m * p.ldc + n;
  if (e.accumulate) {
    float c[8];
    if (prim == EP_PRIM_ACC) ep_unpack_primary(p, prim, in, c);
    else load8_any(p.C, PENEO_F32, ci, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += c[i];
  }
  store8_any(p.C, p.c_dtype, ci, v);
}
// (The tiled kernels of gemm.hip / gemm_big.hip keep this single-granule form -- the same arithmetic in the same order as
//  epilogue_apply8 above; written out rather than composed from the two halves because their register allocation sits at the
//  edge where one more live scalar sends an LDS-DMA base pointer to a VGPR.)
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
// gemm_sk.hip (persistent stream-k launch): same return convention
int launch_gemm_sk(const GemmParams& p, bool b_kmajor, hipStream_t st);

}  // namespace peneo
