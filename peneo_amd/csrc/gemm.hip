// Dense contraction for gfx950: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each
// wave a 64x64 sub-tile = 2x2 MFMA 32x32 accumulators), k-tiles of 128 bytes per row staged
// global -> registers -> LDS (XOR-swizzled 16-byte slots, conflict-free ds_read_b128), double
// buffered with one barrier per k-tile.  Operands may be k-major or mn-major in memory
// (mn-major tiles are transposed in registers on their way to LDS), which gives forward
// (x W^T), dgrad (dy W) and wgrad (dy^T x) from one template.  bf16 -> v_mfma_f32_32x32x16_bf16,
// fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32; the parity mode).
#include <cstdlib>
#include "common.h"
#include "gemm_common.h"

namespace peneo {

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4); }

// ---- global -> register staging ---------------------------------------------------------------
template <typename T> struct Stage { uint4 v[4]; };

template <typename T>
__device__ __forceinline__ uint4 guarded_vec_load(const T* base, int64_t ld, int r, int c, int rmax, int cmax) {
  // 16-byte vector at (r, c..c+VEC) of a row-major matrix; zero outside [rmax, cmax)
  constexpr int VEC = Elem<T>::kVec;
  uint4 z = make_uint4(0, 0, 0, 0);
  if (r >= rmax || c >= cmax) return z;
  const T* p = base + (int64_t)r * ld + c;
  if (c + VEC <= cmax && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const uint4*>(p);
  float f[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) f[e] = (c + e < cmax) ? Elem<T>::load(p + e) : 0.f;
  return pack16<T>(f);
}

// k-major operand: matrix [rows, K]; tile rows r0.., k from k0..; thread t owns vectors t + 256*i
// `fast`: the whole tile is in bounds and every vector is 16-byte aligned (block-uniform) -> plain loads.
template <typename T>
__device__ __forceinline__ void load_kmajor(Stage<T>& s, const T* base, int64_t ld, int r0, int k0, int rmax, int kmax,
                                            int tid, bool fast) {
  constexpr int VEC = Elem<T>::kVec;
  if (fast) {
    const T* p0 = base + (int64_t)(r0 + (tid >> 3)) * ld + k0 + (tid & 7) * VEC;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = *reinterpret_cast<const uint4*>(p0 + (int64_t)(32 * i) * ld);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int v = tid + 256 * i;
    int row = v >> 3, slot = v & 7;
    s.v[i] = guarded_vec_load<T>(base, ld, r0 + row, k0 + slot * VEC, rmax, kmax);
  }
}
template <typename T>
__device__ __forceinline__ void store_kmajor(const Stage<T>& s, char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int v = tid + 256 * i;
    int row = v >> 3, slot = v & 7;
    *reinterpret_cast<uint4*>(tile + lds_off(row, slot)) = s.v[i];
  }
}

// mn-major operand: matrix [K, rows] (rows contiguous).  Each thread loads 4 consecutive k-rows of one
// 16-byte vector of tile-rows and writes them k-contiguous.
template <typename T>
__device__ __forceinline__ void load_mnmajor(Stage<T>& s, const T* base, int64_t ld, int r0, int k0, int rmax, int kmax,
                                             int tid, bool fast) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int MG = GB / VEC;  // vectors along the tile-row dim: 16 (bf16) / 32 (fp32)
  int mg = tid % MG, kg = tid / MG;
  if (fast) {
    const T* p0 = base + (int64_t)(k0 + kg * 4) * ld + r0 + mg * VEC;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = *reinterpret_cast<const uint4*>(p0 + (int64_t)i * ld);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s.v[i] = guarded_vec_load<T>(base, ld, k0 + kg * 4 + i, r0 + mg * VEC, kmax, rmax);
}
template <typename T> __device__ __forceinline__ void store_mnmajor(const Stage<T>& s, char* tile, int tid);
template <>
__device__ __forceinline__ void store_mnmajor<bf16_t>(const Stage<bf16_t>& s, char* tile, int tid) {
  int mg = tid % 16, kg = tid / 16;  // kg: group of 4 k (8 bytes), 16 groups per 64-k tile
  const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&s.v[0]);
  const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&s.v[1]);
  const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&s.v[2]);
  const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&s.v[3]);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int sh = (e & 1) * 16;
    uint32_t a = (w0[e >> 1] >> sh) & 0xffffu, b = (w1[e >> 1] >> sh) & 0xffffu;
    uint32_t c = (w2[e >> 1] >> sh) & 0xffffu, d = (w3[e >> 1] >> sh) & 0xffffu;
    int row = mg * 8 + e;
    *reinterpret_cast<uint2*>(tile + lds_off(row, kg >> 1) + (kg & 1) * 8) = make_uint2(a | (b << 16), c | (d << 16));
  }
}
template <>
__device__ __forceinline__ void store_mnmajor<float>(const Stage<float>& s, char* tile, int tid) {
  int mg = tid % 32, kg = tid / 32;  // kg: group of 4 k (16 bytes), 8 groups per 32-k tile
  const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&s.v[0]);
  const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&s.v[1]);
  const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&s.v[2]);
  const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&s.v[3]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int row = mg * 4 + e;
    *reinterpret_cast<uint4*>(tile + lds_off(row, kg)) = make_uint4(w0[e], w1[e], w2[e], w3[e]);
  }
}

// ---- LDS -> MFMA fragments -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ Frag<T> read_frag(const char* tile, int row, int ks, int lane);
template <>
__device__ __forceinline__ Frag<bf16_t> read_frag<bf16_t>(const char* tile, int row, int ks, int lane) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const uint4*>(tile + lds_off(row, 2 * ks + (lane >> 5)));
  return f;
}
template <>
__device__ __forceinline__ Frag<float> read_frag<float>(const char* tile, int row, int ks, int lane) {
  Frag<float> f;
  int s = 4 * ks + 2 * (lane >> 5);
  f.v[0] = *reinterpret_cast<const uint4*>(tile + lds_off(row, s));
  f.v[1] = *reinterpret_cast<const uint4*>(tile + lds_off(row, s + 1));
  return f;
}

// Per-thread operands of the pair-dz epilogue, fetched at kernel start so their latency hides under the k loop:
// the scaled dlogits of row (tid & 127) for the tile's first (tid < 128) / second head, and the second-layer weights
// and first-layer bias of the thread's 8 columns.
struct DzPre { float g[3]; float w2c[3][8]; float b1v[8]; };
__device__ __forceinline__ void dz_prefetch(const GemmParams& p, int m0, int n0, int tid, DzPre& d) {
  const peneo_pair_dz_args& a = p.dz;
  const int nrem = min(GB, p.N - n0), mrem = min(GB, p.M - m0);
  const int h_lo = n0 / a.D, h_hi = min((n0 + nrem - 1) / a.D, a.num_heads - 1);
  const int hh = h_lo + (tid >> 7), r = tid & (GB - 1);
  d.g[0] = 0.f; d.g[1] = 0.f; d.g[2] = 0.f;
  if (hh <= h_hi && r < mrem) {
    const int Cn = a.classes[hh];
    const float sc = a.scale[hh];
    const float* dl = a.dlogits[hh] + (int64_t)(m0 + r) * Cn;
    d.g[0] = dl[0] * sc;
    if (Cn > 1) d.g[1] = dl[1] * sc;
    if (Cn > 2) d.g[2] = dl[2] * sc;
  }
  const int c0 = (tid & 15) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int n = min(n0 + c0 + e, p.N - 1);
    const int h = n / a.D, k = n - h * a.D;
    const int Cn = a.classes[h];
    d.w2c[0][e] = a.w2[h][k];
    d.w2c[1][e] = Cn > 1 ? a.w2[h][(int64_t)a.D + k] : 0.f;
    d.w2c[2][e] = Cn > 2 ? a.w2[h][(int64_t)2 * a.D + k] : 0.f;
    d.b1v[e] = p.ep.bias ? p.ep.bias[n] : 0.f;
  }
}

// epilogue: park the accumulators in LDS (the staging buffers are dead now), then walk the tile
// row-major so that consecutive lanes own consecutive columns (coalesced C / residual traffic)
// and the fused epilogue stays a compact rolled loop.
// DZ: the pair-head backward epilogue (peneo_gemm_epilogue.pair_dz) is compiled into its own instantiations only - 15 KB of code
// that every other GEMM kernel used to carry (round 6: the instruction cache of a CU pair is 64 KB and a layer's kernels alternate)
template <bool DZ>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x16_t (&acc)[2][2], char* smem, int m0, int n0, int tid,
                                              int lane, int wm, int wn, const DzPre& pre, int zsplit) {
  float* sC = reinterpret_cast<float*>(smem);  // [128][128] fp32 = 64 KiB
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sC[(wm * 64 + i * 32 + acc_row(r, lane)) * GB + wn * 64 + j * 32 + acc_col(lane)] = acc[i][j][r];
  __syncthreads();
  const int mrem = min(GB, p.M - m0), nrem = min(GB, p.N - n0);
  if constexpr (DZ) {
    // z = acc + b1 -> dz.  Thread = 8 consecutive columns (one 16-byte store per row) x 8 rows; the per-row dlogits of
    // the tile's head(s) are staged in LDS behind the C tile; column sums (dW2 / db1 partials) are reduced over the 16
    // row-lanes through LDS (the C tile is dead by then) and leave as one atomic per column and quantity.
    const peneo_pair_dz_args& a = p.dz;
    float* sG = sC + GB * GB;                       // [2 heads][128 rows][4]: scale * dlogits (zero padded classes)
    const int h_lo = n0 / a.D, h_hi = min((n0 + nrem - 1) / a.D, a.num_heads - 1);
    const bool staged = (h_hi - h_lo) <= 1;         // tile spans at most 2 heads (always when D >= 128)
    if (staged) {
      *reinterpret_cast<float4*>(sG + tid * 4) = make_float4(pre.g[0], pre.g[1], pre.g[2], 0.f);
      __syncthreads();
    }
    const int cg = tid & 15, rl = tid >> 4;           // 16 column groups x 16 row lanes
    const int c0 = cg * 8;
    float w2c[3][8], b1v[8], s0[8], s1[8], s2[8], sb[8];
    int hsel[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = min(n0 + c0 + e, p.N - 1);
      const int h = n / a.D, k = n - h * a.D;
      const int Cn = a.classes[h];
      hsel[e] = h - h_lo;
      (void)k; (void)Cn;
      w2c[0][e] = pre.w2c[0][e]; w2c[1][e] = pre.w2c[1][e]; w2c[2][e] = pre.w2c[2][e];
      b1v[e] = pre.b1v[e];
      s0[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; sb[e] = 0.f;
    }
    const bool one_head = hsel[0] == hsel[7];
    const bool vec_store = (c0 + 8 <= nrem) && p.c_dtype == PENEO_BF16 && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                           ((p.ldc * 2) % 16 == 0);
    if (staged && h_lo == h_hi) {
      // common case (decoder width a multiple of the tile): one head per tile.  The loop is VALU-bound, so the
      // full-rate arithmetic runs as packed fp32 pairs (v_pk_fma/mul/add_f32); only exp2 / rcp stay scalar.
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 w0p[4], w1p[4], w2p[4], b1p[4], s0p[4], s1p[4], s2p[4], sbp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w0p[j] = f2{w2c[0][2 * j], w2c[0][2 * j + 1]}; w1p[j] = f2{w2c[1][2 * j], w2c[1][2 * j + 1]};
        w2p[j] = f2{w2c[2][2 * j], w2c[2][2 * j + 1]}; b1p[j] = f2{b1v[2 * j], b1v[2 * j + 1]};
        s0p[j] = f2{0.f, 0.f}; s1p[j] = s0p[j]; s2p[j] = s0p[j]; sbp[j] = s0p[j];
      }
      const f2 one2 = f2{1.f, 1.f}, nl2e = f2{-1.4426950408889634f, -1.4426950408889634f};
#pragma unroll 2
      for (int i = 0; i < GB / 16; ++i) {
        const int r = rl + 16 * i;
        if (r < mrem && c0 < nrem) {
          const float4 g4 = *reinterpret_cast<const float4*>(sG + r * 4);
          const f2 g0 = f2{g4.x, g4.x}, g1 = f2{g4.y, g4.y}, g2 = f2{g4.z, g4.z};
          const float4 za = *reinterpret_cast<const float4*>(sC + r * GB + c0);
          const float4 zb = *reinterpret_cast<const float4*>(sC + r * GB + c0 + 4);
          const f2 zin[4] = {f2{za.x, za.y}, f2{za.z, za.w}, f2{zb.x, zb.y}, f2{zb.z, zb.w}};
          float o[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f2 z = zin[j] + b1p[j];
            const f2 t = z * nl2e;
            const f2 ex = f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            const f2 den = ex + one2;
            const f2 sg = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            const f2 y = z * sg;
            const f2 dy = __builtin_elementwise_fma(g2, w2p[j], __builtin_elementwise_fma(g1, w1p[j], g0 * w0p[j]));
            const f2 dz = dy * (sg * __builtin_elementwise_fma(z, one2 - sg, one2));
            s0p[j] = __builtin_elementwise_fma(g0, y, s0p[j]);
            s1p[j] = __builtin_elementwise_fma(g1, y, s1p[j]);
            s2p[j] = __builtin_elementwise_fma(g2, y, s2p[j]);
            sbp[j] = sbp[j] + dz;
            o[2 * j] = dz.x; o[2 * j + 1] = dz.y;
          }
          const int64_t ci = (int64_t)(m0 + r) * p.ldc + n0 + c0;
          if (vec_store) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + ci) = pack16<bf16_t>(o);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e < nrem) store_any(p.C, p.c_dtype, ci + e, o[e]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0[2 * j] = s0p[j].x; s0[2 * j + 1] = s0p[j].y; s1[2 * j] = s1p[j].x; s1[2 * j + 1] = s1p[j].y;
        s2[2 * j] = s2p[j].x; s2[2 * j + 1] = s2p[j].y; sb[2 * j] = sbp[j].x; sb[2 * j + 1] = sbp[j].y;
      }
    } else
#pragma unroll 2
    for (int i = 0; i < GB / 16; ++i) {
      const int r = rl + 16 * i;
      if (r < mrem && c0 < nrem) {
        float zv[8], o[8];
        const uint4* q = reinterpret_cast<const uint4*>(sC + r * GB + c0);
        unpack16<float>(q[0], zv); unpack16<float>(q[1], zv + 4);
        float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
        if (staged) {
          ga = *reinterpret_cast<const float4*>(sG + (hsel[0] * GB + r) * 4);
          gb = one_head ? ga : *reinterpret_cast<const float4*>(sG + (hsel[7] * GB + r) * 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float g0, g1, g2;
          if (staged) {
            const bool lo = hsel[e] == hsel[0];
            g0 = lo ? ga.x : gb.x; g1 = lo ? ga.y : gb.y; g2 = lo ? ga.z : gb.z;
          } else {   // tiny decoder widths: a tile spans 3+ heads, read the row's dlogits directly
            const int h = h_lo + hsel[e];
            const int Cn = a.classes[h];
            const float sc = a.scale[h];
            const float* dl = a.dlogits[h] + (int64_t)(m0 + r) * Cn;
            g0 = dl[0] * sc; g1 = Cn > 1 ? dl[1] * sc : 0.f; g2 = Cn > 2 ? dl[2] * sc : 0.f;
          }
          // explicit fma: the library is built with -ffp-contract=off and this loop is VALU-bound
          const float z = zv[e] + b1v[e];
          const float sg = sigmoid_f(z);
          const float y = z * sg;
          const float dy = fmaf(g2, w2c[2][e], fmaf(g1, w2c[1][e], g0 * w2c[0][e]));
          const float dz = dy * (sg * fmaf(z, 1.f - sg, 1.f));
          s0[e] = fmaf(g0, y, s0[e]); s1[e] = fmaf(g1, y, s1[e]); s2[e] = fmaf(g2, y, s2[e]); sb[e] += dz;
          o[e] = dz;
        }
        const int64_t ci = (int64_t)(m0 + r) * p.ldc + n0 + c0;
        if (vec_store) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + ci) = pack16<bf16_t>(o);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (c0 + e < nrem) store_any(p.C, p.c_dtype, ci + e, o[e]);
        }
      }
    }
    __syncthreads();                                  // every thread is done with the C tile: reuse it for the reduction
    float* red = sC;                                  // [16 row lanes][4 quantities][128 columns]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rl * 4 + 0) * GB + c0 + e] = s0[e];
      red[(rl * 4 + 1) * GB + c0 + e] = s1[e];
      red[(rl * 4 + 2) * GB + c0 + e] = s2[e];
      red[(rl * 4 + 3) * GB + c0 + e] = sb[e];
    }
    __syncthreads();
    const int64_t ncol = (int64_t)a.num_heads * a.D;
    float* slot = p.dz_ws + (int64_t)((m0 / GB) % GEMM_DZ_SLOTS) * 4 * ncol;
    for (int i = tid; i < 4 * GB; i += 256) {
      const int qn = i >> 7, c = i & (GB - 1);
      if (c < nrem) {
        float sum = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) sum += red[(l * 4 + qn) * GB + c];
        atomicAdd(slot + qn * ncol + n0 + c, sum);
      }
    }
    return;
  }
  // vector path: 8 columns per thread per step (16 threads per row), when every row segment is 16-byte aligned
  const peneo_gemm_epilogue& e = p.ep;
  const int csz = p.c_dtype == PENEO_F32 ? 4 : 2;
  auto al = [&](const void* ptr, int64_t ld, int esz) {
    return ptr == nullptr || (((reinterpret_cast<uintptr_t>(ptr) & 15) == 0) && ((ld * esz) % 16 == 0));
  };
  const bool vec_ok = p.split_k <= 1 && nrem == GB && al(p.C, p.ldc, csz) && al(e.preact, e.ld_preact, csz) &&
                      al(e.grad_src, e.ld_grad, csz) && al(e.residual, e.ld_res, csz) && al(e.bias, 0, 4);
  if (vec_ok) {
    for (int idx = tid; idx < GB * (GB / 8); idx += 256) {
      const int r = idx >> 4, c = (idx & 15) * 8;
      if (r < mrem) {
        float v[8];
        const uint4* q = reinterpret_cast<const uint4*>(sC + r * GB + c);
        unpack16<float>(q[0], v); unpack16<float>(q[1], v + 4);
        epilogue_store8(p, m0 + r, n0 + c, v);
      }
    }
  } else if (p.split_k > 1 && nrem == GB && (p.N & 3) == 0) {
    // split-k partials: raw fp32 rows of the tile, 16 bytes per store (ws is 256-byte aligned, N % 4 == 0)
    float* wsz = p.ws + ((int64_t)zsplit * p.M + m0) * p.N + n0;
    for (int idx = tid; idx < GB * (GB / 4); idx += 256) {
      const int r = idx >> 5, c = (idx & 31) * 4;
      if (r < mrem) *reinterpret_cast<float4*>(wsz + (int64_t)r * p.N + c) = *reinterpret_cast<const float4*>(sC + r * GB + c);
    }
  } else {
    for (int idx = tid; idx < GB * GB; idx += 256) {
      const int r = idx >> 7, c = idx & (GB - 1);
      if (r < mrem && c < nrem) {
        const float v = sC[idx];
        if (p.split_k > 1) p.ws[((int64_t)zsplit * p.M + (m0 + r)) * p.N + (n0 + c)] = v;
        else epilogue_store(p, m0 + r, n0 + c, v);
      }
    }
  }
}

template <typename T, bool AK, bool BK, bool DZ = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KT = ROWB / sizeof(T);             // k elements per tile: 64 (bf16) / 32 (fp32)
  constexpr int KSTEPS = KT / 16;
  char* sA = smem;                                 // [2][TILE_BYTES]
  char* sB = smem + 2 * TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
  DzPre dzpre = {};
  if constexpr (DZ) dz_prefetch(p, m0, n0, tid, dzpre);
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);

  const int ktiles = (p.K + KT - 1) / KT;
  int kt_begin = 0, kt_end = ktiles;
  if (p.split_k > 1) {
    kt_begin = blockIdx.z * p.kt_per_split;
    kt_end = min(ktiles, kt_begin + p.kt_per_split);
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // block-uniform fast-path predicates: tile rows in bounds, vectors 16-byte aligned
  const bool alignA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((p.lda * (int64_t)sizeof(T)) % 16 == 0);
  const bool alignB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((p.ldb * (int64_t)sizeof(T)) % 16 == 0);
  const bool rowsA = m0 + GB <= p.M, rowsB = n0 + GB <= p.N;
  Stage<T> ra, rb;
  auto gload = [&](int kt) {
    const int k0 = kt * KT;
    const bool kfull = k0 + KT <= p.K;
    const bool fa = alignA && rowsA && kfull, fb = alignB && rowsB && kfull;
    if (AK) load_kmajor<T>(ra, A, p.lda, m0, k0, p.M, p.K, tid, fa); else load_mnmajor<T>(ra, A, p.lda, m0, k0, p.M, p.K, tid, fa);
    if (BK) load_kmajor<T>(rb, B, p.ldb, n0, k0, p.N, p.K, tid, fb); else load_mnmajor<T>(rb, B, p.ldb, n0, k0, p.N, p.K, tid, fb);
  };
  auto lstore = [&](int buf) {
    if (AK) store_kmajor<T>(ra, sA + buf * TILE_BYTES, tid); else store_mnmajor<T>(ra, sA + buf * TILE_BYTES, tid);
    if (BK) store_kmajor<T>(rb, sB + buf * TILE_BYTES, tid); else store_mnmajor<T>(rb, sB + buf * TILE_BYTES, tid);
  };

  if (kt_begin < kt_end) {
    gload(kt_begin);
    lstore(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) gload(kt + 1);
    const char* tA = sA + buf * TILE_BYTES;
    const char* tB = sB + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = read_frag<T>(tA, wm * 64 + i * 32 + (lane & 31), ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag<T>(tB, wn * 64 + j * 32 + (lane & 31), ks, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_step(fa[i], fb[j], acc[i][j]);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  tile_epilogue<DZ>(p, acc, smem, m0, n0, tid, lane, wm, wn, dzpre, blockIdx.z);
}

// ================================================================================================
// bf16 fast path: operand tiles go global -> LDS by LDS-DMA (no VGPR staging, no ds_write pass), two
// 32 KiB stages (2 workgroups per CU), one barrier per k-tile.
//   k-major operand  [rows][K]: LDS image = the register-staged kernel's (row pitch 128 B, 16-byte slot s of row r at
//                     slot s ^ ((r >> 1) & 7)); the swizzle is applied to the SOURCE address of each DMA lane, the
//                     destination is lane-linear (1 KiB piece = 8 rows).  Fragments by ds_read_b128.
//   mn-major operand [K][rows]: 1 KiB pieces of [8 k][64 rows] (full 128-byte lines from memory), the two 64-byte
//                     halves of a line swapped when (k >> 1) & 1; fragments by the hardware transpose read
//                     ds_read_b64_tr_b16 (each 16-lane group reads a [4 k][16 rows] block, conflict-free with that swap).
// Requirements (checked by the launcher, else the register-staged kernel runs): 16-byte aligned bases and row strides,
// extents along the contiguous dimension multiples of 8.  Ragged M/N edges read clamped rows (results discarded),
// the k tail reads a zero line.
// ================================================================================================
__device__ __attribute__((aligned(16))) uint32_t g_zero_line[4] = {0u, 0u, 0u, 0u};

typedef short s16x4_t __attribute__((ext_vector_type(4)));

struct DmaSrc { const char* p[4]; };

// source pointers of this wave's 4 pieces of a k-major tile at k-tile `kt` (guard: zero-fill chunks beyond K)
__device__ __forceinline__ void dma_src_kmajor(DmaSrc& s, const bf16_t* X, int64_t ld, int r0, int rmax, int K, int kt,
                                               int wave, int lane, bool guard) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    const int sg = (lane & 7) ^ ((row >> 1) & 7);
    const int k = kt * 64 + sg * 8;
    const int r = min(r0 + row, rmax - 1);
    const char* q = reinterpret_cast<const char*>(X + (int64_t)r * ld + k);
    if (guard && k >= K) q = reinterpret_cast<const char*>(g_zero_line);
    s.p[j] = q;
  }
}
__device__ __forceinline__ void dma_src_mnmajor(DmaSrc& s, const bf16_t* X, int64_t ld, int r0, int rmax, int K, int kt,
                                                int wave, int lane, bool guard) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pc = wave * 4 + j;
    const int kr = lane >> 3;
    const int k = kt * 64 + (pc >> 1) * 8 + kr;
    const int cg = (lane & 7) ^ (((kr >> 1) & 1) << 2);
    const int col = min(r0 + (pc & 1) * 64 + cg * 8, rmax - 8);
    const char* q = reinterpret_cast<const char*>(X + (int64_t)k * ld + col);
    if (guard && k >= K) q = reinterpret_cast<const char*>(g_zero_line);
    s.p[j] = q;
  }
}

// ================================================================================================
// Two-stage kernel with the LDS -> register fragment reads scheduled by hand: the compiler's schedule re-uses one
// fragment register set and waits (lgkmcnt) for each k-step's reads right in front of its MFMAs, exposing the LDS
// latency four times per k-tile.  Here the reads of k-step ks+1 are issued (inline asm, two register sets) before
// the four MFMAs of k-step ks, with counted lgkmcnt waits.
// ================================================================================================
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
#define PENEO_DSR128(dst_, addr_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(OFF_))
#define PENEO_DSRTR(dst_, addr_, OFF_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(OFF_))
#define PENEO_LGKM(n_)                                           \
  {                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #n_ ")" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                           \
  }

// one operand fragment (32 rows x 16 k) as 4 dwords; k-major: one b128 read, mn-major: two transpose reads
struct PFrag { u32x4_t v; };
template <bool KMAJ, int OFF>
__device__ __forceinline__ void pfrag_read(PFrag& f, uint32_t addr) {
  if constexpr (KMAJ) {
    PENEO_DSR128(f.v, addr, OFF);
  } else {
    u32x2_t lo, hi;
    PENEO_DSRTR(lo, addr, OFF);
    PENEO_DSRTR(hi, addr, OFF + 512);
    f.v = u32x4_t{lo.x, lo.y, hi.x, hi.y};
  }
}
__device__ __forceinline__ void pmma(const PFrag& a, const PFrag& b, f32x16_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), acc, 0, 0, 0);
}

// the tile (m0, n0, split slice) is chosen by the __global__ wrappers below: one problem per launch, or several
// independent problems side by side in one launch (peneo_gemm_group)
template <bool AK, bool BK, bool CS = false, bool DZ = false>
__device__ __forceinline__ void gemm_dma_pipe_body(const GemmParams& p, char* smem, const int m0, const int n0, const int zsplit,
                                                   const int cs_col = 0, const int cs_mod = 1) {
  constexpr int STAGE = 2 * TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  DzPre dzpre = {};
  if constexpr (DZ) dz_prefetch(p, m0, n0, tid, dzpre);

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const int ktiles = (p.K + 63) / 64;
  int kt_begin = 0, kt_end = ktiles;
  if (p.split_k > 1) {
    kt_begin = zsplit * p.kt_per_split;
    kt_end = min(ktiles, kt_begin + p.kt_per_split);
  }
  const bool ragged_k = (p.K & 63) != 0;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = lds_addr(smem);
  const uint32_t lbase = lds0 + wave * 4096;
  const int64_t stepA = AK ? 128 : (int64_t)64 * p.lda * 2, stepB = BK ? 128 : (int64_t)64 * p.ldb * 2;
  DmaSrc sa, sb;
  auto sources = [&](int kt) {
    const bool guard = ragged_k && kt == ktiles - 1;
    if (AK) dma_src_kmajor(sa, A, p.lda, m0, p.M, p.K, kt, wave, lane, guard);
    else dma_src_mnmajor(sa, A, p.lda, m0, p.M, p.K, kt, wave, lane, guard);
    if (BK) dma_src_kmajor(sb, B, p.ldb, n0, p.N, p.K, kt, wave, lane, guard);
    else dma_src_mnmajor(sb, B, p.ldb, n0, p.N, p.K, kt, wave, lane, guard);
  };
  // Operand addressing of the k loop: per-lane 32-bit byte offsets inside the tile (constant over k) + a uniform 64-bit base
  // per operand that walks k with scalar adds.  (The ragged last k-tile keeps the per-lane pointer form: its guard swaps
  // single lanes' sources for a zero line.)
  uint32_t oa[4], ob[4];
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (AK) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int sg = (lane & 7) ^ ((row >> 1) & 7);
        oa[j] = (uint32_t)(((int64_t)(min(m0 + row, p.M - 1) - m0) * p.lda + sg * 8) * 2);
      } else {
        const int pc = wave * 4 + j, kr = lane >> 3;
        const int cg = (lane & 7) ^ (((kr >> 1) & 1) << 2);
        oa[j] = (uint32_t)(((int64_t)((pc >> 1) * 8 + kr) * p.lda + (min(m0 + (pc & 1) * 64 + cg * 8, p.M - 8) - m0)) * 2);
      }
      if (BK) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int sg = (lane & 7) ^ ((row >> 1) & 7);
        ob[j] = (uint32_t)(((int64_t)(min(n0 + row, p.N - 1) - n0) * p.ldb + sg * 8) * 2);
      } else {
        const int pc = wave * 4 + j, kr = lane >> 3;
        const int cg = (lane & 7) ^ (((kr >> 1) & 1) << 2);
        ob[j] = (uint32_t)(((int64_t)((pc >> 1) * 8 + kr) * p.ldb + (min(n0 + (pc & 1) * 64 + cg * 8, p.N - 8) - n0)) * 2);
      }
    }
  }
  // uniform tile bases at k-tile kt (AK: row m0, column kt * 64; mn-major: row kt * 64, column m0)
  auto base_of = [&](const bf16_t* X, int64_t ld, int r0, bool kmaj, int kt) -> const char* {
    const char* q = reinterpret_cast<const char*>(kmaj ? X + (int64_t)r0 * ld + (int64_t)kt * 64 : X + (int64_t)kt * 64 * ld + r0);
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  const char* bA = base_of(A, p.lda, m0, AK, kt_begin);
  const char* bB = base_of(B, p.ldb, n0, BK, kt_begin);
  bool by_pointer = false;       // the tile about to be issued is the guarded (ragged) one: sa / sb hold per-lane pointers
  auto issue = [&](int buf) {
    const uint32_t d = lbase + buf * STAGE;
    if (by_pointer) {
      lds_dma_1k<0>(sa.p[0], d);
      lds_dma_1k<0>(sa.p[1], d + 1024);
      lds_dma_1k<0>(sa.p[2], d + 2048);
      lds_dma_1k<0>(sa.p[3], d + 3072);
      lds_dma_1k<0>(sb.p[0], d + TILE_BYTES);
      lds_dma_1k<0>(sb.p[1], d + TILE_BYTES + 1024);
      lds_dma_1k<0>(sb.p[2], d + TILE_BYTES + 2048);
      lds_dma_1k<0>(sb.p[3], d + TILE_BYTES + 3072);
    } else {
      lds_dma_1k_s<0>(oa[0], bA, d);
      lds_dma_1k_s<0>(oa[1], bA, d + 1024);
      lds_dma_1k_s<0>(oa[2], bA, d + 2048);
      lds_dma_1k_s<0>(oa[3], bA, d + 3072);
      lds_dma_1k_s<0>(ob[0], bB, d + TILE_BYTES);
      lds_dma_1k_s<0>(ob[1], bB, d + TILE_BYTES + 1024);
      lds_dma_1k_s<0>(ob[2], bB, d + TILE_BYTES + 2048);
      lds_dma_1k_s<0>(ob[3], bB, d + TILE_BYTES + 3072);
    }
  };
  auto advance = [&]() { bA += stepA; bB += stepB; };
  // one quarter of the next k-tile's pieces (A piece q, B piece q): issued BETWEEN the MFMA steps of the current k-tile, not as
  // a block of eight in front of its first fragment reads (each issue holds the wave for 60+ cycles while MFMAs run)
  bool nxt_on = false;
  uint32_t nxt_d = 0;
  auto issue_q = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    if (nxt_on) {
      lds_dma_1k_s<0>(oa[q], bA, nxt_d + q * 1024);
      lds_dma_1k_s<0>(ob[q], bB, nxt_d + TILE_BYTES + q * 1024);
    }
  };

  // LDS byte addresses of the fragments inside stage 0 (the stage and the second 32-row block are immediates):
  //   k-major : one address per k-step (the XOR swizzle depends on ks), second row block = +4096
  //   mn-major: one address per row block, k-step = +4096 (pieces), second transpose read = +512
  uint32_t aaddr[4], baddr[4];
  {
    const int half = lane >> 5;
    if (AK) {
      const int row = wm * 64 + (lane & 31), swz = (row >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) aaddr[ks] = lds0 + row * ROWB + (((2 * ks + half) ^ swz) << 4);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = wm * 64 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        aaddr[i] = lds0 + (half * 2 + (n >> 6)) * 1024 + ((lane & 15) >> 2) * 128 + (((n & 63) * 2) ^ (((lane >> 3) & 1) << 6));
      }
      aaddr[2] = aaddr[3] = 0;
    }
    if (BK) {
      const int row = wn * 64 + (lane & 31), swz = (row >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) baddr[ks] = lds0 + TILE_BYTES + row * ROWB + (((2 * ks + half) ^ swz) << 4);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = wn * 64 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        baddr[i] = lds0 + TILE_BYTES + (half * 2 + (n >> 6)) * 1024 + ((lane & 15) >> 2) * 128 +
                   (((n & 63) * 2) ^ (((lane >> 3) & 1) << 6));
      }
      baddr[2] = baddr[3] = 0;
    }
  }
  PFrag fa[2][2], fb[2][2];   // [register set][row block]
  // CS: a_colsum[m] += sum_k A(m, k) (the bias gradient beside a weight gradient): one extra MFMA per k-step against a
  // ones operand -- every column of accb is the row sum.  The work is dealt evenly: the workgroups of one tile row share
  // the same A tiles, tile column j takes the k-tiles with kt % (tile columns) == j, and of a workgroup's two waves on the
  // same 64 rows wave wn takes the 32-row block wn; everybody adds its partial sums with fp32 atomics.
  const bool cs_on = CS && p.ep.a_colsum != nullptr;
  bool cs_now = false;
  f32x16_t accb;
  PFrag ones;
  ones.v = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  if constexpr (CS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  }
#define PENEO_READ_STEP(SET_, KS_, BUFOFF_)                                                                     \
  {                                                                                                            \
    if constexpr (AK) {                                                                                        \
      pfrag_read<true, BUFOFF_>(fa[SET_][0], aaddr[KS_]);                                                      \
      pfrag_read<true, BUFOFF_ + 4096>(fa[SET_][1], aaddr[KS_]);                                               \
    } else {                                                                                                   \
      pfrag_read<false, BUFOFF_ + (KS_) * 4096>(fa[SET_][0], aaddr[0]);                                        \
      pfrag_read<false, BUFOFF_ + (KS_) * 4096>(fa[SET_][1], aaddr[1]);                                        \
    }                                                                                                          \
    if constexpr (BK) {                                                                                        \
      pfrag_read<true, BUFOFF_>(fb[SET_][0], baddr[KS_]);                                                      \
      pfrag_read<true, BUFOFF_ + 4096>(fb[SET_][1], baddr[KS_]);                                               \
    } else {                                                                                                   \
      pfrag_read<false, BUFOFF_ + (KS_) * 4096>(fb[SET_][0], baddr[0]);                                        \
      pfrag_read<false, BUFOFF_ + (KS_) * 4096>(fb[SET_][1], baddr[1]);                                        \
    }                                                                                                          \
  }
#define PENEO_MMA_STEP(SET_)                 \
  {                                          \
    pmma(fa[SET_][0], fb[SET_][0], acc[0][0]); \
    pmma(fa[SET_][0], fb[SET_][1], acc[0][1]); \
    pmma(fa[SET_][1], fb[SET_][0], acc[1][0]); \
    pmma(fa[SET_][1], fb[SET_][1], acc[1][1]); \
    if constexpr (CS) {                        \
      if (cs_now) { if (wn == 0) pmma(fa[SET_][0], ones, accb); else pmma(fa[SET_][1], ones, accb); } \
    }                                          \
  }
  // DS instructions per read step (for the counted waits): 4 with two k-major operands, 8 with two mn-major ones
  constexpr int RPS = (AK ? 2 : 4) + (BK ? 2 : 4);
#define PENEO_WAIT_NEWER()                                      \
  {                                                             \
    if constexpr (RPS == 4) PENEO_LGKM(4)                       \
    else if constexpr (RPS == 6) PENEO_LGKM(6)                  \
    else PENEO_LGKM(8)                                          \
  }
#define PENEO_KTILE(BUFOFF_)             \
  {                                      \
    PENEO_READ_STEP(0, 0, BUFOFF_)       \
    PENEO_READ_STEP(1, 1, BUFOFF_)       \
    issue_q(std::integral_constant<int, 0>{}); \
    PENEO_WAIT_NEWER()                   \
    PENEO_MMA_STEP(0)                    \
    PENEO_READ_STEP(0, 2, BUFOFF_)       \
    issue_q(std::integral_constant<int, 1>{}); \
    PENEO_WAIT_NEWER()                   \
    PENEO_MMA_STEP(1)                    \
    PENEO_READ_STEP(1, 3, BUFOFF_)       \
    issue_q(std::integral_constant<int, 2>{}); \
    PENEO_WAIT_NEWER()                   \
    PENEO_MMA_STEP(0)                    \
    issue_q(std::integral_constant<int, 3>{}); \
    PENEO_LGKM(0)                        \
    PENEO_MMA_STEP(1)                    \
  }

  if (kt_begin < kt_end) {
    by_pointer = ragged_k && kt_begin == ktiles - 1;
    if (by_pointer) sources(kt_begin);
    issue(0);
  }
  wait_vm<0>();
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; kt += 2) {
    nxt_on = false;
    if (kt + 1 < kt_end) {
      by_pointer = ragged_k && kt + 1 == ktiles - 1;
      if (by_pointer) { sources(kt + 1); issue(1); }
      else { advance(); nxt_on = true; nxt_d = lbase + STAGE; }
    }
    if constexpr (CS) cs_now = cs_on && (kt % cs_mod) == cs_col;
    PENEO_KTILE(0)
    wait_vm<0>();
    __syncthreads();
    if (kt + 1 < kt_end) {
      nxt_on = false;
      if (kt + 2 < kt_end) {
        by_pointer = ragged_k && kt + 2 == ktiles - 1;
        if (by_pointer) { sources(kt + 2); issue(0); }
        else { advance(); nxt_on = true; nxt_d = lbase; }
      }
      if constexpr (CS) cs_now = cs_on && ((kt + 1) % cs_mod) == cs_col;
      PENEO_KTILE(32768)
      wait_vm<0>();
      __syncthreads();
    }
  }
#undef PENEO_KTILE
#undef PENEO_WAIT_NEWER
#undef PENEO_MMA_STEP
#undef PENEO_READ_STEP
  if constexpr (CS) {
    if (cs_on && (lane & 31) == 0) {      // column 0 of the 32 x 32 block: lanes 0 and 32 hold 16 rows each
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + wn * 32 + acc_row(r, lane);
        if (m < p.M) atomicAdd(p.ep.a_colsum + m, accb[r]);
      }
    }
  }
  tile_epilogue<DZ>(p, acc, smem, m0, n0, tid, lane, wm, wn, dzpre, zsplit);
}

template <bool AK, bool BK, bool CS = false, bool DZ = false>
__global__ __launch_bounds__(256, 2) void gemm_dma_pipe_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int gx = gridDim.x, gxy = gx * gridDim.y, total = gxy * gridDim.z;
  const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, slot = lin >> 3;
  const int tile3 = xcd * q8 + min(xcd, r8) + slot;
  const int zsplit = tile3 / gxy, tile = tile3 - zsplit * gxy;
  gemm_dma_pipe_body<AK, BK, CS, DZ>(p, smem, (tile / gx) * GB, (tile % gx) * GB, zsplit, tile % gx, gx);
}

// Several independent GEMMs of one operand layout in ONE launch (no split-k): the tiles of all problems are numbered
// consecutively, the XCD-aware order runs over the whole launch.  Used for the four weight gradients of an encoder layer:
// 432 tiles with the full K = tokens each, instead of four split-k launches of ~450 short workgroups plus four reductions.
constexpr int GEMM_GROUP_MAX = 4;
struct GemmGroup {
  GemmParams p[GEMM_GROUP_MAX];
  int first_tile[GEMM_GROUP_MAX + 1];
  int gx[GEMM_GROUP_MAX];
  int n;
};
template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void gemm_dma_pipe_group_kernel(GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = gridDim.x, lin = blockIdx.x;
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, slot = lin >> 3;
  const int tile3 = xcd * q8 + min(xcd, r8) + slot;
  int i = 0;
#pragma unroll
  for (int k = 1; k < GEMM_GROUP_MAX; ++k)
    if (k < g.n && tile3 >= g.first_tile[k]) i = k;
  const int tile = tile3 - g.first_tile[i], gx = g.gx[i];
  gemm_dma_pipe_body<AK, BK>(g.p[i], smem, (tile / gx) * GB, (tile % gx) * GB, 0);
}

__global__ void splitk_reduce_kernel(GemmParams p) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)p.M * p.N;
  if (idx >= total) return;
  float s = 0.f;
  for (int z = 0; z < p.split_k; ++z) s += p.ws[(int64_t)z * total + idx];
  epilogue_store(p, (int)(idx / p.N), (int)(idx % p.N), s);
}
// 8 consecutive columns per thread (N % 8 == 0, every operand of the epilogue 16-byte aligned per row): two 16-byte loads
// per slice, four slices in flight.  The scalar kernel above moved 4 bytes per load and reached ~1.2 TB/s on the encoder's
// weight gradients (14 slices of 2.4 MB: 26 us); same summation order (slice 0 first), so results are bit-identical.
__global__ __launch_bounds__(256) void splitk_reduce8_kernel(GemmParams p) {
  const int64_t v8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)p.M * p.N;
  const int64_t idx = v8 * 8;
  if (idx >= total) return;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const float* src = p.ws + idx;
  int z = 0;
  for (; z + 4 <= p.split_k; z += 4) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const float4*>(src + (int64_t)(z + u) * total);
      b[u] = *reinterpret_cast<const float4*>(src + (int64_t)(z + u) * total + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] += a[u].x; acc[1] += a[u].y; acc[2] += a[u].z; acc[3] += a[u].w;
      acc[4] += b[u].x; acc[5] += b[u].y; acc[6] += b[u].z; acc[7] += b[u].w;
    }
  }
  for (; z < p.split_k; ++z) {
    const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)z * total);
    const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)z * total + 4);
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
  }
  epilogue_store8(p, (int)(idx / p.N), (int)(idx % p.N), acc);
}

template <typename T>
static int launch_gemm(const GemmParams& p, bool ak, bool bk, dim3 grid, hipStream_t st) {
  size_t shmem = 4 * TILE_BYTES + (p.dz_on ? 2 * GB * 4 * sizeof(float) : 0);
  if (p.dz_on) {
    if (!(ak && bk)) { set_error("peneo_gemm: pair_dz needs k-major A and B"); return PENEO_ERR_INVALID; }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) { set_error("peneo_gemm: cannot raise dynamic LDS"); return PENEO_ERR_LAUNCH; }
    hipLaunchKernelGGL((gemm_kernel<T, true, true, true>), grid, dim3(256), shmem, st, p);
    return check_launch("peneo_gemm");
  }
  if (ak && bk) hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(256), shmem, st, p);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(256), shmem, st, p);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, dim3(256), shmem, st, p);
  else hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(256), shmem, st, p);
  return check_launch("peneo_gemm");
}

static int launch_gemm_dma_pipe(const GemmParams& p, bool ak, bool bk, dim3 grid, hipStream_t st) {
  size_t shmem = 4 * TILE_BYTES + (p.dz_on ? 2 * GB * 4 * sizeof(float) : 0);
  if (p.dz_on) {
    if (!(ak && bk)) { set_error("peneo_gemm: pair_dz needs k-major A and B"); return PENEO_ERR_INVALID; }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_pipe_kernel<true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) { set_error("peneo_gemm: cannot raise dynamic LDS"); return PENEO_ERR_LAUNCH; }
    hipLaunchKernelGGL((gemm_dma_pipe_kernel<true, true, false, true>), grid, dim3(256), shmem, st, p);
    return check_launch("peneo_gemm");
  }
  if (ak && bk) hipLaunchKernelGGL((gemm_dma_pipe_kernel<true, true>), grid, dim3(256), shmem, st, p);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_dma_pipe_kernel<true, false>), grid, dim3(256), shmem, st, p);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_dma_pipe_kernel<false, true>), grid, dim3(256), shmem, st, p);
  else if (p.ep.a_colsum) hipLaunchKernelGGL((gemm_dma_pipe_kernel<false, false, true>), grid, dim3(256), shmem, st, p);
  else hipLaunchKernelGGL((gemm_dma_pipe_kernel<false, false>), grid, dim3(256), shmem, st, p);
  return check_launch("peneo_gemm");
}


}  // namespace peneo

using namespace peneo;

extern "C" size_t peneo_gemm_workspace_bytes(int M, int N, int K, int split_k) {
  (void)K;
  return split_k > 1 ? (size_t)split_k * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int peneo_gemm_group(int dtype, int a_kmajor, int b_kmajor, int c_dtype, const peneo_gemm_problem* problems, int n,
                                peneo_stream_t stream) {
  PENEO_REQUIRE(dtype == PENEO_BF16, "peneo_gemm_group: bf16 operands only");
  PENEO_REQUIRE(c_dtype == PENEO_F32 || c_dtype == PENEO_BF16, "peneo_gemm_group: bad c_dtype %d", c_dtype);
  PENEO_REQUIRE(problems && n >= 1 && n <= GEMM_GROUP_MAX, "peneo_gemm_group: 1..%d problems (got %d)", GEMM_GROUP_MAX, n);
  auto dma_ok = [](const void* ptr, int64_t ld, bool kmajor, int rows, int k) {
    return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (ld % 8) == 0 && ((kmajor ? k : rows) % 8) == 0 && rows >= 8;
  };
  GemmGroup g;
  g.n = n;
  int tiles = 0;
  for (int i = 0; i < GEMM_GROUP_MAX; ++i) {
    const peneo_gemm_problem& q = problems[i < n ? i : n - 1];
    if (i < n) {
      PENEO_REQUIRE(q.M > 0 && q.N > 0 && q.K > 0 && q.A && q.B && q.C, "peneo_gemm_group: problem %d is empty or has a null operand", i);
      PENEO_REQUIRE(q.lda >= (a_kmajor ? q.K : q.M) && q.ldb >= (b_kmajor ? q.K : q.N) && q.ldc >= q.N,
                    "peneo_gemm_group: problem %d: leading dim too small", i);
      PENEO_REQUIRE(dma_ok(q.A, q.lda, a_kmajor != 0, q.M, q.K) && dma_ok(q.B, q.ldb, b_kmajor != 0, q.N, q.K),
                    "peneo_gemm_group: problem %d: operands must be 16-byte aligned with extents in whole 16-byte chunks", i);
      PENEO_REQUIRE(!q.accumulate || c_dtype == PENEO_F32, "peneo_gemm_group: accumulate needs an fp32 C");
    }
    GemmParams& p = g.p[i];
    p.A = q.A; p.B = q.B; p.C = q.C; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.M = q.M; p.N = q.N; p.K = q.K;
    p.c_dtype = c_dtype;
    peneo_gemm_epilogue z = {};
    p.ep = (i < n && q.ep) ? *q.ep : z;
    if (i < n && q.ep) {
      PENEO_REQUIRE(!q.ep->pair_dz && !q.ep->a_colsum, "peneo_gemm_group: problem %d: pair_dz / a_colsum are not available in a group", i);
      PENEO_REQUIRE(q.ep->drop_p >= 0.f && q.ep->drop_p < 1.f, "peneo_gemm_group: problem %d: drop_p out of range", i);
      PENEO_REQUIRE(!q.ep->accumulate || c_dtype == PENEO_F32, "peneo_gemm_group: accumulate needs an fp32 C");
    }
    if (p.ep.alpha == 0.f) p.ep.alpha = 1.f;
    if (q.accumulate) p.ep.accumulate = 1;
    p.split_k = 1; p.ws = nullptr; p.kt_per_split = (q.K + 63) / 64; p.dz_on = 0; p.dz_ws = nullptr;
    peneo_pair_dz_args za = {};
    p.dz = za;
    g.first_tile[i] = tiles;
    g.gx[i] = (q.N + GB - 1) / GB;
    if (i < n) tiles += g.gx[i] * ((q.M + GB - 1) / GB);
  }
  g.first_tile[GEMM_GROUP_MAX] = tiles;
  for (int i = n; i < GEMM_GROUP_MAX; ++i) g.first_tile[i] = tiles;
  const int shmem = 2 * 2 * TILE_BYTES;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool ak = a_kmajor != 0, bk = b_kmajor != 0;
#define PENEO_GROUP_LAUNCH(AK_, BK_)                                                                                         \
  {                                                                                                                          \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_pipe_group_kernel<AK_, BK_>),  \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, shmem);                   \
    if (attr != hipSuccess) { set_error("peneo_gemm_group: cannot raise dynamic LDS to %d bytes", shmem); return PENEO_ERR_LAUNCH; } \
    hipLaunchKernelGGL((gemm_dma_pipe_group_kernel<AK_, BK_>), dim3((unsigned)tiles), dim3(256), shmem, st, g);              \
  }
  if (ak && bk) PENEO_GROUP_LAUNCH(true, true)
  else if (ak && !bk) PENEO_GROUP_LAUNCH(true, false)
  else if (!ak && bk) PENEO_GROUP_LAUNCH(false, true)
  else PENEO_GROUP_LAUNCH(false, false)
#undef PENEO_GROUP_LAUNCH
  return check_launch("peneo_gemm_group");
}

extern "C" int peneo_gemm(int dtype, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int64_t lda,
                          const void* B, int64_t ldb, void* C, int64_t ldc, int c_dtype,
                          const peneo_gemm_epilogue* ep, int split_k, void* workspace, size_t workspace_bytes,
                          peneo_stream_t stream) {
  PENEO_REQUIRE(dtype == PENEO_F32 || dtype == PENEO_BF16, "peneo_gemm: bad dtype %d", dtype);
  PENEO_REQUIRE(c_dtype == PENEO_F32 || c_dtype == PENEO_BF16, "peneo_gemm: bad c_dtype %d", c_dtype);
  PENEO_REQUIRE(M > 0 && N > 0 && K > 0, "peneo_gemm: empty problem %dx%dx%d", M, N, K);
  PENEO_REQUIRE(A && B && C, "peneo_gemm: null operand");
  PENEO_REQUIRE(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N) && ldc >= N, "peneo_gemm: leading dim too small");
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.c_dtype = c_dtype;
  if (ep) p.ep = *ep; else { peneo_gemm_epilogue z = {}; p.ep = z; }
  if (p.ep.alpha == 0.f) p.ep.alpha = 1.f;
  p.dz_on = 0; p.dz_ws = nullptr;
  if (p.ep.pair_dz) {
    const peneo_pair_dz_args* a = p.ep.pair_dz;
    PENEO_REQUIRE(p.ep.pair_dz_ws, "peneo_gemm: pair_dz needs its workspace");
    PENEO_REQUIRE(a->drop_p == 0.f, "peneo_gemm: the pair_dz epilogue has no classifier dropout (peneo_pair_bwd_fused / peneo_pair_dz do)");
    PENEO_REQUIRE(a->num_heads > 0 && a->num_heads <= PENEO_MAX_HEADS && a->D > 0 && (int64_t)a->num_heads * a->D == N && a->scale,
                  "peneo_gemm: pair_dz expects N == num_heads * D");
    for (int h = 0; h < a->num_heads; ++h)
      PENEO_REQUIRE(a->dlogits[h] && a->w2[h] && a->classes[h] >= 1 && a->classes[h] <= 3, "peneo_gemm: pair_dz head %d incomplete", h);
    PENEO_REQUIRE(p.ep.act == PENEO_ACT_NONE && !p.ep.preact && !p.ep.grad_src && !p.ep.residual && !p.ep.accumulate &&
                  p.ep.drop_p == 0.f && p.ep.alpha == 1.f, "peneo_gemm: pair_dz cannot be combined with other epilogue options");
    split_k = 1;
    p.dz_on = 1; p.dz = *a; p.dz_ws = p.ep.pair_dz_ws;
  }
  PENEO_REQUIRE(!p.ep.accumulate || c_dtype == PENEO_F32, "peneo_gemm: accumulate needs an fp32 C");
  PENEO_REQUIRE(!p.ep.a_colsum || !a_kmajor, "peneo_gemm: a_colsum needs A as [K, M] (a_kmajor = 0)");
  bool cs_fused = false;
  PENEO_REQUIRE(p.ep.drop_p >= 0.f && p.ep.drop_p < 1.f, "peneo_gemm: drop_p out of range");
  const int KT = dtype == PENEO_BF16 ? 64 : 32;
  const int ktiles = (K + KT - 1) / KT;
  if (split_k < 1) split_k = 1;
  if (split_k > ktiles) split_k = ktiles;
  p.kt_per_split = (ktiles + split_k - 1) / split_k;
  split_k = (ktiles + p.kt_per_split - 1) / p.kt_per_split;
  p.split_k = split_k;
  p.ws = reinterpret_cast<float*>(workspace);
  if (split_k > 1)
    PENEO_REQUIRE(workspace && workspace_bytes >= peneo_gemm_workspace_bytes(M, N, K, split_k),
                  "peneo_gemm: split-k workspace too small");
  dim3 grid((N + GB - 1) / GB, (M + GB - 1) / GB, split_k);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // bf16 LDS-DMA path: 16-byte aligned bases / row strides, contiguous extents in whole 16-byte chunks
  auto dma_ok = [](const void* ptr, int64_t ld, bool kmajor, int rows, int k) {
    return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (ld % 8) == 0 && ((kmajor ? k : rows) % 8) == 0 && rows >= 8;
  };
  int rc = PENEO_OK;
  bool launched = false;
  if (dtype == PENEO_BF16 && a_kmajor && dma_ok(A, lda, true, M, K) && dma_ok(B, ldb, b_kmajor != 0, N, K)) {
    // large forward problems: one 8-wave workgroup per CU on 256 x 256 / 384 x 192 / 256 x 128 tiles (gemm_big.hip, split_k == 1)
    // first choice: the persistent launch (gemm_sk.hip) where its rules pick the problem -- it balances a deep reduction itself
    // (stream-k ranges), so a split the caller asked for is dropped with it; then the tiled 8-wave kernels
    GemmParams q = p;
    q.split_k = 1;
    int big = launch_gemm_sk(q, b_kmajor != 0, st);
    if (big < 0) return big;
    if (big == 1) split_k = 1;
    if (big == 0) big = launch_gemm_big(p, b_kmajor != 0, st);
    if (big < 0) return big;
    launched = big == 1;
  }
  if (launched) {
  } else if (dtype == PENEO_BF16 && dma_ok(A, lda, a_kmajor != 0, M, K) && dma_ok(B, ldb, b_kmajor != 0, N, K))
  {
    // the two-stage LDS-DMA kernel with hand-scheduled fragment reads (the compiler-scheduled two-stage form and a four-stage
    // k-tile-32 form were measured behind it on every shape and removed in round 4)
    rc = launch_gemm_dma_pipe(p, a_kmajor != 0, b_kmajor != 0, grid, st);
    cs_fused = !a_kmajor && !b_kmajor;       // the (mn-major, mn-major) instantiation sums A's columns itself
  }
  else
    rc = dtype == PENEO_BF16 ? launch_gemm<bf16_t>(p, a_kmajor != 0, b_kmajor != 0, grid, st)
                             : launch_gemm<float>(p, a_kmajor != 0, b_kmajor != 0, grid, st);
  if (rc != PENEO_OK) return rc;
  if (p.ep.a_colsum && !cs_fused) {     // every other kernel: the column sums as their own pass over A = [K, M]
    rc = peneo_colsum(dtype, A, lda, K, M, p.ep.a_colsum, 1, stream);
    if (rc != PENEO_OK) return rc;
  }
  if (split_k > 1) {
    int64_t total = (int64_t)M * N;
    const peneo_gemm_epilogue& e = p.ep;
    const int csz = p.c_dtype == PENEO_F32 ? 4 : 2;
    auto al = [&](const void* ptr, int64_t ld, int esz) {
      return ptr == nullptr || (((reinterpret_cast<uintptr_t>(ptr) & 15) == 0) && ((ld * esz) % 16 == 0));
    };
    const bool vec = (N % 8 == 0) && al(p.ws, 0, 4) && al(p.C, p.ldc, csz) && al(e.preact, e.ld_preact, csz) &&
                     al(e.grad_src, e.ld_grad, csz) && al(e.residual, e.ld_res, csz) && al(e.bias, 0, 4);
    if (vec) hipLaunchKernelGGL(splitk_reduce8_kernel, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return check_launch("peneo_gemm(split-k reduce)");
  }
  return PENEO_OK;
}
