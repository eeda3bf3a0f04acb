// K6: attention core for gfx950, flash-style (the [T, T] score matrix never reaches HBM).
//
// Layout trick used throughout: every product is arranged so that the *query* (forward, dQ kernel)
// or the *key* (dK/dV kernel) index is the MFMA "n" index = lane & 31.  Then all per-row softmax
// state (running max / sum, lse, delta) is lane-local, the probability tile that comes out of one
// MFMA in C-layout (lane = column, 16 rows in registers) is directly the B operand of the next MFMA
// (reduction index = those rows, under a fixed permutation that the A-side tiles follow), and no
// cross-lane shuffles other than one lane^32 exchange per reduction are needed.
//
//   forward :  S^T[key, q] = K . Q^T          O^T[d, q]  = V^T . P^T
//   dQ      :  S^T, dP^T[key, q] = V . dO^T   dQ^T[d, q] = K^T . dS^T
//   dK/dV   :  S[q, key] = Q . K^T, dP[q, key] = dO . V^T,  dV^T[d, key] = dO^T . P,  dK^T[d, key] = Q^T . dS
//
// scores = scale * q.k + bias[b, h, q, key]; keys with key_mask == 0 (and the tail beyond T) get
// probability exactly 0 (reference adds finfo.min: modeling_layoutlmv3.py:383-389, 1126-1128).
#include "common.h"

namespace peneo {

constexpr int AQ = 128;   // rows of the "n" side per workgroup (4 waves x 32)
constexpr int AK = 64;    // rows of the streamed side per tile
constexpr float NEG_BIG = -3.0e38f;

template <typename T> struct Sz { static constexpr int v = sizeof(T); };

// ---- LDS tiles: row-major [rows][cols] of T, row pitch = cols*sizeof(T) + 16 bytes (odd number of
//      16-byte slots => 16 consecutive rows hit 16 different slots: conflict-free b128 column reads)
template <typename T, int COLS> struct Pitch { static constexpr int v = COLS * (int)sizeof(T) + 16; };

// FragReader::straight : 8 consecutive elements starting at column c (c multiple of 8) -> MFMA fragment
template <typename T, int COLS> struct FragReader;
template <int COLS> struct FragReader<bf16_t, COLS> {
  __device__ static __forceinline__ Frag<bf16_t> straight(const char* tile, int row, int c) {
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const uint4*>(tile + row * Pitch<bf16_t, COLS>::v + c * 2);
    return f;
  }
  // elements {c..c+3} and {c+8..c+11}
  __device__ static __forceinline__ Frag<bf16_t> perm(const char* tile, int row, int c) {
    const char* p = tile + row * Pitch<bf16_t, COLS>::v + c * 2;
    uint2 a = *reinterpret_cast<const uint2*>(p);
    uint2 b = *reinterpret_cast<const uint2*>(p + 16);
    Frag<bf16_t> f;
    f.v = make_uint4(a.x, a.y, b.x, b.y);
    return f;
  }
};
template <int COLS> struct FragReader<float, COLS> {
  __device__ static __forceinline__ Frag<float> straight(const char* tile, int row, int c) {
    const char* p = tile + row * Pitch<float, COLS>::v + c * 4;
    Frag<float> f;
    f.v[0] = *reinterpret_cast<const uint4*>(p);
    f.v[1] = *reinterpret_cast<const uint4*>(p + 16);
    return f;
  }
  __device__ static __forceinline__ Frag<float> perm(const char* tile, int row, int c) {
    const char* p = tile + row * Pitch<float, COLS>::v + c * 4;
    Frag<float> f;
    f.v[0] = *reinterpret_cast<const uint4*>(p);
    f.v[1] = *reinterpret_cast<const uint4*>(p + 32);
    return f;
  }
};

// pack_frag8 of 8 accumulator registers (r0..r0+7) of a C-layout tile gives a fragment whose element t is row
// 2*r0 + (t&3) + 8*(t>>2) + 4*(lane>>5): exactly what FragReader::perm(col = 2*r0 + 4*half) reads on the A side.
template <typename T> __device__ __forceinline__ Frag<T> pack_acc(const float* v) { return pack_frag8<T>(v); }

// ---- cooperative tile staging (256 threads) ---------------------------------------------------------
// dst[r][c] = src[(r0 + r) * ld + c] for r < ROWS, c < COLS; zero where r0 + r >= rmax or c >= cmax
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void stage_rowmajor(char* tile, const T* src, int64_t ld, int r0, int rmax, int cmax, int tid) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int VPR = COLS / VEC;
  for (int v = tid; v < ROWS * VPR; v += 256) {
    int r = v / VPR, c = (v % VPR) * VEC;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (r0 + r < rmax && c < cmax) {
      const T* p = src + (int64_t)(r0 + r) * ld + c;
      if (c + VEC <= cmax && (reinterpret_cast<uintptr_t>(p) & 15) == 0) val = *reinterpret_cast<const uint4*>(p);
      else {
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = (c + e < cmax) ? Elem<T>::load(p + e) : 0.f;
        val = pack16<T>(f);
      }
    }
    *reinterpret_cast<uint4*>(tile + r * Pitch<T, COLS>::v + c * (int)sizeof(T)) = val;
  }
}
// transposed: dst[c][r] = src[(r0 + r) * ld + c]; dst is a [COLS][ROWS] tile
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void stage_transposed(char* tile, const T* src, int64_t ld, int r0, int rmax, int cmax, int tid) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int VPR = COLS / VEC;
  for (int v = tid; v < (ROWS / 4) * VPR; v += 256) {
    int rg = v / VPR, c = (v % VPR) * VEC;
    float f[4][VEC];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = r0 + rg * 4 + i;
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[i][e] = 0.f;
      if (r < rmax && c < cmax) {
        const T* p = src + (int64_t)r * ld + c;
        if (c + VEC <= cmax && (reinterpret_cast<uintptr_t>(p) & 15) == 0) unpack16<T>(*reinterpret_cast<const uint4*>(p), f[i]);
        else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[i][e] = (c + e < cmax) ? Elem<T>::load(p + e) : 0.f;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      char* d = tile + (c + e) * Pitch<T, ROWS>::v + rg * 4 * (int)sizeof(T);
      if (sizeof(T) == 2) *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16x2(f[0][e], f[1][e]), pack_bf16x2(f[2][e], f[3][e]));
      else *reinterpret_cast<uint4*>(d) = make_uint4(__float_as_uint(f[0][e]), __float_as_uint(f[1][e]),
                                                      __float_as_uint(f[2][e]), __float_as_uint(f[3][e]));
    }
  }
}

// bias tile [AQ q][AK keys] -> fp32 LDS tile with pitch AK + 2 floats... kept in T with 8-byte pad
template <typename T> struct BiasPitch { static constexpr int v = AK * (int)sizeof(T) + 8; };
template <typename T>
__device__ __forceinline__ void stage_bias(char* tile, const T* bias_bh, int Tn, int q0, int k0, int tid) {
  // element-granular (rows of the [T, T] map are not 16-byte aligned in general); 64 consecutive keys per row
  for (int v = tid; v < AQ * AK; v += 256) {
    int r = v / AK, c = v % AK;
    T val = (T)0;
    if (q0 + r < Tn && k0 + c < Tn) val = bias_bh[(int64_t)(q0 + r) * Tn + k0 + c];
    *reinterpret_cast<T*>(tile + r * BiasPitch<T>::v + c * (int)sizeof(T)) = val;
  }
}

struct AttnParams {
  const void* q; const void* k; const void* v; int64_t ld;
  int B, nh, T, d; float scale;
  const void* bias; const int32_t* mask;
  void* out; int64_t ld_out; float* lse;
  float drop_p; uint32_t seed;
  // backward only
  const void* d_out; void* dq; void* dk; void* dv; int64_t ld_d; float* g_bias; float* delta;
};

// ================================================================================================
// forward
// ================================================================================================
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16;       // k-steps over the head dim
  constexpr int DT = DP / 32;       // 32-row tiles of the head dim (output rows)
  char* sK = smem;                                   // [AK][DP]
  char* sVt = sK + AK * Pitch<T, DP>::v;             // [DP][AK]
  char* sB = sVt + DP * Pitch<T, AK>::v;             // [AQ][AK] bias
  int* sValid = reinterpret_cast<int*>(sB + AQ * BiasPitch<T>::v);  // [AK] key validity
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d;
  const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * Tn : nullptr;
  const int32_t* mask = p.mask ? p.mask + (int64_t)b * Tn : nullptr;
  const int myq = q0 + wave * 32 + (lane & 31);
  const uint32_t thresh = (uint32_t)fminf(p.drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  // Q fragments straight from global: lane (q, half) holds d-elements 16*ks + 8*half .. +8
  Frag<T> qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = 16 * ks + 8 * half + e;
      f[e] = (myq < Tn && c < d) ? Elem<T>::load(Q + (int64_t)myq * p.ld + c) : 0.f;
    }
    qf[ks] = pack_acc<T>(f);
  }

  f32x16_t o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  for (int k0 = 0; k0 < Tn; k0 += AK) {
    __syncthreads();  // previous tile fully consumed
    stage_rowmajor<T, AK, DP>(sK, K, p.ld, k0, Tn, d, tid);
    stage_transposed<T, AK, DP>(sVt, V, p.ld, k0, Tn, d, tid);
    if (bias) stage_bias<T>(sB, bias, Tn, q0, k0, tid);
    if (tid < AK) sValid[tid] = (k0 + tid < Tn && (!mask || mask[k0 + tid] != 0)) ? 1 : 0;
    __syncthreads();

    // S^T[key, q] for the 64 keys of this tile
    f32x16_t s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> kf = FragReader<T, DP>::straight(sK, kt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(kf, qf[ks], s[kt]);
      }
    }
    // scale + bias + mask, running max
    float mt = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = kt * 32 + acc_row(r, lane);  // key within tile
        const int key = k0 + kl;
        float v = s[kt][r] * p.scale;
        if (bias) v += Elem<T>::load(reinterpret_cast<const T*>(sB + (wave * 32 + (lane & 31)) * BiasPitch<T>::v) + kl);
        (void)key;
        v = sValid[kl] ? v : NEG_BIG;
        s[kt][r] = v;
        mt = fmaxf(mt, v);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);   // m_run = NEG_BIG first time: exp(-inf-ish) = 0 (l_run, o are 0 anyway)
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = s[kt][r];
        float e = (v <= NEG_BIG * 0.5f) ? 0.f : __expf(v - m_new);
        ls += e;
        if (p.drop_p > 0.f) {
          const int key = k0 + kt * 32 + acc_row(r, lane);
          uint64_t idx = (((uint64_t)b * p.nh + h) * Tn + (uint64_t)myq) * Tn + key;
          e = dropout_keep(p.seed, idx, thresh) ? e * keep_scale : 0.f;
        }
        s[kt][r] = e;
      }
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // O^T[d, q] += V^T[d, key] . P^T[key, q]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float pv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) pv[t] = s[kk >> 1][8 * (kk & 1) + t];
      Frag<T> pf = pack_acc<T>(pv);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        Frag<T> vf = FragReader<T, AK>::perm(sVt, t * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(vf, pf, o[t]);
      }
    }
  }

  // normalise and write O[q, d] through LDS (transpose to row-major rows of 16-byte vectors)
  __syncthreads();
  float* sO = reinterpret_cast<float*>(smem);  // per wave [32 q][DP + 1] fp32
  float* myO = sO + wave * 32 * (DP + 1);
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) myO[(lane & 31) * (DP + 1) + t * 32 + acc_row(r, lane)] = o[t][r] * inv;
  if (half == 0 && myq < Tn && p.lse) p.lse[((int64_t)b * p.nh + h) * Tn + myq] = (l_run > 0.f) ? m_run + __logf(l_run) : NEG_BIG;
  __syncthreads();
  T* O = reinterpret_cast<T*>(p.out) + (int64_t)b * Tn * p.ld_out + h * d;
  for (int v = lane; v < 32 * d; v += 64) {
    int r = v / d, c = v % d;
    int qq = q0 + wave * 32 + r;
    if (qq < Tn) Elem<T>::store(O + (int64_t)qq * p.ld_out + c, myO[r * (DP + 1) + c]);
  }
}

// ================================================================================================
// backward, part 0: delta[b, h, q] = sum_d O[q, d] * dO[q, d]
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b * nh + h) * T + q
  const int64_t total = (int64_t)p.B * p.nh * p.T;
  if (row >= total) return;
  const int q = (int)(row % p.T);
  const int h = (int)((row / p.T) % p.nh);
  const int64_t b = row / ((int64_t)p.T * p.nh);
  const T* O = reinterpret_cast<const T*>(p.out) + (b * p.T + q) * p.ld_out + h * p.d;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (b * p.T + q) * p.ld_out + h * p.d;
  float s = 0.f;
  for (int c = lane; c < p.d; c += 64) s += Elem<T>::load(O + c) * Elem<T>::load(dO + c);
  s = wave_sum(s);
  if (lane == 0) p.delta[row] = s;
}

// ================================================================================================
// backward, part 1: dQ (and the accumulated bias gradient).  Workgroup = 128 queries, streams key tiles.
// ================================================================================================
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16, DT = DP / 32;
  char* sK = smem;                                   // [AK][DP]
  char* sV = sK + AK * Pitch<T, DP>::v;              // [AK][DP]
  char* sKt = sV + AK * Pitch<T, DP>::v;             // [DP][AK]
  char* sB = sKt + DP * Pitch<T, AK>::v;             // [AQ][AK] bias (T)
  float* sG = reinterpret_cast<float*>(sB + AQ * BiasPitch<T>::v);  // [AQ][AK + 1] fp32 dS staging
  int* sValid = reinterpret_cast<int*>(sG + AQ * (AK + 1));         // [AK] key validity
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (int64_t)b * Tn * p.ld_out + h * d;
  const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * Tn : nullptr;
  float* G = p.g_bias ? p.g_bias + ((int64_t)b * p.nh + h) * Tn * Tn : nullptr;
  const int32_t* mask = p.mask ? p.mask + (int64_t)b * Tn : nullptr;
  const int myq = q0 + wave * 32 + (lane & 31);
  const uint32_t thresh = (uint32_t)fminf(p.drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  Frag<T> qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float f[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = 16 * ks + 8 * half + e;
      bool ok = myq < Tn && c < d;
      f[e] = ok ? Elem<T>::load(Q + (int64_t)myq * p.ld + c) : 0.f;
      g[e] = ok ? Elem<T>::load(dO + (int64_t)myq * p.ld_out + c) : 0.f;
    }
    qf[ks] = pack_acc<T>(f);
    dof[ks] = pack_acc<T>(g);
  }
  const int64_t rowid = ((int64_t)b * p.nh + h) * Tn + myq;
  const float my_lse = myq < Tn ? p.lse[rowid] : 0.f;
  const float my_delta = myq < Tn ? p.delta[rowid] : 0.f;

  f32x16_t dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  for (int k0 = 0; k0 < Tn; k0 += AK) {
    __syncthreads();
    stage_rowmajor<T, AK, DP>(sK, K, p.ld, k0, Tn, d, tid);
    stage_rowmajor<T, AK, DP>(sV, V, p.ld, k0, Tn, d, tid);
    stage_transposed<T, AK, DP>(sKt, K, p.ld, k0, Tn, d, tid);
    if (bias) stage_bias<T>(sB, bias, Tn, q0, k0, tid);
    if (tid < AK) sValid[tid] = (k0 + tid < Tn && (!mask || mask[k0 + tid] != 0)) ? 1 : 0;
    __syncthreads();

    f32x16_t s[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> kf = FragReader<T, DP>::straight(sK, kt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(kf, qf[ks], s[kt]);
        Frag<T> vf = FragReader<T, DP>::straight(sV, kt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(vf, dof[ks], dp[kt]);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = kt * 32 + acc_row(r, lane);
        const int key = k0 + kl;
        float v = s[kt][r] * p.scale;
        if (bias) v += Elem<T>::load(reinterpret_cast<const T*>(sB + (wave * 32 + (lane & 31)) * BiasPitch<T>::v) + kl);
        const bool valid = sValid[kl] != 0 && myq < Tn;
        float pr = valid ? __expf(v - my_lse) : 0.f;
        float dpv = dp[kt][r];
        if (p.drop_p > 0.f) {
          uint64_t idx = (((uint64_t)b * p.nh + h) * Tn + (uint64_t)myq) * Tn + key;
          dpv = dropout_keep(p.seed, idx, thresh) ? dpv * keep_scale : 0.f;
        }
        float ds = pr * (dpv - my_delta);
        s[kt][r] = ds;
        if (G) sG[(wave * 32 + (lane & 31)) * (AK + 1) + kl] = ds;
      }
    // dQ^T[d, q] += K^T[d, key] . dS^T[key, q]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float pv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) pv[t] = s[kk >> 1][8 * (kk & 1) + t];
      Frag<T> pf = pack_acc<T>(pv);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        Frag<T> kf = FragReader<T, AK>::perm(sKt, t * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(kf, pf, dq[t]);
      }
    }
    if (G) {  // coalesced read-modify-write of the bias-gradient tile (this workgroup owns it)
      __syncthreads();
      for (int v = tid; v < AQ * AK; v += 256) {
        int r = v / AK, c = v % AK;
        if (q0 + r < Tn && k0 + c < Tn) G[(int64_t)(q0 + r) * Tn + k0 + c] += sG[r * (AK + 1) + c];
      }
    }
  }

  __syncthreads();
  float* sO = reinterpret_cast<float*>(smem);
  float* myO = sO + wave * 32 * (DP + 1);
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) myO[(lane & 31) * (DP + 1) + t * 32 + acc_row(r, lane)] = dq[t][r] * p.scale;
  __syncthreads();
  T* DQ = reinterpret_cast<T*>(p.dq) + (int64_t)b * Tn * p.ld_d + h * d;
  for (int v = lane; v < 32 * d; v += 64) {
    int r = v / d, c = v % d;
    int qq = q0 + wave * 32 + r;
    if (qq < Tn) Elem<T>::store(DQ + (int64_t)qq * p.ld_d + c, myO[r * (DP + 1) + c]);
  }
}

// ================================================================================================
// backward, part 2: dK, dV.  Workgroup = 128 keys (lane = key), streams query tiles of 64.
// ================================================================================================
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16, DT = DP / 32;
  char* sQ = smem;                                    // [AK q][DP]
  char* sdO = sQ + AK * Pitch<T, DP>::v;              // [AK q][DP]
  char* sQt = sdO + AK * Pitch<T, DP>::v;             // [DP][AK q]
  char* sdOt = sQt + DP * Pitch<T, AK>::v;            // [DP][AK q]
  float* sLse = reinterpret_cast<float*>(sdOt + DP * Pitch<T, AK>::v);  // [AK]
  float* sDelta = sLse + AK;                          // [AK]
  char* sB = reinterpret_cast<char*>(sDelta + AK);    // bias^T staging: [AK q][AQ keys] T, pitch AQ*sizeof(T)+8
  constexpr int BP = AQ * (int)sizeof(T) + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (int64_t)b * Tn * p.ld_out + h * d;
  const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * Tn : nullptr;
  const int32_t* mask = p.mask ? p.mask + (int64_t)b * Tn : nullptr;
  const int mykey = key0 + wave * 32 + (lane & 31);
  const bool key_valid = mykey < Tn && (!mask || mask[mykey] != 0);
  const uint32_t thresh = (uint32_t)fminf(p.drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  Frag<T> kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float f[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = 16 * ks + 8 * half + e;
      bool ok = mykey < Tn && c < d;
      f[e] = ok ? Elem<T>::load(K + (int64_t)mykey * p.ld + c) : 0.f;
      g[e] = ok ? Elem<T>::load(V + (int64_t)mykey * p.ld + c) : 0.f;
    }
    kf[ks] = pack_acc<T>(f);
    vf[ks] = pack_acc<T>(g);
  }
  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  for (int q0 = 0; q0 < Tn; q0 += AK) {
    __syncthreads();
    stage_rowmajor<T, AK, DP>(sQ, Q, p.ld, q0, Tn, d, tid);
    stage_rowmajor<T, AK, DP>(sdO, dO, p.ld_out, q0, Tn, d, tid);
    stage_transposed<T, AK, DP>(sQt, Q, p.ld, q0, Tn, d, tid);
    stage_transposed<T, AK, DP>(sdOt, dO, p.ld_out, q0, Tn, d, tid);
    if (tid < AK) {
      int qq = q0 + tid;
      int64_t rowid = ((int64_t)b * p.nh + h) * Tn + qq;
      sLse[tid] = qq < Tn ? p.lse[rowid] : 0.f;
      sDelta[tid] = qq < Tn ? p.delta[rowid] : 0.f;
    }
    if (bias) {
      for (int v = tid; v < AK * AQ; v += 256) {
        int r = v / AQ, c = v % AQ;  // r: query within tile, c: key within workgroup (contiguous in memory)
        T val = (T)0;
        if (q0 + r < Tn && key0 + c < Tn) val = bias[(int64_t)(q0 + r) * Tn + key0 + c];
        *reinterpret_cast<T*>(sB + r * BP + c * (int)sizeof(T)) = val;
      }
    }
    __syncthreads();

    // S[q, key] and dP[q, key]: A = Q / dO tiles (rows = q), B = K / V fragments (lane = key)
    f32x16_t s[2], dp[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[qt][r] = 0.f; dp[qt][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> a = FragReader<T, DP>::straight(sQ, qt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(a, kf[ks], s[qt]);
        Frag<T> a2 = FragReader<T, DP>::straight(sdO, qt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(a2, vf[ks], dp[qt]);
      }
    }
    f32x16_t pr[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qt * 32 + acc_row(r, lane);
        const int qq = q0 + ql;
        float v = s[qt][r] * p.scale;
        if (bias) v += Elem<T>::load(reinterpret_cast<const T*>(sB + ql * BP) + wave * 32 + (lane & 31));
        const bool valid = key_valid && qq < Tn;
        float pv = valid ? __expf(v - sLse[ql]) : 0.f;
        float dpv = dp[qt][r];
        float pdrop = pv;
        if (p.drop_p > 0.f) {
          uint64_t idx = (((uint64_t)b * p.nh + h) * Tn + (uint64_t)qq) * Tn + mykey;
          bool keep = dropout_keep(p.seed, idx, thresh);
          dpv = keep ? dpv * keep_scale : 0.f;
          pdrop = keep ? pv * keep_scale : 0.f;
        }
        pr[qt][r] = pdrop;                      // what multiplied V in the forward
        s[qt][r] = pv * (dpv - sDelta[ql]);     // dS
      }
    // dV^T[d, key] += dO^T[d, q] . P[q, key] ;  dK^T[d, key] += Q^T[d, q] . dS[q, key]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[8], c[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { a[t] = pr[kk >> 1][8 * (kk & 1) + t]; c[t] = s[kk >> 1][8 * (kk & 1) + t]; }
      Frag<T> pf = pack_acc<T>(a), dsf = pack_acc<T>(c);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        Frag<T> dot = FragReader<T, AK>::perm(sdOt, t * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(dot, pf, dv[t]);
        Frag<T> qt_ = FragReader<T, AK>::perm(sQt, t * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(qt_, dsf, dk[t]);
      }
    }
  }

  __syncthreads();
  float* sO = reinterpret_cast<float*>(smem);
  float* myO = sO + wave * 32 * (DP + 1);
  T* DK = reinterpret_cast<T*>(p.dk) + (int64_t)b * Tn * p.ld_d + h * d;
  T* DV = reinterpret_cast<T*>(p.dv) + (int64_t)b * Tn * p.ld_d + h * d;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        myO[(lane & 31) * (DP + 1) + t * 32 + acc_row(r, lane)] = which == 0 ? dk[t][r] * p.scale : dv[t][r];
    __syncthreads();
    T* dst = which == 0 ? DK : DV;
    for (int v = lane; v < 32 * d; v += 64) {
      int r = v / d, c = v % d;
      int kk = key0 + wave * 32 + r;
      if (kk < Tn) Elem<T>::store(dst + (int64_t)kk * p.ld_d + c, myO[r * (DP + 1) + c]);
    }
    __syncthreads();
  }
}

template <typename T, int DP> static size_t fwd_smem() {
  size_t a = (size_t)AK * Pitch<T, DP>::v + (size_t)DP * Pitch<T, AK>::v + (size_t)AQ * BiasPitch<T>::v + AK * sizeof(int);
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}
template <typename T, int DP> static size_t dq_smem() {
  size_t a = (size_t)2 * AK * Pitch<T, DP>::v + (size_t)DP * Pitch<T, AK>::v + (size_t)AQ * BiasPitch<T>::v +
             (size_t)AQ * (AK + 1) * sizeof(float) + AK * sizeof(int);
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}
template <typename T, int DP> static size_t dkv_smem() {
  size_t a = (size_t)2 * AK * Pitch<T, DP>::v + (size_t)2 * DP * Pitch<T, AK>::v + 2 * AK * sizeof(float) +
             (size_t)AK * (AQ * sizeof(T) + 8);
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}

template <typename KernelT>
static int set_smem(KernelT kern, size_t bytes) {
  if (bytes > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      set_error("attention: cannot raise dynamic LDS to %zu bytes", bytes);
      return PENEO_ERR_LAUNCH;
    }
  }
  return PENEO_OK;
}

template <typename T, int DP>
static int launch_fwd(const AttnParams& p, hipStream_t st) {
  size_t sh = fwd_smem<T, DP>();
  int rc = set_smem(attn_fwd_kernel<T, DP>, sh);
  if (rc) return rc;
  dim3 grid((p.T + AQ - 1) / AQ, p.nh, p.B);
  hipLaunchKernelGGL((attn_fwd_kernel<T, DP>), grid, dim3(256), sh, st, p);
  return check_launch("peneo_attn_fwd");
}
template <typename T, int DP>
static int launch_bwd(const AttnParams& p, hipStream_t st) {
  int64_t rows = (int64_t)p.B * p.nh * p.T;
  hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
  int rc = check_launch("peneo_attn_bwd(delta)");
  if (rc) return rc;
  dim3 grid((p.T + AQ - 1) / AQ, p.nh, p.B);
  size_t s1 = dq_smem<T, DP>();
  rc = set_smem(attn_bwd_dq_kernel<T, DP>, s1);
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DP>), grid, dim3(256), s1, st, p);
  rc = check_launch("peneo_attn_bwd(dq)");
  if (rc) return rc;
  size_t s2 = dkv_smem<T, DP>();
  rc = set_smem(attn_bwd_dkv_kernel<T, DP>, s2);
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DP>), grid, dim3(256), s2, st, p);
  return check_launch("peneo_attn_bwd(dkv)");
}

template <typename T>
static int dispatch(const AttnParams& p, bool bwd, hipStream_t st) {
  const int dp = (p.d + 31) / 32 * 32;
  switch (dp) {
    case 32: return bwd ? launch_bwd<T, 32>(p, st) : launch_fwd<T, 32>(p, st);
    case 64: return bwd ? launch_bwd<T, 64>(p, st) : launch_fwd<T, 64>(p, st);
    case 96: return bwd ? launch_bwd<T, 96>(p, st) : launch_fwd<T, 96>(p, st);
    case 128: return bwd ? launch_bwd<T, 128>(p, st) : launch_fwd<T, 128>(p, st);
    default: set_error("attention: head dim %d not supported (<= 128)", p.d); return PENEO_ERR_INVALID;
  }
}

}  // namespace peneo
using namespace peneo;

extern "C" int peneo_attn_fwd(int dtype, const void* q, const void* k, const void* v, int64_t ld_qkv, int B, int nh, int T,
                              int d, float scale, const void* bias, const int32_t* key_mask, void* out, int64_t ld_out,
                              float* lse, float drop_p, uint32_t drop_seed, peneo_stream_t stream) {
  PENEO_REQUIRE(dtype == PENEO_F32 || dtype == PENEO_BF16, "peneo_attn_fwd: bad dtype");
  PENEO_REQUIRE(q && k && v && out && B > 0 && nh > 0 && T > 0 && d > 0, "peneo_attn_fwd: bad arguments");
  PENEO_REQUIRE(ld_qkv >= (int64_t)nh * d && ld_out >= (int64_t)nh * d, "peneo_attn_fwd: leading dims too small");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "peneo_attn_fwd: drop_p out of range");
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.B = B; p.nh = nh; p.T = T; p.d = d; p.scale = scale; p.bias = bias;
  p.mask = key_mask; p.out = out; p.ld_out = ld_out; p.lse = lse; p.drop_p = drop_p; p.seed = drop_seed;
  return dtype == PENEO_BF16 ? dispatch<bf16_t>(p, false, (hipStream_t)stream) : dispatch<float>(p, false, (hipStream_t)stream);
}

extern "C" int peneo_attn_bwd(int dtype, const void* q, const void* k, const void* v, int64_t ld_qkv, const void* out,
                              const void* d_out, int64_t ld_out, const float* lse, int B, int nh, int T, int d, float scale,
                              const void* bias, const int32_t* key_mask, void* dq, void* dk, void* dv, int64_t ld_dqkv,
                              float* g_bias, float* delta, float drop_p, uint32_t drop_seed, peneo_stream_t stream) {
  PENEO_REQUIRE(dtype == PENEO_F32 || dtype == PENEO_BF16, "peneo_attn_bwd: bad dtype");
  PENEO_REQUIRE(q && k && v && out && d_out && lse && dq && dk && dv && delta, "peneo_attn_bwd: null pointer");
  PENEO_REQUIRE(B > 0 && nh > 0 && T > 0 && d > 0, "peneo_attn_bwd: bad sizes");
  PENEO_REQUIRE(ld_qkv >= (int64_t)nh * d && ld_out >= (int64_t)nh * d && ld_dqkv >= (int64_t)nh * d, "peneo_attn_bwd: leading dims too small");
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.B = B; p.nh = nh; p.T = T; p.d = d; p.scale = scale; p.bias = bias;
  p.mask = key_mask; p.out = const_cast<void*>(out); p.ld_out = ld_out; p.lse = const_cast<float*>(lse);
  p.drop_p = drop_p; p.seed = drop_seed; p.d_out = d_out; p.dq = dq; p.dk = dk; p.dv = dv; p.ld_d = ld_dqkv;
  p.g_bias = g_bias; p.delta = delta;
  return dtype == PENEO_BF16 ? dispatch<bf16_t>(p, true, (hipStream_t)stream) : dispatch<float>(p, true, (hipStream_t)stream);
}
