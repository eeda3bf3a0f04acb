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
// The A-side operands that need the reduction index contiguous (V^T, K^T, Q^T, dO^T) come from
// per-head transposed copies [B, nh, DP, Tp] made by peneo_head_transpose (one HBM-bound pass per
// layer), so every LDS tile is filled with plain 16-byte row copies, prefetched into registers one
// tile ahead of the MFMAs.  The additive bias [B, nh, T, Tp] (row stride Tp = T rounded up to 64,
// padding and masked keys = -1e30) is staged through LDS the same way.  Soft-max runs in the exp2
// domain in fp32.  scores = scale * q.k + bias (reference: modeling_layoutlmv3.py:365-389).
#include <cstdlib>
#include "common.h"
#include "attention.h"

namespace peneo {

constexpr int AQ = 128;   // rows of the "n" side per workgroup (4 waves x 32)
constexpr int AK = 64;    // rows of the streamed side per tile
constexpr float MASKED = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
// raw v_exp_f32: arguments here are <= 0 (or hugely negative for masked keys), results in [0, 1]; no denormal fix-up needed
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Attention-probability dropout.  The keep bits of a call are made ONCE by peneo_attn_drop_words (one bit per (query, key),
// Bernoulli(1 - p) from the counter-based chain of the classifier dropout, common.h) and every kernel here READS them:
//
//     words[((b * nh + h) * nqb + (q >> 5)) * Tk + attn_kslot(key)]   bit (q & 31)
//     nqb = 4 * ceil(T / 128) query blocks, Tk = 128 * ceil(T / 128) key slots (peneo_attn_drop_words_dims)
//
// One dword = the 32 queries of a block for one key.  attn_kslot orders the 32 keys of a key block the way the forward's
// accumulator registers see them (lane = query, register r of half-wave hf = key 8 (r / 4) + 4 hf + r % 4): slots 2r and
// 2r + 1 hold the keys of register r for hf = 0 / 1, so the 64-bit word at slot 2r IS the 64-lane select mask of register r.
// The forward keeps a tile's 64 words one per lane (one coalesced 256-byte load per wave and tile, fetched a tile ahead),
// moves register r's pair into SGPRs with two v_readlane and masks with one v_cndmask on that pair - no hash, no compare,
// three VALU instructions per element (the in-kernel hash it replaces was 8.5 of the kernel's 20 VALU slots per element).  The
// single-pass backward (lane = key, registers = queries) loads its key's dword per 32-query block and takes register r's
// bit with one v_bfe_i32 (0 / -1), used as an AND mask on P and on keep_scale.  1 / (1 - p) is applied once to the
// accumulators (O, dV), not per element.
// (attn_kslot: attention.h)
__device__ __forceinline__ bool attn_word_keep(const uint32_t* words_bh, int Tk, int q, int key) {
  return (words_bh[(int64_t)(q >> 5) * Tk + attn_kslot(key)] >> (q & 31)) & 1u;
}
// x where the lane's bit of the 64-bit mask is set, else 0 (mask in an SGPR pair).  s_nop 1: the pair comes from v_readlane,
// and gfx950 wants two wait states between a VALU write of an SGPR and a VALU read of it; the compiler pads that hazard for
// instructions it can see, not inside inline asm (the fp32 kernel's schedule put the two back to back: wrong masks)
__device__ __forceinline__ float mask_keep(float x, uint64_t m) {
  float r;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
  return r;
}
// register mask i of a forward tile from the wave's 64 keep words (lane L holds word L of the tile): two v_readlane into
// an SGPR pair.  (Scalar loads of the words straight into SGPRs were tried first: left to the compiler they sit right in
// front of the first use with a full s_waitcnt each - SMEM returns out of order - and the forward ran slower than with the
// hash; issued by hand at the top of the tile, the register allocator spilled the destination SGPRs between the load and
// the wait, i.e. before the data had arrived.)
__device__ __forceinline__ uint64_t lane_words_mask(uint32_t w, int word /* even, wave-uniform */) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)w, word), hi = (uint32_t)__builtin_amdgcn_readlane((int)w, word + 1);
  return (uint64_t)lo | ((uint64_t)hi << 32);
#else
  return 0;
#endif
}
__device__ __forceinline__ float and_mask(float x, int m) { return __uint_as_float(__float_as_uint(x) & (uint32_t)m); }
// the running maximum of the online softmax only moves when a tile beats it by more than this (natural units): exponentials
// stay below e^4 and the rescaling of the accumulators - one multiply per element - is skipped for almost every tile
constexpr float RESCALE_TAU = 4.0f;

__global__ __launch_bounds__(256) void attn_drop_words_kernel(uint32_t* words, int64_t n, uint32_t thr16, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t st = mix32(seed ^ ((uint32_t)i * 0x9e3779b9u) ^ ((uint32_t)(i >> 32) * 0x85ebca6bu));
    uint32_t w = 0;
    const uint32_t inc = pair_drop_inc_of(st);   // (bits 24..31 of the seed word select the chain's increment: 2^32 distinct words' chains)
#pragma unroll
    for (int bit = 0; bit < 32; ++bit) {
      st = pair_drop_step(st, inc);
      w |= ((st >> 16) >= thr16 ? 1u : 0u) << bit;
    }
    words[i] = w;
  }
}

// ---- LDS tiles: row-major [rows][COLS] of T, pitch = COLS*sizeof(T) + 16 bytes (odd number of 16-byte
//      slots => 16 consecutive rows hit 16 different slots: conflict-free b128 column reads)
template <typename T, int COLS> struct Pitch { static constexpr int v = COLS * (int)sizeof(T) + 16; };

template <typename T, int COLS> struct FragReader;
template <int COLS> struct FragReader<bf16_t, COLS> {
  // 8 consecutive elements starting at column c (multiple of 8)
  __device__ static __forceinline__ Frag<bf16_t> straight(const char* tile, int row, int c) {
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const uint4*>(tile + row * Pitch<bf16_t, COLS>::v + c * 2);
    return f;
  }
  // elements {c..c+3} and {c+8..c+11} (c multiple of 4)
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

typedef short s16x4_t __attribute__((ext_vector_type(4)));
// fragment for an MFMA operand whose row index is the contiguous LDS dimension: rows (lane&31) + col0, reduction
// indices {k0..k0+3} and {k1..k1+3} (LDS rows) -> two transpose reads of a [4 k][16 rows] block per 16-lane group
template <int PITCH>
__device__ __forceinline__ Frag<bf16_t> frag_tr(const char* tile, int col0, int k0, int k1, int lane) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_s4p;
  const int c = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int kr = (lane & 15) >> 2;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(tile + (k0 + kr) * PITCH + c * 2));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(tile + (k1 + kr) * PITCH + c * 2));
  uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  Frag<bf16_t> f;
  f.v = make_uint4(l2.x, l2.y, h2.x, h2.y);
  return f;
}

// pack_frag8 of 8 accumulator registers (r0..r0+7) of a C-layout tile gives a fragment whose element t is row
// 2*r0 + (t&3) + 8*(t>>2) + 4*(lane>>5): exactly what FragReader::perm(col = 2*r0 + 4*half) reads on the A side.

// ---- register-staged tile copies -------------------------------------------------------------------
// A [ROWS][COLS] tile of 16-byte vectors split over 256 threads; NV vectors per thread.
template <typename T, int ROWS, int COLS> struct TileRegs {
  static constexpr int VEC = Elem<T>::kVec;
  static constexpr int VPR = COLS / VEC;                       // vectors per row
  static constexpr int NV = (ROWS * VPR + 255) / 256;
  uint4 v[NV];
};

// src rows r0.. (row stride ld elements), columns c0..c0+COLS; rows >= rmax or columns >= cmax read as zero.
// `fast`: everything in bounds and 16-byte aligned (block-uniform).
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void tile_load(TileRegs<T, ROWS, COLS>& t, const T* src, int64_t ld, int r0, int rmax, int c0,
                                          int cmax, int tid, bool fast) {
  using TR = TileRegs<T, ROWS, COLS>;
#pragma unroll
  for (int i = 0; i < TR::NV; ++i) {
    const int v = tid + 256 * i;
    const int r = v / TR::VPR, c = (v % TR::VPR) * TR::VEC;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (TR::NV * 256 == ROWS * TR::VPR || v < ROWS * TR::VPR) {
      if (fast) {
        val = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * ld + c0 + c);
      } else if (r0 + r < rmax && c0 + c < cmax) {
        const T* p = src + (int64_t)(r0 + r) * ld + c0 + c;
        if (c0 + c + TR::VEC <= cmax && (reinterpret_cast<uintptr_t>(p) & 15) == 0) val = *reinterpret_cast<const uint4*>(p);
        else {
          float f[TR::VEC];
#pragma unroll
          for (int e = 0; e < TR::VEC; ++e) f[e] = (c0 + c + e < cmax) ? Elem<T>::load(p + e) : 0.f;
          val = pack16<T>(f);
        }
      }
    }
    t.v[i] = val;
  }
}
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void tile_store(const TileRegs<T, ROWS, COLS>& t, char* tile, int tid) {
  using TR = TileRegs<T, ROWS, COLS>;
#pragma unroll
  for (int i = 0; i < TR::NV; ++i) {
    const int v = tid + 256 * i;
    const int r = v / TR::VPR, c = (v % TR::VPR) * TR::VEC;
    if (TR::NV * 256 == ROWS * TR::VPR || v < ROWS * TR::VPR)
      *reinterpret_cast<uint4*>(tile + r * Pitch<T, COLS>::v + c * (int)sizeof(T)) = t.v[i];
  }
}

// bias tile [AQ q][AK keys]: pitch chosen for conflict-light 4-key reads by 32 consecutive rows
template <typename T, int KW = AK> struct BiasPitch { static constexpr int v = KW * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16); };
template <typename T, int ROWS, int KW = AK> struct BiasRegs {
  static constexpr int VEC = Elem<T>::kVec;
  static constexpr int VPR = KW / VEC;
  static constexpr int NV = ROWS * VPR / 256;
  uint4 v[NV];
};
// rows are clamped to rmax-1 (their results are discarded), the key range is always inside the padded row
template <typename T, int ROWS, int KW = AK>
__device__ __forceinline__ void bias_load(BiasRegs<T, ROWS, KW>& t, const T* src, int64_t ld, int r0, int rmax, int c0, int tid) {
  using BR = BiasRegs<T, ROWS, KW>;
#pragma unroll
  for (int i = 0; i < BR::NV; ++i) {
    const int v = tid + 256 * i;
    const int r = min(r0 + v / BR::VPR, rmax - 1), c = (v % BR::VPR) * BR::VEC;
    t.v[i] = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + c0 + c);
  }
}
template <typename T, int ROWS, int KW = AK>
__device__ __forceinline__ void bias_store(const BiasRegs<T, ROWS, KW>& t, char* tile, int tid) {
  using BR = BiasRegs<T, ROWS, KW>;
#pragma unroll
  for (int i = 0; i < BR::NV; ++i) {
    const int v = tid + 256 * i;
    const int r = v / BR::VPR, c = (v % BR::VPR) * BR::VEC;
    char* d = tile + r * BiasPitch<T, KW>::v + c * (int)sizeof(T);
    if (sizeof(T) == 2) {
      *reinterpret_cast<uint2*>(d) = make_uint2(t.v[i].x, t.v[i].y);
      *reinterpret_cast<uint2*>(d + 8) = make_uint2(t.v[i].z, t.v[i].w);
    } else {
      *reinterpret_cast<uint4*>(d) = t.v[i];
    }
  }
}
// 4 consecutive bias values (fp32) at (row, col) of the staged tile, col multiple of 4
template <typename T, int KW = AK>
__device__ __forceinline__ void bias_read4(const char* tile, int row, int col, float* out) {
  const char* p = tile + row * BiasPitch<T, KW>::v + col * (int)sizeof(T);
  if (sizeof(T) == 2) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    out[0] = __uint_as_float(u.x << 16); out[1] = __uint_as_float(u.x & 0xffff0000u);
    out[2] = __uint_as_float(u.y << 16); out[3] = __uint_as_float(u.y & 0xffff0000u);
  } else {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    out[0] = __uint_as_float(u.x); out[1] = __uint_as_float(u.y); out[2] = __uint_as_float(u.z); out[3] = __uint_as_float(u.w);
  }
}


// 8 elements of row `row` starting at column c of a [rows, ld] matrix -> fragment (zero beyond rmax / cmax)
template <typename T>
__device__ __forceinline__ Frag<T> frag_from_global(const T* base, int64_t ld, int row, int rmax, int c, int cmax) {
  float f[8];
  const T* p = base + (int64_t)row * ld + c;
  if (row < rmax && c + 8 <= cmax && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    if (sizeof(T) == 2) unpack16<T>(*reinterpret_cast<const uint4*>(p), f);
    else { unpack16<T>(*reinterpret_cast<const uint4*>(p), f); unpack16<T>(*reinterpret_cast<const uint4*>(p + 4), f + 4); }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (row < rmax && c + e < cmax) ? Elem<T>::load(p + e) : 0.f;
  }
  return pack_frag8<T>(f);
}

// per-wave [32 rows][DP] fp32 tile in LDS -> rows of a [*, ld] matrix (16-byte stores when possible)
template <typename T, int DP>
__device__ __forceinline__ void store_rows(const float* myO, T* dst, int64_t ld, int row0, int rmax, int d, int lane) {
  constexpr int VEC = Elem<T>::kVec;
  const bool vec_ok = (d % VEC == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) && ((ld * (int64_t)sizeof(T)) % 16 == 0);
  if (vec_ok) {
    const int vpr = d / VEC;
    for (int v = lane; v < 32 * vpr; v += 64) {
      const int r = v / vpr, c = (v % vpr) * VEC;
      if (row0 + r < rmax) {
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = myO[r * (DP + 1) + c + e];
        *reinterpret_cast<uint4*>(dst + (int64_t)(row0 + r) * ld + c) = pack16<T>(f);
      }
    }
  } else {
    for (int v = lane; v < 32 * d; v += 64) {
      const int r = v / d, c = v % d;
      if (row0 + r < rmax) Elem<T>::store(dst + (int64_t)(row0 + r) * ld + c, myO[r * (DP + 1) + c]);
    }
  }
}

// ================================================================================================
// per-head transposed copy: dst[b, h, c, t] = src[(b*T + t) * ld + h*d + c]  (zero padded to DP x Tp)
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void head_transpose_kernel(const T* src, int64_t ld, int Tn, int d, int nh, T* dst, int DP, int Tp) {
  __shared__ float tile[32][33];
  const int bh = blockIdx.z, b = bh / nh, h = bh % nh;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (t < Tn && c < d) ? Elem<T>::load(src + ((int64_t)b * Tn + t) * ld + h * d + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (c < DP && t < Tp) Elem<T>::store(dst + (((int64_t)bh * DP + c) * Tp + t), tile[tx][i]);
  }
}

// ================================================================================================
// forward
// ================================================================================================
// VTR (bf16): V comes as the row-major [keys][d] tile and the V^T operand is read with the hardware transpose read, so
// no per-head transposed copy of V is made (peneo_attn_fwd then takes v instead of vt)
// FK = keys per tile, 64 or (VTR only) 32.  With 32 the staging registers of the next tile halve (K, V, bias: 32 -> 16) and
// the kernel fits 168 registers = three workgroups per CU: the 576 workgroups of 8 documents are resident at once instead of
// 512 + a second round of 64 (the 12 key-tile loop of a wave is the unit of time, so a second round costs half a first one).
template <typename T, int DP, bool DROP, bool VTR, int FK>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && DP <= 64) ? (FK == 32 ? 3 : 2) : 1) void attn_fwd_kernel(AttnParams p) {
  static_assert(FK == AK || (FK == 32 && VTR), "32-key tiles: row-major V only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16, DT = DP / 32;
  char* sK = smem;                                   // [FK][DP]
  char* sVt = sK + FK * Pitch<T, DP>::v;             // [DP][FK]
  constexpr int VT_BYTES = (DP * Pitch<T, FK>::v > FK * Pitch<T, DP>::v) ? DP * Pitch<T, FK>::v : FK * Pitch<T, DP>::v;
  char* sB = sVt + VT_BYTES;                         // [AQ][FK] bias (after V^T [DP][FK] or, VTR, V [FK][DP])
  float* sKb = reinterpret_cast<float*>(sB + AQ * BiasPitch<T, FK>::v);  // [FK] additive key bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d, Tp = p.Tp;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* Vt = VTR ? reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d
                    : reinterpret_cast<const T*>(p.vt) + ((int64_t)b * p.nh + h) * DP * Tp;
  const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * p.bias_ld : nullptr;
  const float* kb = p.key_bias ? p.key_bias + (int64_t)b * Tp : nullptr;
  const int myq = q0 + wave * 32 + (lane & 31);
  const int qrow = wave * 32 + (lane & 31);
  const float keep_scale = DROP ? p.keep_scale : 1.0f;
  const bool add_kb = kb != nullptr || bias == nullptr;   // with a bias tensor its padding columns already mask keys >= T
  // this wave's 32-query block of keep words: lane L reads word L of each 64-key tile
  const uint32_t* wq = nullptr;
  if (DROP) wq = p.words + (((int64_t)b * p.nh + h) * p.nqb + (blockIdx.x * 4 + wave)) * (int64_t)p.Tk + lane;
  uint32_t rw = 0u;
  const bool k_al = ((reinterpret_cast<uintptr_t>(K) & 15) == 0) && ((p.ld * (int64_t)sizeof(T)) % 16 == 0) && (d == DP);

  Frag<T> qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_from_global<T>(Q, p.ld, myq, Tn, 16 * ks + 8 * half, d);

  f32x16_t o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = MASKED, l_run = 0.f;

  TileRegs<T, FK, DP> rk;
  TileRegs<T, DP, FK> rv;
  TileRegs<T, FK, DP> rv2;
  BiasRegs<T, AQ, FK> rb;
  const bool v_al = VTR && ((reinterpret_cast<uintptr_t>(Vt) & 15) == 0) && ((p.ld * (int64_t)sizeof(T)) % 16 == 0) && (d == DP);
  const int ntile = (Tn + FK - 1) / FK;
#define FWD_PREFETCH(t_)                                                                 \
  {                                                                                      \
    const int k0_ = (t_) * FK;                                                           \
    tile_load<T, FK, DP>(rk, K, p.ld, k0_, Tn, 0, d, tid, k_al && k0_ + FK <= Tn);       \
    if constexpr (VTR) tile_load<T, FK, DP>(rv2, Vt, p.ld, k0_, Tn, 0, d, tid, v_al && k0_ + FK <= Tn); \
    else tile_load<T, DP, FK>(rv, Vt, Tp, 0, DP, k0_, Tp, tid, true);                    \
    if (bias) bias_load<T, AQ, FK>(rb, bias, p.bias_ld, q0, Tn, k0_, tid);               \
    if constexpr (DROP) rw = wq[k0_ & ~63];   /* lane L: word L of the 64-key group */    \
  }
  FWD_PREFETCH(0)
  for (int t = 0; t < ntile; ++t) {
    const int k0 = t * FK;
    __syncthreads();  // previous tile fully consumed
    tile_store<T, FK, DP>(rk, sK, tid);
    if constexpr (VTR) tile_store<T, FK, DP>(rv2, sVt, tid); else tile_store<T, DP, FK>(rv, sVt, tid);
    if (bias) bias_store<T, AQ, FK>(rb, sB, tid);
    if (tid < FK) sKb[tid] = (k0 + tid < Tn) ? (kb ? kb[k0 + tid] : 0.f) : MASKED;
    __syncthreads();
    const uint32_t cw = rw;                   // this tile's keep words (lane L: word L of its 64-key group)
    const int wsel = (FK == 32) ? 32 * (t & 1) : 0;   // 32-key tiles: the odd tile uses the group's second half
    FWD_PREFETCH(t + 1 < ntile ? t + 1 : t)   // unconditional: keeps the staging registers out of scratch

    // The tile's two 32-key blocks go through the online softmax one after the other (one 16-register score tile live at
    // a time - the running maximum moves lazily, so the second pass costs a compare and a ballot).  A block that lies
    // entirely past T (T = 709: the second half of the twelfth tile) is all masked and skipped.
    auto block = [&](auto kt_c) {
      constexpr int kt = decltype(kt_c)::value;
      f32x16_t s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {       // S^T[key, q]
        Frag<T> kf = FragReader<T, DP>::straight(sK, kt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(kf, qf[ks], s);
      }
      float mt = MASKED;                      // scores (natural units) and the block's row maximum
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kl = kt * 32 + 8 * g + 4 * half;   // 4 consecutive keys: regs 4g .. 4g+3
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) bias_read4<T, FK>(sB, qrow, kl, bb);
        if (add_kb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bb[e] += sKb[kl + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(s[4 * g + e], p.scale, bb[e]);
          s[4 * g + e] = v;
          mt = fmaxf(mt, v);
        }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = (mt > m_run + RESCALE_TAU) ? mt : m_run;
      if (__builtin_amdgcn_ballot_w64(m_new != m_run)) {   // rare after the first tiles (wave-uniform branch)
        const float alpha = fast_exp2((m_run - m_new) * LOG2E);
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t2][r] *= alpha;
      }
      const float nm = -m_run * LOG2E;
      float ls = 0.f;
      auto soft = [&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        float e = fast_exp2(fmaf(s[r], LOG2E, nm));
        ls += e;
        if constexpr (DROP) e = mask_keep(e, lane_words_mask(cw, 2 * (16 * kt + r) + wsel));
        s[r] = e;
      };
      soft(std::integral_constant<int, 0>{}); soft(std::integral_constant<int, 1>{});
      soft(std::integral_constant<int, 2>{}); soft(std::integral_constant<int, 3>{});
      soft(std::integral_constant<int, 4>{}); soft(std::integral_constant<int, 5>{});
      soft(std::integral_constant<int, 6>{}); soft(std::integral_constant<int, 7>{});
      soft(std::integral_constant<int, 8>{}); soft(std::integral_constant<int, 9>{});
      soft(std::integral_constant<int, 10>{}); soft(std::integral_constant<int, 11>{});
      soft(std::integral_constant<int, 12>{}); soft(std::integral_constant<int, 13>{});
      soft(std::integral_constant<int, 14>{}); soft(std::integral_constant<int, 15>{});
      ls += __shfl_xor(ls, 32, 64);
      l_run += ls;
      // O^T[d, q] += V^T[d, key] . P^T[key, q]
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int kk = 2 * kt + kh;
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = s[8 * kh + e];
        Frag<T> pf = pack_frag8<T>(pv);
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2) {
          Frag<T> vf;
          if constexpr (VTR) vf = frag_tr<Pitch<T, DP>::v>(sVt, t2 * 32, 16 * kk + 4 * half, 16 * kk + 4 * half + 8, lane);
          else vf = FragReader<T, FK>::perm(sVt, t2 * 32 + (lane & 31), 16 * kk + 4 * half);
          mma_step(vf, pf, o[t2]);
        }
      }
    };
    block(std::integral_constant<int, 0>{});
    if constexpr (FK == 64) {
      if (k0 + 32 < Tn) block(std::integral_constant<int, 1>{});
    }
  }
#undef FWD_PREFETCH

  // normalise and write O[q, d] through LDS (transpose to row-major rows)
  __syncthreads();
  float* myO = reinterpret_cast<float*>(smem) + wave * 32 * (DP + 1);
  const bool any = m_run > 0.5f * MASKED;
  const float inv = (any && l_run > 0.f) ? keep_scale / l_run : 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) myO[(lane & 31) * (DP + 1) + t * 32 + acc_row(r, lane)] = o[t][r] * inv;
  if (half == 0 && myq < Tn && p.lse) p.lse[((int64_t)b * p.nh + h) * Tn + myq] = any ? fmaf(m_run, LOG2E, log2f(l_run)) : MASKED;  // log2 units
  __syncthreads();
  T* O = reinterpret_cast<T*>(p.out) + (int64_t)b * Tn * p.ld_out + h * d;
  store_rows<T, DP>(myO, O, p.ld_out, q0 + wave * 32, Tn, d, lane);
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

// vector variant (head dim = 8 * 2^k elements of T-vectors): G = d / VEC lanes share one (token, head) row with 16-byte loads
// and meet in G-lane shuffles; a wave covers 64 / G heads of consecutive (token, head) pairs -> full 128-byte lines
template <typename T, int G>
__global__ __launch_bounds__(256) void attn_delta_vec_kernel(AttnParams p) {
  constexpr int VEC = Elem<T>::kVec;
  const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;       // (b * T + q) * nh + h
  const int sub = threadIdx.x % G;
  const int64_t total = (int64_t)p.B * p.T * p.nh;
  const bool ok = item < total;
  const int64_t it = ok ? item : total - 1;
  const int h = (int)(it % p.nh);
  const int64_t tok = it / p.nh;                                             // b * T + q
  const T* O = reinterpret_cast<const T*>(p.out) + tok * p.ld_out + h * p.d + sub * VEC;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + tok * p.ld_out + h * p.d + sub * VEC;
  float a[VEC], g[VEC];
  if constexpr (sizeof(T) == 2) {
    unpack16<T>(*reinterpret_cast<const uint4*>(O), a);
    unpack16<T>(*reinterpret_cast<const uint4*>(dO), g);
  } else {
    unpack16<T>(*reinterpret_cast<const uint4*>(O), a);
    unpack16<T>(*reinterpret_cast<const uint4*>(dO), g);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) s += a[e] * g[e];
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (ok && sub == 0) {
    const int64_t b = tok / p.T, q = tok % p.T;
    p.delta[(b * p.nh + h) * p.T + q] = s;
  }
}

// ================================================================================================
// backward, part 1: dQ (and the accumulated bias gradient).  Workgroup = 128 queries, streams key tiles.
// ================================================================================================
template <typename T, int DP, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16, DT = DP / 32;
  char* sK = smem;                                   // [AK][DP]
  char* sV = sK + AK * Pitch<T, DP>::v;              // [AK][DP]
  char* sKt = sV + AK * Pitch<T, DP>::v;             // [DP][AK]
  char* sB = sKt + DP * Pitch<T, AK>::v;             // [AQ][AK] bias (T)
  float* sKb = reinterpret_cast<float*>(sB + AQ * BiasPitch<T>::v);  // [AK]
  float* sG = sKb + AK;                              // [AQ][AK + 4] fp32 dS staging
  constexpr int GP = AK + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d, Tp = p.Tp;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d;
  const T* Kt = reinterpret_cast<const T*>(p.kt) + ((int64_t)b * p.nh + h) * DP * Tp;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (int64_t)b * Tn * p.ld_out + h * d;
  const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * p.bias_ld : nullptr;
  float* G = p.g_bias ? p.g_bias + ((int64_t)b * p.nh + h) * Tn * p.bias_ld : nullptr;
  const float* kb = p.key_bias ? p.key_bias + (int64_t)b * Tp : nullptr;
  const int myq = q0 + wave * 32 + (lane & 31);
  const int qrow = wave * 32 + (lane & 31);
  const float keep_scale = DROP ? p.keep_scale : 1.0f;
  const float sc2 = p.scale * LOG2E;
  const uint32_t* wbh = DROP ? p.words + ((int64_t)b * p.nh + h) * p.nqb * (int64_t)p.Tk : nullptr;
  const bool k_al = ((reinterpret_cast<uintptr_t>(K) & 15) == 0) && ((reinterpret_cast<uintptr_t>(V) & 15) == 0) &&
                    ((p.ld * (int64_t)sizeof(T)) % 16 == 0) && (d == DP);

  Frag<T> qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = frag_from_global<T>(Q, p.ld, myq, Tn, 16 * ks + 8 * half, d);
    dof[ks] = frag_from_global<T>(dO, p.ld_out, myq, Tn, 16 * ks + 8 * half, d);
  }
  const int64_t rowid = ((int64_t)b * p.nh + h) * Tn + myq;
  const float my_lse = myq < Tn ? p.lse[rowid] : 0.f;      // log2 units
  const float my_delta = myq < Tn ? p.delta[rowid] : 0.f;

  f32x16_t dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  TileRegs<T, AK, DP> rk, rv;
  TileRegs<T, DP, AK> rkt;
  BiasRegs<T, AQ> rb;
  const int ntile = (Tn + AK - 1) / AK;
#define DQ_PREFETCH(t_)                                                                  \
  {                                                                                      \
    const int k0_ = (t_) * AK;                                                           \
    const bool fast_ = k_al && k0_ + AK <= Tn;                                           \
    tile_load<T, AK, DP>(rk, K, p.ld, k0_, Tn, 0, d, tid, fast_);                        \
    tile_load<T, AK, DP>(rv, V, p.ld, k0_, Tn, 0, d, tid, fast_);                        \
    tile_load<T, DP, AK>(rkt, Kt, Tp, 0, DP, k0_, Tp, tid, true);                        \
    if (bias) bias_load<T, AQ>(rb, bias, p.bias_ld, q0, Tn, k0_, tid);                   \
  }
  DQ_PREFETCH(0)
  for (int t = 0; t < ntile; ++t) {
    const int k0 = t * AK;
    __syncthreads();
    tile_store<T, AK, DP>(rk, sK, tid);
    tile_store<T, AK, DP>(rv, sV, tid);
    tile_store<T, DP, AK>(rkt, sKt, tid);
    if (bias) bias_store<T, AQ>(rb, sB, tid);
    if (tid < AK) sKb[tid] = (k0 + tid < Tn) ? (kb ? kb[k0 + tid] * LOG2E : 0.f) : MASKED;
    __syncthreads();
    DQ_PREFETCH(t + 1 < ntile ? t + 1 : t)

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
      for (int g = 0; g < 4; ++g) {
        const int kl = kt * 32 + 8 * g + 4 * half;
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) bias_read4<T>(sB, qrow, kl, bb);
        float dsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(s[kt][4 * g + e], sc2, fmaf(bb[e], LOG2E, sKb[kl + e]));
          const float pr = (myq < Tn) ? fast_exp2(v - my_lse) : 0.f;
          float dpv = dp[kt][4 * g + e];
          if (DROP) dpv = attn_word_keep(wbh, p.Tk, myq, k0 + kl + e) ? dpv * keep_scale : 0.f;
          dsv[e] = pr * (dpv - my_delta);
          s[kt][4 * g + e] = dsv[e];
        }
        if (G) *reinterpret_cast<float4*>(sG + qrow * GP + kl) = make_float4(dsv[0], dsv[1], dsv[2], dsv[3]);
      }
    // dQ^T[d, q] += K^T[d, key] . dS^T[key, q]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = s[kk >> 1][8 * (kk & 1) + e];
      Frag<T> pf = pack_frag8<T>(pv);
#pragma unroll
      for (int t2 = 0; t2 < DT; ++t2) {
        Frag<T> kf = FragReader<T, AK>::perm(sKt, t2 * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(kf, pf, dq[t2]);
      }
    }
    if (G) {  // coalesced read-modify-write of the bias-gradient tile (this workgroup owns it); rows are 16-byte aligned
      __syncthreads();
      for (int v = tid; v < AQ * (AK / 4); v += 256) {
        const int r = v / (AK / 4), c = (v % (AK / 4)) * 4;
        if (q0 + r < Tn) {
          float4* gp = reinterpret_cast<float4*>(G + (int64_t)(q0 + r) * p.bias_ld + k0 + c);
          float4 a = *gp;
          const float4 sgv = *reinterpret_cast<const float4*>(sG + r * GP + c);
          a.x += sgv.x; a.y += sgv.y; a.z += sgv.z; a.w += sgv.w;
          *gp = a;
        }
      }
    }
  }
#undef DQ_PREFETCH

  __syncthreads();
  float* myO = reinterpret_cast<float*>(smem) + wave * 32 * (DP + 1);
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) myO[(lane & 31) * (DP + 1) + t * 32 + acc_row(r, lane)] = dq[t][r] * p.scale;
  __syncthreads();
  T* DQ = reinterpret_cast<T*>(p.dq) + (int64_t)b * Tn * p.ld_d + h * d;
  store_rows<T, DP>(myO, DQ, p.ld_d, q0 + wave * 32, Tn, d, lane);
}

// ================================================================================================
// backward, part 2: dK, dV.  Workgroup = 128 keys (lane = key), streams query tiles of 64.
// ================================================================================================
template <typename T, int DP, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16, DT = DP / 32;
  char* sQ = smem;                                    // [AK q][DP]
  char* sdO = sQ + AK * Pitch<T, DP>::v;              // [AK q][DP]
  char* sQt = sdO + AK * Pitch<T, DP>::v;             // [DP][AK q]
  char* sdOt = sQt + DP * Pitch<T, AK>::v;            // [DP][AK q]
  float* sLse = reinterpret_cast<float*>(sdOt + DP * Pitch<T, AK>::v);  // [AK]
  float* sDelta = sLse + AK;                          // [AK]
  char* sB = reinterpret_cast<char*>(sDelta + AK);    // bias: [AK q][AQ keys] T
  constexpr int BP = AQ * (int)sizeof(T) + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d, Tp = p.Tp;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (int64_t)b * Tn * p.ld_out + h * d;
  const T* Qt = reinterpret_cast<const T*>(p.qt) + ((int64_t)b * p.nh + h) * DP * Tp;
  const T* dOt = reinterpret_cast<const T*>(p.dot) + ((int64_t)b * p.nh + h) * DP * Tp;
  const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * p.bias_ld : nullptr;
  const int mykey = key0 + wave * 32 + (lane & 31);
  const float my_kb = (mykey < Tn) ? (p.key_bias ? p.key_bias[(int64_t)b * Tp + mykey] * LOG2E : 0.f) : MASKED;
  const float keep_scale = DROP ? p.keep_scale : 1.0f;
  const float sc2 = p.scale * LOG2E;
  const uint32_t* wbh = DROP ? p.words + ((int64_t)b * p.nh + h) * p.nqb * (int64_t)p.Tk : nullptr;
  const bool q_al = ((reinterpret_cast<uintptr_t>(Q) & 15) == 0) && ((p.ld * (int64_t)sizeof(T)) % 16 == 0) && (d == DP);
  const bool do_al = ((reinterpret_cast<uintptr_t>(dO) & 15) == 0) && ((p.ld_out * (int64_t)sizeof(T)) % 16 == 0) && (d == DP);

  Frag<T> kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = frag_from_global<T>(K, p.ld, mykey, Tn, 16 * ks + 8 * half, d);
    vf[ks] = frag_from_global<T>(V, p.ld, mykey, Tn, 16 * ks + 8 * half, d);
  }
  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  TileRegs<T, AK, DP> rq, rdo;
  TileRegs<T, DP, AK> rqt, rdot;
  constexpr int VEC = Elem<T>::kVec;
  constexpr int BNV = AK * (AQ / VEC) / 256;   // bias^T tile [AK q rows][AQ keys]
  uint4 rbias[BNV];
  const int ntile = (Tn + AK - 1) / AK;
#define DKV_PREFETCH(t_)                                                                 \
  {                                                                                      \
    const int q0_ = (t_) * AK;                                                           \
    tile_load<T, AK, DP>(rq, Q, p.ld, q0_, Tn, 0, d, tid, q_al && q0_ + AK <= Tn);       \
    tile_load<T, AK, DP>(rdo, dO, p.ld_out, q0_, Tn, 0, d, tid, do_al && q0_ + AK <= Tn); \
    tile_load<T, DP, AK>(rqt, Qt, Tp, 0, DP, q0_, Tp, tid, true);                        \
    tile_load<T, DP, AK>(rdot, dOt, Tp, 0, DP, q0_, Tp, tid, true);                      \
    _Pragma("unroll") for (int i_ = 0; i_ < BNV; ++i_) {                                 \
      const int v_ = tid + 256 * i_;                                                     \
      const int r_ = min(q0_ + v_ / (AQ / VEC), Tn - 1), c_ = (v_ % (AQ / VEC)) * VEC;   \
      const int cc_ = min(key0 + c_, (int)p.bias_ld - VEC);  /* window may pass the padded row end: masked anyway */ \
      /* unconditional load (a predicated one sends rbias to scratch): without bias read a harmless valid address */ \
      const T* bp_ = bias ? bias + (int64_t)r_ * p.bias_ld + cc_ : Qt;                   \
      rbias[i_] = *reinterpret_cast<const uint4*>(bp_);                                  \
    }                                                                                    \
  }
  DKV_PREFETCH(0)
  for (int t = 0; t < ntile; ++t) {
    const int q0 = t * AK;
    __syncthreads();
    tile_store<T, AK, DP>(rq, sQ, tid);
    tile_store<T, AK, DP>(rdo, sdO, tid);
    tile_store<T, DP, AK>(rqt, sQt, tid);
    tile_store<T, DP, AK>(rdot, sdOt, tid);
    if (tid < AK) {
      const int qq = q0 + tid;
      const int64_t rowid = ((int64_t)b * p.nh + h) * Tn + qq;
      sLse[tid] = qq < Tn ? p.lse[rowid] : 0.f;
      sDelta[tid] = qq < Tn ? p.delta[rowid] : 0.f;
    }
    if (bias) {
#pragma unroll
      for (int i = 0; i < BNV; ++i) {
        const int v = tid + 256 * i;
        const int r = v / (AQ / VEC), c = (v % (AQ / VEC)) * VEC;
        *reinterpret_cast<uint4*>(sB + r * BP + c * (int)sizeof(T)) = rbias[i];
      }
    }
    __syncthreads();
    DKV_PREFETCH(t + 1 < ntile ? t + 1 : t)

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
        float bv = 0.f;
        if (bias) bv = Elem<T>::load(reinterpret_cast<const T*>(sB + ql * BP) + wave * 32 + (lane & 31));
        const float v = fmaf(s[qt][r], sc2, fmaf(bv, LOG2E, my_kb));
        const float pv = (qq < Tn) ? fast_exp2(v - sLse[ql]) : 0.f;
        float dpv = dp[qt][r];
        float pdrop = pv;
        if (DROP) {
          const bool keep = attn_word_keep(wbh, p.Tk, qq, mykey);
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
      for (int e = 0; e < 8; ++e) { a[e] = pr[kk >> 1][8 * (kk & 1) + e]; c[e] = s[kk >> 1][8 * (kk & 1) + e]; }
      Frag<T> pf = pack_frag8<T>(a), dsf = pack_frag8<T>(c);
#pragma unroll
      for (int t2 = 0; t2 < DT; ++t2) {
        Frag<T> dot = FragReader<T, AK>::perm(sdOt, t2 * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(dot, pf, dv[t2]);
        Frag<T> qt_ = FragReader<T, AK>::perm(sQt, t2 * 32 + (lane & 31), 16 * kk + 4 * half);
        mma_step(qt_, dsf, dk[t2]);
      }
    }
  }
#undef DKV_PREFETCH

  __syncthreads();
  float* myO = reinterpret_cast<float*>(smem) + wave * 32 * (DP + 1);
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
    store_rows<T, DP>(myO, which == 0 ? DK : DV, p.ld_d, key0 + wave * 32, Tn, d, lane);
    __syncthreads();
  }
}

// ================================================================================================
// backward, bf16 single pass: one workgroup = 128 keys (lane = key) streaming query tiles of 64.  S and dP are computed
// once; dK / dV accumulate in registers; dS goes through LDS (stored key-major) into the dQ product of the tile, which
// is added to an fp32 dQ accumulator in HBM with coalesced no-return atomics (6 key tiles contribute per element at
// T = 709); the bias gradient is added to its fp32 accumulator the same way (this workgroup owns its columns).
// Every operand that needs the reduction index strided comes from the hardware transpose read ds_read_b64_tr_b16 on
// the plain row-major tiles, so no transposed copies of Q / K / dO are made.
// ================================================================================================
#ifdef ATTN_PROF
// tools/attn_cycles.py: per-phase s_memtime ticks of the single-pass backward, summed over all waves (a -DATTN_PROF build only)
__device__ unsigned long long g_attn_prof[16];
#define AP_DECL unsigned long long ap_t = __builtin_amdgcn_s_memtime(), ap_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long ap_t0 = ap_t;
#define AP_MARK(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ap_acc[i] += n_ - ap_t; ap_t = n_; }
#define AP_FLUSH { if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 10; ++i) atomicAdd(&g_attn_prof[i], ap_acc[i]); atomicAdd(&g_attn_prof[10], __builtin_amdgcn_s_memtime() - ap_t0); atomicAdd(&g_attn_prof[11], 1ull); } }
#else
#define AP_DECL
#define AP_MARK(i)
#define AP_FLUSH
#endif
constexpr int FQ = 64;     // queries per step
constexpr int FKEYS = 128; // keys per workgroup
// bias tile [FQ queries][FKEYS keys] of bf16: 16-byte vector i of thread tid (4 per thread).  Kept as four named
// registers on purpose: as an array the set ended up in scratch memory.
__device__ __forceinline__ uint4 fused_bias_load(const bf16_t* bias, int64_t ld, int q0, int Tn, int key0, int tid, int i) {
  const int v = tid + 256 * i;
  const int r = min(q0 + v / (FKEYS / 8), Tn - 1), c = (v % (FKEYS / 8)) * 8;
  if (key0 + c >= (int)ld) return make_uint4(0xF14AF14Au, 0xF14AF14Au, 0xF14AF14Au, 0xF14AF14Au);   // past the padded row end: bf16(-1e30)
  return *reinterpret_cast<const uint4*>(bias + (int64_t)r * ld + key0 + c);
}
template <int PB>
__device__ __forceinline__ void fused_bias_store(uint4 val, char* sB, int tid, int i) {
  const int v = tid + 256 * i;
  const int r = v / (FKEYS / 8), c = (v % (FKEYS / 8)) * 8;
  *reinterpret_cast<uint4*>(sB + r * PB + c * 2) = val;
}

template <int DP, bool DROP, bool HAS_BIAS, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_fused_kernel(AttnParams p, float* dq_acc) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DP / 16, DT = DP / 32;
  constexpr int PQ = Pitch<T, DP>::v;              // pitch of the [*][DP] tiles
  constexpr int PS = FQ * 2 + 16;                  // pitch of dS^T [keys][FQ]
  constexpr int PB = FKEYS * 2 + 16;               // pitch of the bias tile [FQ][keys]
  char* sQ = smem;                                  // [FQ][DP]
  char* sdO = sQ + FQ * PQ;                         // [FQ][DP]
  char* sK = sdO + FQ * PQ;                         // [FKEYS][DP]
  char* sS = sK + FKEYS * PQ;                       // [FKEYS][FQ]  dS^T
  char* sB = sS + FKEYS * PS;                       // [FQ][FKEYS]  bias
  float* sLse = reinterpret_cast<float*>(sB + FQ * PB);   // [FQ]
  float* sDelta = sLse + FQ;                        // [FQ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * FKEYS;
  const int Tn = p.T, d = p.d, Tp = p.Tp;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * d;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * d;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (int64_t)b * Tn * p.ld_out + h * d;
  const T* bias = HAS_BIAS ? reinterpret_cast<const T*>(p.bias) + ((int64_t)b * p.nh + h) * Tn * p.bias_ld : nullptr;
  float* G = (HAS_BIAS && p.g_bias) ? p.g_bias + ((int64_t)b * p.nh + h) * Tn * p.bias_ld : nullptr;
  float* DQ = dq_acc ? dq_acc + (int64_t)b * Tn * ((int64_t)p.nh * d) + h * d : nullptr;
  const int64_t ldq = (int64_t)p.nh * d;
  const int keyl = wave * 32 + (lane & 31);
  const int mykey = key0 + keyl;
  const float my_kb = (mykey < Tn) ? (p.key_bias ? p.key_bias[(int64_t)b * Tp + mykey] : 0.f) : MASKED;   // natural units
  const bool add_kb = HAS_BIAS && p.key_bias != nullptr;   // (with a bias tensor its padding columns mask keys >= T)
  const float keep_scale = DROP ? p.keep_scale : 1.0f;
  // this lane's key column of the keep words: one dword per 32-query block
  const uint32_t* wl = DROP ? p.words + ((int64_t)b * p.nh + h) * p.nqb * (int64_t)p.Tk + attn_kslot(mykey) : nullptr;
  const int trk = (lane & 15) >> 2, trc = wave * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);   // transpose-read address parts
  const bool q_al = ((reinterpret_cast<uintptr_t>(Q) & 15) == 0) && ((p.ld * 2) % 16 == 0) && (d == DP);
  const bool do_al = ((reinterpret_cast<uintptr_t>(dO) & 15) == 0) && ((p.ld_out * 2) % 16 == 0) && (d == DP);

  // this workgroup's K rows -> LDS once (B operand of the dQ product); K / V fragments (lane = key) stay in registers
  {
    TileRegs<T, FKEYS, DP> rk;
    tile_load<T, FKEYS, DP>(rk, K, p.ld, key0, Tn, 0, d, tid, q_al && key0 + FKEYS <= Tn);
    tile_store<T, FKEYS, DP>(rk, sK, tid);
  }
  Frag<T> kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = frag_from_global<T>(K, p.ld, mykey, Tn, 16 * ks + 8 * half, d);
    vf[ks] = frag_from_global<T>(V, p.ld, mykey, Tn, 16 * ks + 8 * half, d);
  }
  f32x16_t dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  TileRegs<T, FQ, DP> rq, rdo;
  static_assert(FQ * (FKEYS / 8) / 256 == 4, "bias tile = 4 vectors per thread");
  uint4 rb0 = make_uint4(0, 0, 0, 0), rb1 = rb0, rb2 = rb0, rb3 = rb0;
  uint32_t rw0 = 0u, rw1 = 0u;   // keep words of the tile's two query blocks
  const int ntile = (Tn + FQ - 1) / FQ;
#define FUSED_PREFETCH(t_)                                                               \
  {                                                                                      \
    const int q0_ = (t_) * FQ;                                                           \
    tile_load<T, FQ, DP>(rq, Q, p.ld, q0_, Tn, 0, d, tid, q_al && q0_ + FQ <= Tn);       \
    tile_load<T, FQ, DP>(rdo, dO, p.ld_out, q0_, Tn, 0, d, tid, do_al && q0_ + FQ <= Tn); \
    if constexpr (HAS_BIAS) {                                                            \
      rb0 = fused_bias_load(bias, p.bias_ld, q0_, Tn, key0, tid, 0);                     \
      rb1 = fused_bias_load(bias, p.bias_ld, q0_, Tn, key0, tid, 1);                     \
      rb2 = fused_bias_load(bias, p.bias_ld, q0_, Tn, key0, tid, 2);                     \
      rb3 = fused_bias_load(bias, p.bias_ld, q0_, Tn, key0, tid, 3);                     \
    }                                                                                    \
    if constexpr (DROP) {                                                                \
      rw0 = wl[(int64_t)(2 * (t_)) * p.Tk];                                              \
      rw1 = wl[(int64_t)(2 * (t_) + 1) * p.Tk];                                          \
    }                                                                                    \
  }
  FUSED_PREFETCH(0)
  AP_DECL
  for (int t = 0; t < ntile; ++t) {
    const int q0 = t * FQ;
    AP_MARK(9)
    __syncthreads();
    AP_MARK(0)
    tile_store<T, FQ, DP>(rq, sQ, tid);
    tile_store<T, FQ, DP>(rdo, sdO, tid);
    if (tid < FQ) {
      const int qq = q0 + tid;
      const int64_t rowid = ((int64_t)b * p.nh + h) * Tn + qq;
      sLse[tid] = qq < Tn ? -p.lse[rowid] : MASKED;   // minus lse (log2 units); rows past T give P = exp2(-huge) = 0
      sDelta[tid] = qq < Tn ? p.delta[rowid] : 0.f;
    }
    const uint32_t cw0 = rw0 >> (4 * half), cw1 = rw1 >> (4 * half);   // bit (8 (r / 4) + r % 4) = this lane's query of register r
    if constexpr (HAS_BIAS) {
      fused_bias_store<PB>(rb0, sB, tid, 0);
      fused_bias_store<PB>(rb1, sB, tid, 1);
      fused_bias_store<PB>(rb2, sB, tid, 2);
      fused_bias_store<PB>(rb3, sB, tid, 3);
    }
    AP_MARK(1)
    __syncthreads();
    AP_MARK(2)
    FUSED_PREFETCH(t + 1 < ntile ? t + 1 : t)
    AP_MARK(3)

    float* Gt = G ? G + (int64_t)q0 * p.bias_ld + key0 : nullptr;   // uniform tile base; lanes add 32-bit offsets
    // a query block that lies entirely past T (T = 709: the second half of the twelfth tile) contributes nothing: skipped
    // (its dS^T columns are zero-filled below)
    const int nqt = (q0 + 32 < Tn) ? 2 : 1;
#pragma unroll 1
    for (int qt = 0; qt < nqt; ++qt) {
      // S[q, key], dP[q, key]: A = Q / dO rows (q), B = K / V fragments (lane = key)
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag<T> a = FragReader<T, DP>::straight(sQ, qt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(a, kf[ks], s);
        Frag<T> a2 = FragReader<T, DP>::straight(sdO, qt * 32 + (lane & 31), 16 * ks + 8 * half);
        mma_step(a2, vf[ks], dp);
      }
      AP_MARK(4)
      f32x16_t pr;
      const uint32_t cw = qt ? cw1 : cw0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qb = qt * 32 + 8 * g + 4 * half;           // 4 consecutive query rows: r = 4g .. 4g+3
        const float4 l4 = *reinterpret_cast<const float4*>(sLse + qb);
        const float4 d4 = *reinterpret_cast<const float4*>(sDelta + qb);
        const float nl4[4] = {l4.x, l4.y, l4.z, l4.w}, del4[4] = {d4.x, d4.y, d4.z, d4.w};
        float bv4[4] = {my_kb, my_kb, my_kb, my_kb};
        if constexpr (HAS_BIAS) {   // the four queries' bias of this lane's key: one transpose read of the [q][key] tile
          typedef __attribute__((address_space(3))) s16x4_t* lds_s4p;
          const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(sB + (qb + trk) * PB + trc * 2)));
          bv4[0] = __uint_as_float(u.x << 16); bv4[1] = __uint_as_float(u.x & 0xffff0000u);
          bv4[2] = __uint_as_float(u.y << 16); bv4[3] = __uint_as_float(u.y & 0xffff0000u);
          if (add_kb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv4[e] += my_kb;
          }
        }
        float ds4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e, ql = qb + e, qq = q0 + ql;
          const float v = fmaf(s[r], p.scale, bv4[e]);
          const float pv = fast_exp2(fmaf(v, LOG2E, nl4[e]));
          float pdrop = pv, ks = keep_scale;
          if (DROP) {
            const int m = __builtin_amdgcn_sbfe((int)cw, 8 * g + e, 1);   // 0 / -1
            pdrop = and_mask(pv, m);
            ks = and_mask(keep_scale, m);
          }
          pr[r] = pdrop;                                    // (1 / (1 - p) goes onto dV once, at the end)
          const float dsv = pv * fmaf(dp[r], ks, -del4[e]);
          s[r] = dsv;
          ds4[e] = dsv;
          if (G && qq < Tn && mykey < Tn) atomicAdd(Gt + ((uint32_t)ql * (uint32_t)p.bias_ld + (uint32_t)keyl), dsv);
        }
        // dS^T[key][q]: 4 consecutive queries of this key
        *reinterpret_cast<uint2*>(sS + keyl * PS + qb * 2) = make_uint2(pack_bf16x2(ds4[0], ds4[1]), pack_bf16x2(ds4[2], ds4[3]));
        __builtin_amdgcn_sched_barrier(0);   // one group's loads and temporaries at a time (register pressure)
      }
      AP_MARK(5)
      // dV^T[d, key] += dO^T[d, q] . P[q, key] ;  dK^T[d, key] += Q^T[d, q] . dS[q, key]
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float a[8], c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = pr[8 * kk + e]; c[e] = s[8 * kk + e]; }
        Frag<T> pf = pack_frag8<T>(a), dsf = pack_frag8<T>(c);
        const int g0 = qt * 32 + 16 * kk + 4 * half;
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2) {
          Frag<T> dot = frag_tr<PQ>(sdO, t2 * 32, g0, g0 + 8, lane);
          mma_step(dot, pf, dv[t2]);
          Frag<T> qtf = frag_tr<PQ>(sQ, t2 * 32, g0, g0 + 8, lane);
          mma_step(qtf, dsf, dk[t2]);
        }
      }
      AP_MARK(6)
    }
    if (nqt == 1 && half == 0) {   // the skipped query block: its dS^T columns are zero (the padding columns of the slab stay 0)
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sS + keyl * PS + 64 + 16 * i) = z4;
    }
    AP_MARK(9)
    __syncthreads();   // dS^T tile complete
    AP_MARK(7)
    if (p.ds_out) {    // this layer's dS^T rows (key-major, 128 B per key and step): the bias-table gradient is reduced from
                       // the per-layer bf16 copies once per step (peneo_relpos_bias_bwd_layers) instead of a fp32 RMW per layer
      T* dsg = reinterpret_cast<T*>(p.ds_out) + (((int64_t)b * p.nh + h) * Tn + key0) * (int64_t)Tp + q0;
#pragma unroll
      for (int i = 0; i < FKEYS * (FQ / 8) / 256; ++i) {
        const int v = tid + 256 * i;
        const int kr = v >> 3, ch = v & 7;
        if (key0 + kr < Tn)
          *reinterpret_cast<uint4*>(dsg + (int64_t)kr * Tp + ch * 8) = *reinterpret_cast<const uint4*>(sS + kr * PS + ch * 16);
      }
    }
    AP_MARK(8)
    // dQ[q, dcol] += dS[q, keys] . K[keys, dcol]: 2 x DT output tiles of 32 x 32 over the 4 waves
    for (int tile = wave; tile < 2 * DT && dq_acc != nullptr; tile += 4) {   // (no accumulator: dQ comes from the stored dS^T)
      const int qt = tile / DT, dt = tile % DT;
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < FKEYS / 16; ++kk) {
        const int k0 = 16 * kk + 8 * half;
        Frag<T> af = frag_tr<PS>(sS, qt * 32, k0, k0 + 4, lane);     // rows = q, reduction = keys
        Frag<T> bf = frag_tr<PQ>(sK, dt * 32, k0, k0 + 4, lane);     // cols = dcol, reduction = keys
        mma_step(af, bf, acc);
      }
      const int dcol = dt * 32 + (lane & 31);
      float* DQt = DQ + (int64_t)(q0 + qt * 32) * ldq;   // uniform
      if (dcol < d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = acc_row(r, lane);
          if (q0 + qt * 32 + ql < Tn) atomicAdd(DQt + ((uint32_t)ql * (uint32_t)ldq + (uint32_t)dcol), acc[r] * p.scale);
        }
      }
    }
  }
#undef FUSED_PREFETCH
  AP_FLUSH

  __syncthreads();
  float* myO = reinterpret_cast<float*>(smem) + wave * 32 * (DP + 1);
  T* DK = reinterpret_cast<T*>(p.dk) + (int64_t)b * Tn * p.ld_d + h * d;
  T* DV = reinterpret_cast<T*>(p.dv) + (int64_t)b * Tn * p.ld_d + h * d;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        myO[(lane & 31) * (DP + 1) + t * 32 + acc_row(r, lane)] = which == 0 ? dk[t][r] * p.scale : dv[t][r] * keep_scale;
    __syncthreads();
    store_rows<T, DP>(myO, which == 0 ? DK : DV, p.ld_d, key0 + wave * 32, Tn, d, lane);
    __syncthreads();
  }
}

// dQ[q, :] = scale * sum_key dS[q, key] K[key, :] from the stored dS^T slab of this layer (key-major [B, nh, T, Tp]):
// workgroup = 128 queries of one (b, h), streams key tiles of 64; both operands are "reduction-index-major" tiles read
// with the hardware transpose read.  Replaces the 6-fold fp32 atomic accumulation of the single-pass kernel (a quarter of
// its run time) by one extra read of the slab and plain bf16 stores.
template <int DP>
__global__ __launch_bounds__(256, 2) void attn_dq_from_ds_kernel(AttnParams p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DT = DP / 32;
  constexpr int PA = AQ * 2 + 16;                 // dS^T tile [AK keys][AQ queries]
  constexpr int PK = Pitch<T, DP>::v;             // K tile [AK keys][DP]
  char* sA = smem;
  char* sK = sA + AK * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int Tn = p.T, d = p.d, Tp = p.Tp;
  const T* DS = reinterpret_cast<const T*>(p.ds_out) + ((int64_t)b * p.nh + h) * Tn * (int64_t)Tp;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * d;
  const bool k_al = ((reinterpret_cast<uintptr_t>(K) & 15) == 0) && ((p.ld * 2) % 16 == 0) && (d == DP);
  const bool a_full = q0 + AQ <= Tp;
  f32x16_t acc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // the slab is read once from HBM: tiles are fetched TWO steps ahead into two register sets (one step ahead left a single
  // 24 KB tile per workgroup in flight and the kernel at 1.6 TB/s: every step waited out the full memory latency)
  TileRegs<T, AK, AQ> ra0, ra1;
  TileRegs<T, AK, DP> rk0, rk1;
  const int ntile = (Tn + AK - 1) / AK;
  auto prefetch = [&](TileRegs<T, AK, AQ>& ra, TileRegs<T, AK, DP>& rk, int t_) {
    const int k0_ = min(t_, ntile - 1) * AK;
    tile_load<T, AK, AQ>(ra, DS, Tp, k0_, Tn, q0, Tp, tid, a_full && k0_ + AK <= Tn);
    tile_load<T, AK, DP>(rk, K, p.ld, k0_, Tn, 0, d, tid, k_al && k0_ + AK <= Tn);
  };
  auto step = [&](TileRegs<T, AK, AQ>& ra, TileRegs<T, AK, DP>& rk, int t) {
    __syncthreads();
    tile_store<T, AK, AQ>(ra, sA, tid);
    tile_store<T, AK, DP>(rk, sK, tid);
    __syncthreads();
    prefetch(ra, rk, t + 2);
#pragma unroll
    for (int kk = 0; kk < AK / 16; ++kk) {
      const int k0 = 16 * kk + 8 * half;
      Frag<T> af = frag_tr<PA>(sA, wave * 32, k0, k0 + 4, lane);          // rows = q, reduction = keys
#pragma unroll
      for (int t2 = 0; t2 < DT; ++t2) {
        Frag<T> bf = frag_tr<PK>(sK, t2 * 32, k0, k0 + 4, lane);          // cols = dcol, reduction = keys
        mma_step(af, bf, acc[t2]);
      }
    }
  };
  prefetch(ra0, rk0, 0);
  prefetch(ra1, rk1, 1);
  for (int t = 0; t < ntile; t += 2) {
    step(ra0, rk0, t);
    if (t + 1 < ntile) step(ra1, rk1, t + 1);
  }
  __syncthreads();
  // acc: rows = q (register index), cols = dcol (lane) -> per-wave [32 q][DP] fp32 tile -> bf16 rows of dq
  float* myO = reinterpret_cast<float*>(smem) + wave * 32 * (DP + 1);
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) myO[acc_row(r, lane) * (DP + 1) + t * 32 + (lane & 31)] = acc[t][r] * p.scale;
  __syncthreads();
  T* DQ = reinterpret_cast<T*>(p.dq) + (int64_t)b * Tn * p.ld_d + h * d;
  store_rows<T, DP>(myO, DQ, p.ld_d, q0 + wave * 32, Tn, d, lane);
}
template <int DP> static size_t dqs_smem() {
  size_t a = (size_t)AK * (AQ * 2 + 16) + (size_t)AK * Pitch<bf16_t, DP>::v;
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}

// fp32 dQ accumulator [rows][cols] -> bf16 dq rows (leading dim ld)
__global__ __launch_bounds__(256) void dq_finish_kernel(const float* acc, int64_t rows, int cols, bf16_t* dq, int64_t ld) {
  const int64_t total = rows * (cols / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (cols / 8);
    const int c = (int)(i % (cols / 8)) * 8;
    float f[8];
    const uint4* q = reinterpret_cast<const uint4*>(acc + r * cols + c);
    unpack16<float>(q[0], f); unpack16<float>(q[1], f + 4);
    *reinterpret_cast<uint4*>(dq + r * ld + c) = pack16<bf16_t>(f);
  }
}

template <int DP> static size_t fused_smem() {
  size_t a = (size_t)2 * FQ * Pitch<bf16_t, DP>::v + (size_t)FKEYS * Pitch<bf16_t, DP>::v + (size_t)FKEYS * (FQ * 2 + 16) +
             (size_t)FQ * (FKEYS * 2 + 16) + 2 * FQ * sizeof(float);
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}

template <typename T, int DP, int FK> static size_t fwd_smem() {
  size_t vt = (size_t)DP * Pitch<T, FK>::v, vr = (size_t)FK * Pitch<T, DP>::v;
  size_t a = (size_t)FK * Pitch<T, DP>::v + (vt > vr ? vt : vr) + (size_t)AQ * BiasPitch<T, FK>::v + FK * sizeof(float);
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}
template <typename T, int DP> static size_t dq_smem() {
  size_t a = (size_t)2 * AK * Pitch<T, DP>::v + (size_t)DP * Pitch<T, AK>::v + (size_t)AQ * BiasPitch<T>::v +
             AK * sizeof(float) + (size_t)AQ * (AK + 4) * sizeof(float);
  size_t o = (size_t)4 * 32 * (DP + 1) * sizeof(float);
  return a > o ? a : o;
}
template <typename T, int DP> static size_t dkv_smem() {
  size_t a = (size_t)2 * AK * Pitch<T, DP>::v + (size_t)2 * DP * Pitch<T, AK>::v + 2 * AK * sizeof(float) +
             (size_t)AK * (AQ * sizeof(T) + 16);
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

template <typename T, int DP, bool DROP>
static int launch_fwd_d(const AttnParams& p, hipStream_t st) {
  dim3 grid((p.T + AQ - 1) / AQ, p.nh, p.B);
  if constexpr (sizeof(T) == 2 && DP == 64) {
    if (attn_fwd_pipe_supported(p)) return launch_attn_fwd_pipe(p, st);   // attn_fwd_pipe.hip (round 5)
  }
  if constexpr (sizeof(T) == 2) {
    if (p.vt == nullptr) {   // V given row-major: transpose reads
      if constexpr (DP <= 64) {
        // 32-key tiles (three workgroups per CU) when the grid does not fit the 512 slots of two per CU: 8 documents of 709
        // tokens, 576 workgroups, 46 instead of 60 us; a grid that fits is a few per cent faster on 64-key tiles
        const bool narrow = (int64_t)grid.x * grid.y * grid.z > 512;
        if (narrow) {
          const size_t sh = fwd_smem<T, DP, 32>();
          int rc = set_smem(attn_fwd_kernel<T, DP, DROP, true, 32>, sh);
          if (rc) return rc;
          hipLaunchKernelGGL((attn_fwd_kernel<T, DP, DROP, true, 32>), grid, dim3(256), sh, st, p);
          return check_launch("peneo_attn_fwd");
        }
      }
      const size_t sh = fwd_smem<T, DP, AK>();
      int rc = set_smem(attn_fwd_kernel<T, DP, DROP, true, AK>, sh);
      if (rc) return rc;
      hipLaunchKernelGGL((attn_fwd_kernel<T, DP, DROP, true, AK>), grid, dim3(256), sh, st, p);
      return check_launch("peneo_attn_fwd");
    }
  }
  const size_t sh = fwd_smem<T, DP, AK>();
  int rc = set_smem(attn_fwd_kernel<T, DP, DROP, false, AK>, sh);
  if (rc) return rc;
  hipLaunchKernelGGL((attn_fwd_kernel<T, DP, DROP, false, AK>), grid, dim3(256), sh, st, p);
  return check_launch("peneo_attn_fwd");
}
template <typename T, int DP>
static int launch_fwd(const AttnParams& p, hipStream_t st) {
  return p.drop_p > 0.f ? launch_fwd_d<T, DP, true>(p, st) : launch_fwd_d<T, DP, false>(p, st);
}
template <typename T, int DP, bool DROP>
static int launch_bwd_d(const AttnParams& p, float* dq_acc, hipStream_t st) {
  int64_t rows = (int64_t)p.B * p.nh * p.T;
  {
    constexpr int VEC = Elem<T>::kVec;
    const int g = p.d / VEC;
    const bool vec = (p.d % VEC == 0) && (g == 1 || g == 2 || g == 4 || g == 8 || g == 16) &&
                     ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.d_out)) & 15) == 0 &&
                     ((p.ld_out * (int64_t)sizeof(T)) % 16 == 0);
    const unsigned vblocks = (unsigned)((rows * (g > 0 ? g : 1) + 255) / 256);
    if (vec && g == 8) hipLaunchKernelGGL((attn_delta_vec_kernel<T, 8>), dim3(vblocks), dim3(256), 0, st, p);
    else if (vec && g == 16) hipLaunchKernelGGL((attn_delta_vec_kernel<T, 16>), dim3(vblocks), dim3(256), 0, st, p);
    else if (vec && g == 4) hipLaunchKernelGGL((attn_delta_vec_kernel<T, 4>), dim3(vblocks), dim3(256), 0, st, p);
    else if (vec && g == 2) hipLaunchKernelGGL((attn_delta_vec_kernel<T, 2>), dim3(vblocks), dim3(256), 0, st, p);
    else if (vec && g == 1) hipLaunchKernelGGL((attn_delta_vec_kernel<T, 1>), dim3(vblocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
  }
  int rc = check_launch("peneo_attn_bwd(delta)");
  if (rc) return rc;
  if constexpr (sizeof(T) == 2) {
    const int64_t cols = (int64_t)p.nh * p.d;
    const bool dq_from_ds = dq_acc == nullptr && p.ds_out != nullptr;   // single pass, dQ by a second kernel from dS^T
    if (dq_acc || dq_from_ds) {   // eligibility was decided by peneo_attn_bwd
      const int64_t R = (int64_t)p.B * p.T;
      if (dq_acc && hipMemsetAsync(dq_acc, 0, sizeof(float) * R * cols, st) != hipSuccess) {
        set_error("peneo_attn_bwd: memset of the dQ accumulator failed");
        return PENEO_ERR_LAUNCH;
      }
      if constexpr (DP == 64) {
        if (dq_from_ds && attn_bwd_pipe_supported(p)) {   // the pipelined form (attn_bwd_pipe.hip) + dQ from its dS^T slab
          rc = launch_attn_bwd_pipe(p, st);
          if (rc <= 0) return rc;      // 0: dK, dV, the slab AND dQ are launched
          size_t sq = dqs_smem<DP>();
          rc = set_smem(attn_dq_from_ds_kernel<DP>, sq);
          if (rc) return rc;
          dim3 qgrid((p.T + AQ - 1) / AQ, p.nh, p.B);
          hipLaunchKernelGGL((attn_dq_from_ds_kernel<DP>), qgrid, dim3(256), sq, st, p);
          return check_launch("peneo_attn_bwd(dq from dS)");
        }
      }
      size_t sf = fused_smem<DP>();
      dim3 fgrid((p.T + FKEYS - 1) / FKEYS, p.nh, p.B);
      const int occ = DP <= 64 ? 2 : 1;        // workgroups per CU (256 registers at head dims above 64)
#define PENEO_LAUNCH_FUSED(HB_, OCC_)                                                                        \
  {                                                                                                          \
    rc = set_smem(attn_bwd_fused_kernel<DP, DROP, HB_, OCC_>, sf);                                           \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL((attn_bwd_fused_kernel<DP, DROP, HB_, OCC_>), fgrid, dim3(256), sf, st, p, dq_acc);    \
  }
      if (p.bias) { if (occ >= 2) PENEO_LAUNCH_FUSED(true, 2) else PENEO_LAUNCH_FUSED(true, 1) }
      else { if (occ >= 2) PENEO_LAUNCH_FUSED(false, 2) else PENEO_LAUNCH_FUSED(false, 1) }
#undef PENEO_LAUNCH_FUSED
      rc = check_launch("peneo_attn_bwd(fused)");
      if (rc) return rc;
      if (dq_from_ds) {
        size_t sq = dqs_smem<DP>();
        rc = set_smem(attn_dq_from_ds_kernel<DP>, sq);
        if (rc) return rc;
        dim3 qgrid((p.T + AQ - 1) / AQ, p.nh, p.B);
        hipLaunchKernelGGL((attn_dq_from_ds_kernel<DP>), qgrid, dim3(256), sq, st, p);
        return check_launch("peneo_attn_bwd(dq from dS)");
      }
      int64_t blocks = (R * (cols / 8) + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(dq_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dq_acc, R, (int)cols,
                         reinterpret_cast<bf16_t*>(p.dq), p.ld_d);
      return check_launch("peneo_attn_bwd(dq finish)");
    }
  }
  dim3 grid((p.T + AQ - 1) / AQ, p.nh, p.B);
  size_t s1 = dq_smem<T, DP>();
  rc = set_smem(attn_bwd_dq_kernel<T, DP, DROP>, s1);
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DP, DROP>), grid, dim3(256), s1, st, p);
  rc = check_launch("peneo_attn_bwd(dq)");
  if (rc) return rc;
  size_t s2 = dkv_smem<T, DP>();
  rc = set_smem(attn_bwd_dkv_kernel<T, DP, DROP>, s2);
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DP, DROP>), grid, dim3(256), s2, st, p);
  return check_launch("peneo_attn_bwd(dkv)");
}
template <typename T, int DP>
static int launch_bwd(const AttnParams& p, float* dq_acc, hipStream_t st) {
  return p.drop_p > 0.f ? launch_bwd_d<T, DP, true>(p, dq_acc, st) : launch_bwd_d<T, DP, false>(p, dq_acc, st);
}

template <typename T>
static int dispatch(const AttnParams& p, bool bwd, hipStream_t st, float* dq_acc = nullptr) {
  const int dp = (p.d + 31) / 32 * 32;
  switch (dp) {
    case 32: return bwd ? launch_bwd<T, 32>(p, dq_acc, st) : launch_fwd<T, 32>(p, st);
    case 64: return bwd ? launch_bwd<T, 64>(p, dq_acc, st) : launch_fwd<T, 64>(p, st);
    case 96: return bwd ? launch_bwd<T, 96>(p, dq_acc, st) : launch_fwd<T, 96>(p, st);
    case 128: return bwd ? launch_bwd<T, 128>(p, dq_acc, st) : launch_fwd<T, 128>(p, st);
    default: set_error("attention: head dim %d not supported (<= 128)", p.d); return PENEO_ERR_INVALID;
  }
}

}  // namespace peneo
using namespace peneo;

#ifdef ATTN_PROF
extern "C" int peneo_attn_prof_read(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_attn_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
extern "C" int peneo_attn_padded_len(int T) { return (T + 63) / 64 * 64; }
extern "C" int peneo_attn_padded_dim(int d) { return (d + 31) / 32 * 32; }

extern "C" int peneo_head_transpose(int dtype, const void* src, int64_t ld, int B, int nh, int T, int d, void* dst,
                                    peneo_stream_t stream) {
  PENEO_REQUIRE(dtype == PENEO_F32 || dtype == PENEO_BF16, "peneo_head_transpose: bad dtype");
  PENEO_REQUIRE(src && dst && B > 0 && nh > 0 && T > 0 && d > 0 && ld >= (int64_t)nh * d, "peneo_head_transpose: bad arguments");
  const int DP = peneo_attn_padded_dim(d), Tp = peneo_attn_padded_len(T);
  dim3 grid(Tp / 32, DP / 32, B * nh);
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(head_transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld, T, d, nh, (bf16_t*)dst, DP, Tp);
  else
    hipLaunchKernelGGL(head_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, ld, T, d, nh, (float*)dst, DP, Tp);
  return check_launch("peneo_head_transpose");
}

static int attn_common_check(const char* who, int dtype, int B, int nh, int T, int d, const void* bias, int64_t bias_ld) {
  PENEO_REQUIRE(dtype == PENEO_F32 || dtype == PENEO_BF16, "%s: bad dtype", who);
  PENEO_REQUIRE(B > 0 && nh > 0 && T > 0 && d > 0 && d <= 128, "%s: bad sizes", who);
  if (bias) {
    PENEO_REQUIRE(bias_ld == peneo_attn_padded_len(T), "%s: bias row stride must be peneo_attn_padded_len(T) = %d", who,
                  peneo_attn_padded_len(T));
    PENEO_REQUIRE((reinterpret_cast<uintptr_t>(bias) & 15) == 0, "%s: bias must be 16-byte aligned", who);
  }
  return PENEO_OK;
}

extern "C" void peneo_attn_drop_words_dims(int T, int* n_query_blocks, int* n_key_slots) {
  const int t128 = (T + 127) / 128;
  if (n_query_blocks) *n_query_blocks = 4 * t128;
  if (n_key_slots) *n_key_slots = 128 * t128;
}
extern "C" int64_t peneo_attn_drop_words_count(int B, int nh, int T) {
  int nqb = 0, tk = 0;
  peneo_attn_drop_words_dims(T, &nqb, &tk);
  return (int64_t)B * nh * nqb * tk;
}
extern "C" int peneo_attn_drop_words(uint32_t* words, int B, int nh, int T, float drop_p, uint32_t seed, peneo_stream_t stream) {
  PENEO_REQUIRE(words && B > 0 && nh > 0 && T > 0, "peneo_attn_drop_words: bad arguments");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "peneo_attn_drop_words: drop_p out of range");
  const int64_t n = peneo_attn_drop_words_count(B, nh, T);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(attn_drop_words_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, words, n,
                     pair_drop_thr16_host(drop_p), seed);
  return check_launch("peneo_attn_drop_words");
}
static void set_drop(AttnParams& p, float drop_p, const uint32_t* words) {
  p.drop_p = drop_p; p.keep_scale = pair_drop_scale_host(drop_p); p.words = words;
  peneo_attn_drop_words_dims(p.T, &p.nqb, &p.Tk);
}

extern "C" int peneo_attn_fwd(int dtype, const void* q, const void* k, const void* v, int64_t ld_qk, const void* vt, int B, int nh, int T,
                              int d, float scale, const void* bias, int64_t bias_ld, const float* key_bias, void* out,
                              int64_t ld_out, float* lse, float drop_p, const uint32_t* drop_words, peneo_stream_t stream) {
  int rc = attn_common_check("peneo_attn_fwd", dtype, B, nh, T, d, bias, bias_ld);
  if (rc) return rc;
  PENEO_REQUIRE(q && k && out && (vt || (v && dtype == PENEO_BF16)), "peneo_attn_fwd: null pointer (fp32 needs the transposed copy vt)");
  PENEO_REQUIRE(ld_qk >= (int64_t)nh * d && ld_out >= (int64_t)nh * d, "peneo_attn_fwd: leading dims too small");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "peneo_attn_fwd: drop_p out of range");
  PENEO_REQUIRE(drop_p == 0.f || (drop_words && (reinterpret_cast<uintptr_t>(drop_words) & 3) == 0),
                "peneo_attn_fwd: drop_p > 0 needs the keep words of peneo_attn_drop_words");
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.ld = ld_qk; p.vt = vt; p.B = B; p.nh = nh; p.T = T; p.d = d; p.Tp = peneo_attn_padded_len(T);
  p.scale = scale; p.bias = bias; p.bias_ld = bias_ld; p.key_bias = key_bias; p.out = out; p.ld_out = ld_out; p.lse = lse;
  set_drop(p, drop_p, drop_words);
  return dtype == PENEO_BF16 ? dispatch<bf16_t>(p, false, (hipStream_t)stream) : dispatch<float>(p, false, (hipStream_t)stream);
}

extern "C" int peneo_attn_bwd(int dtype, const void* q, const void* k, const void* v, int64_t ld_qkv, const void* kt,
                              const void* qt, const void* dot, const void* out, const void* d_out, int64_t ld_out,
                              const float* lse, int B, int nh, int T, int d, float scale, const void* bias, int64_t bias_ld,
                              const float* key_bias, void* dq, void* dk, void* dv, int64_t ld_dqkv, float* g_bias, float* delta,
                              float* dq_accum, void* ds_out, float drop_p, const uint32_t* drop_words, peneo_stream_t stream) {
  int rc = attn_common_check("peneo_attn_bwd", dtype, B, nh, T, d, bias, bias_ld);
  if (rc) return rc;
  PENEO_REQUIRE(q && k && v && out && d_out && lse && dq && dk && dv && delta, "peneo_attn_bwd: null pointer");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || drop_words),
                "peneo_attn_bwd: drop_p > 0 needs the keep words the forward used (peneo_attn_drop_words)");
  // single-pass eligibility (bf16, accumulator given, dq rows writable as 16-byte vectors); otherwise the two-kernel path
  bool fused = dtype == PENEO_BF16 && (dq_accum != nullptr || ds_out != nullptr) && (((int64_t)nh * d) % 8 == 0) &&
               ((reinterpret_cast<uintptr_t>(dq) & 15) == 0) && ((ld_dqkv * 2) % 16 == 0);
  if (!fused) dq_accum = nullptr;
  PENEO_REQUIRE(fused || (kt && qt && dot), "peneo_attn_bwd: the two-kernel path needs the transposed copies kt / qt / dot");
  PENEO_REQUIRE(!ds_out || fused, "peneo_attn_bwd: ds_out is written by the single-pass (bf16) kernel only");
  PENEO_REQUIRE(!ds_out || (reinterpret_cast<uintptr_t>(ds_out) & 15) == 0, "peneo_attn_bwd: ds_out must be 16-byte aligned");
  PENEO_REQUIRE(ld_qkv >= (int64_t)nh * d && ld_out >= (int64_t)nh * d && ld_dqkv >= (int64_t)nh * d, "peneo_attn_bwd: leading dims too small");
  PENEO_REQUIRE(!g_bias || bias, "peneo_attn_bwd: g_bias needs bias (it shares its row stride)");
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.kt = kt; p.qt = qt; p.dot = dot; p.B = B; p.nh = nh; p.T = T; p.d = d;
  p.Tp = peneo_attn_padded_len(T); p.scale = scale; p.bias = bias; p.bias_ld = bias_ld; p.key_bias = key_bias;
  p.out = const_cast<void*>(out); p.ld_out = ld_out; p.lse = const_cast<float*>(lse);
  set_drop(p, drop_p, drop_words); p.d_out = d_out; p.dq = dq; p.dk = dk; p.dv = dv; p.ld_d = ld_dqkv;
  p.g_bias = g_bias; p.delta = delta; p.ds_out = ds_out;
  return dtype == PENEO_BF16 ? dispatch<bf16_t>(p, true, (hipStream_t)stream, dq_accum)
                             : dispatch<float>(p, true, (hipStream_t)stream, nullptr);
}
