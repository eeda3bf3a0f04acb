// Shared device helpers for the peneo_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/peneo_hip.h"

namespace peneo {

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kWave = 64;

// ---------------------------------------------------------------- error plumbing (host)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define PENEO_REQUIRE(cond, ...)                \
  do {                                          \
    if (!(cond)) {                              \
      ::peneo::set_error(__VA_ARGS__);          \
      return PENEO_ERR_INVALID;                 \
    }                                           \
  } while (0)

// ---------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// fp32 -> bf16, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
// fp32 -> f16 pair, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kVec = 4;  // elements per 16-byte vector
  static constexpr int kDtype = PENEO_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kVec = 8;
  static constexpr int kDtype = PENEO_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// 16-byte vector <-> floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y);
  out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  out[4] = __uint_as_float(v.z << 16); out[5] = __uint_as_float(v.z & 0xffff0000u);
  out[6] = __uint_as_float(v.w << 16); out[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* in);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* in) {
  return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* in) {
  return make_uint4(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]), pack_bf16x2(in[4], in[5]),
                    pack_bf16x2(in[6], in[7]));
}

// ---------------------------------------------------------------- activations
// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE-division / denormal-safe library forms: an fp32 divide is
// ~10 VALU instructions, and SiLU runs on every one of the nh*D hidden units of every token pair.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// bf16 throughput mode: erf(z) = z P(z^2) on |z| <= 3 (degree-8 P, weighted least squares on Chebyshev nodes, then rescaled
// and nudged by a few ulps so that the fp32 Horner evaluation at z = 3 is EXACTLY 1.0f: arguments beyond 3 are clamped, so erf
// saturates at +-1 and GELU(x) -> 0 for x -> -inf instead of -1.1e-5 |x|; |error| <= 2.5e-5 on [-3, 3], 2.2e-5 beyond (erf(3) =
// 1 - 2.2e-5), i.e. GELU to 5e-5 absolute for every x - a tenth of a bf16 rounding step of a typical activation).  Ten
// full-rate instructions and no transcendental: the Abramowitz-Stegun form used before (one rcp, one exp2, 1.5e-7) cost 17 of
// FFN1's 62 us in the epilogue, where nothing else overlaps.  fp32 parity mode keeps the library erff.
__device__ __forceinline__ float fast_erf(float x) {
  const float z = __builtin_amdgcn_fmed3f(x, -3.0f, 3.0f);
  const float u = z * z;
  float p = fmaf(0x1.5032b6p-25f, u, -0x1.f99df6p-20f);
  p = fmaf(p, u, 0x1.505c6ap-15f);
  p = fmaf(p, u, -0x1.07b80ap-11f);
  p = fmaf(p, u, 0x1.1312f8p-8f);
  p = fmaf(p, u, -0x1.999afep-6f);
  p = fmaf(p, u, 0x1.c6626ep-4f);
  p = fmaf(p, u, -0x1.803ac4p-2f);
  p = fmaf(p, u, 0x1.20d59ap+0f);
  return p * z;
}
__device__ __forceinline__ float gelu_fast_f(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
  return fmaf(x, pdf, cdf);
}
__device__ __forceinline__ float act_f(int act, float x) {
  return act == PENEO_ACT_GELU ? gelu_f(x) : (act == PENEO_ACT_SILU ? silu_f(x) : x);
}
__device__ __forceinline__ float act_grad_f(int act, float x) {
  return act == PENEO_ACT_GELU ? gelu_grad_f(x) : (act == PENEO_ACT_SILU ? silu_grad_f(x) : 1.0f);
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, idx) is a pure function, so backward kernels regenerate the forward mask.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool dropout_keep(uint32_t seed, uint64_t idx, uint32_t thresh /* p * 2^32 */) {
  uint32_t h = mix32((uint32_t)idx ^ mix32(seed ^ (uint32_t)(idx >> 32) * 0x9e3779b9u));
  return h >= thresh;
}
// The same function with the inner hash hoisted: base = dropout_base(seed, idx >> 32) is constant over any run of elements
// that does not cross a 2^32 boundary, leaving one mix32 (two quarter-rate multiplies) per element.
__device__ __forceinline__ uint32_t dropout_base(uint32_t seed, uint64_t idx) { return mix32(seed ^ (uint32_t)(idx >> 32) * 0x9e3779b9u); }
__device__ __forceinline__ bool dropout_keep_b(uint32_t base, uint32_t lo, uint32_t thresh) { return mix32(lo ^ base) >= thresh; }
// ---- K12 dropout: the Dropout between the two layers of every pair classifier (model/peneo_decoder.py:261), acting on the
// [B, P, nh*D] hidden that never exists in memory.  The hidden units of one pair are walked in slabs of 32 (the kernels'
// weight slabs); a slab splits into two halves of 16 units, half h = units 8g + 4h + e (g = 0..3, e = 0..3): what ONE lane
// of the forward kernel holds (accumulator rows 8g + 4 half + e of its pair).  Each (document, pair, slab, half) owns a
// chain of 16 fields: a two-round 24-bit-multiply mixer (full-rate v_mul_u32_u24, as the attention mask) of the counter
// seeds it, then one v_mad_u32_u24 per field:
//     w = premix(((p * nslab + slab) * 2 + h) ^ key(seed, b));  st_0 = mix24(w);  st_{i+1} = (st_i[23:0] * 0xC2B2AF + inc(w[31:24])) mod 2^32
//     keep(unit 8 (i >> 2) + 4 h + (i & 3)) = (st_{i+1} >> 16) >= p_drop * 2^16
// i.e. 1.5 integer operations per element instead of a hash each (the mask generation was 27 % of the forward kernel).
// Backward kernels regenerate the same bits.  tools/check_dropout_hash.py: keep rate, neighbour / slab / pair correlations
// and count statistics at the noise floor.
__device__ __forceinline__ uint32_t pair_drop_key(uint32_t seed, int b) { return mix32(seed ^ ((uint32_t)(b + 1) * 0x9e3779b9u)); }
// One full 32-bit multiply in front of the 24-bit mixer (round 5): the v_mul_u32_u24 rounds read 24 bits, so without it bits 24..31
// of the counter entered only through the x >> 16 fold and counters c and c ^ 0x01000100 seeded the SAME chain - a structural
// duplicate at a fixed far offset for every chain once the counter passes 2^24 (config 4: N = 1023, 80 slabs -> 83.8 M chains).
// An odd multiplier is a bijection on 32 bits and carries every counter bit into the bits the mixer reads.
__device__ __forceinline__ uint32_t pair_drop_premix(uint32_t key, uint32_t counter /* (p * nslab + slab) * 2 + half */) {
  uint32_t x = (counter ^ key) * 0x9E3779B1u;
  return x ^ (x >> 16);
}
__device__ __forceinline__ uint32_t pair_drop_seed(uint32_t key, uint32_t counter) {
  uint32_t x = pair_drop_premix(key, counter);
  x = __umul24(x, 0x9E3779u); x ^= x >> 13; x = __umul24(x, 0x85EBCBu); x ^= x >> 16;
  return x;
}
// A chain is an LCG on the low 24 bits of its state, so the mixer above (which reads 24 bits of the premixed word) can start at most
// 2^24 different chains.  Round 5: the chain's INCREMENT comes from the 8 premixed bits the mixer does NOT read (odd: bit 0 of
// 0x9E3779 is kept), so the 2^32 premixed words give 2^32 different chains and duplicates among config 4's 83.8 M chains are at the
// birthday level (tools/check_dropout_hash.py --far).  The v_mad_u32_u24 of a field takes the increment from a register instead of a
// constant: same instruction count.  (The attention keep words take the increment from bits 24..31 of their mix32 seed word.)
__device__ __forceinline__ uint32_t pair_drop_inc_of(uint32_t word) { return 0x9E3779u ^ ((word >> 24) << 8); }
__device__ __forceinline__ uint32_t pair_drop_inc(uint32_t key, uint32_t counter) { return pair_drop_inc_of(pair_drop_premix(key, counter)); }
__device__ __forceinline__ uint32_t pair_drop_step(uint32_t st, uint32_t inc) { return __umul24(st, 0xC2B2AFu) + inc; }
__device__ __forceinline__ uint32_t pair_drop_thr16_dev(float p) { return p > 0.f ? (uint32_t)(p * 65536.f + 0.5f) : 0u; }
// threshold and the matching scale: p is realised as thr16 / 2^16 (0.1 -> 6554 / 65536) and the kept values are scaled by
// 2^16 / (2^16 - thr16), so the mask is unbiased for the probability it really uses
inline uint32_t pair_drop_thr16_host(float p) { return p > 0.f ? (uint32_t)(p * 65536.f + 0.5f) : 0u; }
inline float pair_drop_scale_host(float p) { const uint32_t t = pair_drop_thr16_host(p); return t ? 65536.f / (65536.f - (float)t) : 1.f; }
// Kernels whose lane owns ONE hidden unit (chain position i) and walks many pairs jump to that position: the 24-bit state
// after i steps is affine in the seed, st_i[23:0] = (A^i st_0 + C_i) mod 2^24, so an element costs the seed + two multiply-adds.
struct PairDropJump { uint32_t a, s; };   // st_i = a st_0 + inc s   (mod 2^24):  a = A^i,  s = 1 + A + ... + A^(i-1)
__device__ __forceinline__ PairDropJump pair_drop_jump(int i /* chain position 0..15 */) {
  uint32_t a = 1u, sm = 0u;
  for (int k = 0; k < i; ++k) { sm = (sm * 0xC2B2AFu + 1u) & 0xFFFFFFu; a = (a * 0xC2B2AFu) & 0xFFFFFFu; }
  return PairDropJump{a, sm};
}
__device__ __forceinline__ bool pair_drop_keep_at(uint32_t seedword, uint32_t inc, PairDropJump j, uint32_t thr16) {
  const uint32_t si = __umul24(seedword, j.a) + __umul24(inc, j.s);   // low 24 bits = state after i steps
  return (pair_drop_step(si, inc) >> 16) >= thr16;
}
// one element (the chunked / fp32 kernels, where speed does not matter): n = column in [0, nslab * 32)
__device__ __forceinline__ bool pair_drop_keep(uint32_t key, int64_t pair, int n, int nslab, uint32_t thr16) {
  const int slab = n >> 5, w = n & 31, half = (w >> 2) & 1, i = 4 * (w >> 3) + (w & 3);
  const uint32_t cnt = (uint32_t)((pair * nslab + slab) * 2 + half);
  uint32_t st = pair_drop_seed(key, cnt);
  const uint32_t inc = pair_drop_inc(key, cnt);
  for (int k = 0; k <= i; ++k) st = pair_drop_step(st, inc);
  return (st >> 16) >= thr16;
}

// 8 consecutive elements starting at idx0 (handles the rare run that crosses a 2^32 boundary)
__device__ __forceinline__ uint32_t dropout_keep8(uint32_t seed, uint64_t idx0, uint32_t thresh) {
  const uint32_t lo0 = (uint32_t)idx0;
  uint32_t mask = 0;
  if (lo0 <= 0xfffffff8u) {
    const uint32_t base = dropout_base(seed, idx0);
#pragma unroll
    for (int i = 0; i < 8; ++i) mask |= (dropout_keep_b(base, lo0 + i, thresh) ? 1u : 0u) << i;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) mask |= (dropout_keep(seed, idx0 + i, thresh) ? 1u : 0u) << i;
  }
  return mask;
}

// ---------------------------------------------------------------- MFMA wrappers
// One "k-step" covers 16 reduction elements: lanes 0-31 hold k 0..7, lanes 32-63 hold k 8..15 of the step,
// for row/col (lane & 31).  For bf16 that is one v_mfma_f32_32x32x16_bf16; for fp32 eight
// v_mfma_f32_32x32x2_f32 (exact fp32, element t of each lane pairs k=t with k=8+t).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { uint4 v; };
template <> struct Frag<float> { uint4 v[2]; };

__device__ __forceinline__ void mma_step(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x16_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const Frag<float>& a, const Frag<float>& b, f32x16_t& acc) {
  const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a.v[0]);
  const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b.v[0]);
#pragma unroll
  for (int t = 0; t < 8; ++t)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(pa[t]), __uint_as_float(pb[t]), acc, 0, 0, 0);
}

// 8 floats -> one MFMA fragment (bf16: rounded to nearest-even; fp32: verbatim)
template <typename T> __device__ __forceinline__ Frag<T> pack_frag8(const float* v);
template <> __device__ __forceinline__ Frag<bf16_t> pack_frag8<bf16_t>(const float* v) {
  Frag<bf16_t> f; f.v = pack16<bf16_t>(v); return f;
}
template <> __device__ __forceinline__ Frag<float> pack_frag8<float>(const float* v) {
  Frag<float> f; f.v[0] = pack16<float>(v); f.v[1] = pack16<float>(v + 4); return f;
}

// accumulator element r of lane l of a 32x32 tile sits at (row, col):
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

// ---------------------------------------------------------------- block walk of the pair triangle (pair_bwd.hip; the saving forward)
// One block = 8 rows i x 16 columns j of the triangle = 128 pairs = four groups of 32 (group g: rows 2g, 2g + 1); blocks on the
// diagonal / the last row of blocks carry pairs outside the triangle (i > j, or beyond N).
constexpr int PB_TI = 8, PB_TJ = 16;
constexpr int PB_ROWS = PB_TI * PB_TJ;
__host__ __device__ inline int pb_row_tiles(int N) { return (N + PB_TI - 1) / PB_TI; }
__host__ __device__ inline int pb_col_tiles(int N) { return (N + PB_TJ - 1) / PB_TJ; }
// blocks of row-tile ti: tj = first(ti) .. pb_col_tiles - 1, first(ti) = 8 ti / 16
__host__ __device__ inline int pb_tiles_before(int ti, int N) {
  const int m = ti >> 1;
  return ti * pb_col_tiles(N) - (m * (m - 1) + ((ti & 1) ? m : 0));
}
__host__ __device__ inline int pb_num_tiles(int N) { return pb_tiles_before(pb_row_tiles(N), N); }
// Saved activations of the classifier heads (peneo_pair_heads_fwd_save -> peneo_pair_bwd_saved): per document, block, 32-unit slab
// of the nh * D hidden units and group one 2 KiB record = the pre-activations z = W1 x + b1 of 32 pairs x 32 units as f16, a dropped
// unit's as -30000 (the K12 dropout; SiLU and SiLU' of that are 0).  Record = [2 halves of 16 units][32 pairs][16 units]: 32-byte
// rows in the group's pair order, the rows of the second half stored at pair ^ 4 (bank spread of the backward's transposing reads)
constexpr int PB_REC_BYTES = 2048;

// ---------------------------------------------------------------- LDS-DMA (global -> LDS without VGPRs)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// One 1 KiB piece (global_load_lds_dwordx4: 64 lanes x 16 B; LDS destination = M0 base + OFF + lane*16, global source =
// the lane's own pointer + OFF).  Issued from inline asm on purpose: hipcc drains the whole vm counter in front of every
// ds_read while an LDS-DMA *it knows about* is in flight.  The caller orders it: counted s_waitcnt vmcnt + barrier.
template <int OFF>
__device__ __forceinline__ void lds_dma_1k(const char* gsrc_lane, uint32_t lds_base_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_base_uniform), "n"(OFF) : "memory");
}
// The same piece addressed as UNIFORM 64-bit base (SGPR pair) + per-lane 32-bit byte offset (global "saddr" form): the issue
// reads one address VGPR per lane instead of two, and a kernel that walks k keeps its per-lane offsets constant and advances
// the base with scalar adds (no 64-bit VALU pointer arithmetic per piece and k-tile).
template <int OFF>
__device__ __forceinline__ void lds_dma_1k_s(uint32_t voff_lane, const char* base_uniform, uint32_t lds_base_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_lane), "s"(base_uniform), "s"(lds_base_uniform), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// UPW consecutive 1 KiB pieces (this wave's share of a weight slab).  Hidden in asm, the DMA is ours to order: counted
// s_waitcnt vmcnt + s_barrier before the slab is read (cdna guide §5.7).
template <int U, int UPW>
__device__ __forceinline__ void lds_dma_units(const char* gsrc_lane, uint32_t lds_base_uniform) {
  if constexpr (U < UPW) {
    lds_dma_1k<(U % 4) * 1024>(gsrc_lane + (U / 4) * 4096, lds_base_uniform + (U / 4) * 4096);
    lds_dma_units<U + 1, UPW>(gsrc_lane, lds_base_uniform);
  }
}

// fragment `frag_index` of a buffer in MFMA fragment order (64 lanes x 8 elements per fragment)
template <typename T> struct FragBytes { static constexpr int v = 8 * (int)sizeof(T); };  // per lane per fragment
template <typename T>
__device__ __forceinline__ Frag<T> load_frag_linear(const char* base, int frag_index, int lane) {
  Frag<T> f;
  const char* p = base + ((int64_t)frag_index * 64 + lane) * FragBytes<T>::v;
  if constexpr (sizeof(T) == 2) f.v = *reinterpret_cast<const uint4*>(p);
  else { f.v[0] = *reinterpret_cast<const uint4*>(p); f.v[1] = *reinterpret_cast<const uint4*>(p + 16); }
  return f;
}

// ---------------------------------------------------------------- packed upper-triangular pair index
// p(i, j) = i*n - i(i-1)/2 + (j - i),  0 <= i <= j < n   (reference: model/peneo_decoder.py:129-147)
__device__ __host__ __forceinline__ int64_t pair_row_start(int64_t i, int64_t n) { return i * n - i * (i - 1) / 2; }
__device__ __forceinline__ void pair_decode(int64_t p, int n, int& i, int& j) {
  // largest i with row_start(i) <= p
  double nn = 2.0 * n + 1.0;
  int ii = (int)((nn - sqrt(nn * nn - 8.0 * (double)p)) * 0.5);
  if (ii < 0) ii = 0;
  if (ii > n - 1) ii = n - 1;
  while (ii > 0 && pair_row_start(ii, n) > p) --ii;
  while (ii < n - 1 && pair_row_start(ii + 1, n) <= p) ++ii;
  i = ii;
  j = ii + (int)(p - pair_row_start(ii, n));
}

}  // namespace peneo
