// Decoder backward through the pair space in ONE kernel (bf16): for every token pair (i, j), i <= j, of every document
//
//   x   = SiLU(a_i + b_j)                                   rebuilt in registers (MFMA operand fragments)
//   z   = x W1cat^T + b1                                    matrix cores, accumulator = z[pair, hidden]
//   dz  = (sum_c scale_h dlogits_h[p, c] W2_h[c, :]) * SiLU'(z)        -> written once ([rows, nh*D] bf16, for dW1 = dz^T x)
//   du  = dz W1cat                                          matrix cores again, accumulated over all hidden slabs in registers
//   d_a[i] += sum_j du * SiLU'(a_i + b_j),  d_b[j] += sum_i du * SiLU'(a_i + b_j)
//   dW2 / db1 column sums                                   (as peneo_pair_dz_fused)
//
// i.e. the autograd graph through model/peneo_decoder.py:149-177 (HandshakingKernel) and :231-292 (the five classifier
// heads) except the weight gradient of the first layers, which stays a GEMM over the dz / x this kernel leaves behind.
// Against the kernel chain it replaces (peneo_pair_x_fwd, peneo_pair_dz_fused, the du GEMM with its SiLU' epilogue,
// peneo_pair_x_bwd) dz is read back from HBM once instead of twice, and neither a_i + b_j nor du ever go through memory.
//
// Work unit: one workgroup owns a block of 8 rows i x 16 columns j of the pair triangle, wave w the rows 2w, 2w+1.  With
// pairs blocked in 2-D the sums over j (d_a) stay inside a wave and the sums over i (d_b) need 8x fewer atomics than a
// row-major walk of the triangle.  dz and x are written in this block order ("rows" below); the dW1 GEMM only needs both
// in the SAME order.  Pairs of a block outside the triangle (i > j, or beyond N) carry dlogits = 0: their dz rows are 0.
//
// Weights: per 32-column slab of hidden units 2*KS fragments of 1 KiB (KS = D / 16), packed by peneo_pair_bwd_pack:
//   fragment ks < KS          : W1cat[slab*32 + (lane&31)][16 ks + 8 (lane>>5) + e]          B operand of z   (k = decoder dim)
//   fragment KS + 2 dt + kk   : W1cat[slab*32 + 16 kk + 8 (lane>>5) + e][32 dt + (lane&31)]  B operand of du  (k = hidden unit)
// streamed L2 -> LDS by LDS-DMA, each half through its own two-slot ring (the du half of slab s is loaded one iteration
// after the z half, it is needed one phase later).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace peneo {

constexpr int PB_WAVES = 4;
constexpr int PB_SLOTS = 256;                       // rows of the dW2 / db1 workspace the atomics are spread over
// (PB_TI / PB_TJ / PB_ROWS and the block walk pb_*: common.h - the forward that saves activations walks the same blocks)

struct PackBwdSrc { const float* w1[PENEO_MAX_HEADS]; int num_heads; int D; };

__global__ void pack_bwd_weights_kernel(PackBwdSrc s, bf16_t* out) {
  const int D = s.D, KS = D / 16;
  const int nslab = s.num_heads * D / 32;
  const int64_t stride = (int64_t)3 * KS * 512;     // elements per slab: z fragments | du fragments | z fragments, swizzled
  const int64_t total = (int64_t)nslab * stride;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int slab = (int)(q / stride);
    const int64_t i = q % stride;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), f = (int)(i >> 9);
    int row, col;                                   // element of W1cat [nh*D, D]
    if (f < KS) { row = slab * 32 + (lane & 31); col = 16 * f + 8 * (lane >> 5) + e; }
    else if (f < 2 * KS) { const int g = f - KS, dt = g >> 1, kk = g & 1; row = slab * 32 + 16 * kk + 8 * (lane >> 5) + e; col = 32 * dt + (lane & 31); }
    else {
      // third part (wave-specialised kernel): the z fragments again, 16-byte slot p of fragment ks holding the row
      // n = (p & 31) ^ (4 h | 8 (ks & 1)), h = p >> 5: the same bytes serve z (one ds_read_b128 per lane, any slot order) and,
      // read with ds_read_b64_tr_b16, the transposed du operand without bank conflicts
      const int ks = f - 2 * KS, h = lane >> 5, n = (lane & 31) ^ (4 * h + 8 * (ks & 1));
      row = slab * 32 + n; col = 16 * ks + 8 * h + e;
    }
    const int h = row / D;
    out[q] = f32_to_bf16(s.w1[h][(int64_t)(row - h * D) * D + col]);
  }
}

struct PairBwdParams {
  const bf16_t* ab; int B, N, D; int64_t P;
  const void* wp; const float* b1;
  peneo_pair_dz_args a;          // dlogits[h]: [B, P, classes[h]] of the whole batch
  bf16_t* dz; bf16_t* x;         // [B][ntiles * 128][nh*D] / [.. ][D]
  const char* act;               // saved-activation form: the forward's records (common.h: PB_REC_BYTES), else NULL
  float* part_a; float* part_b;  // per-block partial sums [B][ntiles][8][D] / [B][ntiles][16][D] fp32
  float* ws;                     // [PB_SLOTS][4 * nh*D]
  int ntiles;
  uint32_t drop_thr16, drop_seed; float drop_scale;   // K12 dropout of the forward (0 = off): mask regenerated, never stored
  unsigned long long* dbg;       // optional (tools/): per wave of the first 256 blocks: cycles in the loop, cycles waiting at the top
};

// hand-issued fragment reads (the compiler would wait for every ds_read right in front of its MFMA: with one wave per SIMD
// nothing else covers that latency).  The reader owns lgkmcnt: pb_lgkm0 waits and ties the fragment registers to the wait.
typedef __attribute__((ext_vector_type(4))) unsigned int pb_u32x4;
template <int OFF> __device__ __forceinline__ void pb_dsr(pb_u32x4& d, uint32_t a) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
__device__ __forceinline__ void pb_mma(const pb_u32x4& a, const pb_u32x4& b, f32x16_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void pb_mma(const Frag<bf16_t>& a, const pb_u32x4& b, f32x16_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
// fragments per chunk: a slab's KS fragments are dealt to 8 chunks, fragment f goes to chunk f * 8 / KS
template <int KS> constexpr int pb_chunk_count(int j) { int n = 0; for (int f = 0; f < KS; ++f) n += (f * 8 / KS == j) ? 1 : 0; return n; }
template <int KS> constexpr int pb_chunk_first(int j) { for (int f = 0; f < KS; ++f) if (f * 8 / KS == j) return f; return KS; }
constexpr int PB_MAXC = 3;   // fragments per chunk (KS = 24)

#ifndef PB_DBG
#define PB_DBG 0      // tools/ab_pb_dbg.sh: 65536 = packed-fp32 dz arithmetic in the wave-specialised kernel (reproduces the corruption)
#endif
#ifndef PB_ABLATE
#define PB_ABLATE 0   // timing experiments (tools/): 1 no dz stores, 2 no du MFMAs, 4 no z MFMAs, 8 no epilogue, 16 no du-half DMA, 32 no DMA, 64 no z fragment reads, 128 no mask generation (ws kernel)
#endif

// d_ab[b, i, :D] = sum over the blocks of row-tile i / 8 of part_a ; d_ab[b, j, D:] = sum over the blocks of column-tile j / 16 of part_b
__global__ __launch_bounds__(256) void pair_bwd_reduce_kernel(const float* part_a, const float* part_b, int N, int D, int ntiles,
                                                              float* d_ab) {
  const int n = blockIdx.x, b = blockIdx.y;
  const int ncolt = pb_col_tiles(N);
  const float* pa = part_a + (int64_t)b * ntiles * PB_TI * D;
  const float* pb = part_b + (int64_t)b * ntiles * PB_TJ * D;
  float* out = d_ab + ((int64_t)b * N + n) * 2 * D;
  const int ti = n / PB_TI, first = pb_tiles_before(ti, N), cnt = ncolt - (ti >> 1);
  const int tj = n / PB_TJ, ti_max = min(pb_row_tiles(N) - 1, 2 * tj + 1);
  // 16 bytes per lane and four partial rows in flight (D % 4 == 0: D is a multiple of 32): the rows of one sum lie 12 / 24 KiB
  // apart, one scalar load at a time left the launch at 1.9 TB/s for 312 MB
  const int nv = D / 4;
  for (int v = threadIdx.x; v < 2 * nv; v += blockDim.x) {
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [](float4& a, const float4 x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; };
    if (v < nv) {
      const float* src = pa + ((int64_t)first * PB_TI + (n % PB_TI)) * D + 4 * v;
      const int64_t step = (int64_t)PB_TI * D;
      int k = 0;
      for (; k + 4 <= cnt; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) add(acc[u], *reinterpret_cast<const float4*>(src + (k + u) * step));
      }
      for (; k < cnt; ++k) add(acc[0], *reinterpret_cast<const float4*>(src + k * step));
    } else {
      const int c4 = 4 * (v - nv);
      auto row = [&](int t) {
        const int tile = pb_tiles_before(t, N) + (tj - (t >> 1));
        return pb + ((int64_t)tile * PB_TJ + (n % PB_TJ)) * D + c4;
      };
      int t = 0;
      for (; t + 4 <= ti_max + 1; t += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) add(acc[u], *reinterpret_cast<const float4*>(row(t + u)));
      }
      for (; t <= ti_max; ++t) add(acc[0], *reinterpret_cast<const float4*>(row(t)));
    }
    add(acc[0], acc[1]); add(acc[2], acc[3]); add(acc[0], acc[2]);
    *reinterpret_cast<float4*>(out + 4 * v) = acc[0];
  }
}

// ================================================================================================
// Wave-specialised form (default): 8 waves per workgroup, two per SIMD.  Waves 0-3 ("producers", pair group w) own
// x, z and the dz arithmetic; waves 4-7 ("consumers", pair group w - 4) own the du accumulator.  A producer hands the dz
// tile of a slab to its consumer through a double-buffered LDS tile; the ONE workgroup barrier per slab that already
// orders the weight ring orders that hand-off too.  With two waves on every SIMD the hardware overlaps the producer's
// VALU-bound dz arithmetic with the consumer's MFMAs, and both roles fit 256 registers.
//   iteration s = -1 .. nslab:   producer: Z(s+1) interleaved with E(s) -> tile[s & 1]     consumer: U(s-1) <- tile[(s-1) & 1]
//   weight ring: ONE copy of a slab's first-layer weights (its z fragments, 16-byte slots XOR-swizzled) serves both products:
//   the producers read it with ds_read_b128, the consumers read the SAME bytes transposed (ds_read_b64_tr_b16) as the du
//   operand.  Slab s lives in slot s & 3 of a four-slot ring; slab s+2 is issued at the top of iteration s by all 8 waves
//   (KS/8 pieces each: an LDS-DMA piece costs its issuer hundreds of cycles once a few are in flight) and has landed by the
//   top of iteration s+1.
// Two rules this kernel obeys because two waves share each SIMD's matrix pipe (both found the hard way: sporadic wrong
// 16-byte pieces of dz that vanished as soon as the partner wave issued no MFMAs):
//   * nothing reads an MFMA accumulator shortly after the chain that wrote it: z is double buffered (Z(s+1) writes one
//     accumulator while E(s) reads the other, a whole barrier later), du is only read after the loop.  The wait states the
//     compiler inserts between an MFMA and a reader assume the MFMA started when it was issued; with a partner wave's
//     MFMAs queued in front of it, it did not.
//   * operand fragments are re-loaded only after the NEXT chunk's MFMAs have been issued behind the ones that read them.
// ================================================================================================
constexpr int PW_WAVES = 8;

template <int KS, bool DROP>
__global__ __launch_bounds__(PW_WAVES * 64, 2) void pair_bwd_ws_kernel(PairBwdParams p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HALF_BYTES = KS * 1024;
  constexpr int NDT = KS / 2;
  constexpr int NPIECE = KS;                                 // 1 KiB pieces per iteration: ONE copy of the slab's weights serves z and du
  constexpr int PPW = (NPIECE + PW_WAVES - 1) / PW_WAVES;    // pieces per wave
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int grp = wave & 3;                                  // pair group: rows 2 grp, 2 grp + 1 of the block
  const int D = p.D, N = p.N, nh = p.a.num_heads, ncol = nh * D;
  char* sA = smem;                                                          // [4][HALF_BYTES] weight ring: slab s in slot s & 3
  uint2* sW2p = reinterpret_cast<uint2*>(smem + 4 * HALF_BYTES);            // [ncol]: the column's W2 rows as bf16 (w0, w1 | w2, 0)
  float* sB1 = reinterpret_cast<float*>(sW2p + ncol);                       // [ncol]: first-layer bias
  float4* sG = reinterpret_cast<float4*>(sB1 + ncol);                       // [4][32]
  float4* sPart = sG + 4 * 32;                                              // [2][4][32]
  char* sT = reinterpret_cast<char*>(sPart + 2 * 4 * 32);                   // [2][4][32 rows][64 B]
  // classifier-dropout masks of a slab as f16 addends (0 = keep, -30000 = drop) for the pre-activation: [2][4 groups][32 units]
  // [2 halves][16 registers]: a producer lane (unit, half) reads its 16 register values with two ds_read_b128
  _Float16* sMask = reinterpret_cast<_Float16*>(sT + 2 * 4 * 2048);

  int ti = 0;
  {
    const int nti = pb_row_tiles(N);
    while (ti + 1 < nti && pb_tiles_before(ti + 1, N) <= (int)blockIdx.x) ++ti;
  }
  const int tj = (ti >> 1) + ((int)blockIdx.x - pb_tiles_before(ti, N));
  const int b = blockIdx.y;
  const int i0 = ti * PB_TI + 2 * grp, j0 = tj * PB_TJ;
  const int pi = i0 + (r32 >> 4), pj = j0 + (r32 & 15);
  const bool pair_ok = pi < N && pj < N && pi <= pj;
  const int ci = min(pi, N - 1), cj = min(pj, N - 1);
  const int64_t mypair = pair_row_start(ci, N) + (cj - ci);
  const int64_t rows_per_doc = (int64_t)p.ntiles * PB_ROWS;
  const int64_t row = (int64_t)b * rows_per_doc + (int64_t)blockIdx.x * PB_ROWS + grp * 32 + r32;
  const int nslab = ncol / 32, spb = D / 32;
  const T* abd = p.ab + (int64_t)b * N * 2 * D;

  for (int n = tid; n < ncol; n += PW_WAVES * 64) {
    const int h = n / D, k = n - h * D, Cn = p.a.classes[h];
    const float ds = DROP ? p.drop_scale : 1.f;     // dy = g W2 / (1 - p): the scale of the kept units rides on the W2 rows
    sW2p[n] = make_uint2(pack_bf16x2(ds * p.a.w2[h][k], Cn > 1 ? ds * p.a.w2[h][(int64_t)D + k] : 0.f),
                         pack_bf16x2(Cn > 2 ? ds * p.a.w2[h][(int64_t)2 * D + k] : 0.f, 0.f));
    sB1[n] = p.b1[n];
  }

  // weight stream: piece q of an iteration (q < KS: z fragments, else du fragments); wave w carries pieces w, w + 8, ...
  const char* wbase = reinterpret_cast<const char*>(p.wp) + lane * 16;
  const uint32_t ring = lds_addr(sA);
  auto dma_z = [&](int slab, int q) {
    if (!(PB_ABLATE & 32))
      lds_dma_1k<0>(wbase + (int64_t)slab * (3 * HALF_BYTES) + 2 * HALF_BYTES + q * 1024, __builtin_amdgcn_readfirstlane(ring + (slab & 3) * HALF_BYTES + q * 1024));
  };
  // the consumer waves carry the weight stream (the producers' instruction stream is the critical path, tools/pb_cycles.py)
  constexpr int PPC = (NPIECE + 3) / 4;
  auto dma_iter = [&](int s) {           // slab s+2 -> slot (s+2) & 3 (last read by U(s-2) during iteration s-1)
    if (wave < 4) return;
#pragma unroll
    for (int k = 0; k < PPC; ++k) {
      const int q = (wave - 4) + k * 4;
      if (q < NPIECE && s >= 0 && s + 2 < nslab) dma_z(s + 2, q);
    }
  };
#pragma unroll
  for (int k = 0; k < PPW; ++k) {        // z fragments of slabs 0 and 1 before the loop
    const int q = wave + k * PW_WAVES;
    if (q < KS) { dma_z(0, q); if (nslab > 1) dma_z(1, q); }
  }

#if !(PB_DBG & 65536)
  // Scalar pairs, and this file is built with -fno-slp-vectorize: NO packed-fp32 VALU (v_pk_mul / v_pk_fma / v_pk_add_f32) in
  // this kernel.  With the partner wave's MFMAs running on the same SIMD the packed forms returned wrong low halves in lanes
  // 48-63 (sporadic wrong 16-byte pieces of dz, gone with the partner's MFMAs removed, gone with scalar arithmetic; the
  // one-wave-per-SIMD form of this kernel (removed in round 3), where nothing shared the SIMD, was unaffected).
  struct f2 {
    float x, y;
    __device__ f2 operator+(const f2& o) const { return f2{x + o.x, y + o.y}; }
    __device__ f2 operator-(const f2& o) const { return f2{x - o.x, y - o.y}; }
    __device__ f2 operator*(const f2& o) const { return f2{x * o.x, y * o.y}; }
  };
  auto fma2 = [](const f2& a, const f2& b, const f2& c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; };
#else
  typedef float f2 __attribute__((ext_vector_type(2)));
  auto fma2 = [](const f2& a, const f2& b, const f2& c) { return __builtin_elementwise_fma(a, b, c); };
#endif
  const uint32_t fbase = ring + lane * 16;
  const int t_swz = (r32 >> 2) & 3;
  // top of an iteration, both roles: everything this wave put in flight has completed (LDS-DMA pieces: vmcnt; tile /
  // column-sum stores that OTHER waves read: lgkmcnt), then the workgroup meets
  unsigned long long t_wait = 0, t_vm = 0, t_land = 0;
  auto top = [&]() {
    if (p.dbg) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      t_vm += t1 - t0; t_wait += t2 - t1;
      return;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  const unsigned long long t_begin = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
  auto report = [&]() {
    if (p.dbg && blockIdx.y == 0 && blockIdx.x < 256 && lane == 0) {
      unsigned long long* d = p.dbg + ((int64_t)blockIdx.x * PW_WAVES + wave) * 4;
      d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_vm; d[2] = t_wait; d[3] = t_land;
    }
  };

  if (producer) {
    // ---------------------------------------------------------------- producer: x, z, dz
    const T* arow = abd + (int64_t)ci * 2 * D;
    const T* brow = abd + (int64_t)cj * 2 * D + D;
    T* x_row = p.x + row * D + 8 * half;
    Frag<T> xf[KS];
    {
      constexpr int G = KS % 4 == 0 ? 4 : 2;
      uint4 ra[2][G], rb[2][G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        ra[0][i] = *reinterpret_cast<const uint4*>(arow + 16 * i + 8 * half);
        rb[0][i] = *reinterpret_cast<const uint4*>(brow + 16 * i + 8 * half);
      }
#pragma unroll
      for (int g = 0; g < KS / G; ++g) {
        if (g + 1 < KS / G) {
#pragma unroll
          for (int i = 0; i < G; ++i) {
            ra[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(arow + 16 * (G * (g + 1) + i) + 8 * half);
            rb[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(brow + 16 * (G * (g + 1) + i) + 8 * half);
          }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
          float a[8], bb[8];
          unpack16<T>(ra[g & 1][i], a);
          unpack16<T>(rb[g & 1][i], bb);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e] + bb[e]);
          xf[G * g + i] = pack_frag8<T>(a);
          asm volatile("" : "+v"(xf[G * g + i].v.x), "+v"(xf[G * g + i].v.y), "+v"(xf[G * g + i].v.z), "+v"(xf[G * g + i].v.w) :: "memory");
          *reinterpret_cast<uint4*>(x_row + 16 * (G * g + i)) = xf[G * g + i].v;
        }
      }
    }
    float* slot = p.ws + (int64_t)(blockIdx.x % PB_SLOTS) * 4 * ncol;
    auto flush = [&](int s) {
      if (wave == (s & 3) && lane < 32) {
        const float4* src = sPart + (s & 1) * (4 * 32) + lane;
        float4 t = src[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 u = src[w * 32]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        float* dst = slot + s * 32 + lane;
        atomicAdd(dst, t.x); atomicAdd(dst + ncol, t.y); atomicAdd(dst + 2 * (int64_t)ncol, t.z); atomicAdd(dst + 3 * (int64_t)ncol, t.w);
      }
    };
    // MFMA operands of the current head, rebuilt every D / 32 slabs from scale_h * dlogits_h of the wave's 32 pairs:
    //   gA      A operand of dy[pair, hid] = sum_c g[pair, c] W2[c, hid]: row = pair (lanes 0-31 hold (g0, g1, g2, 0 ...), k = class)
    //   gT[kk]  A operand of the dW2 sums out[c, hid] = sum_pair g[pair, c] y[pair, hid]: row = class, k = pair in the order
    //           in which a C-layout accumulator register set presents the pairs (pi(kk, half, e) = 16 kk + 8 (e >> 2) + 4 half + (e & 3))
    pb_u32x4 gA = pb_u32x4{0u, 0u, 0u, 0u}, gT0 = gA, gT1 = gA;
    auto stage_g = [&](int h) {
      const int Cn = p.a.classes[h];
      float gx = 0.f, gy = 0.f, gz = 0.f, sc = 0.f;
      if (lane < 32 && pair_ok) {
        sc = p.a.scale[h];
        const float* dl = p.a.dlogits[h] + ((int64_t)b * p.P + mypair) * Cn;
        gx = dl[0];
        if (Cn > 1) gy = dl[1];
        if (Cn > 2) gz = dl[2];
      }
      // explicit wait: the compiler's counted vmcnt does not know about the LDS-DMA pieces in flight around these loads
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(gx), "+v"(gy), "+v"(gz), "+v"(sc) :: "memory");
      gx *= sc; gy *= sc; gz *= sc;
      gA = pb_u32x4{pack_bf16x2(gx, gy), pack_bf16x2(gz, 0.f), 0u, 0u};         // lanes >= 32 (k = 8..15) and bad pairs: zeros
      float* gt = reinterpret_cast<float*>(sG + grp * 32);                        // [3][32] floats: g_c of pair
      if (lane < 32) { gt[lane] = gx; gt[32 + lane] = gy; gt[64 + lane] = gz; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int c = r32;                                                          // class row of this lane in gT
      float v[16];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
          const float4 q = c < 3 ? *reinterpret_cast<const float4*>(gt + c * 32 + 16 * kk + 8 * e4 + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[8 * kk + 4 * e4 + 0] = q.x; v[8 * kk + 4 * e4 + 1] = q.y; v[8 * kk + 4 * e4 + 2] = q.z; v[8 * kk + 4 * e4 + 3] = q.w;
        }
      gT0 = pb_u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      gT1 = pb_u32x4{pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    const f2 nl2e = f2{-1.4426950408889634f, -1.4426950408889634f};

    // one producer iteration: Z(s+1) accumulates into zw (fragments hand-issued one chunk ahead, two register sets, each
    // re-loaded only BEHIND the following chunk's MFMAs); E(s) reads zr (written during the previous iteration)
    auto iteration = [&](auto z_c, auto e_c, int s, f32x16_t& zr, f32x16_t& zw, auto pre) {
      constexpr bool DOZ = decltype(z_c)::value, DOE = decltype(e_c)::value;
      const uint32_t zs = ring + ((s + 1) & 3) * HALF_BYTES + half * 512;
      const uint32_t za0 = zs + ((r32 ^ (4 * half)) << 4), za1 = zs + ((r32 ^ (4 * half + 8)) << 4);   // even / odd fragments
      pb_u32x4 fa[PB_MAXC], fb[PB_MAXC];
      // (defined without an instruction: the "+v" ties in `landed` want a value, any value, in the sets a short chunk leaves unused)
#pragma unroll
      for (int i = 0; i < PB_MAXC; ++i) { asm volatile("" : "=v"(fa[i])); asm volatile("" : "=v"(fb[i])); }
      if constexpr (DOZ) {
        // The accumulator of Z(s+1) starts at b1 + m instead of 0: the first-layer bias of the lane's hidden unit and, with the K12
        // dropout, the element's f16 addend (0 = keep, -30000 = drop; written by the consumer waves one iteration ago) - the 16
        // moves that zeroed it become 16 v_fma_mix_f32 and the E phase needs neither the bias add nor the mask.  Adding -30000 to
        // the pre-activation zeroes y = z sigmoid(z) AND SiLU'(z) (so dz).
        const float nb1 = sB1[(s + 1) * 32 + r32];
        if constexpr (DROP) {
          const pb_u32x4* mp = reinterpret_cast<const pb_u32x4*>(sMask + (((((s + 1) & 1) * 4 + grp) * 32 + r32) * 2 + half) * 16);
          const pb_u32x4 mk0 = mp[0], mk1 = mp[1];
          const uint32_t mw[8] = {mk0.x, mk0.y, mk0.z, mk0.w, mk1.x, mk1.y, mk1.z, mk1.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) {     // registers 2i, 2i + 1 = halves (lo, hi) of dword i of the 16 addends
            float lo, hi;
            asm volatile("v_fma_mix_f32 %0, %2, 1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mix_f32 %1, %2, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                         : "=&v"(lo), "=&v"(hi) : "v"(mw[i]), "v"(nb1));
            zw[2 * i] = lo; zw[2 * i + 1] = hi;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) zw[r] = nb1;
        }
        asm volatile("" : "+v"(zw));   // opaque: a srcC the compiler can see through lets it overlap the chain's destination with its B operand
      }
      uint2 cw2 = make_uint2(0u, 0u);
      if constexpr (DOE) cw2 = sW2p[s * 32 + r32];
      // dz tile of this slab: element (pair row R, unit r32) is one bf16 at byte 2 r32 of the row, its 16-byte slot XORed with
      // (R >> 2) & 3 = (rowc >> 2) + half; rowc >> 2 is even for the rows of registers 0-7 / 16-.. and takes two values mod 4
      const uint32_t tbase = lds_addr(sT) + ((s & 1) * 4 + grp) * 2048 + half * 256 + ((2 * r32) & 15);
      const uint32_t tA = tbase + ((((r32 >> 3) ^ half)) << 4), tB = tbase + ((((r32 >> 3) ^ half ^ 2)) << 4);
      // dy[pair, hid] = sum_c g[pair, c] W2[c, hid] on the matrix cores: B operand = this lane's column of W2 (k = class:
      // lanes 0-31 hold (w0, w1, w2, 0 ...), lanes 32-63 the zero half), same accumulator layout as z
      f32x16_t dy;
#pragma unroll
      for (int r = 0; r < 16; ++r) dy[r] = 0.f;
      float sbx = 0.f, sby = 0.f;
      float yv[16];
      auto issue = [&](auto jc, pb_u32x4 (&d_)[PB_MAXC]) {
        constexpr int J = decltype(jc)::value;
        constexpr int C = pb_chunk_count<KS>(J), F0 = pb_chunk_first<KS>(J);
        static_assert(C <= PB_MAXC, "chunk too large");
        if constexpr (DOZ && !(PB_ABLATE & 64)) {
          if constexpr (C > 0) pb_dsr<(F0 + 0) * 1024>(d_[0], ((F0 + 0) & 1) ? za1 : za0);
          if constexpr (C > 1) pb_dsr<(F0 + 1) * 1024>(d_[1], ((F0 + 1) & 1) ? za1 : za0);
          if constexpr (C > 2) pb_dsr<(F0 + 2) * 1024>(d_[2], ((F0 + 2) & 1) ? za1 : za0);
        }
      };
      auto landed = [&](pb_u32x4 (&d_)[PB_MAXC], auto in_chunk) {
#ifdef PB_PROF
        // d[3] of the debug buffer: ticks the producer spends in these waits (fragment reads of the next chunk + its own tile stores)
        const unsigned long long tl0 = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
#endif
        // the next chunk's fragments have landed; the two dz tile stores this chunk just issued (younger, and the LDS completes a wave's
        // operations in order) may still be in flight: waiting for them too cost the producers a store latency per chunk
        asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(d_[0]), "+v"(d_[1]), "+v"(d_[2]) : [n] "n"((DOE && decltype(in_chunk)::value) ? 2 : 0) : "memory");
#ifdef PB_PROF
        if (p.dbg) { const unsigned long long tl1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_land += tl1 - tl0; }
#endif
        __builtin_amdgcn_sched_barrier(0);
      };
      auto chunk = [&](auto jc, pb_u32x4 (&cur)[PB_MAXC], pb_u32x4 (&nxt)[PB_MAXC]) {
        constexpr int J = decltype(jc)::value;
        constexpr int C = pb_chunk_count<KS>(J), F0 = pb_chunk_first<KS>(J);
        if constexpr (DOZ && !(PB_ABLATE & 4)) {
          if constexpr (C > 0) pb_mma(xf[F0 + 0], cur[0], zw);
          if constexpr (C > 1) pb_mma(xf[F0 + 1], cur[1], zw);
          if constexpr (C > 2) pb_mma(xf[F0 + 2], cur[2], zw);
        }
        if constexpr (J + 1 < 8) issue(std::integral_constant<int, J + 1>{}, nxt);
        if constexpr (DOE) {
          constexpr int r0 = 2 * J, rowc = (r0 & 3) + 8 * (r0 >> 2);
          const f2 zz = f2{zr[r0], zr[r0 + 1]};       // bias and dropout addend are in the accumulator since its first MFMA
          const f2 t = zz * nl2e;
          const f2 den = f2{__builtin_amdgcn_exp2f(t.x) + 1.f, __builtin_amdgcn_exp2f(t.y) + 1.f};
          const f2 sg = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
          const f2 y = zz * sg;
          // SiLU'(z) = sg (1 + z (1 - sg)) = (sg + y) - y sg
          const f2 dzv = f2{dy[r0], dy[r0 + 1]} * fma2(f2{-y.x, -y.y}, sg, sg + y);
          yv[r0] = y.x; yv[r0 + 1] = y.y;
          if constexpr (J == 0) { sbx = dzv.x; sby = dzv.y; } else { sbx += dzv.x; sby += dzv.y; }
          // rows rowc, rowc + 1 (+ 4 half) of the lane's unit: one conversion, two 16-bit stores (the halves of one register); the
          // form with one 32-bit store per lane pair cost four more VALU instructions per two elements (exchange + selects)
          const uint32_t packed = pack_bf16x2(dzv.x, dzv.y);
          asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3"
                       :: "v"(((rowc >> 2) & 2) ? tB : tA), "v"(packed), "n"(rowc * 64), "n"(rowc * 64 + 64) : "memory");
        }
        if constexpr (J + 1 < 8) landed(nxt, std::true_type{});
      };
      // the first chunk's fragments are in LDS since the barrier: their read latency runs under the LDS-DMA issue block,
      // the column-sum flush and the dlogits staging (`pre`)
      issue(std::integral_constant<int, 0>{}, fa);
      pre();                          // (may stage the next head's dlogits: gA / gT change here)
      if constexpr (DOE) {
        const pb_u32x4 w2f = pb_u32x4{half ? 0u : cw2.x, half ? 0u : cw2.y, 0u, 0u};
        pb_mma(gA, w2f, dy);
      }
      landed(fa, std::false_type{});
      chunk(std::integral_constant<int, 0>{}, fa, fb);
      chunk(std::integral_constant<int, 1>{}, fb, fa);
      chunk(std::integral_constant<int, 2>{}, fa, fb);
      chunk(std::integral_constant<int, 3>{}, fb, fa);
      chunk(std::integral_constant<int, 4>{}, fa, fb);
      chunk(std::integral_constant<int, 5>{}, fb, fa);
      chunk(std::integral_constant<int, 6>{}, fa, fb);
      chunk(std::integral_constant<int, 7>{}, fb, fa);
      if constexpr (DOE) {
        // dW2 sums out[c, hid] = sum_pair g[pair, c] y[pair, hid]: y as the B operand straight from its accumulator-layout
        // registers (k = pairs in the order gT was built for); rows 0..2 of the result = registers 0..2 of lanes 0-31
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const pb_u32x4 y0 = pb_u32x4{pack_bf16x2(yv[0], yv[1]), pack_bf16x2(yv[2], yv[3]), pack_bf16x2(yv[4], yv[5]), pack_bf16x2(yv[6], yv[7])};
        const pb_u32x4 y1 = pb_u32x4{pack_bf16x2(yv[8], yv[9]), pack_bf16x2(yv[10], yv[11]), pack_bf16x2(yv[12], yv[13]), pack_bf16x2(yv[14], yv[15])};
        pb_mma(gT0, y0, acc);
        pb_mma(gT1, y1, acc);
        float sbt = sbx + sby;
        sbt += __shfl_xor(sbt, 32);
        const float ys = DROP ? p.drop_scale : 1.f;     // dW2 = sum g (y m) / (1 - p)
        if (lane < 32) sPart[(s & 1) * (4 * 32) + grp * 32 + lane] = make_float4(acc[0] * ys, acc[1] * ys, acc[2] * ys, sbt);
      }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    f32x16_t z0, z1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { z0[r] = 0.f; z1[r] = 0.f; }
    __syncthreads();                                          // sCol visible (matches the consumers' barrier)
    // slab s lives in z0 when s is even, in z1 when odd; nslab >= 2 (launcher)
    auto pre = [&](int s) {
      return [&, s]() {
        dma_iter(s);
        if (s >= 1) flush(s - 1);
        if (s >= 0 && s < nslab && s % spb == 0) stage_g(s / spb);
      };
    };
    top(); iteration(yes{}, no{}, -1, z1, z0, pre(-1));
    int s = 0;
    for (; s + 2 < nslab; s += 2) {
      top(); iteration(yes{}, yes{}, s, z0, z1, pre(s));
      top(); iteration(yes{}, yes{}, s + 1, z1, z0, pre(s + 1));
    }
    for (; s < nslab; ++s) {                                  // the last one or two slabs: no Z(s+1) for the very last
      top();
      if (s + 1 < nslab) { if (s & 1) iteration(yes{}, yes{}, s, z1, z0, pre(s)); else iteration(yes{}, yes{}, s, z0, z1, pre(s)); }
      else { if (s & 1) iteration(no{}, yes{}, s, z1, z0, pre(s)); else iteration(no{}, yes{}, s, z0, z1, pre(s)); }
    }
    top(); dma_iter(nslab); flush(nslab - 1);                 // iteration nslab: the consumers' U(nslab - 1)
    report();
    __syncthreads();     // the consumers' reduction buffer (the rings) is free
    __syncthreads();     // ... and filled
  } else {
    // ---------------------------------------------------------------- consumer: du, d_a / d_b
    f32x16_t du[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) du[t][r] = 0.f;
    T* dz_row = p.dz + row * ncol + 8 * half;
    const uint32_t drop_key = DROP ? pair_drop_key(p.drop_seed, b) : 0u;
    const uint32_t drop_base = (uint32_t)(mypair * nslab * 2) + (uint32_t)half;
    // this lane's pair as the producers see it: accumulator register r of the lanes of half hp, R = (r & 3) + 8 (r >> 2) + 4 hp
    const int mreg = (r32 & 3) + 4 * (r32 >> 3), mhp = (r32 >> 2) & 1;
    // addends of slab q for the producers' Z(q) (the accumulator's start value): lane = (pair r32 of the group, half) walks the
    // forward's chain of 16 fields = units 8g + 4 half + e and stores one f16 per unit at [unit][half of the pair's register][register]
    auto masks_of = [&](int q) {
      _Float16* mrow = sMask + (((q & 1) * 4 + grp) * 32 * 2 + mhp) * 16 + mreg;
      uint32_t st = pair_drop_seed(drop_key, drop_base + 2u * (uint32_t)q);
      const uint32_t inc = pair_drop_inc(drop_key, drop_base + 2u * (uint32_t)q);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        st = pair_drop_step(st, inc);
        const int unit = 8 * (i >> 2) + 4 * half + (i & 3);
        mrow[unit * 32] = (st >> 16) >= p.drop_thr16 ? (_Float16)0.f : (_Float16)(-30000.f);
      }
    };
    if constexpr (DROP && !(PB_ABLATE & 128)) masks_of(0);    // Z(0) runs in iteration -1
    __syncthreads();                                          // matches the producers' barrier
    for (int s = -1; s <= nslab; ++s) {
      top();
      // Z(s + 2) starts at the top of iteration s + 1: its addends are written during iteration s into the buffer Z(s) read its
      // start values from at the top of iteration s - 1
      if constexpr (DROP && !(PB_ABLATE & 128)) { if (s + 2 < nslab) masks_of(s + 2); }
      if (s < 1) dma_iter(s);
      if (s >= 1) {
        const int u = s - 1;
        const uint32_t ta = lds_addr(sT + ((u & 1) * 4 + grp) * 2048) + r32 * 64;
        // du operand = the slab's z fragments read transposed (ds_read_b64_tr_b16: a 16-lane group reads a [4 hid][16 d] block,
        // each lane 4 consecutive d of one hidden row, and receives 4 consecutive hidden units of one d).  Lane (g = lane >> 4,
        // r = (lane & 15) >> 2, c = lane & 3): fragment ks' = 2 dt + (g & 1), k half hh = g >> 1, source half h = c >> 1,
        // hidden row 16 kk + 8 hh + r (+ 4 for the second read), 8 bytes at (c & 1) * 8 of the swizzled slot.
        const int g_ = lane >> 4, hh_ = g_ >> 1, rr_ = (lane & 15) >> 2, cc_ = lane & 3, h_ = cc_ >> 1;
        const uint32_t ub = ring + (u & 3) * HALF_BYTES + (g_ & 1) * 1024 + ((h_ * 32 + 8 * (hh_ ^ (g_ & 1)) + rr_) << 4) + (cc_ & 1) * 8;
        uint32_t ua0 = ub + (h_ ? 64 : 0), ua1 = ub + (h_ ? 0 : 64);
        typedef __attribute__((ext_vector_type(2))) unsigned int pb_u32x2;
        pb_u32x4 a0, a1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(ta + (((0 + half) ^ t_swz) << 4)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(a1) : "v"(ta + (((2 + half) ^ t_swz) << 4)));
        pb_u32x2 la[PB_MAXC], ha[PB_MAXC], lb[PB_MAXC], hb[PB_MAXC];
#pragma unroll
        for (int i = 0; i < PB_MAXC; ++i) { la[i] = pb_u32x2{0u, 0u}; ha[i] = la[i]; lb[i] = la[i]; hb[i] = la[i]; }
        auto trd = [ua0, ua1](auto off_c, pb_u32x2& lo, pb_u32x2& hi) {
          constexpr int OFF = decltype(off_c)::value;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ua0), "n"(OFF));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ua1), "n"(OFF));
        };
        auto uissue = [&](auto jc, pb_u32x2 (&lo_)[PB_MAXC], pb_u32x2 (&hi_)[PB_MAXC]) {
          constexpr int J = decltype(jc)::value;
          constexpr int C = pb_chunk_count<KS>(J), F0 = pb_chunk_first<KS>(J);
          // fragment f = 2 dt + kk -> byte offset dt * 2048 + kk * 256
          if constexpr (C > 0) trd(std::integral_constant<int, ((F0 + 0) >> 1) * 2048 + ((F0 + 0) & 1) * 256>{}, lo_[0], hi_[0]);
          if constexpr (C > 1) trd(std::integral_constant<int, ((F0 + 1) >> 1) * 2048 + ((F0 + 1) & 1) * 256>{}, lo_[1], hi_[1]);
          if constexpr (C > 2) trd(std::integral_constant<int, ((F0 + 2) >> 1) * 2048 + ((F0 + 2) & 1) * 256>{}, lo_[2], hi_[2]);
        };
        auto ulanded = [&](pb_u32x2 (&lo_)[PB_MAXC], pb_u32x2 (&hi_)[PB_MAXC]) {
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(lo_[0]), "+v"(lo_[1]), "+v"(lo_[2]), "+v"(hi_[0]), "+v"(hi_[1]), "+v"(hi_[2]), "+v"(a0), "+v"(a1) :: "memory");
          __builtin_amdgcn_sched_barrier(0);
        };
        auto uchunk = [&](auto jc, pb_u32x2 (&clo)[PB_MAXC], pb_u32x2 (&chi)[PB_MAXC], pb_u32x2 (&nlo)[PB_MAXC], pb_u32x2 (&nhi)[PB_MAXC]) {
          constexpr int J = decltype(jc)::value;
          constexpr int C = pb_chunk_count<KS>(J), F0 = pb_chunk_first<KS>(J);
          if constexpr (!(PB_ABLATE & 2)) {
            // the 4-register operands are put together AFTER the wait (a copy made before it would copy stale registers)
            if constexpr (C > 0) pb_mma(((F0 + 0) & 1) ? a1 : a0, pb_u32x4{clo[0].x, clo[0].y, chi[0].x, chi[0].y}, du[(F0 + 0) >> 1]);
            if constexpr (C > 1) pb_mma(((F0 + 1) & 1) ? a1 : a0, pb_u32x4{clo[1].x, clo[1].y, chi[1].x, chi[1].y}, du[(F0 + 1) >> 1]);
            if constexpr (C > 2) pb_mma(((F0 + 2) & 1) ? a1 : a0, pb_u32x4{clo[2].x, clo[2].y, chi[2].x, chi[2].y}, du[(F0 + 2) >> 1]);
          }
          if constexpr (J + 1 < 8) { uissue(std::integral_constant<int, J + 1>{}, nlo, nhi); ulanded(nlo, nhi); }
        };
        uissue(std::integral_constant<int, 0>{}, la, ha);
        dma_iter(s);                 // under the latency of the reads just issued
        ulanded(la, ha);
        if constexpr (!(PB_ABLATE & 1)) {
          *reinterpret_cast<pb_u32x4*>(dz_row + u * 32) = a0;
          *reinterpret_cast<pb_u32x4*>(dz_row + u * 32 + 16) = a1;
        }
        uchunk(std::integral_constant<int, 0>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 1>{}, lb, hb, la, ha);
        uchunk(std::integral_constant<int, 2>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 3>{}, lb, hb, la, ha);
        uchunk(std::integral_constant<int, 4>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 5>{}, lb, hb, la, ha);
        uchunk(std::integral_constant<int, 6>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 7>{}, lb, hb, la, ha);
      }
    }
    report();
    __syncthreads();     // every wave is done with the rings
    // ---- du * SiLU'(a_i + b_j): sums over j stay in the wave, sums over i meet in LDS (the rings are dead) ----
    float* red = reinterpret_cast<float*>(smem);                      // [4][NDT][16][32]
    const int ia = min(i0, N - 1), ib = min(i0 + 1, N - 1);
    float* pa = p.part_a + (((int64_t)b * p.ntiles + blockIdx.x) * PB_TI + 2 * grp + half) * D;
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      const int d = 32 * t + r32;
      const float a_lo = bf16_to_f32(abd[(int64_t)ia * 2 * D + d]), a_hi = bf16_to_f32(abd[(int64_t)ib * 2 * D + d]);
      float bj[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = min(j0 + 8 * (q >> 2) + 4 * half + (q & 3), N - 1);
        bj[q] = bf16_to_f32(abd[(int64_t)j * 2 * D + D + d]);
      }
      float sa0 = 0.f, sa1 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v0 = du[t][q] * silu_grad_f(a_lo + bj[q]);
        const float v1 = du[t][8 + q] * silu_grad_f(a_hi + bj[q]);
        sa0 += v0; sa1 += v1;
        red[((grp * NDT + t) * 16 + 8 * (q >> 2) + 4 * half + (q & 3)) * 32 + r32] = v0 + v1;
      }
      sa0 += __shfl_xor(sa0, 32);
      sa1 += __shfl_xor(sa1, 32);
      pa[d] = half ? sa1 : sa0;
    }
    __syncthreads();
    float* pb = p.part_b + ((int64_t)b * p.ntiles + blockIdx.x) * PB_TJ * D;
    for (int e = tid - 256; e < NDT * 16 * 32; e += 256) {
      const int c = e & 31, jl = (e >> 5) & 15, t = e >> 9;
      float v = red[e];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += red[w * NDT * 512 + e];
      pb[(int64_t)jl * D + 32 * t + c] = v;
    }
  }
}

// ================================================================================================
// One-wave form (D = 512, KS = 32): the du tile of a wave's 32 pairs is 32 x 512 fp32 = 256 accumulator registers, a wave's
// whole share of a SIMD's register file at two waves per SIMD.  Here ONE wave per SIMD (four per workgroup, 512 registers
// each) does both roles: du lives in the 256 AGPRs (its MFMAs are inline asm with "+a" operands: the compiler would
// otherwise pick the register class of every MFMA of the kernel at once; this file is built with -amdgpu-mfma-vgpr-form so
// that the builtin MFMAs - z, dy, the dW2 sums - keep VGPR results the VALU can read), x / z / everything else in the 256
// VGPRs.  Per slab s:   Z(s): 32 MFMAs into z (pure matrix phase, operand reads three chunks deep),
//                       E(s) (VALU: dz from z) interleaved with U(s-1) (du += dz(s-1) W1, the matrix cores): the VALU work of
//                       one slab hides behind the MFMAs of the previous slab's du product,
// same data layout, weight packing, LDS tiles and dropout stream as the wave-specialised kernel above (its rows, partial
// sums and workspace are interchangeable).  The weight ring has three slots (slab s+1 lands while Z(s) / U(s-1) read).
// ================================================================================================
template <int N, int I = 0, typename F> __device__ __forceinline__ void pb_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); pb_static_for<N, I + 1>(f); }
}
typedef __attribute__((ext_vector_type(2))) unsigned int pb_u32x2;
template <int OFF> __device__ __forceinline__ void pb_trd(pb_u32x2& d, uint32_t a) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void pb_dsw16(uint32_t a, uint32_t v) {
  asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(OFF) : "memory");
}
// ("+v": the destination counts as live before the read, so the register allocator cannot fold two rotating sets into one)
template <int OFF> __device__ __forceinline__ void pb_trd_tied(pb_u32x2& d, uint32_t a) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "+v"(d) : "v"(a), "n"(OFF));
}
template <int N> __device__ __forceinline__ void pb_lgkm(pb_u32x2& a, pb_u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void pb_lgkm4(pb_u32x4& a, pb_u32x4& b, pb_u32x2& c, pb_u32x2& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
constexpr int pb_uoff(int f) { return (f >> 1) * 2048 + (f & 1) * 256; }
constexpr int pb_mask_steps(int nz, int j) { return j * 16 / nz; }   // mask steps done before chunk j of nz
// registers an inline-asm read fills are pinned behind the wait that covers it
template <int C, typename V> __device__ __forceinline__ void pb_pin(V (&d)[C]) {
#pragma unroll
  for (int i = 0; i < C; ++i) asm volatile("" : "+v"(d[i]));
}
__device__ __forceinline__ void pb_lgkm0(pb_u32x4& a, pb_u32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) :: "memory"); }
__device__ __forceinline__ void pb_mma_v(const uint4& a, const pb_u32x4& b, f32x16_t& acc) {
  const pb_u32x4 av = pb_u32x4{a.x, a.y, a.z, a.w};
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(b));
}
__device__ __forceinline__ void pb_mma_v0(const uint4& a, const pb_u32x4& b, f32x16_t& acc) {   // acc = a b (no accumulator read)
  const pb_u32x4 av = pb_u32x4{a.x, a.y, a.z, a.w};
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(av), "v"(b));
}
// one LDS-DMA piece, uniform 64-bit base + per-lane offset; M0 is this kernel's to clobber (nothing else in it reads M0)
__device__ __forceinline__ void pb_dma_piece(uint32_t voff_lane, const char* base_uniform, uint32_t lds_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff_lane), "s"(base_uniform), "s"(lds_uniform) : "memory", "m0");
}
__device__ __forceinline__ void pb_mma_acc(const pb_u32x4& a, const pb_u32x4& b, f32x16_t& acc) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// ================================================================================================
// Saved-activation form (bf16, D = 384; peneo_pair_bwd_saved).  The forward (pair_heads_fwd_hand_kernel<.., SAVE>) walked the pairs
// in THESE blocks and left, per group and 32-unit slab, the pre-activations z (f16, a dropped unit's as -30000) as a [32 pairs][32 units]
// tile, and x = SiLU(a_i + b_j) in block-row order.  What is left of the kernel above:
//   producers (waves 0-3, group g):  dy = g W2 (one MFMA), y and SiLU'(z) from the saved z (the E phase of the kernel above, z read as
//       the f16 half it was saved as), dz = dy SiLU'(z), the dz tile for the consumers, the db1 sums, dW2 sums = g^T y (two MFMAs) -
//       no first-layer product (24 of the 51 MFMAs per slab and their 24 KiB of fragment reads), no dropout chain, no x
//   consumers (waves 4-7):           du += dz W1 (24 MFMAs per slab), dz rows to memory, the column-sum flush, at the end d_a / d_b
// A group's record (2 KiB) arrives by LDS-DMA three slabs ahead through a ring of four; the producer reads its accumulator-layout
// values with ds_read_b64_tr_b16 (16 lanes x 4 consecutive units of one pair row in, 4 consecutive pairs of one unit out: lane & 31 =
// unit, registers = pairs, the layout dy has).  The weight ring holds three slabs (only the consumers read it).
// vm counter: the producers only ever have record pieces in flight (2 per slab: all loads, in order -> vmcnt(4) = "slab s has
// landed"); the consumers have weight pieces (6 per wave and slab), the dz row stores and the flush atomics: loads and writes may
// retire out of order with each other, so the count that proves "the 6 pieces of slab s-1 are in" is 6.
// ================================================================================================
#ifndef PSV_NR
#define PSV_NR 4        // record ring slots (requests run PSV_NR - 1 slabs ahead)
#endif
#ifndef PSV_SPREAD
#define PSV_SPREAD 0    // 1: a consumer's weight pieces go out one per chunk instead of all behind the first reads
#endif
template <int KS, bool DROP>
__global__ __launch_bounds__(PW_WAVES * 64, 2) void pair_bwd_sv_kernel(PairBwdParams p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HALF_BYTES = KS * 1024;
  constexpr int NDT = KS / 2;
  constexpr int PPC = KS / 4;                                // weight pieces per consumer wave and slab
  static_assert(KS % 4 == 0, "weight pieces are dealt to four consumer waves");
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int grp = wave & 3;
  const int D = p.D, N = p.N, nh = p.a.num_heads, ncol = nh * D;
  char* sA = smem;                                                          // [3][HALF_BYTES] weight ring: slab s in slot s % 3
  char* sR = smem + 3 * HALF_BYTES;                                         // [PSV_NR][4 groups][PB_REC_BYTES] record ring: slab s in slot s % PSV_NR
  uint2* sW2p = reinterpret_cast<uint2*>(sR + PSV_NR * 4 * PB_REC_BYTES);   // [ncol]: the column's W2 rows as bf16 (w0, w1 | w2, 0)
  float4* sG = reinterpret_cast<float4*>(sW2p + ncol);                      // [4][32]
  float4* sPart = sG + 4 * 32;                                              // [2][4][32]
  char* sT = reinterpret_cast<char*>(sPart + 2 * 4 * 32);                   // [2][4][32 rows][64 B]

  int ti = 0;
  {
    const int nti = pb_row_tiles(N);
    while (ti + 1 < nti && pb_tiles_before(ti + 1, N) <= (int)blockIdx.x) ++ti;
  }
  const int tj = (ti >> 1) + ((int)blockIdx.x - pb_tiles_before(ti, N));
  const int b = blockIdx.y;
  const int i0 = ti * PB_TI + 2 * grp, j0 = tj * PB_TJ;
  const int pi = i0 + (r32 >> 4), pj = j0 + (r32 & 15);
  const bool pair_ok = pi < N && pj < N && pi <= pj;
  const int ci = min(pi, N - 1), cj = min(pj, N - 1);
  const int64_t mypair = pair_row_start(ci, N) + (cj - ci);
  const int64_t rows_per_doc = (int64_t)p.ntiles * PB_ROWS;
  const int64_t row = (int64_t)b * rows_per_doc + (int64_t)blockIdx.x * PB_ROWS + grp * 32 + r32;
  const int nslab = ncol / 32, spb = D / 32;
  const T* abd = p.ab + (int64_t)b * N * 2 * D;

  for (int n = tid; n < ncol; n += PW_WAVES * 64) {
    const int h = n / D, k = n - h * D, Cn = p.a.classes[h];
    const float ds = DROP ? p.drop_scale : 1.f;     // dy = g W2 / (1 - p): the scale of the kept units rides on the W2 rows
    sW2p[n] = make_uint2(pack_bf16x2(ds * p.a.w2[h][k], Cn > 1 ? ds * p.a.w2[h][(int64_t)D + k] : 0.f),
                         pack_bf16x2(Cn > 2 ? ds * p.a.w2[h][(int64_t)2 * D + k] : 0.f, 0.f));
  }
  const uint32_t ring = lds_addr(sA), recs = lds_addr(sR);
  auto top = [&](auto n_c) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(decltype(n_c)::value) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (producer) {
    // ---------------------------------------------------------------- producer: dz, db1 / dW2 sums
    // scale_h * dlogits_h of the lane's pair for every head, read once (lanes 0-31; zero for pairs outside the triangle)
    float gl[PENEO_MAX_HEADS][3];
#pragma unroll
    for (int h = 0; h < PENEO_MAX_HEADS; ++h) {
      gl[h][0] = 0.f; gl[h][1] = 0.f; gl[h][2] = 0.f;
      if (h < nh && lane < 32 && pair_ok) {
        const int Cn = p.a.classes[h];
        const float sc = p.a.scale[h];
        const float* dl = p.a.dlogits[h] + ((int64_t)b * p.P + mypair) * Cn;
        gl[h][0] = dl[0] * sc;
        if (Cn > 1) gl[h][1] = dl[1] * sc;
        if (Cn > 2) gl[h][2] = dl[2] * sc;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // from here on this wave's vm counter only sees its record pieces
    // record of slab s: 2 pieces of 1 KiB into slot s & 3
    const char* rsrc = p.act + ((((int64_t)b * p.ntiles + blockIdx.x) * nslab * 4 + grp) * PB_REC_BYTES) + lane * 16;
    auto request = [&](int s, int slot) {
      const char* src = rsrc + (int64_t)min(s, nslab - 1) * (4 * PB_REC_BYTES);     // (past the end: the last slab again, into a free slot)
      const uint32_t dst = __builtin_amdgcn_readfirstlane(recs + (slot * 4 + grp) * PB_REC_BYTES);
      if constexpr (!(PB_ABLATE & 16)) { lds_dma_1k<0>(src, dst); lds_dma_1k<1024>(src, dst); }
    };
#pragma unroll
    for (int k = 0; k < PSV_NR - 1; ++k) request(k, k);
    // MFMA operands of the current head (see the kernel above): gA = A operand of dy, gT0 / gT1 = A operands of the dW2 sums
    pb_u32x4 gA = pb_u32x4{0u, 0u, 0u, 0u}, gT0 = gA, gT1 = gA;
    auto stage_g = [&](int h) {
      float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
      for (int k = 0; k < PENEO_MAX_HEADS; ++k) { gx = (k == h) ? gl[k][0] : gx; gy = (k == h) ? gl[k][1] : gy; gz = (k == h) ? gl[k][2] : gz; }
      gA = pb_u32x4{pack_bf16x2(gx, gy), pack_bf16x2(gz, 0.f), 0u, 0u};         // lanes >= 32 (k = 8..15) and bad pairs: zeros
      float* gt = reinterpret_cast<float*>(sG + grp * 32);                        // [3][32] floats: g_c of pair
      if (lane < 32) { gt[lane] = gx; gt[32 + lane] = gy; gt[64 + lane] = gz; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int c = r32;                                                          // class row of this lane in gT
      float v[16];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
          const float4 q = c < 3 ? *reinterpret_cast<const float4*>(gt + c * 32 + 16 * kk + 8 * e4 + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[8 * kk + 4 * e4 + 0] = q.x; v[8 * kk + 4 * e4 + 1] = q.y; v[8 * kk + 4 * e4 + 2] = q.z; v[8 * kk + 4 * e4 + 3] = q.w;
        }
      gT0 = pb_u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      gT1 = pb_u32x4{pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    // transposing reads of a record (two 1 KiB halves hg of 16 units, 32-byte rows, the second half's rows at pair ^ 4: common.h): the
    // 16-lane group (hg = (lane >> 4) & 1, half) supplies, for quad qd, rows 8 qd + 4 half + r (r = (lane & 15) >> 2), units 16 hg + 4 c ..
    // + 3 (c = lane & 3) and receives pairs 8 qd + 4 half + 0..3 of unit 16 hg + (lane & 15) = lane & 31
    const uint32_t tr_l = (uint32_t)(((lane >> 4) & 1) * 1024 + ((4 * (half ^ ((lane >> 4) & 1)) + ((lane & 15) >> 2)) * 32) + (lane & 3) * 8);
    const uint32_t tbase = lds_addr(sT) + grp * 2048 + half * 256 + ((2 * r32) & 15);
    const uint32_t tA0 = tbase + ((((r32 >> 3) ^ half)) << 4), tB0 = tbase + ((((r32 >> 3) ^ half ^ 2)) << 4);
    const float nl2e = -1.4426950408889634f;
    __syncthreads();                                          // sW2p visible (matches the consumers' barrier)
    int rslot = 0;                                            // s % PSV_NR
    for (int s = 0; s < nslab; ++s) {
      top(std::integral_constant<int, 2 * (PSV_NR - 2)>{});   // record s has landed (records s + 1 .. s + PSV_NR - 2 may be on their way)
      request(s + PSV_NR - 1, rslot == 0 ? PSV_NR - 1 : rslot - 1);   // slot (s - 1) % PSV_NR: last read by E(s - 1)
      if (s % spb == 0) stage_g(s / spb);
      const uint2 cw2 = sW2p[s * 32 + r32];
      const uint32_t ra = recs + (rslot * 4 + grp) * PB_REC_BYTES + tr_l;
      rslot = rslot == PSV_NR - 1 ? 0 : rslot + 1;
      pb_u32x2 zv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(zv[i]));
      pb_trd<0>(zv[0], ra); pb_trd<256>(zv[1], ra); pb_trd<512>(zv[2], ra); pb_trd<768>(zv[3], ra);
      f32x16_t dy, acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dy[r] = 0.f; acc[r] = 0.f; }
      {
        const pb_u32x4 w2f = pb_u32x4{half ? 0u : cw2.x, half ? 0u : cw2.y, 0u, 0u};
        pb_mma(gA, w2f, dy);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zv[0]), "+v"(zv[1]), "+v"(zv[2]), "+v"(zv[3]) :: "memory");
      const uint32_t tsel[2] = {tA0 + (s & 1) * (4 * 2048), tB0 + (s & 1) * (4 * 2048)};
      float sbx = 0.f, sby = 0.f;
      float yv[16];
      pb_static_for<8>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        constexpr int r0 = 2 * J, rowc = (r0 & 3) + 8 * (r0 >> 2);
        const uint32_t zw2 = (J & 1) ? zv[J >> 1].y : zv[J >> 1].x;  // registers r0, r0 + 1 = halves (lo, hi) of dword J of the 16 saved values
        // (plain C++, compiled to v_fma_mix_f32: the compiler then also places the wait states between the dy MFMA and its first reader -
        // an inline-asm reader gets none and read zeros for rows 0, 1)
        const f16x2_t zh = __builtin_bit_cast(f16x2_t, zw2);
        const float t0 = __builtin_fmaf((float)zh[0], nl2e, 0.f), t1 = __builtin_fmaf((float)zh[1], nl2e, 0.f);
        const float sg0 = (PB_ABLATE & 8) ? t0 : __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(t0) + 1.f), sg1 = (PB_ABLATE & 8) ? t1 : __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(t1) + 1.f);
        const float y0 = __builtin_fmaf((float)zh[0], sg0, 0.f), y1 = __builtin_fmaf((float)zh[1], sg1, 0.f);
        // SiLU'(z) = sg (1 + z (1 - sg)) = (sg + y) - y sg
        const float d0 = dy[r0] * fmaf(-y0, sg0, sg0 + y0), d1 = dy[r0 + 1] * fmaf(-y1, sg1, sg1 + y1);
        yv[r0] = y0; yv[r0 + 1] = y1;
        if constexpr (J == 0) { sbx = d0; sby = d1; } else { sbx += d0; sby += d1; }
        const uint32_t packed = pack_bf16x2(d0, d1);
        const uint32_t taddr = tsel[((rowc >> 2) & 2) ? 1 : 0];      // (a plain use: asm operands alone do not capture in a generic lambda)
        asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3"
                     :: "v"(taddr), "v"(packed), "n"(rowc * 64), "n"(rowc * 64 + 64) : "memory");
      });
      // dW2 sums out[c, unit] = sum_pair g[pair, c] y[pair, unit]: y as the B operand straight from its accumulator-layout registers
      const pb_u32x4 yb0 = pb_u32x4{pack_bf16x2(yv[0], yv[1]), pack_bf16x2(yv[2], yv[3]), pack_bf16x2(yv[4], yv[5]), pack_bf16x2(yv[6], yv[7])};
      const pb_u32x4 yb1 = pb_u32x4{pack_bf16x2(yv[8], yv[9]), pack_bf16x2(yv[10], yv[11]), pack_bf16x2(yv[12], yv[13]), pack_bf16x2(yv[14], yv[15])};
      pb_mma(gT0, yb0, acc);
      pb_mma(gT1, yb1, acc);
      float sbt = sbx + sby;
      sbt += __shfl_xor(sbt, 32);
      const float ys = DROP ? p.drop_scale : 1.f;     // dW2 = sum g (y m) / (1 - p)
      if (lane < 32) sPart[(s & 1) * (4 * 32) + grp * 32 + lane] = make_float4(acc[0] * ys, acc[1] * ys, acc[2] * ys, sbt);
    }
    top(std::integral_constant<int, 0>{});                    // iteration nslab: the consumers' U(nslab - 1); every piece of this wave has landed
    __syncthreads();     // the consumers' reduction buffer (the rings) is free
    __syncthreads();     // ... and filled
  } else {
    // ---------------------------------------------------------------- consumer: du, d_a / d_b
    f32x16_t du[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) du[t][r] = 0.f;
    T* dz_row = p.dz + row * ncol + 8 * half;
    const char* wbase = reinterpret_cast<const char*>(p.wp) + lane * 16;
    // slab's pieces (the z-fragment image of the packed weights, third part) -> ring slot; wave w carries pieces w - 4, w, w + 4, ...
    auto wrequest = [&](int slab, int slot) {
      const char* src = wbase + (int64_t)min(slab, nslab - 1) * (3 * HALF_BYTES) + 2 * HALF_BYTES;
#pragma unroll
      for (int k = 0; k < PPC; ++k) {
        const int q = (wave - 4) + k * 4;
        if constexpr (!(PB_ABLATE & 32)) lds_dma_1k<0>(src + q * 1024, __builtin_amdgcn_readfirstlane(ring + slot * HALF_BYTES + q * 1024));
      }
    };
    auto wpiece = [&](int slab, int slot, int k) {
      const int q = (wave - 4) + k * 4;
      const char* src = wbase + (int64_t)min(slab, nslab - 1) * (3 * HALF_BYTES) + 2 * HALF_BYTES;
      if constexpr (!(PB_ABLATE & 32)) lds_dma_1k<0>(src + q * 1024, __builtin_amdgcn_readfirstlane(ring + slot * HALF_BYTES + q * 1024));
    };
    wrequest(0, 0);
    wrequest(1, 1);
    float* wslot = p.ws + (int64_t)(blockIdx.x % PB_SLOTS) * 4 * ncol;
    const int t_swz = (r32 >> 2) & 3;
    __syncthreads();                                          // matches the producers' barrier
    int slot = 0;                                             // s % 3
    for (int s = 0; s <= nslab; ++s) {
      // the 6 pieces of slab s - 1 are in: of this wave's operations only its newest 6 may be outstanding (header)
      if (s < nslab) top(std::integral_constant<int, 6>{}); else top(std::integral_constant<int, 0>{});
      if (s >= 1) {
        const int u = s - 1, uslot = slot == 0 ? 2 : slot - 1;
        if (wave - 4 == (u & 3) && lane < 32) {               // the column sums of slab u: four groups -> one row of the workspace
          const float4* src = sPart + (u & 1) * (4 * 32) + lane;
          float4 t = src[0];
#pragma unroll
          for (int w = 1; w < 4; ++w) { const float4 v = src[w * 32]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
          float* dst = wslot + u * 32 + lane;
          atomicAdd(dst, t.x); atomicAdd(dst + ncol, t.y); atomicAdd(dst + 2 * (int64_t)ncol, t.z); atomicAdd(dst + 3 * (int64_t)ncol, t.w);
        }
        const uint32_t ta = lds_addr(sT + ((u & 1) * 4 + grp) * 2048) + r32 * 64;
        const int g_ = lane >> 4, hh_ = g_ >> 1, rr_ = (lane & 15) >> 2, cc_ = lane & 3, h_ = cc_ >> 1;
        const uint32_t ub = ring + uslot * HALF_BYTES + (g_ & 1) * 1024 + ((h_ * 32 + 8 * (hh_ ^ (g_ & 1)) + rr_) << 4) + (cc_ & 1) * 8;
        uint32_t ua0 = ub + (h_ ? 64 : 0), ua1 = ub + (h_ ? 0 : 64);
        pb_u32x4 a0, a1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(ta + (((0 + half) ^ t_swz) << 4)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(a1) : "v"(ta + (((2 + half) ^ t_swz) << 4)));
        pb_u32x2 la[PB_MAXC], ha[PB_MAXC], lb[PB_MAXC], hb[PB_MAXC];
#pragma unroll
        for (int i = 0; i < PB_MAXC; ++i) { asm volatile("" : "=v"(la[i])); asm volatile("" : "=v"(ha[i])); asm volatile("" : "=v"(lb[i])); asm volatile("" : "=v"(hb[i])); }
        auto trd = [ua0, ua1](auto off_c, pb_u32x2& lo, pb_u32x2& hi) {
          constexpr int OFF = decltype(off_c)::value;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ua0), "n"(OFF));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ua1), "n"(OFF));
        };
        auto uissue = [&](auto jc, pb_u32x2 (&lo_)[PB_MAXC], pb_u32x2 (&hi_)[PB_MAXC]) {
          constexpr int J = decltype(jc)::value;
          constexpr int C = pb_chunk_count<KS>(J), F0 = pb_chunk_first<KS>(J);
          if constexpr (C > 0) trd(std::integral_constant<int, ((F0 + 0) >> 1) * 2048 + ((F0 + 0) & 1) * 256>{}, lo_[0], hi_[0]);
          if constexpr (C > 1) trd(std::integral_constant<int, ((F0 + 1) >> 1) * 2048 + ((F0 + 1) & 1) * 256>{}, lo_[1], hi_[1]);
          if constexpr (C > 2) trd(std::integral_constant<int, ((F0 + 2) >> 1) * 2048 + ((F0 + 2) & 1) * 256>{}, lo_[2], hi_[2]);
        };
        auto ulanded = [&](pb_u32x2 (&lo_)[PB_MAXC], pb_u32x2 (&hi_)[PB_MAXC]) {
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(lo_[0]), "+v"(lo_[1]), "+v"(lo_[2]), "+v"(hi_[0]), "+v"(hi_[1]), "+v"(hi_[2]), "+v"(a0), "+v"(a1) :: "memory");
          __builtin_amdgcn_sched_barrier(0);
        };
        auto uchunk = [&](auto jc, pb_u32x2 (&clo)[PB_MAXC], pb_u32x2 (&chi)[PB_MAXC], pb_u32x2 (&nlo)[PB_MAXC], pb_u32x2 (&nhi)[PB_MAXC]) {
          constexpr int J = decltype(jc)::value;
          constexpr int C = pb_chunk_count<KS>(J), F0 = pb_chunk_first<KS>(J);
          if constexpr (!(PB_ABLATE & 2)) {
          if constexpr (C > 0) pb_mma(((F0 + 0) & 1) ? a1 : a0, pb_u32x4{clo[0].x, clo[0].y, chi[0].x, chi[0].y}, du[(F0 + 0) >> 1]);
          if constexpr (C > 1) pb_mma(((F0 + 1) & 1) ? a1 : a0, pb_u32x4{clo[1].x, clo[1].y, chi[1].x, chi[1].y}, du[(F0 + 1) >> 1]);
          if constexpr (C > 2) pb_mma(((F0 + 2) & 1) ? a1 : a0, pb_u32x4{clo[2].x, clo[2].y, chi[2].x, chi[2].y}, du[(F0 + 2) >> 1]);
          }
          if constexpr (J + 1 < 8) uissue(std::integral_constant<int, J + 1>{}, nlo, nhi);
          if constexpr (PSV_SPREAD && J < PPC) { if (s < nslab) wpiece(s + 1, slot == 2 ? 0 : slot + 1, J); }   // under the latency of the reads just issued
          if constexpr (J + 1 < 8) ulanded(nlo, nhi);
        };
        uissue(std::integral_constant<int, 0>{}, la, ha);
        if constexpr (!PSV_SPREAD) { if (s < nslab) wrequest(s + 1, slot == 2 ? 0 : slot + 1); }   // (s + 1) % 3: last read by U(s - 2)
        ulanded(la, ha);
        if constexpr (!(PB_ABLATE & 1)) {
          *reinterpret_cast<pb_u32x4*>(dz_row + u * 32) = a0;
          *reinterpret_cast<pb_u32x4*>(dz_row + u * 32 + 16) = a1;
        }
        uchunk(std::integral_constant<int, 0>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 1>{}, lb, hb, la, ha);
        uchunk(std::integral_constant<int, 2>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 3>{}, lb, hb, la, ha);
        uchunk(std::integral_constant<int, 4>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 5>{}, lb, hb, la, ha);
        uchunk(std::integral_constant<int, 6>{}, la, ha, lb, hb);
        uchunk(std::integral_constant<int, 7>{}, lb, hb, la, ha);
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
    __syncthreads();     // every wave is done with the rings
    // ---- du * SiLU'(a_i + b_j): sums over j stay in the wave, sums over i meet in LDS (the rings are dead) ----
    float* red = reinterpret_cast<float*>(smem);                      // [4][NDT][16][32]
    const int ia = min(i0, N - 1), ib = min(i0 + 1, N - 1);
    float* pa = p.part_a + (((int64_t)b * p.ntiles + blockIdx.x) * PB_TI + 2 * grp + half) * D;
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      const int d = 32 * t + r32;
      const float a_lo = bf16_to_f32(abd[(int64_t)ia * 2 * D + d]), a_hi = bf16_to_f32(abd[(int64_t)ib * 2 * D + d]);
      float bj[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = min(j0 + 8 * (q >> 2) + 4 * half + (q & 3), N - 1);
        bj[q] = bf16_to_f32(abd[(int64_t)j * 2 * D + D + d]);
      }
      float sa0 = 0.f, sa1 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v0 = du[t][q] * silu_grad_f(a_lo + bj[q]);
        const float v1 = du[t][8 + q] * silu_grad_f(a_hi + bj[q]);
        sa0 += v0; sa1 += v1;
        red[((grp * NDT + t) * 16 + 8 * (q >> 2) + 4 * half + (q & 3)) * 32 + r32] = v0 + v1;
      }
      sa0 += __shfl_xor(sa0, 32);
      sa1 += __shfl_xor(sa1, 32);
      pa[d] = half ? sa1 : sa0;
    }
    __syncthreads();
    float* pb = p.part_b + ((int64_t)b * p.ntiles + blockIdx.x) * PB_TJ * D;
    for (int e = tid - 256; e < NDT * 16 * 32; e += 256) {
      const int c = e & 31, jl = (e >> 5) & 15, t = e >> 9;
      float v = red[e];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += red[w * NDT * 512 + e];
      pb[(int64_t)jl * D + 32 * t + c] = v;
    }
  }
}

template <int KS, bool DROP>
__global__ __launch_bounds__(PB_WAVES * 64, 1) void pair_bwd_one_kernel(PairBwdParams p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HALF_BYTES = KS * 1024;
  constexpr int NDT = KS / 2;
  static_assert(KS == 32, "one-wave form: built and scheduled for D = 512 (16 Z chunks, 16 U sub-chunks, 16 mask steps)");
  constexpr int PPW = KS / PB_WAVES;                         // 1 KiB weight pieces per wave and slab
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave;
  const int D = p.D, N = p.N, nh = p.a.num_heads, ncol = nh * D;
  char* sA = smem;                                                          // [3][HALF_BYTES] weight ring: slab s in slot s % 3
  uint2* sW2p = reinterpret_cast<uint2*>(smem + 3 * HALF_BYTES);            // [ncol]
  float* sB1 = reinterpret_cast<float*>(sW2p + ncol);                       // [ncol]
  float4* sG = reinterpret_cast<float4*>(sB1 + ncol);                       // [4][32]
  float4* sPart = sG + 4 * 32;                                              // [2][4][32]
  char* sT = reinterpret_cast<char*>(sPart + 2 * 4 * 32);                   // [2][4][32 rows][64 B]
  // [4 groups][32 units][80 B: 2 halves x 16 registers of f16, padded so that units u and u + 4 (what the two halves of a
  // wave write in one instruction) sit in different banks]
  char* sMask = sT + 2 * 4 * 2048;
  constexpr int MROW = 80;

  int ti = 0;
  {
    const int nti = pb_row_tiles(N);
    while (ti + 1 < nti && pb_tiles_before(ti + 1, N) <= (int)blockIdx.x) ++ti;
  }
  const int tj = (ti >> 1) + ((int)blockIdx.x - pb_tiles_before(ti, N));
  const int b = blockIdx.y;
  const int i0 = ti * PB_TI + 2 * grp, j0 = tj * PB_TJ;
  const int pi = i0 + (r32 >> 4), pj = j0 + (r32 & 15);
  const bool pair_ok = pi < N && pj < N && pi <= pj;
  const int ci = min(pi, N - 1), cj = min(pj, N - 1);
  const int64_t mypair = pair_row_start(ci, N) + (cj - ci);
  const int64_t rows_per_doc = (int64_t)p.ntiles * PB_ROWS;
  const int64_t row = (int64_t)b * rows_per_doc + (int64_t)blockIdx.x * PB_ROWS + grp * 32 + r32;
  const int nslab = ncol / 32, spb = D / 32;
  const T* abd = p.ab + (int64_t)b * N * 2 * D;

  for (int n = tid; n < ncol; n += PB_WAVES * 64) {
    const int h = n / D, k = n - h * D, Cn = p.a.classes[h];
    const float ds = DROP ? p.drop_scale : 1.f;
    sW2p[n] = make_uint2(pack_bf16x2(ds * p.a.w2[h][k], Cn > 1 ? ds * p.a.w2[h][(int64_t)D + k] : 0.f),
                         pack_bf16x2(Cn > 2 ? ds * p.a.w2[h][(int64_t)2 * D + k] : 0.f, 0.f));
    sB1[n] = p.b1[n];
  }

  const uint32_t ring = lds_addr(sA);
  // weight stream: wave w carries pieces w, w + 4, ... of a slab.  Slab s + 1 is requested piece by piece from inside the Z
  // phase of iteration s (an LDS-DMA instruction holds its issuer for tens to hundreds of cycles once a few are in flight:
  // as a block at the top of the iteration that was 15 % of the kernel; between two chunks of queued MFMAs it is free)
  const char* const wuni = reinterpret_cast<const char*>(p.wp) + 2 * HALF_BYTES + wave * 1024;   // uniform: this wave's first piece of slab 0
  const uint32_t wlane = lane * 16;
  auto dma_piece = [&](int slab, int k) {
    lds_dma_1k_s<0>(wlane, wuni + (int64_t)slab * (3 * HALF_BYTES) + k * (PB_WAVES * 1024),
                    __builtin_amdgcn_readfirstlane(ring + (slab % 3) * HALF_BYTES + (wave + k * PB_WAVES) * 1024));
  };
  auto dma_slab = [&](int slab) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_piece(slab, k);
  };
  dma_slab(0);

  // scalar pairs: the packed forms (v_pk_add / mul / fma_f32) were tried here - exact with one wave per SIMD, but every one of
  // them wants a wait state after a transcendental or before its consumer, and a lone wave pays an issue slot for each s_nop
  // (8.0 against 7.25 ms); two scalar chains interleaved fill those slots with work
  struct f2 {
    float x, y;
    __device__ f2 operator+(const f2& o) const { return f2{x + o.x, y + o.y}; }
    __device__ f2 operator*(const f2& o) const { return f2{x * o.x, y * o.y}; }
  };
  auto fma2 = [](const f2& a, const f2& b, const f2& c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; };
  // ---- x = SiLU(a_i + b_j): operand fragments of z, and what the dW1 GEMM reads ----
  Frag<T> xf[KS];
  {
    const T* arow = abd + (int64_t)ci * 2 * D;
    const T* brow = abd + (int64_t)cj * 2 * D + D;
    T* x_row = p.x + row * D + 8 * half;
    constexpr int G = 4;
    uint4 ra[2][G], rb[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      ra[0][i] = *reinterpret_cast<const uint4*>(arow + 16 * i + 8 * half);
      rb[0][i] = *reinterpret_cast<const uint4*>(brow + 16 * i + 8 * half);
    }
#pragma unroll
    for (int g = 0; g < KS / G; ++g) {
      if (g + 1 < KS / G) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          ra[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(arow + 16 * (G * (g + 1) + i) + 8 * half);
          rb[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(brow + 16 * (G * (g + 1) + i) + 8 * half);
        }
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        float a[8], bb[8];
        unpack16<T>(ra[g & 1][i], a);
        unpack16<T>(rb[g & 1][i], bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e] + bb[e]);
        xf[G * g + i] = pack_frag8<T>(a);
        asm volatile("" : "+v"(xf[G * g + i].v.x), "+v"(xf[G * g + i].v.y), "+v"(xf[G * g + i].v.z), "+v"(xf[G * g + i].v.w) :: "memory");
        *reinterpret_cast<uint4*>(x_row + 16 * (G * g + i)) = xf[G * g + i].v;
      }
    }
  }

  f32x16_t du[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) du[t][r] = 0.f;
    asm volatile("" : "+a"(du[t]));
  }

  float* slot_ws = p.ws + (int64_t)(blockIdx.x % PB_SLOTS) * 4 * ncol;
  auto flush = [&](int s) {
    if (wave == (s & 3) && lane < 32) {
      const float4* src = sPart + (s & 1) * (4 * 32) + lane;
      float4 t = src[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const float4 u = src[w * 32]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      float* dst = slot_ws + s * 32 + lane;
      atomicAdd(dst, t.x); atomicAdd(dst + ncol, t.y); atomicAdd(dst + 2 * (int64_t)ncol, t.z); atomicAdd(dst + 3 * (int64_t)ncol, t.w);
    }
  };
  // operands built per head from scale_h * dlogits_h of the wave's 32 pairs (see the wave-specialised kernel)
  pb_u32x4 gA = pb_u32x4{0u, 0u, 0u, 0u};
  float* const gt = reinterpret_cast<float*>(sG + grp * 32);       // [3][32] floats: g_c of the wave's pairs (this head)
  auto stage_g = [&](int h) {
    const int Cn = p.a.classes[h];
    float gx = 0.f, gy = 0.f, gz = 0.f, sc = 0.f;
    if (lane < 32 && pair_ok) {
      sc = p.a.scale[h];
      const float* dl = p.a.dlogits[h] + ((int64_t)b * p.P + mypair) * Cn;
      gx = dl[0];
      if (Cn > 1) gy = dl[1];
      if (Cn > 2) gz = dl[2];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(gx), "+v"(gy), "+v"(gz), "+v"(sc) :: "memory");
    gx *= sc; gy *= sc; gz *= sc;
    gA = pb_u32x4{pack_bf16x2(gx, gy), pack_bf16x2(gz, 0.f), 0u, 0u};
    if (lane < 32) { gt[lane] = gx; gt[32 + lane] = gy; gt[64 + lane] = gz; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  auto build_gT = [&](pb_u32x4& gT0, pb_u32x4& gT1) {
    const int c = r32;
    float v[16];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int e4 = 0; e4 < 2; ++e4) {
        const float4 q = c < 3 ? *reinterpret_cast<const float4*>(gt + c * 32 + 16 * kk + 8 * e4 + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[8 * kk + 4 * e4 + 0] = q.x; v[8 * kk + 4 * e4 + 1] = q.y; v[8 * kk + 4 * e4 + 2] = q.z; v[8 * kk + 4 * e4 + 3] = q.w;
      }
    gT0 = pb_u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    gT1 = pb_u32x4{pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])};
  };
  const f2 nl2e = f2{-1.4426950408889634f, -1.4426950408889634f};
  // 16-bit selectors of the dz tile's dword: even lanes (own.lo, partner.lo), odd lanes (partner.hi, own.hi)
  const uint32_t psel = (lane & 1) ? 0x03020706u : 0x05040100u;
  const int t_swz = (r32 >> 2) & 3;
  T* dz_row = p.dz + row * ncol + 8 * half;
  const uint32_t drop_key = DROP ? pair_drop_key(p.drop_seed, b) : 0u;
  const uint32_t drop_base = (uint32_t)(mypair * nslab * 2) + (uint32_t)half;
  const int mreg = (r32 & 3) + 4 * (r32 >> 3), mhp = (r32 >> 2) & 1;

  // tools/pb_cycles.py (a -DPB_PROF build, debug buffer set): ticks per phase, summed over the iterations: 0 wait + barrier at the top, 1 weight
  // stream issue / column-sum flush / dlogits staging, 2 Z, 3 E | U, 4 dW2 sums.  A mark drains the LDS queue.
#ifdef PB_PROF
  unsigned long long t_ph[5] = {0, 0, 0, 0, 0}, t_last = 0;
  auto mark = [&](int k) {
    if (p.dbg) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (k >= 0) t_ph[k] += t - t_last;
      t_last = t;
    }
  };
#else
  auto mark = [](int) {};
#endif
  auto top = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mark(0);
  };

  // one iteration: Z(s) (DOZ), then E(s) (DOZ) interleaved with U(s - 1) (DOU)
  auto iteration = [&](auto z_c, auto u_c, int s) {
    constexpr bool DOZ = decltype(z_c)::value, DOU = decltype(u_c)::value;
    f32x16_t z, dy;
    if constexpr (!DOZ) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { z[r] = 0.f; dy[r] = 0.f; }
    }
    uint2 cw2 = make_uint2(0u, 0u);
    float cb1 = 0.f;
    if constexpr (DOZ) {
      if (s % spb == 0) stage_g(s / spb);
      cw2 = sW2p[s * 32 + r32]; cb1 = sB1[s * 32 + r32];
      // (the compiler's own wait for these two belongs HERE: it cannot see the hand-issued reads below and would drain the
      // LDS queue at their first use, in the middle of the interleaved phase)
      asm volatile("" : "+v"(cw2.x), "+v"(cw2.y), "+v"(cb1));
      mark(1);
      // ---- Z(s): pure matrix phase; a chunk's fragments are requested two chunks ahead, into the set read three chunks ago
      const uint32_t zs = ring + (s % 3) * HALF_BYTES + half * 512;
      const uint32_t za0 = zs + ((r32 ^ (4 * half)) << 4), za1 = zs + ((r32 ^ (4 * half + 8)) << 4);
      // chunks of two fragments, requested three chunks ahead into the set read four chunks ago
      constexpr int NZ = KS / 2;
      pb_u32x4 fs[4][2];
      f32x16_t zb;
      auto zissue = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        pb_dsr<(2 * J) * 1024>(fs[J & 3][0], za0);
        pb_dsr<(2 * J + 1) * 1024>(fs[J & 3][1], za1);
      };
      // K12 dropout: the keep / drop addends of this slab are made under the matrix phase.  Lane = (pair r32, half) walks the
      // forward's chain of 16 fields (units 8g + 4 half + e) and leaves one f16 per unit where the lane that owns (unit, half
      // of the pair's register) reads its 16 registers; chunk J takes steps 16 J / NZ .. 16 (J + 1) / NZ - 1
      // the next slab's weights (the last iteration requests its own slab once more, into the free slot: no branch)
      const int nslot = (s + 1) % 3, nsl = min(s + 1, nslab - 1);
      const char* const nsrc = wuni + (int64_t)nsl * (3 * HALF_BYTES);
      const uint32_t ndst = __builtin_amdgcn_readfirstlane(ring + nslot * HALF_BYTES + wave * 1024);
      const uint32_t mwa = lds_addr(sMask) + grp * (32 * MROW) + half * (4 * MROW) + mhp * 32 + mreg * 2;
      uint32_t mst = DROP ? pair_drop_seed(drop_key, drop_base + 2u * (uint32_t)s) : 0u;
      const uint32_t minc = DROP ? pair_drop_inc(drop_key, drop_base + 2u * (uint32_t)s) : 0u;
      const uint32_t thr32 = p.drop_thr16 << 16;
      // A wave has ONE MFMA in flight: the next one holds the wave until the matrix pipe is free, and only what stands
      // BETWEEN two MFMAs issues in the first one's shadow.  So: wait, MFMA, [next reads + a weight piece], MFMA, [mask steps].
      // LDS operations of the phase, in order:   R0 R1 R2 | chunk k: R(k+3) W(k)      (R = 2 reads, W = the chunk's mask stores)
      auto zchunk = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        // the fragments of chunk J have landed: at most the two younger chunks' reads and the mask stores issued since then
        constexpr int yrd = (NZ - 1 - J) < 2 ? (NZ - 1 - J) : 2;
        constexpr int ywr = DROP ? pb_mask_steps(NZ, J) - pb_mask_steps(NZ, J >= 3 ? J - 3 : 0) : 0;
        asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fs[J & 3][0]), "+v"(fs[J & 3][1]) : [n] "n"(2 * yrd + ywr) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        // (volatile statements, two accumulators: builtin MFMAs carry no ordering of their own against the hand-issued reads
        // around them - instruction selection sank the whole chain behind the last wait)
        if constexpr (J == 0) pb_mma_v0(xf[0].v, fs[0][0], z); else pb_mma_v(xf[2 * J].v, fs[J & 3][0], z);
        if constexpr (J + 3 < NZ) zissue(std::integral_constant<int, J + 3>{});
        if constexpr (J % 2 == 1 && J / 2 < PPW) pb_dma_piece(wlane, nsrc + (J / 2) * (PB_WAVES * 1024), ndst + (J / 2) * (PB_WAVES * 1024));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J == 0) pb_mma_v0(xf[1].v, fs[0][1], zb); else pb_mma_v(xf[2 * J + 1].v, fs[J & 3][1], zb);
        if constexpr (DROP) {
          pb_static_for<pb_mask_steps(NZ, J + 1) - pb_mask_steps(NZ, J)>([&](auto ic) {
            constexpr int I = pb_mask_steps(NZ, J) + decltype(ic)::value;
            mst = pair_drop_step(mst, minc);
            const uint32_t v = mst >= thr32 ? 0u : 0xF753u;   // field = bits 16.. of the state; f16 0 / -30000
            pb_dsw16<(8 * (I >> 2) + (I & 3)) * MROW>(mwa, v);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the counted waits below count only the fragment reads
      zissue(std::integral_constant<int, 0>{});
      zissue(std::integral_constant<int, 1>{});
      zissue(std::integral_constant<int, 2>{});
      pb_static_for<NZ>([&](auto jc) { zchunk(jc); });
      // the compiler does not know that the statements above are MFMAs: the wait states between an 8-pass MFMA and a VALU read
      // of its result are ours (the dy MFMA and these nops)
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(z), "+v"(zb));
#pragma unroll
      for (int r = 0; r < 16; ++r) { z[r] += zb[r]; dy[r] = 0.f; }
      const pb_u32x4 w2f = pb_u32x4{half ? 0u : cw2.x, half ? 0u : cw2.y, 0u, 0u};
      pb_mma(gA, w2f, dy);
      mark(2);
    }
    const f2 b1 = f2{cb1, cb1};
    char* myT = sT + ((s & 1) * 4 + grp) * 2048;
    f2 sb = f2{0.f, 0.f};
    uint32_t yp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) yp[i] = 0u;
    // K12 dropout: the f16 addends (0 / -30000) of accumulator registers 2J, 2J + 1 = dword J of this lane's 16, read one
    // chunk ahead
    const uint32_t mka = lds_addr(sMask) + (grp * 32 + r32) * MROW + half * 32;
    uint32_t mk[2] = {0u, 0u};
    constexpr bool MK = DOZ && DROP;

    // ---- U(s - 1): A operand = the dz tile of slab s - 1 (written by this wave during the previous iteration), B operand =
    // that slab's z fragments read transposed (see the consumer waves of the wave-specialised kernel).  Sub-chunk q = the two
    // fragments of du tile q; two register sets: the reads of sub-chunk q + 2 go out right behind the MFMAs of sub-chunk q.
    // LDS operations complete in order, so the waits count what may still be in flight (reads: 4 per sub-chunk, the mask
    // dword, the dz tile store of the E part).
    pb_u32x4 a0 = pb_u32x4{0u, 0u, 0u, 0u}, a1 = a0;
    pb_u32x2 ul[3][2], uh[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) { ul[k][i] = pb_u32x2{0u, 0u}; uh[k][i] = ul[k][i]; }
    uint32_t ua0 = 0u, ua1 = 0u;
    auto uissue = [&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      // sub-chunk Q = the kk = Q & 1 fragments of du tiles 2 (Q >> 1), 2 (Q >> 1) + 1: its two MFMAs are independent
      if constexpr (DOU && Q < NDT) {
        constexpr int F0 = 4 * (Q >> 1) + (Q & 1);
        pb_trd<pb_uoff(F0)>(ul[Q % 3][0], ua0); pb_trd<pb_uoff(F0)>(uh[Q % 3][0], ua1);
        pb_trd<pb_uoff(F0 + 2)>(ul[Q % 3][1], ua0); pb_trd<pb_uoff(F0 + 2)>(uh[Q % 3][1], ua1);
      }
    };
    if constexpr (MK) asm volatile("ds_read_b32 %0, %1" : "=v"(mk[0]) : "v"(mka));
    if constexpr (DOU) {
      const int u = s - 1;
      const uint32_t ta = lds_addr(sT + ((u & 1) * 4 + grp) * 2048) + r32 * 64;
      const int g_ = lane >> 4, hh_ = g_ >> 1, rr_ = (lane & 15) >> 2, cc_ = lane & 3, h_ = cc_ >> 1;
      const uint32_t ub = ring + (u % 3) * HALF_BYTES + (g_ & 1) * 1024 + ((h_ * 32 + 8 * (hh_ ^ (g_ & 1)) + rr_) << 4) + (cc_ & 1) * 8;
      ua0 = ub + (h_ ? 64 : 0); ua1 = ub + (h_ ? 0 : 64);
      pb_dsr<0>(a0, ta + (((0 + half) ^ t_swz) << 4));
      pb_dsr<0>(a1, ta + (((2 + half) ^ t_swz) << 4));
      uissue(std::integral_constant<int, 0>{});
      uissue(std::integral_constant<int, 1>{});
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a0), "+v"(a1), "+v"(mk[0]) :: "memory");   // all but the two sub-chunks
      *reinterpret_cast<pb_u32x4*>(dz_row + u * 32) = a0;
      *reinterpret_cast<pb_u32x4*>(dz_row + u * 32 + 16) = a1;
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk[0]) :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    // chunk J = sub-chunks 2J, 2J + 1 of U (two MFMAs each) and accumulator registers 2J, 2J + 1 of E.  Every MFMA is followed by
    // a quarter of the chunk's E arithmetic (what issues in its shadow, see Z).  LDS operations of the wave, in order:
    //   [wait a] R(2J+2) MK(J+1) ... R(2J+3) [wait b] ... W(J)       R = 4 transposed reads, MK = mask dword, W = tile store
    // wait a: sub-chunk 2J and MK(J) complete   = at most R(2J+1) W(J-1) outstanding
    // wait b: sub-chunk 2J+1 complete           = at most W(J-1) R(2J+2) MK(J+1) R(2J+3) outstanding
    auto umma1 = [&](auto qc, auto ic) {
      constexpr int Q = decltype(qc)::value, I = decltype(ic)::value;
      if constexpr (DOU)
        pb_mma_acc((Q & 1) ? a1 : a0, pb_u32x4{ul[Q % 3][I].x, ul[Q % 3][I].y, uh[Q % 3][I].x, uh[Q % 3][I].y}, du[2 * (Q >> 1) + I]);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    auto chunk = [&](auto jc) {
      constexpr int J = decltype(jc)::value;
      constexpr int r0 = 2 * J, rowc = (r0 & 3) + 8 * (r0 >> 2);
      constexpr int NW = DOZ ? 1 : 0, NM = MK ? 1 : 0, RA = DOU ? 4 : 0;
      const int row0 = rowc + 4 * half;
      f2 zz = f2{0.f, 0.f}, sg = zz, y = zz, dzv = zz;
      if constexpr (DOU || MK) {
        asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(mk[J & 1]) : [n] "n"(RA + (J > 0 ? NW : 0)) : "memory");
        if constexpr (DOU) { pb_pin<2>(ul[(2 * J) % 3]); pb_pin<2>(uh[(2 * J) % 3]); }
      }
      umma1(std::integral_constant<int, 2 * J>{}, I0{});
      uissue(std::integral_constant<int, 2 * J + 2>{});
      if constexpr (MK && J + 1 < 8) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(mk[(J + 1) & 1]) : "v"(mka), "n"(4 * (J + 1)));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DOZ) {
        float zx = z[r0], zy = z[r0 + 1];
        asm volatile("" : "+v"(zx), "+v"(zy));                 // anchor behind the MFMA above
        zz = f2{zx, zy} + b1;
        if constexpr (DROP) {
          asm volatile("v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                       "v_fma_mix_f32 %1, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                       : "+v"(zz.x), "+v"(zz.y) : "v"(mk[J & 1]));
        }
        const f2 t = zz * nl2e;
        sg = f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        asm volatile("" : "+v"(sg.x), "+v"(sg.y));             // anchor in front of the next MFMA
      }
      __builtin_amdgcn_sched_barrier(0);
      umma1(std::integral_constant<int, 2 * J>{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DOZ) {
        const f2 den = sg + f2{1.f, 1.f};
        sg = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        y = zz * sg;
        asm volatile("" : "+v"(y.x), "+v"(y.y), "+v"(sg.x), "+v"(sg.y));
      }
      uissue(std::integral_constant<int, 2 * J + 3>{});
      if constexpr (DOU) {
        constexpr int nb = J + 1 < 8 ? (J > 0 ? NW : 0) + 4 + NM + 4 : NW;
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(nb) : "memory");
        pb_pin<2>(ul[(2 * J + 1) % 3]); pb_pin<2>(uh[(2 * J + 1) % 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      umma1(std::integral_constant<int, 2 * J + 1>{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DOZ) {
        // SiLU'(z) = sg (1 + z (1 - sg)) = (sg + y) - y sg
        dzv = f2{dy[r0], dy[r0 + 1]} * fma2(f2{-y.x, -y.y}, sg, sg + y);
        yp[J] = pack_bf16x2(y.x, y.y);
        asm volatile("" : "+v"(dzv.x), "+v"(dzv.y), "+v"(yp[J]));
      }
      __builtin_amdgcn_sched_barrier(0);
      umma1(std::integral_constant<int, 2 * J + 1>{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DOZ) {
        sb = sb + dzv;
        // the lane's two values are rows (pairs) r, r + 1 of ONE hidden unit; the tile wants units c, c + 1 of one pair per
        // dword: round both, fetch the neighbour lane's dword, pick the halves
        const uint32_t own = pack_bf16x2(dzv.x, dzv.y);
        const uint32_t oth = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, true);
        const uint32_t packed = __builtin_amdgcn_perm(oth, own, psel);
        const int trow = row0 + (lane & 1);
        const int boff = (r32 & ~1) * 2;
        const int f = ((rowc >> 2) + half) & 3;
        asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr(myT + trow * 64 + ((((boff >> 4) ^ f) << 4) | (boff & 15)))), "v"(packed) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    pb_static_for<8>([&](auto jc) { chunk(jc); });
    mark(3);
    if constexpr (DOZ) {
      // dW2 sums out[c, hid] = sum_pair g[pair, c] y[pair, hid] (y straight from its accumulator-layout registers; the A
      // operand, the head's g transposed, is rebuilt from the wave's staging rows: 8 registers less across the loop) and db1
      pb_u32x4 gT0, gT1;
      build_gT(gT0, gT1);
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16_t acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, gT0), __builtin_bit_cast(bf16x8_t, pb_u32x4{yp[0], yp[1], yp[2], yp[3]}), zero, 0, 0, 0);
      pb_mma(gT1, pb_u32x4{yp[4], yp[5], yp[6], yp[7]}, acc);
      float sbt = sb.x + sb.y;
      sbt += __shfl_xor(sbt, 32);
      const float ys = DROP ? p.drop_scale : 1.f;
      if (lane < 32) sPart[(s & 1) * (4 * 32) + grp * 32 + lane] = make_float4(acc[0] * ys, acc[1] * ys, acc[2] * ys, sbt);
      mark(4);
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  __syncthreads();                                            // sW2p / sB1 visible
  mark(-1);
#ifdef PB_PROF
  const unsigned long long t_begin = t_last;
#endif
  top();                                                      // slab 0 has landed
  iteration(yes{}, no{}, 0);
  for (int s = 1; s < nslab; ++s) {
    top();
    flush(s - 1);
    iteration(yes{}, yes{}, s);
  }
  top();
  flush(nslab - 1);
  iteration(no{}, yes{}, nslab);
#ifdef PB_PROF
  if (p.dbg && blockIdx.y == 0 && blockIdx.x < 256 && lane == 0) {
    unsigned long long* d = p.dbg + ((int64_t)blockIdx.x * PB_WAVES + wave) * 8;
    d[0] = t_last - t_begin;
#pragma unroll
    for (int k = 0; k < 5; ++k) d[1 + k] = t_ph[k];
  }
#endif
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last du MFMAs (inline asm: unknown to the compiler's hazard pass)
  __syncthreads();                                            // every wave is done with the rings
  // ---- du * SiLU'(a_i + b_j): sums over j stay in the wave, sums over i meet in LDS (everything above is dead) ----
  float* red = reinterpret_cast<float*>(smem);                // [4][NDT][16][32]
  const int ia = min(i0, N - 1), ib = min(i0 + 1, N - 1);
  float* pa = p.part_a + (((int64_t)b * p.ntiles + blockIdx.x) * PB_TI + 2 * grp + half) * D;
#pragma unroll
  for (int t = 0; t < NDT; ++t) {
    const int d = 32 * t + r32;
    const float a_lo = bf16_to_f32(abd[(int64_t)ia * 2 * D + d]), a_hi = bf16_to_f32(abd[(int64_t)ib * 2 * D + d]);
    float bj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = min(j0 + 8 * (q >> 2) + 4 * half + (q & 3), N - 1);
      bj[q] = bf16_to_f32(abd[(int64_t)j * 2 * D + D + d]);
    }
    float sa0 = 0.f, sa1 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v0 = du[t][q] * silu_grad_f(a_lo + bj[q]);
      const float v1 = du[t][8 + q] * silu_grad_f(a_hi + bj[q]);
      sa0 += v0; sa1 += v1;
      red[((grp * NDT + t) * 16 + 8 * (q >> 2) + 4 * half + (q & 3)) * 32 + r32] = v0 + v1;
    }
    sa0 += __shfl_xor(sa0, 32);
    sa1 += __shfl_xor(sa1, 32);
    pa[d] = half ? sa1 : sa0;
  }
  __syncthreads();
  float* pb = p.part_b + ((int64_t)b * p.ntiles + blockIdx.x) * PB_TJ * D;
  for (int e = tid; e < NDT * 16 * 32; e += PB_WAVES * 64) {
    const int c = e & 31, jl = (e >> 5) & 15, t = e >> 9;
    float v = red[e];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[w * NDT * 512 + e];
    pb[(int64_t)jl * D + 32 * t + c] = v;
  }
}

template <int KS, bool DROP>
static int launch_pair_bwd_one(const PairBwdParams& p, hipStream_t st) {
  const int ncol = p.a.num_heads * p.D;
  size_t sh = (size_t)3 * KS * 1024 + (size_t)ncol * 12 + (size_t)4 * 32 * 16 * 3 + (size_t)2 * 4 * 2048 + (size_t)4 * 32 * 80;
  const size_t red = (size_t)4 * (KS / 2) * 16 * 32 * sizeof(float);
  if (sh < red) sh = red;
  if (sh > 160 * 1024) { set_error("peneo_pair_bwd_fused: D=%d needs %zu bytes of LDS", p.D, sh); return PENEO_ERR_INVALID; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(pair_bwd_one_kernel<KS, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_error("peneo_pair_bwd_fused: cannot raise dynamic LDS to %zu bytes", sh);
    return PENEO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((pair_bwd_one_kernel<KS, DROP>), dim3((unsigned)p.ntiles, (unsigned)p.B), dim3(PB_WAVES * 64), sh, st, p);
  return check_launch("peneo_pair_bwd_fused");
}

template <int KS, bool DROP>
static int launch_pair_bwd_ws(const PairBwdParams& p, hipStream_t st) {
  const int ncol = p.a.num_heads * p.D;
  const size_t sh = (size_t)4 * KS * 1024 + (size_t)ncol * 12 + (size_t)4 * 32 * 16 * 3 + (size_t)2 * 4 * 2048 + (size_t)2 * 4 * 32 * 32 * 2;
  if (sh > 160 * 1024) { set_error("peneo_pair_bwd_fused: D=%d needs %zu bytes of LDS", p.D, sh); return PENEO_ERR_INVALID; }
  if (sh > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(pair_bwd_ws_kernel<KS, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_error("peneo_pair_bwd_fused: cannot raise dynamic LDS to %zu bytes", sh);
    return PENEO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((pair_bwd_ws_kernel<KS, DROP>), dim3((unsigned)p.ntiles, (unsigned)p.B), dim3(PW_WAVES * 64), sh, st, p);
  return check_launch("peneo_pair_bwd_fused");
}

template <int KS, bool DROP>
static int launch_pair_bwd_sv(const PairBwdParams& p, hipStream_t st) {
  const int ncol = p.a.num_heads * p.D;
  const size_t sh = (size_t)3 * KS * 1024 + (size_t)PSV_NR * 4 * PB_REC_BYTES + (size_t)ncol * 8 + (size_t)4 * 32 * 16 * 3 + (size_t)2 * 4 * 2048;
  if (sh > 160 * 1024) { set_error("peneo_pair_bwd_saved: D=%d with %d heads needs %zu bytes of LDS", p.D, p.a.num_heads, sh); return PENEO_ERR_INVALID; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(pair_bwd_sv_kernel<KS, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_error("peneo_pair_bwd_saved: cannot raise dynamic LDS to %zu bytes", sh);
    return PENEO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((pair_bwd_sv_kernel<KS, DROP>), dim3((unsigned)p.ntiles, (unsigned)p.B), dim3(PW_WAVES * 64), sh, st, p);
  return check_launch("peneo_pair_bwd_saved");
}

template <int KS>
static int launch_pair_bwd(const PairBwdParams& p, hipStream_t st) {
  return p.drop_thr16 ? launch_pair_bwd_ws<KS, true>(p, st) : launch_pair_bwd_ws<KS, false>(p, st);
}
template <int KS>
static int launch_pair_bwd_1w(const PairBwdParams& p, hipStream_t st) {
  return p.drop_thr16 ? launch_pair_bwd_one<KS, true>(p, st) : launch_pair_bwd_one<KS, false>(p, st);
}

}  // namespace peneo
using namespace peneo;

#ifdef PB_PROF
static unsigned long long* g_pb_dbg = nullptr;
/* instrumented builds only (tools/prof_build.sh pairbwd; not part of the product library or its header): device buffer
 * [256 blocks][8 waves][4] that receives per-wave cycle counts of the wave-specialised kernel */
extern "C" void peneo_pair_bwd_debug_buffer(unsigned long long* dev) { g_pb_dbg = dev; }
#else
static unsigned long long* const g_pb_dbg = nullptr;
#endif

// LDS of a launch (the formulas of launch_pair_bwd_one / launch_pair_bwd_ws): the weight ring, 12 bytes per hidden column, the rest
static size_t pair_bwd_lds_bytes(int D, int num_heads) {
  const size_t ks = (size_t)D / 16, ncol = (size_t)num_heads * D;
  if (ks == 32) return 3 * ks * 1024 + ncol * 12 + (size_t)4 * 32 * 16 * 3 + (size_t)2 * 4 * 2048 + (size_t)4 * 32 * 80;
  return 4 * ks * 1024 + ncol * 12 + (size_t)4 * 32 * 16 * 3 + (size_t)2 * 4 * 2048 + (size_t)2 * 4 * 32 * 32 * 2;
}
extern "C" int peneo_pair_bwd_supported(int dtype, int D, int num_heads) {
  const int ks = D / 16;
  if (!(dtype == PENEO_BF16 && D % 32 == 0 && (ks == 2 || ks == 4 || ks == 8 || ks == 24 || ks == 32))) return 0;
  if (num_heads > PENEO_MAX_HEADS) return 0;
  // the per-column table grows with the head count: D = 512 fits 5 heads (161 792 B) but not 6; a caller that gets 0 runs the chunked path
  return num_heads <= 0 || pair_bwd_lds_bytes(D, num_heads) <= (size_t)160 * 1024;
}

extern "C" int64_t peneo_pair_bwd_rows(int N) { return N > 0 ? (int64_t)pb_num_tiles(N) * PB_ROWS : 0; }

extern "C" size_t peneo_pair_bwd_packed_bytes(int num_heads, int D) { return (size_t)(num_heads * D / 32) * 3 * (D / 16) * 1024; }

extern "C" int peneo_pair_bwd_pack(const float* const* w1, int num_heads, int D, void* packed, peneo_stream_t stream) {
  PENEO_REQUIRE(w1 && packed && num_heads > 0 && num_heads <= PENEO_MAX_HEADS && D % 32 == 0, "peneo_pair_bwd_pack: bad arguments");
  PackBwdSrc s;
  s.num_heads = num_heads; s.D = D;
  for (int h = 0; h < num_heads; ++h) { PENEO_REQUIRE(w1[h], "peneo_pair_bwd_pack: null weight"); s.w1[h] = w1[h]; }
  hipLaunchKernelGGL(pack_bwd_weights_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, s, reinterpret_cast<bf16_t*>(packed));
  return check_launch("peneo_pair_bwd_pack");
}

extern "C" size_t peneo_pair_bwd_partial_bytes(int B, int N, int D) {
  return (size_t)B * pb_num_tiles(N) * (PB_TI + PB_TJ) * D * sizeof(float);
}

extern "C" int peneo_pair_bwd_saved(int dtype, const void* ab, int B, int N, int D, const void* w_packed,
                                    const peneo_pair_dz_args* args, const void* act, void* dz, float* d_ab, float* workspace,
                                    float* partials, peneo_stream_t stream) {
  PENEO_REQUIRE(args && peneo_pair_save_supported(dtype, D, args->num_heads), "peneo_pair_bwd_saved: bf16, D = 384 only (peneo_pair_save_supported)");
  PENEO_REQUIRE(ab && w_packed && act && dz && d_ab && workspace && partials && B > 0 && N > 0, "peneo_pair_bwd_saved: bad arguments");
  PENEO_REQUIRE(args->D == D && args->scale && args->num_heads * D >= 64, "peneo_pair_bwd_saved: bad head description");
  PENEO_REQUIRE(((reinterpret_cast<uintptr_t>(ab) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(act)) & 15) == 0,
                "peneo_pair_bwd_saved: pointers must be 16-byte aligned");
  for (int h = 0; h < args->num_heads; ++h)
    PENEO_REQUIRE(args->dlogits[h] && args->w2[h] && args->classes[h] >= 1 && args->classes[h] <= 3, "peneo_pair_bwd_saved: head %d", h);
  PairBwdParams p;
  p.ab = reinterpret_cast<const bf16_t*>(ab); p.B = B; p.N = N; p.D = D; p.P = (int64_t)N * (N + 1) / 2;
  p.wp = w_packed; p.b1 = nullptr; p.a = *args;
  p.dz = reinterpret_cast<bf16_t*>(dz); p.x = nullptr; p.act = static_cast<const char*>(act); p.ws = workspace;
  p.ntiles = pb_num_tiles(N);
  PENEO_REQUIRE(args->drop_p >= 0.f && args->drop_p < 1.f, "peneo_pair_bwd_saved: drop_p must be in [0, 1)");
  p.drop_thr16 = pair_drop_thr16_host(args->drop_p); p.drop_seed = args->drop_seed; p.drop_scale = pair_drop_scale_host(args->drop_p);
  p.dbg = nullptr;
  p.part_a = partials; p.part_b = partials + (size_t)B * p.ntiles * PB_TI * D;
  hipStream_t st = (hipStream_t)stream;
  const int rc = p.drop_thr16 ? launch_pair_bwd_sv<24, true>(p, st) : launch_pair_bwd_sv<24, false>(p, st);
  if (rc != PENEO_OK) return rc;
  hipLaunchKernelGGL(pair_bwd_reduce_kernel, dim3((unsigned)N, (unsigned)B), dim3(256), 0, st, p.part_a, p.part_b, N, D, p.ntiles, d_ab);
  return check_launch("peneo_pair_bwd_saved (reduce)");
}

extern "C" int peneo_pair_bwd_fused(int dtype, const void* ab, int B, int N, int D, const void* w_packed, const float* b1,
                                    const peneo_pair_dz_args* args, void* dz, void* x, float* d_ab, float* workspace,
                                    float* partials, peneo_stream_t stream) {
  PENEO_REQUIRE(peneo_pair_bwd_supported(dtype, D, 0), "peneo_pair_bwd_fused: bf16 and D/16 in {2, 4, 8, 24, 32} only (got D=%d)", D);
  PENEO_REQUIRE(args && peneo_pair_bwd_supported(dtype, D, args->num_heads),
                "peneo_pair_bwd_fused: %d heads at D=%d need more than 160 KiB of LDS (ask peneo_pair_bwd_supported first)", args ? args->num_heads : 0, D);
  PENEO_REQUIRE(ab && w_packed && b1 && args && dz && x && d_ab && workspace && partials && B > 0 && N > 0, "peneo_pair_bwd_fused: bad arguments");
  PENEO_REQUIRE(args->num_heads > 0 && args->num_heads <= PENEO_MAX_HEADS && args->D == D && args->scale, "peneo_pair_bwd_fused: bad head description");
  PENEO_REQUIRE(args->num_heads * D >= 64, "peneo_pair_bwd_fused: needs at least two 32-unit slabs of hidden units");
  PENEO_REQUIRE(((reinterpret_cast<uintptr_t>(ab) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(x)) & 15) == 0,
                "peneo_pair_bwd_fused: pointers must be 16-byte aligned");
  for (int h = 0; h < args->num_heads; ++h)
    PENEO_REQUIRE(args->dlogits[h] && args->w2[h] && args->classes[h] >= 1 && args->classes[h] <= 3, "peneo_pair_bwd_fused: head %d", h);
  PairBwdParams p;
  p.ab = reinterpret_cast<const bf16_t*>(ab); p.B = B; p.N = N; p.D = D; p.P = (int64_t)N * (N + 1) / 2;
  p.wp = w_packed; p.b1 = b1; p.a = *args;
  p.dz = reinterpret_cast<bf16_t*>(dz); p.x = reinterpret_cast<bf16_t*>(x); p.act = nullptr; p.ws = workspace;
  p.ntiles = pb_num_tiles(N);
  PENEO_REQUIRE(args->drop_p >= 0.f && args->drop_p < 1.f, "peneo_pair_bwd_fused: drop_p must be in [0, 1)");
  p.drop_thr16 = pair_drop_thr16_host(args->drop_p); p.drop_seed = args->drop_seed; p.drop_scale = pair_drop_scale_host(args->drop_p);
  p.dbg = g_pb_dbg;
  p.part_a = partials; p.part_b = partials + (size_t)B * p.ntiles * PB_TI * D;
  hipStream_t st = (hipStream_t)stream;
  int rc = PENEO_ERR_INVALID;
  switch (D / 16) {
    case 2: rc = launch_pair_bwd<2>(p, st); break;
    case 4: rc = launch_pair_bwd<4>(p, st); break;
    case 8: rc = launch_pair_bwd<8>(p, st); break;
    case 24: rc = launch_pair_bwd<24>(p, st); break;
    case 32: rc = launch_pair_bwd_1w<32>(p, st); break;
  }
  if (rc != PENEO_OK) return rc;
  hipLaunchKernelGGL(pair_bwd_reduce_kernel, dim3((unsigned)N, (unsigned)B), dim3(256), 0, st, p.part_a, p.part_b, N, D, p.ntiles, d_ab);
  return check_launch("peneo_pair_bwd_fused (reduce)");
}
