// K11 + K12 + K13 fused: handshaking + the five pair-classifier heads + class-weighted CE.
//
// The reference materialises [B, N, N, 2D] and [B, P, D] tensors (model/peneo_decoder.py:164-175)
// and runs five Linear(D->D)+SiLU+Linear(D->C) heads over all P = N(N+1)/2 token pairs
// (:231-292, :355-363).  Here one workgroup owns 128 consecutive pairs of one document; each
// 64-lane wave owns 32 pairs (lane & 31 = pair) and
//   * builds x = SiLU(a_i + b_j) for its pairs straight into MFMA B-operand registers
//     (a_i + b_j == combine_fc(cat(h_i, h_j)), a/b come from one [N, 2D] GEMM),
//   * streams the concatenated first-layer weights (all heads, [nh*D, D], pre-packed in MFMA
//     A-fragment order) through LDS, double buffered, one 32-row slab at a time:
//     z^T[hidden, pair] = W1 . x^T  on v_mfma_f32_32x32x16_bf16 (fp32 accumulate),
//   * applies bias + SiLU in registers; the C-layout accumulator of a slab is, under a fixed
//     row permutation, exactly the B operand of the second layer, whose (block-diagonal, zero
//     padded to 32 classes) weights are pre-packed with that permutation:
//     logits^T[class, pair] += W2 . y   — two more MFMAs per slab, nothing leaves registers,
//   * finishes with soft-max cross-entropy per head (fp32) and writes only the 14 logits per pair
//     (and, for training, the un-normalised dlogits and the loss partial sums).
// HBM traffic per document is the 7.3 MB of logits; everything else is L2/LDS resident.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

namespace peneo {

#ifndef PH_WAVES_N
#define PH_WAVES_N 8
#endif
constexpr int PH_WAVES = PH_WAVES_N;
constexpr int PH_PAIRS = PH_WAVES * 32;   // pairs per workgroup (8 waves x 32)
constexpr int NCP = 16;                   // padded class rows that are ever non-zero (<= 16)

struct PackSrc {
  const float* w1[PENEO_MAX_HEADS];
  const float* w2[PENEO_MAX_HEADS];
  int classes[PENEO_MAX_HEADS];
  int num_heads;
  int D;
};

// Packed weights, one contiguous block per 32-row slab of hidden units, in MFMA A-fragment order
// (64 lanes x 8 elements per fragment):
//   fragment ks < KS   : W1cat[slab*32 + (lane&31)][16*ks + 8*(lane>>5) + e]                       (first layer)
//   fragment KS + kk   : W2full[class = lane&31][hidden = slab*32 + 16*kk + (e&3) + 8*(e>>2) + 4*(lane>>5)]
// where W2full is the block-diagonal [sum classes (zero padded to 32), nh*D] second-layer matrix and the hidden
// permutation is the one under which a C-layout accumulator tile is directly a B operand.
// Slabs are padded to a whole number of 1 KiB DMA units per wave (slab_stride_elems elements per slab), so that
// every wave of the forward kernel issues the same number of LDS-DMA instructions with immediate offsets.
__host__ __device__ inline int64_t slab_stride_bytes(int D, int elem_bytes) {
  const int64_t payload = (int64_t)(D / 16 + 2) * 512 * elem_bytes;
  const int64_t quantum = (int64_t)PH_WAVES * 1024;
  return (payload + quantum - 1) / quantum * quantum;
}
template <typename T>
__global__ void pack_weights_kernel(PackSrc s, T* out) {
  const int D = s.D, KS = D / 16, NF = KS + 2;
  const int nslab = s.num_heads * D / 32;
  const int64_t stride = slab_stride_bytes(D, (int)sizeof(T)) / (int64_t)sizeof(T);
  const int64_t total = (int64_t)nslab * stride;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int slab = (int)(j / stride);
    const int64_t i = j % stride;
    if (i >= (int64_t)NF * 512) { Elem<T>::store(out + j, 0.f); continue; }
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    const int f = (int)(i >> 9);
    float v = 0.f;
    if (f < KS) {
      const int row = slab * 32 + (lane & 31);
      const int col = 16 * f + 8 * (lane >> 5) + e;
      const int h = row / D;
      v = s.w1[h][(int64_t)(row - h * D) * D + col];
    } else {
      const int kk = f - KS;
      const int cls = lane & 31;
      const int hidden = slab * 32 + 16 * kk + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      const int h = hidden / D;
      int off = 0;
      for (int q = 0; q < h; ++q) off += s.classes[q];
      if (cls >= off && cls < off + s.classes[h]) v = s.w2[h][(int64_t)(cls - off) * D + (hidden - h * D)];
    }
    Elem<T>::store(out + j, v);
  }
}

struct PairFwdParams {
  const void* ab; int B, N, D; int64_t P;
  int num_heads; int classes[PENEO_MAX_HEADS]; int total_classes;
  const void* wp; const float* b1; const float* b2;
  float* logits[PENEO_MAX_HEADS];
  const int64_t* tags[PENEO_MAX_HEADS];
  const float* cw[PENEO_MAX_HEADS];
  float* partials;   // [B * gridDim.x][32]: num[8] | den[8] | dl_sum[16] per workgroup
  float* dlogits[PENEO_MAX_HEADS];
  uint32_t drop_thr16, drop_seed; float drop_scale;   // K12 dropout (common.h: pair_drop_*): threshold 0 = off, scale = 1 / (1 - p)
  // saving form (hand kernel only): the pairs are walked in the backward's blocks and the kernel leaves q / y records (common.h:
  // PB_REC_BYTES) and x = SiLU(a_i + b_j) in block-row order for peneo_pair_bwd_saved; NULL = the plain walk, nothing saved
  char* act; bf16_t* x_save; int ntiles;
};

constexpr float NEG_INF_F = -3.0e38f;
// register arrays must only ever be indexed by compile-time constants (dynamic indices go to scratch)
__device__ __forceinline__ float cls_at(const float (&cls)[NCP], int idx) {
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < NCP; ++c) v = (c == idx) ? cls[c] : v;
  return v;
}
__device__ __forceinline__ void dls_add(float (&dls)[NCP], int idx, float g) {
#pragma unroll
  for (int c = 0; c < NCP; ++c) dls[c] += (c == idx) ? g : 0.f;
}


// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: the count must be an immediate
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// logits^T accumulator -> per-pair logits, soft-max CE, un-normalised dlogits and the workgroup's partial sums
__device__ __forceinline__ void pair_epilogue(const PairFwdParams& p, const f32x16_t& lg, char* smem, int tid, int lane, int wave,
                                              int half, int b, int64_t mypair, bool pair_ok) {
  // ---- logits: rows 0..15 of lg live in regs 0..7 (half 0: rows 0-3, 8-11; half 1: rows 4-7, 12-15) ----
  float mine[8], other[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { mine[r] = lg[r]; other[r] = __shfl_xor(lg[r], 32, 64); }
  float cls[NCP];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cls[c] = half == 0 ? mine[c] : other[c];
    cls[4 + c] = half == 0 ? other[c] : mine[c];
    cls[8 + c] = half == 0 ? mine[4 + c] : other[4 + c];
    cls[12 + c] = half == 0 ? other[4 + c] : mine[4 + c];
  }
  float num[PENEO_MAX_HEADS], den[PENEO_MAX_HEADS], dls[NCP];
#pragma unroll
  for (int h = 0; h < PENEO_MAX_HEADS; ++h) { num[h] = 0.f; den[h] = 0.f; }
#pragma unroll
  for (int c = 0; c < NCP; ++c) dls[c] = 0.f;
  const bool writer = pair_ok && half == 0;
  {
    int off = 0;
#pragma unroll
    for (int h = 0; h < PENEO_MAX_HEADS; ++h) {
      if (h < p.num_heads) {
        const int C = p.classes[h];
        float l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) l[c] = (c < C) ? cls_at(cls, off + c) * p.drop_scale + p.b2[off + c] : NEG_INF_F;
        if (writer && p.logits[h]) {
          float* dst = p.logits[h] + ((int64_t)b * p.P + mypair) * C;
          for (int c = 0; c < C; ++c) dst[c] = l[c];
        }
        if (writer && p.tags[h]) {
          const int tag = (int)p.tags[h][(int64_t)b * p.P + mypair];
          float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
          float e[4], se = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) { e[c] = (c < C) ? __expf(l[c] - mx) : 0.f; se += e[c]; }
          const float lse = mx + __logf(se);
          float lt = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) lt = (c == tag) ? l[c] : lt;
          // labels outside [0, C) (the reference's ignore_index = -100 among them) carry zero weight: no loss, no gradient
          const float w = (tag < 0 || tag >= C) ? 0.f : (p.cw[h] ? p.cw[h][tag] : 1.f);
          num[h] = w * (lse - lt);
          den[h] = w;
          const float inv = 1.f / se;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (c < C) {
              float g = w * (e[c] * inv - (c == tag ? 1.f : 0.f));
              if (p.dlogits[h]) p.dlogits[h][((int64_t)b * p.P + mypair) * C + c] = g;
              dls_add(dls, off + c, g);
            }
          }
        }
        off += C;
      }
    }
  }
  if (p.partials) {
    // one row of partial sums per workgroup (plain stores; peneo_loss_finish reduces them): contended atomics
    // on a handful of addresses cost more than the whole MFMA phase
    float* sRed = reinterpret_cast<float*>(smem);  // [8 waves][32] (the weight ring is dead)
    __syncthreads();
#pragma unroll
    for (int h = 0; h < PENEO_MAX_HEADS; ++h) {
      float a = wave_sum(num[h]), d2 = wave_sum(den[h]);
      if (lane == 0) { sRed[wave * 32 + h] = a; sRed[wave * 32 + 8 + h] = d2; }
    }
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
      float a = wave_sum(dls[c]);
      if (lane == 0) sRed[wave * 32 + 16 + c] = a;
    }
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < PH_WAVES; ++w) t += sRed[w * 32 + tid];
      p.partials[((int64_t)b * gridDim.x + blockIdx.x) * 32 + tid] = t;
    }
  }
}


// Generic kernel (bf16 and fp32).  Per workgroup: 8 waves x 32 pairs.  Weight slabs (32 hidden rows: KS first-layer
// fragments + 2 second-layer fragments, padded to UPW KiB per wave) stream L2 -> LDS through a ring of NSTAGE buffers,
// NSTAGE-1 slabs ahead; one s_barrier per slab publishes them.  VARIANT bits (A/B-tested on the GPU):
//   1: LDS-DMA issued from inline asm with counted vmcnt (else the builtin, which hipcc drains before every ds_read)
//   2: bias + SiLU + second layer of slab s-1 software-pipelined into the first-layer MFMA stream of slab s
//   4: two independent first-layer accumulator chains (even / odd k-steps)
template <typename T, int KS, int NSTAGE, int VARIANT, bool DROP>
__global__ __launch_bounds__(PH_WAVES * 64, sizeof(T) == 2 ? (PH_WAVES > 8 ? 3 : 2) : 1) void pair_heads_fwd_kernel(PairFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool ASM_DMA = (VARIANT & 1) != 0, PIPE_EPI = (VARIANT & 2) != 0, DUAL = (VARIANT & 4) != 0;
  constexpr bool SGB = (VARIANT & 16) != 0;       // sched_group_barrier interleave: 1 MFMA : 1 ds_read : few VALU
  constexpr bool NOSTREAM = (VARIANT & 8) != 0;
  constexpr bool G2 = (VARIANT & 128) != 0;       // 4 stages, ONE barrier per TWO slabs (needs ASM_DMA, NSTAGE == 4)
  constexpr bool NO_EPI = (VARIANT & 32) != 0, NO_LDS = (VARIANT & 64) != 0;   // ablations (wrong results)   // timing experiment only (wrong results): no DMA / barrier in the loop
  constexpr int NF = KS + 2;
  constexpr int PAYLOAD = NF * 64 * FragBytes<T>::v;
  constexpr int UPW = (PAYLOAD / 1024 + PH_WAVES - 1) / PH_WAVES;   // 1 KiB DMA pieces per wave per slab
  constexpr int SLAB_BYTES = UPW * PH_WAVES * 1024;                 // == packed slab stride
  static_assert(UPW <= 12, "slab too large");
  char* sW = smem;                                                   // [NSTAGE][SLAB_BYTES]
  float* sB1 = reinterpret_cast<float*>(smem + NSTAGE * SLAB_BYTES); // [nh * D]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D, N = p.N;
  const int b = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * PH_PAIRS;
  const int64_t mypair = p0 + wave * 32 + (lane & 31);
  const bool pair_ok = mypair < p.P;
  int pi, pj;
  pair_decode(pair_ok ? mypair : p.P - 1, N, pi, pj);
  const int nslab = p.num_heads * D / 32;

  for (int i = tid; i < p.num_heads * D; i += PH_WAVES * 64) sB1[i] = p.b1[i];

  // ---- x = SiLU(a_i + b_j) as B-operand fragments (k = decoder dim), kept in registers ----
  const T* abd = reinterpret_cast<const T*>(p.ab) + (int64_t)b * N * 2 * D;
  const T* arow = abd + (int64_t)pi * 2 * D;
  const T* brow = abd + (int64_t)pj * 2 * D + D;
  Frag<T> xf[KS];
  if constexpr (sizeof(T) == 2 && KS % 4 == 0) {
    // groups of 4 k-steps, raw 16-byte gathers double-buffered by hand; the empty asm fences stop the compiler from
    // hoisting every gather to the top (which spilled 760 bytes per thread: 3 GB of scratch traffic per launch)
    uint4 ra[2][4], rb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[0][i] = *reinterpret_cast<const uint4*>(arow + 16 * i + 8 * half);
      rb[0][i] = *reinterpret_cast<const uint4*>(brow + 16 * i + 8 * half);
    }
#pragma unroll
    for (int g = 0; g < KS / 4; ++g) {
      if (g + 1 < KS / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ra[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(arow + 16 * (4 * (g + 1) + i) + 8 * half);
          rb[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(brow + 16 * (4 * (g + 1) + i) + 8 * half);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a[8], bb[8];
        unpack16<T>(ra[g & 1][i], a);
        unpack16<T>(rb[g & 1][i], bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e] + bb[e]);
        xf[4 * g + i] = pack_frag8<T>(a);
        // pin the finished fragment here: otherwise the SiLU math sinks below all the gathers and their raw data spills
        asm volatile("" : "+v"(xf[4 * g + i].v.x), "+v"(xf[4 * g + i].v.y), "+v"(xf[4 * g + i].v.z), "+v"(xf[4 * g + i].v.w) :: "memory");
      }
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 16 * ks + 8 * half;
      float a[8], bb[8];
      if constexpr (sizeof(T) == 2) {
        unpack16<T>(*reinterpret_cast<const uint4*>(arow + c), a);
        unpack16<T>(*reinterpret_cast<const uint4*>(brow + c), bb);
      } else {
        unpack16<T>(*reinterpret_cast<const uint4*>(arow + c), a);
        unpack16<T>(*reinterpret_cast<const uint4*>(arow + c + 4), a + 4);
        unpack16<T>(*reinterpret_cast<const uint4*>(brow + c), bb);
        unpack16<T>(*reinterpret_cast<const uint4*>(brow + c + 4), bb + 4);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e] + bb[e]);
      xf[ks] = pack_frag8<T>(a);
      if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep few gathers in flight (else the raw loads spill)
    }
  }
  // every ordinary global load above has been consumed: from here on the vm counter only sees our DMA pieces
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                          // sB1 visible

  const char* wsrc = reinterpret_cast<const char*>(p.wp) + wave * (UPW * 1024) + lane * 16;
  const uint32_t wdst = lds_addr(sW) + wave * (UPW * 1024);
  char* wdst_p = sW + wave * (UPW * 1024);
  auto dma = [&](int slab_, int buf_) {
    if constexpr (ASM_DMA) {
      lds_dma_units<0, UPW>(wsrc + (int64_t)slab_ * SLAB_BYTES, wdst + buf_ * SLAB_BYTES);
    } else {
#pragma unroll
      for (int u = 0; u < UPW; ++u)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wsrc + (int64_t)slab_ * SLAB_BYTES + u * 1024),
            (__attribute__((address_space(3))) void*)(wdst_p + buf_ * SLAB_BYTES + u * 1024), 16, 0, 0);
    }
  };
  // prologue: NSTAGE-1 slabs in flight
#pragma unroll
  for (int s0 = 0; s0 < (G2 ? 2 : NSTAGE - 1); ++s0)
    if (s0 < nslab) dma(s0, s0);

  f32x16_t lg, zp;   // logits^T[class, pair]; first-layer accumulator of the previous slab
#pragma unroll
  for (int r = 0; r < 16; ++r) { lg[r] = 0.f; zp[r] = 0.f; }
  Frag<T> w2p0, w2p1;  // second-layer fragments of the previous slab (zero for the dummy slab -1)
  {
    float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    w2p0 = pack_frag8<T>(zero); w2p1 = w2p0;
  }
  const uint32_t drop_key = DROP ? pair_drop_key(p.drop_seed, b) : 0u;
  const uint32_t drop_base = (uint32_t)(mypair * nslab * 2) + (uint32_t)half;
  auto second_layer = [&](const f32x16_t& zz, int bias_slab, const Frag<T>& wa_, const Frag<T>& wb_) {
    float y[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // accumulator rows 8g + 4*half + 0..3 -> one 16-byte bias read
      if constexpr (NO_EPI) {
        y[4 * g + 0] = zz[4 * g + 0]; y[4 * g + 1] = zz[4 * g + 1]; y[4 * g + 2] = zz[4 * g + 2]; y[4 * g + 3] = zz[4 * g + 3];
      } else {
        const float4 bv = *reinterpret_cast<const float4*>(sB1 + bias_slab * 32 + 8 * g + 4 * half);
        y[4 * g + 0] = silu_f(zz[4 * g + 0] + bv.x);
        y[4 * g + 1] = silu_f(zz[4 * g + 1] + bv.y);
        y[4 * g + 2] = silu_f(zz[4 * g + 2] + bv.z);
        y[4 * g + 3] = silu_f(zz[4 * g + 3] + bv.w);
      }
    }
    if constexpr (DROP) {
      // K12 dropout (peneo_decoder.py:261): this lane's 16 hidden units of the slab (rows 8g + 4 half + e, y[4g + e]) are the
      // 16 fields of ONE chain (common.h); the 1 / (1 - p) factor is applied to the logits
      uint32_t st = pair_drop_seed(drop_key, drop_base + 2u * (uint32_t)bias_slab);
      const uint32_t inc = pair_drop_inc(drop_key, drop_base + 2u * (uint32_t)bias_slab);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        st = pair_drop_step(st, inc);
        y[i] = (st >> 16) >= p.drop_thr16 ? y[i] : 0.f;
      }
    }
    Frag<T> y0 = pack_frag8<T>(y), y1 = pack_frag8<T>(y + 8);
    mma_step(wa_, y0, lg);
    mma_step(wb_, y1, lg);
  };
  Frag<T> fixed = load_frag_linear<T>(sW, 0, lane);

  for (int slab = 0; slab < nslab; ++slab) {
    if constexpr (NOSTREAM) {
    } else if constexpr (G2) {
      // the waves meet only every second slab: between barriers they drift apart, so one wave's SiLU epilogue (VALU)
      // runs under the other wave's MFMAs on the same SIMD instead of both doing the same phase at the same time
      if ((slab & 1) == 0) {
        wait_vm<0>();                        // slabs slab, slab+1 (issued two slabs ago) have landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (slab + 2 < nslab) dma(slab + 2, (slab + 2) % NSTAGE);
        if (slab + 3 < nslab) dma(slab + 3, (slab + 3) % NSTAGE);
      }
    } else if constexpr (ASM_DMA) {
      // slabs slab .. slab+NSTAGE-2 are in flight (only `slab` itself at the very end): wait for the oldest one
      if (NSTAGE > 2 && slab + 1 < nslab) wait_vm<(NSTAGE - 2) * UPW>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      __syncthreads();   // hipcc places s_waitcnt vmcnt(0) in front of it while a DMA it knows of is in flight
    }
    if (!NOSTREAM && !G2 && slab + NSTAGE - 1 < nslab) dma(slab + NSTAGE - 1, (slab + NSTAGE - 1) % NSTAGE);
    const char* wb = sW + (slab % NSTAGE) * SLAB_BYTES;
    if constexpr (PIPE_EPI) second_layer(zp, slab > 0 ? slab - 1 : 0, w2p0, w2p1);
    f32x16_t z, z2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { z[r] = 0.f; z2[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Frag<T> wf = NO_LDS ? fixed : load_frag_linear<T>(wb, ks, lane);
      if (DUAL && (ks & 1)) mma_step(wf, xf[ks], z2); else mma_step(wf, xf[ks], z);
    }
    if constexpr (DUAL) {
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] += z2[r];
    }
    if constexpr (SGB) {
      // ask the scheduler for a fixed interleave inside this block: the VALU of the pipelined epilogue (and the LDS
      // reads) are spread between the MFMAs instead of being issued as one burst while the matrix pipe idles
#pragma unroll
      for (int i = 0; i < KS + 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);   // up to 7 VALU
      }
    }
    if constexpr (PIPE_EPI) {
      w2p0 = load_frag_linear<T>(wb, KS, lane);
      w2p1 = load_frag_linear<T>(wb, KS + 1, lane);
      zp = z;
    } else {
      Frag<T> w2a = load_frag_linear<T>(wb, KS, lane), w2b = load_frag_linear<T>(wb, KS + 1, lane);
      second_layer(z, slab, w2a, w2b);
    }
  }
  if constexpr (PIPE_EPI) second_layer(zp, nslab - 1, w2p0, w2p1);
  if constexpr (!ASM_DMA) __syncthreads();
  pair_epilogue(p, lg, smem, tid, lane, wave, half, b, mypair, pair_ok);
}

// ================================================================================================
// building blocks of the chunked backward
// ================================================================================================
template <typename T>
__global__ void pair_x_fwd_kernel(const T* abd, int N, int D, int i0, int64_t pbase, int64_t npairs, T* x, T* pre) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = D / VEC;
  const int64_t total = npairs * vpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pr = idx / vpr;
    const int c = (int)(idx % vpr) * VEC;
    int i, j;
    pair_decode(pbase + pr, N, i, j);
    float a[VEC], b[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(abd + (int64_t)i * 2 * D + c), a);
    unpack16<T>(*reinterpret_cast<const uint4*>(abd + (int64_t)j * 2 * D + D + c), b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] += b[e];
    if (pre) *reinterpret_cast<uint4*>(pre + pr * D + c) = pack16<T>(a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] = silu_f(a[e]);
    *reinterpret_cast<uint4*>(x + pr * D + c) = pack16<T>(a);
  }
}

// row-based variant (16-byte aligned, D / VEC <= 256): block = (row i, slice of 64 columns j); a_i stays in registers,
// no per-element pair decode, every store a full 16-byte vector of a contiguous pair row
constexpr int XF_JCH = 64;
template <typename T>
__global__ __launch_bounds__(256) void pair_x_fwd_rows_kernel(const T* abd, int N, int D, int i0, int64_t pbase, T* x, T* pre) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = D / VEC, nj = 256 / vpr;
  const int cv = threadIdx.x % vpr, jl = threadIdx.x / vpr;
  if (jl >= nj) return;
  const int i = i0 + blockIdx.x;
  const int jb = i + blockIdx.y * XF_JCH, je = min(N, jb + XF_JCH);
  if (jb >= N) return;
  const int c = cv * VEC;
  float a[VEC];
  unpack16<T>(*reinterpret_cast<const uint4*>(abd + (int64_t)i * 2 * D + c), a);
  const int64_t prow = pair_row_start(i, N) - pbase - i;
  for (int j = jb + jl; j < je; j += nj) {
    float b[VEC], s[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(abd + (int64_t)j * 2 * D + D + c), b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) b[e] += a[e];
    if (pre) *reinterpret_cast<uint4*>(pre + (prow + j) * D + c) = pack16<T>(b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = silu_f(b[e]);
    *reinterpret_cast<uint4*>(x + (prow + j) * D + c) = pack16<T>(s);
  }
}

// d_a[i, k] += sum_{j >= i} dx[p(i,j), k] * SiLU'(a_i[k] + b_j[k]).  Rows of the triangle are contiguous runs of
// dx rows; a 64-thread block owns (row i, one of JSPLIT slices of j), each thread 8 (bf16) / 4 (fp32) columns with
// 16-byte loads, 4 rows in flight; the JSPLIT partial sums meet in fp32 atomics (JSPLIT * D per row: negligible).
constexpr int JSPLIT = 4;
template <typename T, bool PRE>   // PRE: dx already carries the SiLU'(a_i + b_j) factor (applied by the dx GEMM epilogue)
__global__ __launch_bounds__(256) void pair_x_bwd_a_kernel(const T* abd, int N, int D, int i0, int64_t pbase, const T* dx,
                                                          float* d_ab) {
  constexpr int VEC = Elem<T>::kVec;
  const int i = i0 + blockIdx.x;
  const int c = threadIdx.x * VEC;
  if (c >= D) return;
  const int len = N - i;
  const int per = (len + JSPLIT - 1) / JSPLIT;
  const int j0 = i + blockIdx.y * per, j1 = min(N, j0 + per);
  if (j0 >= j1) return;
  const int64_t prow = pair_row_start(i, N) - pbase;
  float a[VEC], s[VEC];
  unpack16<T>(*reinterpret_cast<const uint4*>(abd + (int64_t)i * 2 * D + c), a);
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  for (int j = j0; j < j1; j += 4) {
    uint4 dv[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jj = min(j + u, j1 - 1);
      dv[u] = *reinterpret_cast<const uint4*>(dx + (prow + (jj - i)) * D + c);
      if constexpr (!PRE) bv[u] = *reinterpret_cast<const uint4*>(abd + (int64_t)jj * 2 * D + D + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j + u < j1) {
        float d[VEC], bj[VEC];
        unpack16<T>(dv[u], d);
        if constexpr (PRE) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) s[e] += d[e];
        } else {
          unpack16<T>(bv[u], bj);
#pragma unroll
          for (int e = 0; e < VEC; ++e) s[e] += d[e] * silu_grad_f(a[e] + bj[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) atomicAdd(d_ab + (int64_t)i * 2 * D + c + e, s[e]);
}
// d_b[j, k] += sum_{i0 <= i < i1, i <= j} dx[p(i,j), k] * SiLU'(a_i[k] + b_j[k]): block = (column j, slice of i)
template <typename T, bool PRE>
__global__ __launch_bounds__(256) void pair_x_bwd_b_kernel(const T* abd, int N, int D, int i0, int i1, int64_t pbase, const T* dx,
                                                          float* d_ab) {
  constexpr int VEC = Elem<T>::kVec;
  const int j = i0 + blockIdx.x;  // columns below i0 have no pair in this chunk
  const int c = threadIdx.x * VEC;
  if (c >= D) return;
  const int iend = min(i1, j + 1);
  const int per = (iend - i0 + JSPLIT - 1) / JSPLIT;
  const int ia = i0 + blockIdx.y * per, ib = min(iend, ia + per);
  if (ia >= ib) return;
  float bj[VEC], s[VEC];
  unpack16<T>(*reinterpret_cast<const uint4*>(abd + (int64_t)j * 2 * D + D + c), bj);
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  for (int i = ia; i < ib; i += 4) {
    uint4 dv[4], av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ii = min(i + u, ib - 1);
      dv[u] = *reinterpret_cast<const uint4*>(dx + (pair_row_start(ii, N) + (j - ii) - pbase) * D + c);
      if constexpr (!PRE) av[u] = *reinterpret_cast<const uint4*>(abd + (int64_t)ii * 2 * D + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u < ib) {
        float d[VEC], ai[VEC];
        unpack16<T>(dv[u], d);
        if constexpr (PRE) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) s[e] += d[e];
        } else {
          unpack16<T>(av[u], ai);
#pragma unroll
          for (int e = 0; e < VEC; ++e) s[e] += d[e] * silu_grad_f(ai[e] + bj[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) atomicAdd(d_ab + (int64_t)j * 2 * D + D + c + e, s[e]);
}

struct DzParams {
  peneo_pair_dz_args a;
};
constexpr int DZ_SLOTS = 1024;  // rows of the partial-sum workspace == max workgroups per launch (4 per CU)
// z -> dz in place; every thread owns one 16-byte column vector of the [npairs, nh*D] matrix for a strided set of rows and
// keeps its dW2 / db1 partial sums in registers; one plain read-modify-write of the block's workspace row at the end
// (workspace [DZ_SLOTS][4 * nh*D]: rows c*ncol.. hold sum_p dlogits[p,c]*y[p,:] for c = 0..2, row 3*ncol.. holds sum_p dz).
template <typename T>
__global__ __launch_bounds__(256) void pair_dz_kernel(T* z, int64_t npairs, DzParams pp, float* ws) {
  constexpr int VEC = Elem<T>::kVec;
  const peneo_pair_dz_args& a = pp.a;
  const int ncol = a.num_heads * a.D;
  const int nvec = ncol / VEC;
  float* slot = ws + (int64_t)(blockIdx.x % DZ_SLOTS) * 4 * ncol;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    const int col = v * VEC;
    const int h = col / a.D, k = col - h * a.D;
    const int C = a.classes[h];
    float w2[3][VEC], dw2[3][VEC], db1[VEC];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { w2[c][e] = (c < C) ? a.w2[h][(int64_t)c * a.D + k + e] : 0.f; dw2[c][e] = 0.f; }
#pragma unroll
    for (int e = 0; e < VEC; ++e) db1[e] = 0.f;
    const float sc = a.scale[h];
    const float* dl = a.dlogits[h];
    // K12 dropout of the forward (common.h): regenerated per element; this kernel is the chunked / fp32 path, speed is secondary
    const uint32_t thr16 = pair_drop_thr16_dev(a.drop_p);
    const uint32_t dkey = pair_drop_key(a.drop_seed, a.drop_doc);
    const float dscale = thr16 ? 65536.f / (65536.f - (float)thr16) : 1.f;
    // four rows in flight per step (every load issued before the first use): the loop is latency-bound otherwise
    for (int64_t r0 = blockIdx.x; r0 < npairs; r0 += 4 * (int64_t)gridDim.x) {
      uint4 raw[4]; float g[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = min(r0 + u * (int64_t)gridDim.x, npairs - 1);
        raw[u] = *reinterpret_cast<const uint4*>(z + r * ncol + col);
#pragma unroll
        for (int c = 0; c < 3; ++c) g[u][c] = (c < C) ? dl[r * C + c] * sc : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u * (int64_t)gridDim.x;
        if (r < npairs) {
          float zv[VEC], o[VEC];
          if constexpr (sizeof(T) == 2) unpack16<T>(raw[u], zv);
          else unpack16<T>(raw[u], zv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float sg = sigmoid_f(zv[e]);
            const float keep = (!thr16 || pair_drop_keep(dkey, a.drop_pair0 + r, col + e, ncol / 32, thr16)) ? dscale : 0.f;
            const float y = zv[e] * sg * keep;
            const float dy = fmaf(g[u][2], w2[2][e], fmaf(g[u][1], w2[1][e], g[u][0] * w2[0][e]));
            const float dz = dy * keep * (sg * fmaf(zv[e], 1.f - sg, 1.f));
            dw2[0][e] = fmaf(g[u][0], y, dw2[0][e]); dw2[1][e] = fmaf(g[u][1], y, dw2[1][e]); dw2[2][e] = fmaf(g[u][2], y, dw2[2][e]);
            db1[e] += dz;
            o[e] = dz;
          }
          *reinterpret_cast<uint4*>(z + r * ncol + col) = pack16<T>(o);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {   // one block per slot row: plain read-modify-write
      slot[0 * ncol + col + e] += dw2[0][e];
      slot[1 * ncol + col + e] += dw2[1][e];
      slot[2 * ncol + col + e] += dw2[2][e];
      slot[3 * ncol + col + e] += db1[e];
    }
  }
}

// partials [n][32] (num[8] | den[8] | dl_sum[16]) -> per-head losses, total, backward scales, dl_sum
__global__ __launch_bounds__(1024) void loss_finish_kernel(const float* partials, int64_t n, const float* ratio, int nh,
                                                           int total_classes, float* out, float* scale, float* dl_sum,
                                                           float* inv_den) {
  // one block (the sums must come out in one fixed order): 128 row lanes x 8 column quads, 16-byte loads, four rows in flight
  // per thread (the 4-byte, one-row-at-a-time form took 33-43 us for 4088 rows between the forward and the backward)
  __shared__ float red[32][33];
  __shared__ float4 part[128][8];
  const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p4 = reinterpret_cast<const float4*>(partials);
  int64_t i = rl;
  for (; i + 384 < n; i += 512) {
    const float4 a = p4[i * 8 + q], b = p4[(i + 128) * 8 + q], c = p4[(i + 256) * 8 + q], d = p4[(i + 384) * 8 + q];
    s4.x += (a.x + b.x) + (c.x + d.x); s4.y += (a.y + b.y) + (c.y + d.y);
    s4.z += (a.z + b.z) + (c.z + d.z); s4.w += (a.w + b.w) + (c.w + d.w);
  }
  for (; i < n; i += 128) { const float4 a = p4[i * 8 + q]; s4.x += a.x; s4.y += a.y; s4.z += a.z; s4.w += a.w; }
  part[rl][q] = s4;
  __syncthreads();
  const int col = threadIdx.x & 31, row = threadIdx.x >> 5;   // 32 x 32: fold the 128 row lanes (4 per thread), then 32 -> 1
  {
    float t = 0.f;
    for (int r = row; r < 128; r += 32) t += reinterpret_cast<const float*>(&part[r][0])[col];
    red[row][col] = t;
  }
  __syncthreads();
  if (row == 0) {
    float t = 0.f;
    for (int r = 0; r < 32; ++r) t += red[r][col];
    red[0][col] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int h = 0; h < nh; ++h) {
      const float l = red[0][h] / red[0][8 + h];
      out[h] = l;
      tot += ratio[h] * l;
      if (scale) scale[h] = ratio[h] / red[0][8 + h];
      if (inv_den) inv_den[h] = 1.0f / red[0][8 + h];
    }
    out[nh] = tot;
  }
  if (dl_sum && threadIdx.x < total_classes) dl_sum[threadIdx.x] = red[0][16 + threadIdx.x];
}

__global__ void weighted_ce_kernel(const float* logits, const int64_t* tags, const float* cw, int64_t rows, int C, float* num,
                                   float* den, float* dlogits) {
  float n = 0.f, d = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float* l = logits + r * C;
    const int tag = (int)tags[r];
    float mx = l[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(l[c] - mx);
    const float lse = mx + __logf(se);
    const bool valid = tag >= 0 && tag < C;                 // ignore_index / garbage labels: zero weight
    const float w = !valid ? 0.f : (cw ? cw[tag] : 1.f);
    n += valid ? w * (lse - l[tag]) : 0.f;
    d += w;
    if (dlogits)
      for (int c = 0; c < C; ++c) dlogits[r * C + c] = w * (__expf(l[c] - lse) - (c == tag ? 1.f : 0.f));
  }
  n = wave_sum(n); d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) { atomicAdd(num, n); atomicAdd(den, d); }
}

// K14: argmax != 0 compaction, in increasing p order (one block; P is ~1e5 so a single-block scan is fine)
__global__ __launch_bounds__(1024) void spots_compact_kernel(const float* logits, int64_t P, int C, int N, int32_t* spots,
                                                             float* scores, int32_t* count, int max_spots) {
  __shared__ int wsum[16];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t p0 = 0; p0 < P; p0 += 1024) {
    const int64_t pp = p0 + threadIdx.x;
    int tag = 0; float sc = 0.f;
    if (pp < P) {
      const float* l = logits + pp * C;
      float mx = l[0]; int am = 0;
      for (int c = 1; c < C; ++c) if (l[c] > mx) { mx = l[c]; am = c; }
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += __expf(l[c] - mx);
      tag = am; sc = 1.f / se;
    }
    const unsigned long long bal = __ballot(tag != 0);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
    if (tag != 0) {
      const int slot = base + woff + before;
      if (slot < max_spots) {
        int i, j;
        pair_decode(pp, N, i, j);
        spots[slot * 3 + 0] = i; spots[slot * 3 + 1] = j; spots[slot * 3 + 2] = tag;
        scores[slot] = sc;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = base;
}

// ================================================================================================
// Hand-interleaved form (bf16, D = 384 / 512; round 4).  The generic kernel above runs a slab as [24 first-layer MFMAs] then
// [bias + SiLU + dropout + second layer: ~200 VALU slots]; its eight waves meet at the slab's barrier, so the two waves of a
// SIMD do the same phase at the same time and neither the matrix pipe nor the VALU ever works beside the other (4400 cycles per
// slab for 1664 of MFMA and ~2000 of VALU per SIMD).  Here the epilogue of slab s-1 is cut into 24 pieces and ONE piece stands
// behind every first-layer MFMA of slab s (a wave has one MFMA in flight; what stands between two MFMAs issues in the first one's
// shadow).  MFMAs, fragment reads and waits are ordered statements: the compiler moves builtin MFMAs wherever its scheduler
// likes (round 4 found the whole chain sunk behind the last wait in the pair backward).  LDS operations of a slab, in order:
//   bias x 4 (previous slab) | R0 R1 R2 | slot k: [wait: <= min(2, KS-1-k) reads outstanding] MFMA(k) R(k+3) piece(s) | W2 x 2
// An accumulator is read by the VALU one whole slab (a barrier) after the chain that wrote it (two waves share the matrix pipe: the
// wait states a compiler would insert assume an MFMA starts when it is issued).
// ================================================================================================
#ifdef PH_PROF
__device__ unsigned long long* g_ph_prof;
#endif
#ifndef PH_ABLATE
#define PH_ABLATE 0   // timing experiments (tools/ab_pair_fwd_save.sh): 1 the saving form without its record stores, 2 without its x stores, 4 record stores plain instead of non-temporal, 8 no exp / rcp in the epilogue (timing only)
#endif
#ifndef PH_HAND_LA
#define PH_HAND_LA 3
#endif
typedef unsigned int ph_u32x4 __attribute__((ext_vector_type(4)));
// ("+v": the destination counts as live before the read, so the register allocator cannot fold two of the rotating fragment sets
// into one - a set is re-loaded only after the NEXT slot's MFMA has been issued behind the one that read it)
template <int OFF> __device__ __forceinline__ void ph_dsr(ph_u32x4& d, uint32_t a) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(d) : "v"(a), "n"(OFF));
}
__device__ __forceinline__ void ph_mma(const ph_u32x4& a, const uint4& b, f32x16_t& acc) {
  const ph_u32x4 bv = ph_u32x4{b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(bv));
}
__device__ __forceinline__ void ph_mma0(const ph_u32x4& a, const uint4& b, f32x16_t& acc) {   // acc = a b
  const ph_u32x4 bv = ph_u32x4{b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(bv));
}
// second layer: the B operand was written by the VALU just before (the compiler does not know this statement is an MFMA)
__device__ __forceinline__ void ph_mma_late(const ph_u32x4& a, const ph_u32x4& b, f32x16_t& acc) {
  asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// ... and the LAST one of the kernel: the compiler copies the accumulator as soon as the statement is over, and does not know that
// the result takes 8 passes (+ whatever a partner wave has queued in front) to arrive: the wait states are part of the statement
__device__ __forceinline__ void ph_mma_late_final(const ph_u32x4& a, const ph_u32x4& b, f32x16_t& acc) {
  asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
               : "+v"(acc) : "v"(a), "v"(b));
}
template <int N, int I = 0, typename F> __device__ __forceinline__ void ph_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ph_static_for<N, I + 1>(f); }
}

template <int KS, bool DROP, bool SAVE>
__global__ __launch_bounds__(PH_WAVES * 64, 2) void pair_heads_fwd_hand_kernel(PairFwdParams p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = KS + 2;
  // four slab buffers where they fit (D = 384): the request for slab + 2 then goes out BEFORE the barrier (see `top`)
  constexpr int NSTAGE = ((NF + PH_WAVES - 1) / PH_WAVES) * PH_WAVES * 1024 * 4 + 5 * 512 * 4 <= 160 * 1024 ? 4 : 3;
  constexpr int UPW = (NF + PH_WAVES - 1) / PH_WAVES;               // 1 KiB DMA pieces per wave per slab
  constexpr int SLAB_BYTES = UPW * PH_WAVES * 1024;                 // == packed slab stride
  static_assert(KS % 4 == 0 && KS >= 8 && UPW <= 12, "hand form: D = 128 .. 512 in steps of 64");
  // fragment reads run LA slots ahead of their MFMA through LA + 1 rotating register sets (what the register budget allows)
  constexpr int LA = KS <= 24 ? PH_HAND_LA : 3, NS = LA + 1;
  char* sW = smem;                                                   // [NSTAGE][SLAB_BYTES]
  float* sB1 = reinterpret_cast<float*>(smem + NSTAGE * SLAB_BYTES); // [nh * D]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D, N = p.N;
  const int b = blockIdx.y;
  int64_t mypair;
  bool pair_ok, tile_ok = false;
  int pi, pj, tile = 0;
  if constexpr (SAVE) {
    // the backward's walk: a workgroup = two blocks of 8 x 16 pairs, wave = group (rows 2g, 2g + 1 of its block) - pair_bwd.hip
    const int T2 = 2 * (int)blockIdx.x + (wave >> 2), grp = wave & 3, r32 = lane & 31;
    tile_ok = T2 < p.ntiles;
    tile = tile_ok ? T2 : p.ntiles - 1;
    int ti = 0;
    const int nti = pb_row_tiles(N);
    while (ti + 1 < nti && pb_tiles_before(ti + 1, N) <= tile) ++ti;
    const int tj = (ti >> 1) + (tile - pb_tiles_before(ti, N));
    const int qi = ti * PB_TI + 2 * grp + (r32 >> 4), qj = tj * PB_TJ + (r32 & 15);
    pair_ok = tile_ok && qi < N && qj < N && qi <= qj;
    pi = min(qi, N - 1); pj = min(qj, N - 1);          // (rows outside the triangle: finite x / q / y, their dlogits are never read)
    mypair = pair_ok ? pair_row_start(pi, N) + (pj - pi) : p.P - 1;
  } else {
    const int64_t p0 = (int64_t)blockIdx.x * PH_PAIRS;
    mypair = p0 + wave * 32 + (lane & 31);
    pair_ok = mypair < p.P;
    pair_decode(pair_ok ? mypair : p.P - 1, N, pi, pj);
  }
  const int nslab = p.num_heads * D / 32;

  for (int i = tid; i < p.num_heads * D; i += PH_WAVES * 64) sB1[i] = p.b1[i];

  // ---- x = SiLU(a_i + b_j) as B-operand fragments (k = decoder dim), kept in registers (as in the generic kernel) ----
  const T* abd = reinterpret_cast<const T*>(p.ab) + (int64_t)b * N * 2 * D;
  const T* arow = abd + (int64_t)pi * 2 * D;
  const T* brow = abd + (int64_t)pj * 2 * D + D;
  Frag<T> xf[KS];
  {
    uint4 ra[2][4], rb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[0][i] = *reinterpret_cast<const uint4*>(arow + 16 * i + 8 * half);
      rb[0][i] = *reinterpret_cast<const uint4*>(brow + 16 * i + 8 * half);
    }
#pragma unroll
    for (int g = 0; g < KS / 4; ++g) {
      if (g + 1 < KS / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ra[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(arow + 16 * (4 * (g + 1) + i) + 8 * half);
          rb[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(brow + 16 * (4 * (g + 1) + i) + 8 * half);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a[8], bb[8];
        unpack16<T>(ra[g & 1][i], a);
        unpack16<T>(rb[g & 1][i], bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e] + bb[e]);
        xf[4 * g + i] = pack_frag8<T>(a);
        asm volatile("" : "+v"(xf[4 * g + i].v.x), "+v"(xf[4 * g + i].v.y), "+v"(xf[4 * g + i].v.z), "+v"(xf[4 * g + i].v.w) :: "memory");
      }
    }
  }
  if constexpr (SAVE) {
    // x rows of the group's 32 pairs in block order (the B operand of the backward's dW1 = dz^T x).  A lane owns 16 bytes of every
    // 32-byte piece of its pair's row: stored from the fragments directly, an instruction writes 64 scattered 16-byte pieces (0.11 ms
    // of the launch and 330 MB of line fills).  Eight fragments at a time go through the wave's corner of the still empty weight ring
    // (rows of 256 + 16 bytes) and leave as four 256-byte runs per instruction.
    constexpr int XPITCH = 272;
    char* const sX = smem + wave * (32 * XPITCH);
    static_assert(PH_WAVES * 32 * XPITCH <= NSTAGE * SLAB_BYTES, "the x staging area lives in the weight ring");
    T* const xdst = p.x_save + (((int64_t)b * p.ntiles + tile) * PB_ROWS + (wave & 3) * 32) * D;
#pragma unroll
    for (int pass = 0; pass < KS / 8; ++pass) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sX + (lane & 31) * XPITCH + 32 * i + 16 * half) = xf[8 * pass + i].v;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4), c = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(sX + r * XPITCH + 16 * c);
        if (tile_ok && !(PH_ABLATE & 2)) *reinterpret_cast<uint4*>(xdst + (int64_t)r * D + 128 * pass + 8 * c) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // from here on the vm counter only sees the DMA pieces (+ the record stores)
  __syncthreads();                                          // sB1 visible

  const char* wsrc = reinterpret_cast<const char*>(p.wp) + wave * (UPW * 1024) + lane * 16;
  const uint32_t wdst = lds_addr(sW) + wave * (UPW * 1024);
  auto dma = [&](int slab_, int buf_) { lds_dma_units<0, UPW>(wsrc + (int64_t)slab_ * SLAB_BYTES, wdst + buf_ * SLAB_BYTES); };
  dma(0, 0);
  dma(min(1, nslab - 1), 1);

  f32x16_t lg, z0, z1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { lg[r] = 0.f; z0[r] = 0.f; z1[r] = 0.f; }
  ph_u32x4 w2a = ph_u32x4{0u, 0u, 0u, 0u}, w2b = w2a;        // second-layer fragments of the slab whose epilogue is pending
  const uint32_t drop_key = DROP ? pair_drop_key(p.drop_seed, b) : 0u;
  const uint32_t drop_base = (uint32_t)(mypair * nslab * 2) + (uint32_t)half;
  const uint32_t thr32 = p.drop_thr16 << 16;                // field = bits 16.. of the chain state
  const uint32_t sW_l = lds_addr(sW) + lane * 16, sB1_l = lds_addr(sB1) + half * 16;
  // records of this wave's group: slab s at + s * 4 groups * PB_REC_BYTES; the lane's 16-byte pieces at row (lane & 31), byte 16 half (+ 32)
  // (tile layout, common.h: two 1 KiB halves of 16 units each, 32-byte rows, the second half's rows XORed with 4: every store below is
  // 1 KiB contiguous, and the backward's transposing reads of the two halves fall on different banks)
  char* const rec_l = SAVE ? p.act + ((((int64_t)b * p.ntiles + tile) * nslab * 4 + (wave & 3)) * PB_REC_BYTES + 16 * half) : nullptr;
  const int rec_r0 = (lane & 31) * 32, rec_r1 = 1024 + ((lane & 31) ^ 4) * 32;

  // K12 dropout.  A dropped unit's accumulator STARTS at -30000 instead of its bias: SiLU(-30000 + ...) = -0, so y needs no mask of
  // its own and the pre-activation the saving form stores is already masked (the backward's SiLU' of it is 0 as well).  The start
  // values of slab s + 1 are written during the epilogue of slab s - 1, into the accumulator registers that epilogue has just read
  // (they are the registers slab s + 1 accumulates in): one v_cndmask per element, as the mask on y was.
  auto start_values = [&](int slab_, f32x16_t& acc) {       // plain form, before the loop (slabs 0 and 1)
    uint32_t st_ = pair_drop_seed(drop_key, drop_base + 2u * (uint32_t)slab_);
    const uint32_t inc_ = pair_drop_inc(drop_key, drop_base + 2u * (uint32_t)slab_);
    const float* bsrc = sB1 + slab_ * 32 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *reinterpret_cast<const float4*>(bsrc + 8 * g);
      const float be[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { st_ = pair_drop_step(st_, inc_); acc[4 * g + e] = st_ >= thr32 ? be[e] : -30000.f; }
    }
  };
  // one slab: DOM = first-layer MFMAs of `slab` into zw; DOE = epilogue of slab - 1 (its accumulator: zr)
  auto body = [&](auto m_c, auto e_c, int slab, f32x16_t& zw, f32x16_t& zr) {
    constexpr bool DOM = decltype(m_c)::value, DOE = decltype(e_c)::value;
    const uint32_t wb = sW_l + (slab % NSTAGE) * SLAB_BYTES;
    // (registers the "+v" ties below want a value in are DEFINED by an empty statement, not by an instruction: the zero moves this
    // replaces were 26 VALU slots per slab)
    ph_u32x4 bv[4], fs[NS];
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(bv[i]));
#pragma unroll
    for (int i = 0; i < NS; ++i) asm volatile("" : "=v"(fs[i]));
    uint32_t st = 0u, sinc = 0u;
    float t[4], u[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { asm volatile("" : "=v"(t[e])); asm volatile("" : "=v"(u[e])); }
    uint32_t yp[8], zp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" : "=v"(yp[i])); if constexpr (SAVE) asm volatile("" : "=v"(zp[i])); }
    char* const rec = SAVE ? rec_l + (int64_t)(slab - 1) * (4 * PB_REC_BYTES) : nullptr;
    // groups G, G + 1 done: the two halves of a pair's lanes hold 4 + 4 consecutive units of each group; a v_permlane32_swap pair makes
    // 16 contiguous bytes per lane (lanes 0-31: group G, lanes 32-63: group G + 1)
    auto save_rows = [&](auto gc) {
      constexpr int G = decltype(gc)::value;
      const auto zx = __builtin_amdgcn_permlane32_swap(zp[2 * G], zp[2 * G + 2], false, false);
      const auto zy = __builtin_amdgcn_permlane32_swap(zp[2 * G + 1], zp[2 * G + 3], false, false);
      if (tile_ok && !(PH_ABLATE & 1)) {
        typedef unsigned int ph_u4 __attribute__((ext_vector_type(4)));
        ph_u4* const dst = reinterpret_cast<ph_u4*>(rec + (G == 0 ? rec_r0 : rec_r1));
        const ph_u4 val = ph_u4{zx[0], zy[0], zx[1], zy[1]};
        // (non-temporal: written once, read by the backward after everything else of the forward has gone through the L2 -
        // 16.39 -> 16.32 ms per step, 3 of 4 interleaved pairs; PH_ABLATE & 4: plain stores)
        if constexpr ((PH_ABLATE & 4) != 0) *dst = val; else __builtin_nontemporal_store(val, dst);
      }
      if constexpr ((PH_ABLATE & 1) != 0) asm volatile("" :: "v"(zx[0]), "v"(zy[0]), "v"(zx[1]), "v"(zy[1]));
    };
    // DROP: the chain run here belongs to slab + 1 (see start_values)
    constexpr bool NEXT = DROP && DOM && DOE;
    if constexpr (NEXT) { st = pair_drop_seed(drop_key, drop_base + 2u * (uint32_t)(slab + 1)); sinc = pair_drop_inc(drop_key, drop_base + 2u * (uint32_t)(slab + 1)); }
    if constexpr (DOM) {
      // The chain's accumulator STARTS at the first-layer bias: registers 4g .. 4g+3 = hidden rows 8g + 4 half + 0..3 of the slab, one
      // 16-byte read per group (no dropout: straight into the accumulator's registers; dropout: the bias rows of slab + 1 for the selects
      // below) - the epilogue needs no bias add (16 VALU slots per slab)
      const uint32_t ba = sB1_l + (uint32_t)(DROP ? min(slab + 1, nslab - 1) : slab) * 128;
      ph_dsr<0>(bv[0], ba); ph_dsr<32>(bv[1], ba); ph_dsr<64>(bv[2], ba); ph_dsr<96>(bv[3], ba);
      ph_static_for<LA>([&](auto ic) { constexpr int I = decltype(ic)::value; ph_dsr<I * 1024>(fs[I], wb); });
    }
    // piece q of the epilogue: group g = q / 6 (accumulator registers 4g .. 4g+3), step q % 6
    auto piece = [&](auto qc) {
      constexpr int Q = decltype(qc)::value, G = Q / 6, S = Q % 6;
      if constexpr (S == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { t[e] = zr[4 * G + e]; u[e] = t[e] * -1.4426950408889634f; }
        if constexpr (SAVE) { zp[2 * G] = pack_f16x2(t[0], t[1]); zp[2 * G + 1] = pack_f16x2(t[2], t[3]); }
      } else if constexpr (S == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (PH_ABLATE & 8) ? u[e] : __builtin_amdgcn_exp2f(u[e]);
      } else if constexpr (S == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = u[e] + 1.f;
        if constexpr (!(PH_ABLATE & 8)) { u[0] = __builtin_amdgcn_rcpf(u[0]); u[1] = __builtin_amdgcn_rcpf(u[1]); }
      } else if constexpr (S == 3) {
        if constexpr (!(PH_ABLATE & 8)) { u[2] = __builtin_amdgcn_rcpf(u[2]); u[3] = __builtin_amdgcn_rcpf(u[3]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = t[e] * u[e];          // y = z sigmoid(z)   (-0 where the unit is dropped)
      } else if constexpr (S == 4) {
        if constexpr (NEXT) {
          // the lane's 16 hidden units of a slab are the 16 fields of ONE chain, in register order
          const float be[3] = {__uint_as_float(bv[G].x), __uint_as_float(bv[G].y), __uint_as_float(bv[G].z)};
#pragma unroll
          for (int e = 0; e < 3; ++e) { st = pair_drop_step(st, sinc); zr[4 * G + e] = st >= thr32 ? be[e] : -30000.f; }
        }
      } else {
        if constexpr (NEXT) { st = pair_drop_step(st, sinc); zr[4 * G + 3] = st >= thr32 ? __uint_as_float(bv[G].w) : -30000.f; }
        yp[2 * G] = pack_bf16x2(t[0], t[1]); yp[2 * G + 1] = pack_bf16x2(t[2], t[3]);
      }
      // anchor: the piece stays between the MFMAs it was written between
      asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(st), "+v"(sinc));
      if constexpr (S == 5) asm volatile("" : "+v"(yp[2 * G]), "+v"(yp[2 * G + 1]));
      if constexpr (S == 0 && SAVE) asm volatile("" : "+v"(zp[2 * G]), "+v"(zp[2 * G + 1]));
      if constexpr (S >= 4 && NEXT) asm volatile("" : "+v"(zr));
      // groups 0, 1 done: y rows 0..15 of the slab are complete -> the first second-layer MFMA
      if constexpr (Q == 11) ph_mma_late(w2a, ph_u32x4{yp[0], yp[1], yp[2], yp[3]}, lg);
      if constexpr (Q == 23) {
        if constexpr (DOM) ph_mma_late(w2b, ph_u32x4{yp[4], yp[5], yp[6], yp[7]}, lg);
        else ph_mma_late_final(w2b, ph_u32x4{yp[4], yp[5], yp[6], yp[7]}, lg);
      }
      if constexpr (SAVE && Q == 11) save_rows(std::integral_constant<int, 0>{});
      if constexpr (SAVE && Q == 23) save_rows(std::integral_constant<int, 2>{});
    };
    if constexpr (DOM) {
      ph_static_for<KS>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr int younger = (KS - 1 - K) < LA - 1 ? (KS - 1 - K) : LA - 1;
        asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fs[K % NS]) : [n] "n"(younger) : "memory");
        // (the bias rows were requested in front of the fragments: they have landed with the first of these waits; the empty
        // statement keeps the compiler from using their registers any earlier)
        if constexpr (K == 0) {
          asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
          if constexpr (!DROP)
            zw = __builtin_bit_cast(f32x16_t, __builtin_shufflevector(__builtin_shufflevector(bv[0], bv[1], 0, 1, 2, 3, 4, 5, 6, 7),
                                                                       __builtin_shufflevector(bv[2], bv[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                                                       0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
          else asm volatile("" : "+v"(zw));      // (its start values were written by the previous epilogue / before the loop)
        }
        ph_mma(fs[K % NS], xf[K].v, zw);
        if constexpr (K + LA < KS) ph_dsr<(K + LA) * 1024>(fs[(K + LA) % NS], wb);
        if constexpr (DOE) {
          ph_static_for<(K + 1) * 24 / KS - K * 24 / KS>([&](auto ic) { piece(std::integral_constant<int, K * 24 / KS + decltype(ic)::value>{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // this slab's second-layer fragments, for its epilogue one iteration later
      ph_u32x4 na, nb;
      asm volatile("" : "=v"(na)); asm volatile("" : "=v"(nb));
      ph_dsr<KS * 1024>(na, wb); ph_dsr<(KS + 1) * 1024>(nb, wb);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na), "+v"(nb) :: "memory");
      w2a = na; w2b = nb;
    } else {
      ph_static_for<24>([&](auto qc) { piece(qc); });
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
#ifdef PH_PROF
  // tools/ (a -DPH_PROF build): ticks in [0] the vm wait, [1] the barrier, [2] the DMA issue, [3] the slab bodies -> p.partials rows (debug only)
  unsigned long long tp[4] = {0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
  auto mark = [&](int k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tp[k] += t - tl; tl = t; };
#else
  auto mark = [](int) {};
#endif
  auto top = [&](int slab) {
    mark(3);
    // slabs slab, slab + 1 are in flight.  All eight waves requesting slab + 2 at once behind the barrier held every one of them ~490
    // ticks per slab (32 pieces x 16 cycles of the texture addresser, whose queue is short; spreading the pieces through the body
    // only moved that time there).  With four buffers slab + 2 goes where slab - 2 was - which every wave has left before the
    // PREVIOUS barrier - so a wave requests it while it waits for the others, and the last wave to arrive finds the queue empty.
    // (The last two slabs request the last slab once more, into a free buffer: no branch, constant counts.)
    if constexpr (NSTAGE == 4) {
      dma(min(slab + 2, nslab - 1), (slab + 2) % NSTAGE);
      mark(2);
      wait_vm<2 * UPW>();                                   // slab has landed: at most slab + 1 and slab + 2 outstanding
      mark(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (slab + 1 < nslab) wait_vm<UPW>(); else wait_vm<0>();
      mark(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (slab + 2 < nslab) dma(slab + 2, (slab + 2) % NSTAGE);
    }
    mark(1);
  };
  if constexpr (DROP) { start_values(0, z0); start_values(min(1, nslab - 1), z1); }
  top(0); body(yes{}, no{}, 0, z0, z1);
  int slab = 1;
  for (; slab + 1 < nslab; slab += 2) {
    top(slab); body(yes{}, yes{}, slab, z1, z0);
    top(slab + 1); body(yes{}, yes{}, slab + 1, z0, z1);
  }
  if (slab < nslab) {                                       // nslab even: one more slab, its accumulator is z1
    top(slab); body(yes{}, yes{}, slab, z1, z0);
    __builtin_amdgcn_s_barrier();                           // (a slab's distance between the MFMAs and the VALU reads of z1)
    body(no{}, yes{}, nslab, z0, z1);
  } else {
    __builtin_amdgcn_s_barrier();
    body(no{}, yes{}, nslab, z1, z0);
  }
  wait_vm<0>();                                             // (the last requests land in a buffer the epilogue's sums reuse)
#ifdef PH_PROF
  mark(3);
  if (g_ph_prof && blockIdx.y == 0 && blockIdx.x < 256 && lane == 0)
    for (int k = 0; k < 4; ++k) g_ph_prof[(blockIdx.x * PH_WAVES + wave) * 4 + k] = tp[k];
#endif
  pair_epilogue(p, lg, smem, tid, lane, wave, half, b, mypair, pair_ok);
}

template <int KS, bool DROP, bool SAVE>
static int launch_pair_fwd_hand(const PairFwdParams& p, hipStream_t st) {
  const size_t slab = (size_t)slab_stride_bytes(KS * 16, 2);
  const size_t nstage = 4 * slab + 5 * 512 * 4 <= 160 * 1024 ? 4 : 3;      // as in the kernel
  size_t sh = nstage * slab + (size_t)p.num_heads * p.D * sizeof(float);
  if (sh > 160 * 1024) { set_error("peneo_pair_heads_fwd: D=%d needs %zu bytes of LDS", p.D, sh); return PENEO_ERR_INVALID; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(pair_heads_fwd_hand_kernel<KS, DROP, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_error("peneo_pair_heads_fwd: cannot raise dynamic LDS to %zu bytes", sh);
    return PENEO_ERR_LAUNCH;
  }
  dim3 grid(SAVE ? (unsigned)((p.ntiles + 1) / 2) : (unsigned)((p.P + PH_PAIRS - 1) / PH_PAIRS), p.B);
  hipLaunchKernelGGL((pair_heads_fwd_hand_kernel<KS, DROP, SAVE>), grid, dim3(PH_WAVES * 64), sh, st, p);
  return check_launch("peneo_pair_heads_fwd");
}

template <typename T, int KS, int VARIANT, bool DROP>
static int launch_pair_fwd_v(const PairFwdParams& p, hipStream_t st) {
  constexpr int NSTAGE = (VARIANT & 128) ? 4 : (sizeof(T) == 2 ? 3 : 2);
  const size_t slab = (size_t)slab_stride_bytes(KS * 16, (int)sizeof(T));
  size_t sh = NSTAGE * slab + (size_t)p.num_heads * p.D * sizeof(float);
  if (sh < (size_t)PH_WAVES * 32 * sizeof(float)) sh = (size_t)PH_WAVES * 32 * sizeof(float);
  if (sh > 160 * 1024) { set_error("peneo_pair_heads_fwd: D=%d needs %zu bytes of LDS", p.D, sh); return PENEO_ERR_INVALID; }
  if (sh > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pair_heads_fwd_kernel<T, KS, NSTAGE, VARIANT, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
      set_error("peneo_pair_heads_fwd: cannot raise dynamic LDS to %zu bytes", sh);
      return PENEO_ERR_LAUNCH;
    }
  }
  dim3 grid((unsigned)((p.P + PH_PAIRS - 1) / PH_PAIRS), p.B);
  hipLaunchKernelGGL((pair_heads_fwd_kernel<T, KS, NSTAGE, VARIANT, DROP>), grid, dim3(PH_WAVES * 64), sh, st, p);
  return check_launch("peneo_pair_heads_fwd");
}

// bf16: LDS-DMA from inline asm with counted vmcnt (variant 1): 1761 us against 1838 us for the builtin at B = 8
// (the compiler drains the whole vm counter in front of every ds_read while a DMA it knows about is in flight)
constexpr int PH_DEFAULT_VARIANT = 1;
// tools/: PENEO_PAIR_FWD_HAND=0 selects the generic kernel at D = 384 / 512 too
static bool pair_fwd_hand() { static const bool v = [] { const char* e = getenv("PENEO_PAIR_FWD_HAND"); return !e || atoi(e) != 0; }(); return v; }
template <typename T, int KS>
static int launch_pair_fwd(const PairFwdParams& p, hipStream_t st) {
  constexpr int V = sizeof(T) == 2 ? PH_DEFAULT_VARIANT : 0;
  if constexpr (sizeof(T) == 2 && (KS == 24 || KS == 32)) {
    if (p.act && KS == 24 && p.num_heads * p.D / 32 >= 2)
      return p.drop_thr16 ? launch_pair_fwd_hand<KS, true, true>(p, st) : launch_pair_fwd_hand<KS, false, true>(p, st);
    if (pair_fwd_hand() && p.num_heads * p.D / 32 >= 2)
      return p.drop_thr16 ? launch_pair_fwd_hand<KS, true, false>(p, st) : launch_pair_fwd_hand<KS, false, false>(p, st);
  }
  if (p.act) { set_error("peneo_pair_heads_fwd: the saving form exists for bf16, D = 384 (peneo_pair_save_supported)"); return PENEO_ERR_INVALID; }
  return p.drop_thr16 ? launch_pair_fwd_v<T, KS, V, true>(p, st) : launch_pair_fwd_v<T, KS, V, false>(p, st);
}

template <typename T>
static int dispatch_pair_fwd(const PairFwdParams& p, hipStream_t st) {
  switch (p.D / 16) {
    case 2: return launch_pair_fwd<T, 2>(p, st);
    case 4: return launch_pair_fwd<T, 4>(p, st);
    case 6: return launch_pair_fwd<T, 6>(p, st);
    case 8: return launch_pair_fwd<T, 8>(p, st);
    case 12: return launch_pair_fwd<T, 12>(p, st);
    case 16: return launch_pair_fwd<T, 16>(p, st);
    case 24: return launch_pair_fwd<T, 24>(p, st);
    case 32: return launch_pair_fwd<T, 32>(p, st);
    default: set_error("peneo_pair_heads_fwd: D=%d not supported (D/16 in {2,4,6,8,12,16,24,32})", p.D); return PENEO_ERR_INVALID;
  }
}

// ================================================================================================
// dz of one pair chunk without x or z in memory.  Same skeleton as pair_heads_fwd_kernel (x = SiLU(a_i + b_j) kept as
// fragments in registers, first-layer weight slabs streamed L2 -> LDS), but the MFMA operands are swapped so that the
// accumulator holds z[pair, hidden] with lane = hidden column and the 16 registers = 16 pairs:
//   * the per-column constants (b1, the three W2 rows) are one 16-byte LDS read per lane per slab,
//   * dW2[c, k] = sum_p g[p, c] y[p, k] and db1[k] = sum_p dz[p, k] reduce over REGISTERS (+ one half-wave swap),
//   * dz leaves as 64-byte row segments straight from the accumulator layout.
// The 8 waves' column sums of a slab meet in LDS and one wave adds them to the workspace (fp32 atomics, 128 per slab).
// ================================================================================================
#ifndef DZF_ABLATE
#define DZF_ABLATE 0   // timing experiments only (tools/): 1 no stores, 2 no dz arithmetic, 4 no g reads, 8 no exp/rcp, 16 no MFMA
#endif
struct DzFusedParams {
  const bf16_t* abd; int N, D; int64_t pbase, npairs;
  const void* wp; const float* b1;
  peneo_pair_dz_args a;
  bf16_t* out; float* ws;
  bf16_t* x_out; bf16_t* pre_out;   // optional [npairs, D]: SiLU(a_i + b_j) and a_i + b_j for the dW1 / dx GEMMs
};
constexpr int DZF_SLOTS = 256;

template <int KS, bool PIPE>
__global__ __launch_bounds__(PH_WAVES * 64, 2) void pair_dz_fused_kernel(DzFusedParams p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSTAGE = KS > 24 ? 2 : 3;   // D = 512: two stages, or the ring + the column table overflow the LDS
  constexpr int NF = KS + 2;
  constexpr int PAYLOAD = NF * 64 * FragBytes<T>::v;
  constexpr int UPW = (PAYLOAD / 1024 + PH_WAVES - 1) / PH_WAVES;
  constexpr int SLAB_BYTES = UPW * PH_WAVES * 1024;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D, N = p.N, nh = p.a.num_heads, ncol = nh * D;
  char* sW = smem;                                                         // [NSTAGE][SLAB_BYTES]
  float4* sCol = reinterpret_cast<float4*>(smem + NSTAGE * SLAB_BYTES);    // [ncol]: W2 rows 0..2 of the column, b1
  float4* sG = sCol + ncol;                                                // [PH_WAVES][32]: scale * dlogits of the wave's pairs
  float4* sPart = sG + PH_WAVES * 32;                                      // [2][PH_WAVES][32]: column sums of one slab
  const int64_t lp0 = (int64_t)blockIdx.x * PH_PAIRS + wave * 32;          // first local pair of the wave
  const int64_t lp = lp0 + (lane & 31);
  const bool pair_ok = lp < p.npairs;
  int pi, pj;
  pair_decode(p.pbase + (pair_ok ? lp : p.npairs - 1), N, pi, pj);
  const int nslab = ncol / 32, spb = D / 32;
  const uint32_t drop_thr = pair_drop_thr16_dev(p.a.drop_p);
  const uint32_t drop_key = pair_drop_key(p.a.drop_seed, p.a.drop_doc);
  const float drop_scale = drop_thr ? 65536.f / (65536.f - (float)drop_thr) : 1.f;
  const uint32_t drop_half = ((lane & 31) >> 2) & 1;                                  // unit w = lane & 31 of every slab:
  const PairDropJump drop_jump = pair_drop_jump(4 * ((lane & 31) >> 3) + (lane & 3));  // half (w >> 2) & 1, position 4 (w >> 3) + (w & 3)

  for (int n = tid; n < ncol; n += PH_WAVES * 64) {
    const int h = n / D, k = n - h * D, Cn = p.a.classes[h];
    sCol[n] = make_float4(p.a.w2[h][k], Cn > 1 ? p.a.w2[h][(int64_t)D + k] : 0.f, Cn > 2 ? p.a.w2[h][(int64_t)2 * D + k] : 0.f,
                          p.b1[n]);
  }

  const T* arow = p.abd + (int64_t)pi * 2 * D;
  const T* brow = p.abd + (int64_t)pj * 2 * D + D;
  // the fragments are also what the dW1 / dx GEMMs need as x and a_i + b_j: each lane owns 16 contiguous bytes per k-step
  const bool want_x = p.x_out != nullptr && pair_ok;
  bf16_t* x_row = p.x_out + lp * D + 8 * half;
  bf16_t* pre_row = p.pre_out + lp * D + 8 * half;
  Frag<T> xf[KS];
  if constexpr (KS % 4 == 0) {
    uint4 ra[2][4], rb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[0][i] = *reinterpret_cast<const uint4*>(arow + 16 * i + 8 * half);
      rb[0][i] = *reinterpret_cast<const uint4*>(brow + 16 * i + 8 * half);
    }
#pragma unroll
    for (int g = 0; g < KS / 4; ++g) {
      if (g + 1 < KS / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ra[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(arow + 16 * (4 * (g + 1) + i) + 8 * half);
          rb[(g + 1) & 1][i] = *reinterpret_cast<const uint4*>(brow + 16 * (4 * (g + 1) + i) + 8 * half);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a[8], bb[8];
        unpack16<T>(ra[g & 1][i], a);
        unpack16<T>(rb[g & 1][i], bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += bb[e];
        if (want_x) *reinterpret_cast<uint4*>(pre_row + 16 * (4 * g + i)) = pack_frag8<T>(a).v;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e]);
        xf[4 * g + i] = pack_frag8<T>(a);
        asm volatile("" : "+v"(xf[4 * g + i].v.x), "+v"(xf[4 * g + i].v.y), "+v"(xf[4 * g + i].v.z), "+v"(xf[4 * g + i].v.w) :: "memory");
        if (want_x) *reinterpret_cast<uint4*>(x_row + 16 * (4 * g + i)) = xf[4 * g + i].v;
      }
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 16 * ks + 8 * half;
      float a[8], bb[8];
      unpack16<T>(*reinterpret_cast<const uint4*>(arow + c), a);
      unpack16<T>(*reinterpret_cast<const uint4*>(brow + c), bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += bb[e];
      if (want_x) *reinterpret_cast<uint4*>(pre_row + 16 * ks) = pack_frag8<T>(a).v;
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e]);
      xf[ks] = pack_frag8<T>(a);
      if (want_x) *reinterpret_cast<uint4*>(x_row + 16 * ks) = xf[ks].v;
    }
  }
  // (no vmcnt(0) here: the x / pre stores may still be in flight; the counted DMA waits below are safe next to stores)
  __syncthreads();                                          // sCol visible

  const char* wsrc = reinterpret_cast<const char*>(p.wp) + wave * (UPW * 1024) + lane * 16;
  const uint32_t wdst = lds_addr(sW) + wave * (UPW * 1024);
  auto dma = [&](int slab_, int buf_) {
    lds_dma_units<0, UPW>(wsrc + (int64_t)slab_ * SLAB_BYTES, __builtin_amdgcn_readfirstlane(wdst + buf_ * SLAB_BYTES));
  };
#pragma unroll
  for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
    if (s0 < nslab) dma(s0, s0);

  float* slot = p.ws + (int64_t)(blockIdx.x % DZF_SLOTS) * 4 * ncol;
  // the wave that owns slab s adds the eight waves' column sums of that slab to the workspace
  auto flush = [&](int s) {
    if (wave == (s & (PH_WAVES - 1)) && lane < 32) {
      const float4* src = sPart + (s & 1) * (PH_WAVES * 32) + lane;
      float4 t = src[0];
#pragma unroll
      for (int w = 1; w < PH_WAVES; ++w) { const float4 u = src[w * 32]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      float* dst = slot + s * 32 + lane;
      atomicAdd(dst, t.x); atomicAdd(dst + ncol, t.y); atomicAdd(dst + 2 * (int64_t)ncol, t.z); atomicAdd(dst + 3 * (int64_t)ncol, t.w);
    }
  };
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 one2 = f2{1.f, 1.f}, nl2e = f2{-1.4426950408889634f, -1.4426950408889634f};
  const float4* myG = sG + wave * 32;
  const int nrows = (int)min((int64_t)32, p.npairs - lp0);   // valid pairs of this wave's tile (may be <= 0)

  // Software pipeline inside every wave: iteration s issues the first-layer MFMAs of slab s (3 per step for D = 384)
  // BETWEEN the eight steps of the dz arithmetic of slab s-1, so the matrix pipe works in the shadow of the VALU-bound
  // epilogue instead of alternating with it.  dz leaves as 4-byte stores: even lanes hold columns (c, c+1) of pair row 2j',
  // odd lanes columns (c-1, c) of row 2j'+1 (neighbour lanes trade one value).
  bf16_t* orow2 = p.out + lp0 * ncol + (int64_t)(4 * half + (lane & 1)) * ncol + ((lane & 31) & ~1);
  const bool full_tile = nrows >= 32;
  const bool odd = (lane & 1) != 0;
  f32x16_t zp;
#pragma unroll
  for (int r = 0; r < 16; ++r) zp[r] = 0.f;

  auto stage_g = [&](int h) {
    const int Cn = p.a.classes[h];
    if (lane < 32) {
      float gx = 0.f, gy = 0.f, gz = 0.f;
      if (pair_ok) {
        const float sc = p.a.scale[h];
        const float* dl = p.a.dlogits[h] + lp * Cn;
        gx = dl[0] * sc;
        if (Cn > 1) gy = dl[1] * sc;
        if (Cn > 2) gz = dl[2] * sc;
      }
      // two adjacent pairs share 8 floats {g0, g0', g1, g1', g2, g2', -, -}: the packed-fp32 operands come out of LDS
      // already paired
      float* gp = reinterpret_cast<float*>(sG + wave * 32) + (lane >> 1) * 8 + (lane & 1);
      gp[0] = gx; gp[2] = gy; gp[4] = gz;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // one iteration: MMA -> first layer of slab s into z ; EPI -> dz arithmetic of slab s-1 on zp
  auto step = [&](auto mma_c, auto epi_c, int s) {
    constexpr bool MMA = decltype(mma_c)::value, EPI = decltype(epi_c)::value;
    const char* wb = sW + (s % NSTAGE) * SLAB_BYTES;
    f32x16_t z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    const int es = s - 1;
    float4 cw = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI) cw = sCol[es * 32 + (lane & 31)];
    const f2 w0 = f2{cw.x, cw.x}, w1 = f2{cw.y, cw.y}, w2 = f2{cw.z, cw.z}, b1 = f2{cw.w, cw.w};
    f2 s0 = f2{0.f, 0.f}, s1 = s0, s2 = s0, sb = s0;
    bf16_t* ocol = orow2 + es * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (MMA) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          if (ks * 8 / KS == j) {
            Frag<T> wf = load_frag_linear<T>(wb, ks, lane);
            if constexpr (DZF_ABLATE & 16) { z[ks & 15] += __uint_as_float(wf.v.x); }
            else mma_step(xf[ks], wf, z);                  // rows = pairs, columns = hidden units
          }
      }
      if constexpr (EPI) {
        const int r0 = 2 * j, rowc = (r0 & 3) + 8 * (r0 >> 2);   // accumulator registers 2j, 2j+1: pair rows rowc + 4*half + {0, 1}
        const int row0 = rowc + 4 * half;
        const float4 g01 = (DZF_ABLATE & 4) ? cw : myG[row0];   // (row0 is even: float4 slots row0, row0 + 1 = that pair's 8 floats)
        const float2 g2v = (DZF_ABLATE & 4) ? make_float2(cw.x, cw.y) : *reinterpret_cast<const float2*>(myG + row0 + 1);
        const f2 g0 = f2{g01.x, g01.y}, g1 = f2{g01.z, g01.w}, g2 = f2{g2v.x, g2v.y};
        const f2 zz = f2{zp[r0], zp[r0 + 1]} + b1;
        const f2 t = zz * nl2e;
        const f2 den = (DZF_ABLATE & 8) ? t + one2 : f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + one2;
        const f2 sg = (DZF_ABLATE & 8) ? den * nl2e : f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        const f2 y = zz * sg;
        f2 dz;
        if constexpr (DZF_ABLATE & 2) { dz = zz * g0; sb = sb + dz; }
        else {
          const f2 dy = __builtin_elementwise_fma(g2, w2, __builtin_elementwise_fma(g1, w1, g0 * w0));
          dz = dy * (sg * __builtin_elementwise_fma(zz, one2 - sg, one2));
          if (drop_thr) {
            // the forward's classifier dropout, one hash per element (this chunked kernel is the path of the widths the
            // batch kernel peneo_pair_bwd_fused does not cover: D = 512)
            // the forward's classifier dropout: this lane's unit sits at a fixed chain position, its pairs vary
            const uint32_t cnt0 = (uint32_t)(((p.a.drop_pair0 + lp0 + row0) * nslab + es) * 2) + drop_half;
            const uint32_t cnt1 = cnt0 + 2u * (uint32_t)nslab;
            const f2 kk = f2{pair_drop_keep_at(pair_drop_seed(drop_key, cnt0), pair_drop_inc(drop_key, cnt0), drop_jump, drop_thr) ? drop_scale : 0.f,
                             pair_drop_keep_at(pair_drop_seed(drop_key, cnt1), pair_drop_inc(drop_key, cnt1), drop_jump, drop_thr) ? drop_scale : 0.f};
            dz = dz * kk;
            const f2 ym = y * kk;
            s0 = __builtin_elementwise_fma(g0, ym, s0); s1 = __builtin_elementwise_fma(g1, ym, s1); s2 = __builtin_elementwise_fma(g2, ym, s2);
          } else {
          s0 = __builtin_elementwise_fma(g0, y, s0);
          s1 = __builtin_elementwise_fma(g1, y, s1);
          s2 = __builtin_elementwise_fma(g2, y, s2);
          }
          sb = sb + dz;
        }
        const float give = odd ? dz.x : dz.y;
        const float got = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(give), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
        const uint32_t packed = odd ? pack_bf16x2(got, dz.y) : pack_bf16x2(dz.x, got);
        if constexpr (DZF_ABLATE & 1) { sb.x += __uint_as_float(packed) * 1e-30f; }
        else if (full_tile || row0 + (lane & 1) < nrows)
          *reinterpret_cast<uint32_t*>(ocol + (int64_t)rowc * ncol) = packed;
      }
    }
    if constexpr (EPI) {
      float4 part = make_float4(s0.x + s0.y, s1.x + s1.y, s2.x + s2.y, sb.x + sb.y);
      part.x += __shfl_xor(part.x, 32); part.y += __shfl_xor(part.y, 32);
      part.z += __shfl_xor(part.z, 32); part.w += __shfl_xor(part.w, 32);
      if (lane < 32) sPart[(es & 1) * (PH_WAVES * 32) + wave * 32 + lane] = part;
    }
    if constexpr (MMA) zp = z;
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  for (int slab = 0; slab < nslab; ++slab) {
    // counted wait although stores and atomics share the vm counter (and may retire out of order with respect to loads):
    // loads retire in order among themselves, so while a piece of slab `slab` is missing all UPW pieces of the younger
    // slab+1 are missing too and the counter stays above UPW whatever the stores do
    if (NSTAGE > 2 && slab + 1 < nslab) wait_vm<(NSTAGE - 2) * UPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (slab + NSTAGE - 1 < nslab) dma(slab + NSTAGE - 1, (slab + NSTAGE - 1) % NSTAGE);
    if constexpr (PIPE) {
      if (slab > 1) flush(slab - 2);                         // its eight partial rows were written before this barrier
      if (slab > 0 && (slab - 1) % spb == 0) stage_g((slab - 1) / spb);
      if (slab == 0) step(yes{}, no{}, slab); else step(yes{}, yes{}, slab);
    } else {
      if (slab > 0) flush(slab - 1);
      if (slab % spb == 0) stage_g(slab / spb);
      step(yes{}, no{}, slab);
      step(no{}, yes{}, slab + 1);
    }
  }
  if constexpr (PIPE) {
    __syncthreads();
    if (nslab > 1) flush(nslab - 2);
    if ((nslab - 1) % spb == 0) stage_g((nslab - 1) / spb);
    step(no{}, yes{}, nslab);
  }
  __syncthreads();
  flush(nslab - 1);
}

template <int KS, bool PIPE>
static int launch_pair_dz_fused_v(const DzFusedParams& p, hipStream_t st) {
  const int ncol = p.a.num_heads * p.D;
  const size_t sh = (KS > 24 ? 2 : 3) * (size_t)slab_stride_bytes(KS * 16, 2) + (size_t)ncol * 16 + (size_t)PH_WAVES * 32 * 16 * 3;
  if (sh > 160 * 1024) { set_error("peneo_pair_dz_fused: D=%d needs %zu bytes of LDS", p.D, sh); return PENEO_ERR_INVALID; }
  if (sh > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(pair_dz_fused_kernel<KS, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    set_error("peneo_pair_dz_fused: cannot raise dynamic LDS to %zu bytes", sh);
    return PENEO_ERR_LAUNCH;
  }
  const int64_t blocks = (p.npairs + PH_PAIRS - 1) / PH_PAIRS;
  hipLaunchKernelGGL((pair_dz_fused_kernel<KS, PIPE>), dim3((unsigned)blocks), dim3(PH_WAVES * 64), sh, st, p);
  return check_launch("peneo_pair_dz_fused");
}
template <int KS>
static int launch_pair_dz_fused(const DzFusedParams& p, hipStream_t st) {
  // software-pipelined form (MFMAs of slab s + 1 between the arithmetic steps of slab s) except at D = 512, where it spills
  // (measured on config 4: 60.6 against 56.9 docs/s)
  if constexpr (KS >= 32) return launch_pair_dz_fused_v<KS, false>(p, st);
  else return launch_pair_dz_fused_v<KS, true>(p, st);
}

// label maps from sparse spots (b, i, j, tag): the dense [B, P] int64 maps the collator builds on the host
// (data/collator.py:156-204 -> spots2shaking_tag4batch, model/peneo_decoder.py:35-73) scattered on the device instead.
// "Last spot wins" like the host loop: a spot is skipped when a later one addresses the same cell.
__global__ __launch_bounds__(256) void spots_to_tags_kernel(const int32_t* spots, int n, int B, int N, int64_t* tags,
                                                            int32_t* bad) {
  const int64_t P = (int64_t)N * (N + 1) / 2;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int b = spots[4 * s], i = spots[4 * s + 1], j = spots[4 * s + 2], t = spots[4 * s + 3];
    if (b < 0 || b >= B || i < 0 || j < 0 || i >= N || j >= N) { atomicExch(bad, 1); continue; }
    const int64_t p = i <= j ? pair_row_start(i, N) + (j - i) : 0;   // the reference's index table is 0 below the diagonal
    bool last = true;
    for (int u = s + 1; u < n && last; ++u) {
      const int b2 = spots[4 * u], i2 = spots[4 * u + 1], j2 = spots[4 * u + 2];
      const int64_t p2 = (i2 <= j2 && i2 >= 0 && j2 < N) ? pair_row_start(i2, N) + (j2 - i2) : 0;
      if (b2 == b && p2 == p) last = false;
    }
    if (last) tags[(int64_t)b * P + p] = t;
  }
}

}  // namespace peneo
using namespace peneo;

static inline bool ok_dt(int d) { return d == PENEO_F32 || d == PENEO_BF16; }
static inline unsigned cap_blocks(int64_t n, int per = 256, int64_t cap = 8192) {
  int64_t b = (n + per - 1) / per;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}
static int total_classes(const int* classes, int nh) { int t = 0; for (int h = 0; h < nh; ++h) t += classes[h]; return t; }

extern "C" size_t peneo_pair_heads_packed_bytes(int dtype, int num_heads, int D) {
  return (size_t)(num_heads * D / 32) * (size_t)slab_stride_bytes(D, dtype == PENEO_BF16 ? 2 : 4);
}

extern "C" int peneo_pair_heads_pack(int dtype, const float* const* w1, const float* const* w2, const int* classes,
                                     int num_heads, int D, void* packed, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && w1 && w2 && classes && packed, "peneo_pair_heads_pack: bad arguments");
  PENEO_REQUIRE(num_heads > 0 && num_heads <= PENEO_MAX_HEADS, "peneo_pair_heads_pack: num_heads out of range");
  PENEO_REQUIRE(D > 0 && D % 32 == 0, "peneo_pair_heads_pack: D must be a multiple of 32");
  PENEO_REQUIRE(total_classes(classes, num_heads) <= NCP, "peneo_pair_heads_pack: more than %d classes in total", NCP);
  PackSrc s = {};
  s.num_heads = num_heads; s.D = D;
  for (int h = 0; h < num_heads; ++h) {
    PENEO_REQUIRE(w1[h] && w2[h], "peneo_pair_heads_pack: null weight pointer for head %d", h);
    PENEO_REQUIRE(classes[h] >= 1 && classes[h] <= 4, "peneo_pair_heads_pack: classes[%d] must be 1..4", h);
    s.w1[h] = w1[h]; s.w2[h] = w2[h]; s.classes[h] = classes[h];
  }
  const int64_t total = (int64_t)(num_heads * D / 32) * (slab_stride_bytes(D, dtype == PENEO_BF16 ? 2 : 4) / (dtype == PENEO_BF16 ? 2 : 4));
  if (dtype == PENEO_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)packed);
  else hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream, s, (float*)packed);
  return check_launch("peneo_pair_heads_pack");
}

static int pair_heads_fwd_impl(int dtype, const void* ab, int B, int N, const peneo_pair_heads_desc* desc,
                               float* const* logits, const peneo_pair_loss* loss, void* act, void* x_rows, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && ab && desc && B > 0 && N > 0, "peneo_pair_heads_fwd: bad arguments");
  PENEO_REQUIRE(desc->num_heads > 0 && desc->num_heads <= PENEO_MAX_HEADS, "peneo_pair_heads_fwd: num_heads out of range");
  PENEO_REQUIRE(desc->D > 0 && desc->D % 32 == 0, "peneo_pair_heads_fwd: D must be a multiple of 32");
  PENEO_REQUIRE(desc->w_packed && desc->b1 && desc->b2, "peneo_pair_heads_fwd: null weights");
  PENEO_REQUIRE((reinterpret_cast<uintptr_t>(desc->w_packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(ab) & 15) == 0,
                "peneo_pair_heads_fwd: ab / packed weights must be 16-byte aligned");
  PairFwdParams p = {};
  p.ab = ab; p.B = B; p.N = N; p.D = desc->D; p.P = (int64_t)N * (N + 1) / 2; p.num_heads = desc->num_heads;
  p.total_classes = total_classes(desc->classes, desc->num_heads);
  PENEO_REQUIRE(p.total_classes <= NCP, "peneo_pair_heads_fwd: more than %d classes in total", NCP);
  p.wp = desc->w_packed; p.b1 = desc->b1; p.b2 = desc->b2;
  PENEO_REQUIRE(desc->drop_p >= 0.f && desc->drop_p < 1.f, "peneo_pair_heads_fwd: drop_p must be in [0, 1)");
  PENEO_REQUIRE((int64_t)p.P * (desc->num_heads * desc->D / 16) < ((int64_t)1 << 32), "peneo_pair_heads_fwd: pair space too large for the dropout counter");
  p.drop_thr16 = pair_drop_thr16_host(desc->drop_p); p.drop_seed = desc->drop_seed;
  p.drop_scale = p.drop_thr16 ? 65536.f / (65536.f - (float)p.drop_thr16) : 1.f;
  for (int h = 0; h < desc->num_heads; ++h) {
    PENEO_REQUIRE(desc->classes[h] >= 1 && desc->classes[h] <= 4, "peneo_pair_heads_fwd: classes[%d] must be 1..4", h);
    p.classes[h] = desc->classes[h];
    p.logits[h] = logits ? logits[h] : nullptr;
    if (loss) { p.tags[h] = loss->tags[h]; p.cw[h] = loss->class_weight[h]; p.dlogits[h] = loss->dlogits[h]; }
  }
  if (loss) {
    p.partials = loss->partials;
    bool any = false;
    for (int h = 0; h < desc->num_heads; ++h) any = any || loss->tags[h];
    if (any) PENEO_REQUIRE(p.partials, "peneo_pair_heads_fwd: loss->partials workspace missing");
  }
  if (act) {
    PENEO_REQUIRE(x_rows && peneo_pair_save_supported(dtype, desc->D, desc->num_heads), "peneo_pair_heads_fwd_save: not supported for this dtype / D (peneo_pair_save_supported)");
    PENEO_REQUIRE((reinterpret_cast<uintptr_t>(act) & 15) == 0 && (reinterpret_cast<uintptr_t>(x_rows) & 15) == 0, "peneo_pair_heads_fwd_save: act / x must be 16-byte aligned");
    p.act = static_cast<char*>(act); p.x_save = static_cast<bf16_t*>(x_rows); p.ntiles = pb_num_tiles(N);
  }
  return dtype == PENEO_BF16 ? dispatch_pair_fwd<bf16_t>(p, (hipStream_t)stream) : dispatch_pair_fwd<float>(p, (hipStream_t)stream);
}

extern "C" int peneo_pair_heads_fwd(int dtype, const void* ab, int B, int N, const peneo_pair_heads_desc* desc,
                                    float* const* logits, const peneo_pair_loss* loss, peneo_stream_t stream) {
  return pair_heads_fwd_impl(dtype, ab, B, N, desc, logits, loss, nullptr, nullptr, stream);
}

extern "C" int peneo_pair_save_supported(int dtype, int D, int num_heads) {
  return dtype == PENEO_BF16 && D == 384 && num_heads >= 1 && num_heads <= PENEO_MAX_HEADS;
}
extern "C" size_t peneo_pair_save_bytes(int B, int N, int num_heads, int D) {
  return B > 0 && N > 0 && num_heads > 0 && D > 0 ? (size_t)B * pb_num_tiles(N) * (num_heads * D / 32) * 4 * PB_REC_BYTES : 0;
}
extern "C" int64_t peneo_pair_loss_partials_save(int B, int N) { return (int64_t)B * ((pb_num_tiles(N) + 1) / 2); }
extern "C" int peneo_pair_heads_fwd_save(int dtype, const void* ab, int B, int N, const peneo_pair_heads_desc* desc,
                                         float* const* logits, const peneo_pair_loss* loss, void* act, void* x_rows,
                                         peneo_stream_t stream) {
  PENEO_REQUIRE(act && x_rows, "peneo_pair_heads_fwd_save: null act / x");
  return pair_heads_fwd_impl(dtype, ab, B, N, desc, logits, loss, act, x_rows, stream);
}

static int chunk_check(const char* who, int dtype, int N, int D, int i0, int i1) {
  PENEO_REQUIRE(ok_dt(dtype), "%s: bad dtype", who);
  PENEO_REQUIRE(N > 0 && D > 0 && D % 8 == 0, "%s: bad N/D", who);
  PENEO_REQUIRE(0 <= i0 && i0 < i1 && i1 <= N, "%s: bad row range [%d, %d)", who, i0, i1);
  return PENEO_OK;
}

extern "C" int peneo_pair_x_fwd(int dtype, const void* ab_doc, int N, int D, int i0, int i1, void* x, void* pre,
                                peneo_stream_t stream) {
  int rc = chunk_check("peneo_pair_x_fwd", dtype, N, D, i0, i1);
  if (rc) return rc;
  PENEO_REQUIRE(ab_doc && x, "peneo_pair_x_fwd: null pointer");
  const int64_t pbase = pair_row_start(i0, N), npairs = pair_row_start(i1, N) - pbase;
  const int64_t total = npairs * (D / (dtype == PENEO_BF16 ? 8 : 4));
  const int vec = dtype == PENEO_BF16 ? 8 : 4;
  if (D / vec <= 256 && (reinterpret_cast<uintptr_t>(ab_doc) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(pre) & 15) == 0) {
    dim3 grid(i1 - i0, (N - i0 + XF_JCH - 1) / XF_JCH);
    if (dtype == PENEO_BF16) hipLaunchKernelGGL(pair_x_fwd_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ab_doc, N, D, i0, pbase, (bf16_t*)x, (bf16_t*)pre);
    else hipLaunchKernelGGL(pair_x_fwd_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)ab_doc, N, D, i0, pbase, (float*)x, (float*)pre);
    return check_launch("peneo_pair_x_fwd");
  }
  if (dtype == PENEO_BF16) hipLaunchKernelGGL(pair_x_fwd_kernel<bf16_t>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ab_doc, N, D, i0, pbase, npairs, (bf16_t*)x, (bf16_t*)pre);
  else hipLaunchKernelGGL(pair_x_fwd_kernel<float>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const float*)ab_doc, N, D, i0, pbase, npairs, (float*)x, (float*)pre);
  return check_launch("peneo_pair_x_fwd");
}

extern "C" int peneo_pair_x_bwd(int dtype, const void* ab_doc, int N, int D, int i0, int i1, const void* dx, float* d_ab_doc,
                                int premultiplied, peneo_stream_t stream) {
  int rc = chunk_check("peneo_pair_x_bwd", dtype, N, D, i0, i1);
  if (rc) return rc;
  PENEO_REQUIRE(ab_doc && dx && d_ab_doc, "peneo_pair_x_bwd: null pointer");
  const int vec = dtype == PENEO_BF16 ? 8 : 4;
  PENEO_REQUIRE(D % vec == 0 && D / vec <= 256, "peneo_pair_x_bwd: D must be a multiple of %d and <= %d", vec, 256 * vec);
  const int threads = (D / vec + 63) / 64 * 64;
  PENEO_REQUIRE((reinterpret_cast<uintptr_t>(ab_doc) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0,
                "peneo_pair_x_bwd: pointers must be 16-byte aligned");
  const int64_t pbase = pair_row_start(i0, N);
  hipStream_t st = (hipStream_t)stream;
#define PENEO_XBWD(T_, PRE_)                                                                                                   \
  {                                                                                                                           \
    hipLaunchKernelGGL((pair_x_bwd_a_kernel<T_, PRE_>), dim3(i1 - i0, JSPLIT), dim3(threads), 0, st, (const T_*)ab_doc, N, D, i0, \
                       pbase, (const T_*)dx, d_ab_doc);                                                                       \
    hipLaunchKernelGGL((pair_x_bwd_b_kernel<T_, PRE_>), dim3(N - i0, JSPLIT), dim3(threads), 0, st, (const T_*)ab_doc, N, D, i0, \
                       i1, pbase, (const T_*)dx, d_ab_doc);                                                                   \
  }
  if (dtype == PENEO_BF16) { if (premultiplied) PENEO_XBWD(bf16_t, true) else PENEO_XBWD(bf16_t, false) }
  else { if (premultiplied) PENEO_XBWD(float, true) else PENEO_XBWD(float, false) }
#undef PENEO_XBWD
  return check_launch("peneo_pair_x_bwd");
}

extern "C" int peneo_pair_dz_fused(int dtype, const void* ab_doc, int N, int D, int i0, int i1, const void* w_packed,
                                   const float* b1, const peneo_pair_dz_args* args, void* dz, float* workspace,
                                   void* x_out, void* pre_out, peneo_stream_t stream) {
  PENEO_REQUIRE((x_out == nullptr) == (pre_out == nullptr), "peneo_pair_dz_fused: x_out and pre_out come together");
  PENEO_REQUIRE(((reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(pre_out)) & 15) == 0,
                "peneo_pair_dz_fused: x_out / pre_out must be 16-byte aligned");
  PENEO_REQUIRE(dtype == PENEO_BF16, "peneo_pair_dz_fused: bf16 only (the fp32 path runs peneo_gemm with a pair_dz epilogue)");
  PENEO_REQUIRE(ab_doc && w_packed && b1 && args && dz && workspace, "peneo_pair_dz_fused: null argument");
  PENEO_REQUIRE(N > 0 && i0 >= 0 && i1 > i0 && i1 <= N, "peneo_pair_dz_fused: bad row range [%d, %d) of %d", i0, i1, N);
  PENEO_REQUIRE(args->D == D && D % 32 == 0, "peneo_pair_dz_fused: D=%d (args D=%d) must be a multiple of 32", D, args->D);
  PENEO_REQUIRE(args->num_heads >= 1 && args->num_heads <= PENEO_MAX_HEADS && args->scale, "peneo_pair_dz_fused: bad head arguments");
  for (int h = 0; h < args->num_heads; ++h)
    PENEO_REQUIRE(args->dlogits[h] && args->w2[h] && args->classes[h] >= 1 && args->classes[h] <= 3,
                  "peneo_pair_dz_fused: head %d arguments invalid (classes must be 1..3)", h);
  PENEO_REQUIRE((reinterpret_cast<uintptr_t>(ab_doc) & 15) == 0, "peneo_pair_dz_fused: ab must be 16-byte aligned");
  PENEO_REQUIRE(args->drop_p >= 0.f && args->drop_p < 1.f, "peneo_pair_dz_fused: drop_p must be in [0, 1)");
  DzFusedParams p;
  p.abd = reinterpret_cast<const bf16_t*>(ab_doc); p.N = N; p.D = D;
  p.pbase = pair_row_start(i0, N); p.npairs = pair_row_start(i1, N) - p.pbase;
  p.wp = w_packed; p.b1 = b1; p.a = *args; p.out = reinterpret_cast<bf16_t*>(dz); p.ws = workspace;
  p.x_out = reinterpret_cast<bf16_t*>(x_out); p.pre_out = reinterpret_cast<bf16_t*>(pre_out);
  hipStream_t st = (hipStream_t)stream;
  switch (D / 16) {
    case 2: return launch_pair_dz_fused<2>(p, st);
    case 4: return launch_pair_dz_fused<4>(p, st);
    case 6: return launch_pair_dz_fused<6>(p, st);
    case 8: return launch_pair_dz_fused<8>(p, st);
    case 12: return launch_pair_dz_fused<12>(p, st);
    case 16: return launch_pair_dz_fused<16>(p, st);
    case 24: return launch_pair_dz_fused<24>(p, st);
    case 32: return launch_pair_dz_fused<32>(p, st);
    default: set_error("peneo_pair_dz_fused: D=%d not supported (D/16 in {2,4,6,8,12,16,24,32})", D); return PENEO_ERR_INVALID;
  }
}

extern "C" size_t peneo_pair_dz_workspace_bytes(int num_heads, int D) { return (size_t)DZ_SLOTS * 4 * num_heads * D * sizeof(float); }

extern "C" int peneo_pair_dz(int dtype, void* z_inout, int64_t npairs, const peneo_pair_dz_args* args, float* workspace,
                             peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && z_inout && args && workspace && npairs > 0, "peneo_pair_dz: bad arguments");
  const int vec = dtype == PENEO_BF16 ? 8 : 4;
  PENEO_REQUIRE(args->num_heads > 0 && args->num_heads <= PENEO_MAX_HEADS && args->D > 0 && args->D % vec == 0,
                "peneo_pair_dz: D must be a multiple of %d (got %d)", vec, args->D);
  PENEO_REQUIRE(args->scale, "peneo_pair_dz: null scale");
  PENEO_REQUIRE((reinterpret_cast<uintptr_t>(z_inout) & 15) == 0, "peneo_pair_dz: z must be 16-byte aligned");
  for (int h = 0; h < args->num_heads; ++h)
    PENEO_REQUIRE(args->dlogits[h] && args->w2[h] && args->classes[h] >= 1 && args->classes[h] <= 3,
                  "peneo_pair_dz: head %d arguments invalid (classes must be 1..3)", h);
  PENEO_REQUIRE(args->drop_p >= 0.f && args->drop_p < 1.f, "peneo_pair_dz: drop_p must be in [0, 1)");
  DzParams pp; pp.a = *args;
  const int blocks = (int)(npairs < DZ_SLOTS ? npairs : DZ_SLOTS);
  if (dtype == PENEO_BF16) hipLaunchKernelGGL(pair_dz_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)z_inout, npairs, pp, workspace);
  else hipLaunchKernelGGL(pair_dz_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)z_inout, npairs, pp, workspace);
  return check_launch("peneo_pair_dz");
}

extern "C" int64_t peneo_pair_loss_partials(int B, int N) {
  const int64_t P = (int64_t)N * (N + 1) / 2;
  return (int64_t)B * ((P + PH_PAIRS - 1) / PH_PAIRS);
}

extern "C" int peneo_loss_finish(const float* partials, int64_t n_partials, const float* ratio, int num_heads, int total_classes,
                                 float* out, float* scale, float* dl_sum, float* inv_den, peneo_stream_t stream) {
  PENEO_REQUIRE(partials && n_partials > 0 && ratio && out && num_heads > 0 && num_heads <= PENEO_MAX_HEADS &&
                total_classes >= 0 && total_classes <= NCP, "peneo_loss_finish: bad arguments");
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, n_partials, ratio, num_heads,
                     total_classes, out, scale, dl_sum, inv_den);
  return check_launch("peneo_loss_finish");
}

extern "C" int peneo_weighted_ce(const float* logits, const int64_t* tags, const float* class_weight, int64_t rows, int C,
                                 float* num, float* den, float* dlogits, peneo_stream_t stream) {
  PENEO_REQUIRE(logits && tags && num && den && rows > 0 && C > 0 && C <= 16, "peneo_weighted_ce: bad arguments");
  hipLaunchKernelGGL(weighted_ce_kernel, dim3(cap_blocks(rows, 256, 1024)), dim3(256), 0, (hipStream_t)stream, logits, tags,
                     class_weight, rows, C, num, den, dlogits);
  return check_launch("peneo_weighted_ce");
}

extern "C" int peneo_spots_compact(const float* logits, int64_t P, int C, int N, int32_t* spots_ijt, float* scores,
                                   int32_t* count, int max_spots, peneo_stream_t stream) {
  PENEO_REQUIRE(logits && spots_ijt && scores && count && P > 0 && C > 1 && N > 0 && max_spots >= 0, "peneo_spots_compact: bad arguments");
  PENEO_REQUIRE(P == (int64_t)N * (N + 1) / 2, "peneo_spots_compact: P != N(N+1)/2");
  hipLaunchKernelGGL(spots_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, P, C, N, spots_ijt, scores, count, max_spots);
  return check_launch("peneo_spots_compact");
}

extern "C" int peneo_spots_to_tags(const int32_t* spots_bijt, int n_spots, int B, int N, int64_t* tags, int32_t* status,
                                   peneo_stream_t stream) {
  PENEO_REQUIRE(tags && B > 0 && N > 0 && n_spots >= 0 && (n_spots == 0 || spots_bijt), "peneo_spots_to_tags: bad arguments");
  const int64_t P = (int64_t)N * (N + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(tags, 0, sizeof(int64_t) * B * P, st) != hipSuccess) { set_error("peneo_spots_to_tags: memset failed"); return PENEO_ERR_LAUNCH; }
  if (n_spots == 0) return PENEO_OK;
  PENEO_REQUIRE(status, "peneo_spots_to_tags: status word missing");
  int blocks = (n_spots + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(spots_to_tags_kernel, dim3(blocks), dim3(256), 0, st, spots_bijt, n_spots, B, N, tags, status);
  return check_launch("peneo_spots_to_tags");
}

#ifdef PH_PROF
extern "C" int peneo_pair_fwd_prof_buffer(unsigned long long* dev) {
  return hipMemcpyToSymbol(HIP_SYMBOL(peneo::g_ph_prof), &dev, sizeof(dev)) == hipSuccess ? 0 : -1;
}
#endif
