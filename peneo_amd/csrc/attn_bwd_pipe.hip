// K6 backward, pipelined single pass (round 5; bf16, head dim 64, bias tensor, dS^T slab requested).
//
// Same arithmetic and the same outputs as attn_bwd_fused_kernel (attention.hip; reference: autograd through
// modeling_layoutlmv3.py:308-321,365-404): a workgroup owns 128 keys of one (document, head) - lane = key, K / V fragments and the
// dK / dV accumulators live in registers - and streams the queries; S and dP are computed once, dS^T leaves as this layer's bf16
// slab [B, nh, T keys, Tp queries] (the bias-table reduction and the dQ kernel read it).  What changed is how the stream runs:
//
//   * tiles of 32 queries (one 32 x 32 score block per wave and tile, 23 tiles at T = 709 instead of 12 x 64 with a skipped
//     half), every operand of a tile - Q, dO, the bias block [32 q][128 keys], lse, delta, the dropout keep words - arrives by
//     LDS-DMA (global_load_lds: no staging registers, no ds_write pass) into a ring of THREE buffers: the request for tile t + 2
//     goes out at the top of iteration t, tile t + 1 has landed by then, so S(t + 1) is computed at the end of iteration t and
//     its softmax arithmetic finds the accumulator ready one barrier later;
//   * ONE s_barrier per tile (the old kernel: three per 64 queries, one of them behind a synchronous lse / delta load);
//   * DMA images are lane-linear, so bank conflicts are removed by permuting the SOURCE 16-byte slots of a row (the read applies the
//     same involution): Q / dO rows (128 B) use slot ^ bitrev3(row >> 1) - conflict-free for the b128 fragment reads of S / dP AND
//     for the transpose reads of dV / dK (rows r, r + 2 land in different 64-byte groups); the bias rows (256 B) use
//     slot ^ ((row & 3) << 2) (the four query rows of one transpose read land in four different 64-byte groups);
//   * rows past T need no predication: their lse is read from a +1e30 constant (P = exp2(x - 1e30) = 0, hence dS = 0 and no
//     contribution to dV / dK), every other operand row is clamped to T - 1 (finite values);
//   * the softmax-backward arithmetic is 10 VALU instructions per score element with dropout (bias bf16 -> fp32, two fma, exp2,
//     bfe + two and, fma, mul, half a cvt_pk each for P and dS) against ~31 in the old kernel;
//   * dS^T of a tile goes through a per-wave LDS patch (no barrier: the wave that writes it reads it back) and is stored at the top
//     of the NEXT iteration, so the top-of-tile s_waitcnt vmcnt(0) finds stores that have had a whole tile to drain;
//   * dK / dV leave as 16-byte row pieces straight from the accumulator layout (v_permlane32_swap pairs), no LDS round trip.
#include <cstdlib>
#include "common.h"
#include "attention.h"

namespace peneo {
namespace {

constexpr int TQ = 32;     // queries per tile
constexpr int WK = 128;    // keys per workgroup (4 waves x 32)
// a tile's buffer: Q [32][128 B], dO [32][128 B], bias [32 q][128 keys] (256-byte rows), lse [64], delta [64], keep words [128]
constexpr int O_Q = 0, O_DO = 4096, O_BIAS = 8192, O_LSE = 16384, O_DELTA = 16640, O_WORDS = 16896, BUF = 17408;
constexpr int STG_PITCH = 80, STG_WAVE = 32 * STG_PITCH;
// Two workgroups per CU (205 registers, 62 KB of LDS).  Measured and removed in round 5 (profiles/r05_attention_bwd.txt): a 168-register
// form at three per CU (ring of two, K fragments parked in LDS, dS^T rows straight from registers: all 576 workgroups of 8 documents
// resident at once) and a 160-key form of five waves (480 workgroups, one round at two per CU) - 76 / 86 us against 76 alone and no
// better in the step: the launch is bound by the CU's global -> LDS fill and by instruction issue, not by workgroups per CU.
constexpr int NBUF = 3;
constexpr int LDS_BYTES = NBUF * BUF + 4 * STG_WAVE;
constexpr float kLog2e = 1.4426950408889634f;

__device__ float g_lse_pad = 1.0e30f;   // lse of query rows past T

__device__ __forceinline__ int bitrev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
__device__ __forceinline__ int qslot_swz(int row) { return bitrev3((row >> 1) & 7); }

typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr64(const char* p) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_s4p;
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)p));
}
__device__ __forceinline__ Frag<bf16_t> tr_frag(const char* lo, const char* hi) {
  const uint2 a = tr64(lo), b = tr64(hi);
  Frag<bf16_t> f;
  f.v = make_uint4(a.x, a.y, b.x, b.y);
  return f;
}
// 64 lanes x 4 bytes: global (uniform base + lane offset) -> LDS (uniform base + 4 * lane)
__device__ __forceinline__ void dma4_s(uint32_t voff_lane, const char* base_uniform, uint32_t lds_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_lane), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
// ... with a full per-lane pointer
__device__ __forceinline__ void dma4_v(const char* ptr_lane, uint32_t lds_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(ptr_lane), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ uint32_t and_u(float x, int m) { return __float_as_uint(x) & (uint32_t)m; }

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_pipe_kernel(AttnParams p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tn = p.T, Tp = p.Tp;
  // unit order: the key blocks of one (document, head) run on ONE XCD (they stream the same Q / dO rows through its L2)
  const int nkb = (Tn + WK - 1) / WK;
  int u;
  {
    const int nwg = gridDim.x, L = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, x = L & 7, i = L >> 3;
    u = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
  }
  const int kb = u % nkb, bh = u / nkb, h = bh % p.nh, b = bh / p.nh;
  const int key0 = kb * WK;
  const int keyl = wave * 32 + l31, mykey = key0 + keyl;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * 64;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * 64;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * 64;
  const T* dO = reinterpret_cast<const T*>(p.d_out) + (int64_t)b * Tn * p.ld_out + h * 64;
  const T* bias = reinterpret_cast<const T*>(p.bias) + (int64_t)bh * Tn * p.bias_ld;
  const float* lse = p.lse + (int64_t)bh * Tn;
  const float* delta = p.delta + (int64_t)bh * Tn;
  const float keep_scale = DROP ? p.keep_scale : 1.0f;
  const int nt = (Tn + TQ - 1) / TQ;

  // ---- K / V fragments of this lane's key (B operands of S and dP) ----
  Frag<T> kf[4], vf[4];
  {
    const bool ok = mykey < Tn;
    const T* kr = K + (int64_t)(ok ? mykey : 0) * p.ld + 8 * half;
    const T* vr = V + (int64_t)(ok ? mykey : 0) * p.ld + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks].v = *reinterpret_cast<const uint4*>(kr + 16 * ks);
      vf[ks].v = *reinterpret_cast<const uint4*>(vr + 16 * ks);
      if (!ok) { kf[ks].v = make_uint4(0u, 0u, 0u, 0u); vf[ks].v = make_uint4(0u, 0u, 0u, 0u); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the vm counter holds the DMA pieces and the slab stores only
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    asm volatile("" : "+v"(kf[ks].v.x), "+v"(kf[ks].v.y), "+v"(kf[ks].v.z), "+v"(kf[ks].v.w));
    asm volatile("" : "+v"(vf[ks].v.x), "+v"(vf[ks].v.y), "+v"(vf[ks].v.z), "+v"(vf[ks].v.w));
  }

  const bool wave_on = key0 + wave * 32 < Tn;      // a wave whose 32 keys all lie past T only serves the DMA stream and the barriers

  // ---- DMA sources of this wave's pieces (lane constants; a tile adds a uniform base) ----
  //   wave w: Q piece w, dO piece w (rows 8 w .. 8 w + 7), bias pieces 2 w, 2 w + 1 (four rows each) + one small piece
  //   (lse | delta | keep words 0..63 | 64..127)
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t ldq2 = (uint32_t)(p.ld * 2), ldo2 = (uint32_t)(p.ld_out * 2), ldb2 = (uint32_t)(p.bias_ld * 2);
  // lane offsets of the four 1 KiB pieces (rows clamped to `lim`: only the last tile has rows past T, and recomputes them)
  auto piece_offsets = [&](int lim, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
    const int bcl = (int)p.bias_ld * 2 - 16;                             // (a key block may pass the padded row end: clamp)
    const int qrow = 8 * wave + (lane >> 3);
    const uint32_t qcol = (uint32_t)(((lane & 7) ^ qslot_swz(qrow)) << 4);
    const int brow0 = 8 * wave + (lane >> 4);
    o0 = (uint32_t)min(qrow, lim) * ldq2 + qcol;
    o1 = (uint32_t)min(qrow, lim) * ldo2 + qcol;
    o2 = (uint32_t)min(brow0, lim) * ldb2 + (uint32_t)min(key0 * 2 + (((lane & 15) ^ ((brow0 & 3) << 2)) << 4), bcl);
    o3 = (uint32_t)min(brow0 + 4, lim) * ldb2 + (uint32_t)min(key0 * 2 + (((lane & 15) ^ (((brow0 + 4) & 3) << 2)) << 4), bcl);
  };
  uint32_t po0, po1, po2, po3;
  piece_offsets(TQ - 1, po0, po1, po2, po3);
  // tiles are requested strictly in order: the uniform source pointers of the NEXT tile to request run along (scalar adds)
  const char* nq = reinterpret_cast<const char*>(Q);
  const char* ndo = reinterpret_cast<const char*>(dO);
  const char* nb = reinterpret_cast<const char*>(bias);
  const char* ndl = reinterpret_cast<const char*>(delta);
  const char* nw = DROP ? reinterpret_cast<const char*>(p.words + (int64_t)bh * p.nqb * (int64_t)p.Tk + key0) : nullptr;
  const float* nl = lse;
  int nq0 = 0;                                                            // first query of that tile
  auto dma_tile = [&](auto buf_c) {
    const int buf = buf_c;                              // an integral_constant (static ring position) or a plain int
    const int lim = Tn - 1 - nq0;
    const uint32_t dst = lds0 + buf * BUF;
    uint32_t o0 = po0, o1 = po1, o2 = po2, o3 = po3;
    if (lim < TQ - 1) piece_offsets(lim, o0, o1, o2, o3);               // the last tile (uniform branch)
    lds_dma_1k_s<0>(o0, nq, dst + O_Q + wave * 1024);
    lds_dma_1k_s<0>(o1, ndo, dst + O_DO + wave * 1024);
    lds_dma_1k_s<0>(o2, nb, dst + O_BIAS + wave * 2048);
    lds_dma_1k_s<0>(o3, nb, dst + O_BIAS + wave * 2048 + 1024);
    if (wave == 0) {
      const char* src = (nq0 + lane < Tn) ? reinterpret_cast<const char*>(nl + lane) : reinterpret_cast<const char*>(&g_lse_pad);
      dma4_v(src, dst + O_LSE);
    } else if (wave == 1) {
      dma4_s((uint32_t)min(lane, lim) * 4u, ndl, dst + O_DELTA);
    } else if (DROP) {
      dma4_s((uint32_t)((wave - 2) * 64 + lane) * 4u, nw, dst + O_WORDS + (wave - 2) * 256);
    }
    nq += (int64_t)TQ * ldq2; ndo += (int64_t)TQ * ldo2; nb += (int64_t)TQ * ldb2; ndl += TQ * 4; nl += TQ; nq0 += TQ;
    if (DROP) nw += (int64_t)p.Tk * 4;
  };

  // ---- LDS read addresses (lane constants relative to a buffer) ----
  // S / dP fragment of k-step ks: row l31, slot (2 ks + half) ^ swz = base ^ (ks << 5)
  const int aS0 = l31 * 128 + ((half ^ qslot_swz(l31)) << 4);
  const int li = lane & 15, lj = (lane >> 4) & 1;
  int aT[2][2];                                      // transpose reads of the Q / dO tile: [d tile][rows +0 / +8]; + 2048 kk
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int w8 = 0; w8 < 2; ++w8) {
      const int row = 4 * half + (li >> 2) + 8 * w8;
      const int slot = 4 * t2 + 2 * lj + ((li & 3) >> 1);
      aT[t2][w8] = row * 128 + ((slot ^ qslot_swz(row)) << 4) + ((li & 1) << 3);
    }
  const int aB = O_BIAS + (4 * half + (li >> 2)) * 256 + (((4 * wave + 2 * lj + ((li & 3) >> 1)) ^ ((li >> 2) << 2)) << 4) + ((li & 1) << 3);
  const int aW = attn_kslot(keyl) * 4;
  char* stg = smem + NBUF * BUF + wave * STG_WAVE;
  char* stg_w = stg + l31 * STG_PITCH + 8 * half;                       // + 16 g
  const char* stg_r = stg + (lane >> 2) * STG_PITCH + (lane & 3) * 16;  // + 16 rows: STG_PITCH * 16
  T* slab = reinterpret_cast<T*>(p.ds_out) + ((int64_t)bh * Tn + key0 + wave * 32) * (int64_t)Tp;   // uniform
  const int slab_l = (lane >> 2) * Tp + (lane & 3) * 8;                                              // + 16 rows: 16 Tp

  f32x16_t dk[2], dv[2], s;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; s[r] = 0.f; }

  auto s_tile = [&](const char* buf) {               // S[q, key] of a tile: A = Q rows, B = K fragments
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag<T> a;
      a.v = *reinterpret_cast<const uint4*>(buf + O_Q + (aS0 ^ (ks << 5)));
      mma_step(a, kf[ks], acc);
    }
    return acc;
  };
  auto flush = [&](int tt) {                         // dS^T of tile tt: the wave's patch -> 64-byte row pieces of the slab
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (lane >> 2) + 16 * i;
      const uint4 v = *reinterpret_cast<const uint4*>(stg_r + i * 16 * STG_PITCH);
      if (key0 + wave * 32 + row < Tn) *reinterpret_cast<uint4*>(slab + tt * TQ + 16 * i * Tp + slab_l) = v;
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  dma_tile(I0{});
  if (nt > 1) dma_tile(I1{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  s = s_tile(smem);

  // one tile; the ring position is a compile-time constant (the loop below is unrolled over the ring), so every LDS address of
  // the body is a lane constant plus an immediate
  auto tile = [&](auto cur_c, int t) {
    const int cur = cur_c, nxt = cur + 1 == NBUF ? 0 : cur + 1, nn = nxt + 1 == NBUF ? 0 : nxt + 1;
    if (t > 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next tile has landed (and the stores of tile t - 2 are out)
      __builtin_amdgcn_s_barrier();
    }
    if (t + 2 < nt) dma_tile(nn);
    if (t > 0) flush(t - 1);
    const char* buf = smem + cur * BUF;
    if (!wave_on) return;                              // (wave-uniform; the barrier and this wave's DMA pieces are above)

    // dP[q, key] = dO . V^T
    f32x16_t dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag<T> a;
      a.v = *reinterpret_cast<const uint4*>(buf + O_DO + (aS0 ^ (ks << 5)));
      mma_step(a, vf[ks], dp);
    }
    uint32_t cw = 0u;
    if constexpr (DROP) cw = *reinterpret_cast<const uint32_t*>(buf + O_WORDS + aW) >> (4 * half);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // P = exp2((S scale + bias) log2e - lse); dS = P (dP keep / (1 - p) - delta); bf16 pairs in MFMA operand order
      uint32_t pp[4], dd[4];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * kk + gg;
        const float4 l4 = *reinterpret_cast<const float4*>(buf + O_LSE + 16 * half + 32 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(buf + O_DELTA + 16 * half + 32 * g);
        const uint2 bu = tr64(buf + aB + 2048 * g);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
        const float bf[4] = {__uint_as_float(bu.x << 16), __uint_as_float(bu.x & 0xffff0000u), __uint_as_float(bu.y << 16),
                             __uint_as_float(bu.y & 0xffff0000u)};
        float pd[4], ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pv = __builtin_amdgcn_exp2f(fmaf(fmaf(s[r], p.scale, bf[e]), kLog2e, -lv[e]));   // (bit for bit the fused kernel's order)
          if constexpr (DROP) {
            const int m = __builtin_amdgcn_sbfe((int)cw, 8 * g + e, 1);   // 0 / -1: bit (8 g + e) = this lane's query of register r
            pd[e] = __uint_as_float(and_u(pv, m));                        // (1 / (1 - p) goes onto dV once, at the end)
            ds[e] = pv * fmaf(dp[r], __uint_as_float(and_u(keep_scale, m)), -dl[e]);
          } else {
            pd[e] = pv;
            ds[e] = pv * (dp[r] - dl[e]);
          }
        }
        pp[2 * gg] = pack_bf16x2(pd[0], pd[1]); pp[2 * gg + 1] = pack_bf16x2(pd[2], pd[3]);
        dd[2 * gg] = pack_bf16x2(ds[0], ds[1]); dd[2 * gg + 1] = pack_bf16x2(ds[2], ds[3]);
        *reinterpret_cast<uint2*>(stg_w + 16 * g) = make_uint2(dd[2 * gg], dd[2 * gg + 1]);
      }
      // dV^T[d, key] += dO^T[d, q] . P[q, key] ;  dK^T[d, key] += Q^T[d, q] . dS[q, key]   (these 16 queries)
      Frag<T> pf, dsf;
      pf.v = make_uint4(pp[0], pp[1], pp[2], pp[3]);
      dsf.v = make_uint4(dd[0], dd[1], dd[2], dd[3]);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const Frag<T> dot = tr_frag(buf + O_DO + 2048 * kk + aT[t2][0], buf + O_DO + 2048 * kk + aT[t2][1]);
        mma_step(dot, pf, dv[t2]);
        const Frag<T> qtf = tr_frag(buf + O_Q + 2048 * kk + aT[t2][0], buf + O_Q + 2048 * kk + aT[t2][1]);
        mma_step(qtf, dsf, dk[t2]);
      }
    }
    if (t + 1 < nt) s = s_tile(smem + nxt * BUF);
  };
  {                            // the ring position is static: three copies of the body
    int t = 0;
    for (; t + 3 <= nt; t += 3) { tile(I0{}, t); tile(I1{}, t + 1); tile(I2{}, t + 2); }
    if (t < nt) tile(I0{}, t);
    if (t + 1 < nt) tile(I1{}, t + 1);
  }
  flush(nt - 1);
  // the slab's columns between the last tile and Tp stay zero (its readers load whole 16-byte groups up to Tp)
  for (int c = nt * TQ; c < Tp; c += TQ) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (lane >> 2) + 16 * i;
      if (key0 + wave * 32 + row < Tn) *reinterpret_cast<uint4*>(slab + c + 16 * i * Tp + slab_l) = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  // ---- dK, dV rows: accumulator = [d rows (registers)][key (lane)]; a lane holds d = 8 g + 4 half + 0..3 of each group g: two
  //      groups and a v_permlane32_swap make 16 contiguous bytes per lane ----
  if (mykey < Tn) {
    T* DK = reinterpret_cast<T*>(p.dk) + ((int64_t)b * Tn + mykey) * p.ld_d + h * 64;
    T* DV = reinterpret_cast<T*>(p.dv) + ((int64_t)b * Tn + mykey) * p.ld_d + h * 64;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float mul = which == 0 ? p.scale : keep_scale;
      T* dst = which == 0 ? DK : DV;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const f32x16_t& a = which == 0 ? dk[t2] : dv[t2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          uint32_t ax = pack_bf16x2(a[8 * m + 0] * mul, a[8 * m + 1] * mul), ay = pack_bf16x2(a[8 * m + 2] * mul, a[8 * m + 3] * mul);
          uint32_t bx = pack_bf16x2(a[8 * m + 4] * mul, a[8 * m + 5] * mul), by = pack_bf16x2(a[8 * m + 6] * mul, a[8 * m + 7] * mul);
          const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          *reinterpret_cast<uint4*>(dst + 32 * t2 + 16 * m + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
      }
    }
  }
}


// ================================================================================================
// dQ[q, :] = scale * sum_key dS[q, key] K[key, :] from this layer's dS^T slab (key-major [B, nh, T, Tp]), pipelined like the kernel
// above (replaces attn_dq_from_ds_kernel of attention.hip for head dim 64; same sums in the same order: bit-identical).
// Workgroup = 128 queries (lane = query: dQ^T[d, q] = K^T[d, key] . dS^T[key, q], so a lane ends up with 4 consecutive d of its row
// per register group and the rows leave by v_permlane32_swap pairs), streaming 32-key tiles: the slab block [32 keys][128 q]
// (256-byte rows, source slots permuted by (row & 3) << 2) and the K rows [32][128 B] (slot ^ bitrev3(row >> 1)) arrive by LDS-DMA
// into a ring of three; both operands are read with the hardware transpose read.  K rows past T come from a zero line (the slab rows
// read beside them are clamped to T - 1: finite x 0).
// ================================================================================================
constexpr int DQ_O_S = 0, DQ_O_K = 8192, DQ_BUF = 12288;
__device__ uint4 g_zero_line[8];    // 128 zero bytes (device globals are zero-initialised)

__global__ __launch_bounds__(256, 2) void attn_dq_pipe_kernel(AttnParams p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tn = p.T, Tp = p.Tp;
  const int nqb = (Tn + 127) / 128;
  int u;
  {
    const int nwg = gridDim.x, L = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, x = L & 7, i = L >> 3;
    u = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
  }
  const int qb = u % nqb, bh = u / nqb, h = bh % p.nh, b = bh / p.nh;
  const int q0 = qb * 128, myq = q0 + wave * 32 + l31;
  const T* DS = reinterpret_cast<const T*>(p.ds_out) + (int64_t)bh * Tn * (int64_t)Tp;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * 64;
  const int nt = (Tn + 31) / 32;
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t lds2 = (uint32_t)(Tp * 2), ldk2 = (uint32_t)(p.ld * 2);
  // DMA: wave w sends slab pieces 2 w, 2 w + 1 (four key rows each) and K piece w (eight key rows)
  const int srow0 = 8 * wave + (lane >> 4), krow = 8 * wave + (lane >> 3);
  const int scl = Tp * 2 - 16;                                   // (a query block may pass the padded row end: clamp; those rows are not stored)
  const uint32_t sc0 = (uint32_t)min(q0 * 2 + (((lane & 15) ^ ((srow0 & 3) << 2)) << 4), scl);
  const uint32_t sc1 = (uint32_t)min(q0 * 2 + (((lane & 15) ^ (((srow0 + 4) & 3) << 2)) << 4), scl);
  const uint32_t kcol = (uint32_t)(((lane & 7) ^ qslot_swz(krow)) << 4);
  const char* ns = reinterpret_cast<const char*>(DS);
  const char* nk = reinterpret_cast<const char*>(K);
  int nk0 = 0;
  auto dma_tile = [&](auto buf_c) {
    const int buf = buf_c;
    const uint32_t dst = lds0 + buf * DQ_BUF;
    const int lim = Tn - 1 - nk0;
    lds_dma_1k_s<0>((uint32_t)min(srow0, lim) * lds2 + sc0, ns, dst + DQ_O_S + wave * 2048);
    lds_dma_1k_s<0>((uint32_t)min(srow0 + 4, lim) * lds2 + sc1, ns, dst + DQ_O_S + wave * 2048 + 1024);
    if (lim >= 31) {
      lds_dma_1k_s<0>((uint32_t)krow * ldk2 + kcol, nk, dst + DQ_O_K + wave * 1024);
    } else {                                                      // the last tile: K rows past T read the zero line
      const char* src = krow <= lim ? nk + (uint32_t)krow * ldk2 + kcol : reinterpret_cast<const char*>(g_zero_line) + (lane & 7) * 16;
      lds_dma_1k<0>(src, dst + DQ_O_K + wave * 1024);
    }
    ns += (int64_t)32 * lds2; nk += (int64_t)32 * ldk2; nk0 += 32;
  };
  // transpose-read addresses: [4 key rows][16 columns] blocks; keys 16 kk + 8 half + {0..3} and + 4
  const int li = lane & 15, lj = (lane >> 4) & 1;
  int aA[2], aK[2][2];
#pragma unroll
  for (int w4 = 0; w4 < 2; ++w4) {
    const int row = 8 * half + 4 * w4 + (li >> 2);
    aA[w4] = DQ_O_S + row * 256 + (((4 * wave + 2 * lj + ((li & 3) >> 1)) ^ ((row & 3) << 2)) << 4) + ((li & 1) << 3);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
      aK[t2][w4] = DQ_O_K + row * 128 + (((4 * t2 + 2 * lj + ((li & 3) >> 1)) ^ qslot_swz(row)) << 4) + ((li & 1) << 3);
  }
  f32x16_t acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  dma_tile(I0{});
  if (nt > 1) dma_tile(I1{});
  auto tile = [&](auto cur_c, int t) {
    const int cur = cur_c, nxt = cur + 1 == 3 ? 0 : cur + 1, nn = nxt + 1 == 3 ? 0 : nxt + 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) dma_tile(nn);
    const char* buf = smem + cur * DQ_BUF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint2 b0 = tr64(buf + aA[0] + 4096 * kk), b1 = tr64(buf + aA[1] + 4096 * kk);
      Frag<T> bf;                                                 // dS^T[key, q]: this lane's query, keys 16 kk + 8 half + 0..7
      bf.v = make_uint4(b0.x, b0.y, b1.x, b1.y);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const uint2 a0 = tr64(buf + aK[t2][0] + 2048 * kk), a1 = tr64(buf + aK[t2][1] + 2048 * kk);
        Frag<T> af;                                               // K^T[d, key]
        af.v = make_uint4(a0.x, a0.y, a1.x, a1.y);
        mma_step(af, bf, acc[t2]);
      }
    }
  };
  {
    int t = 0;
    for (; t + 3 <= nt; t += 3) { tile(I0{}, t); tile(I1{}, t + 1); tile(I2{}, t + 2); }
    if (t < nt) tile(I0{}, t);
    if (t + 1 < nt) tile(I1{}, t + 1);
  }
  if (myq < Tn) {
    T* dst = reinterpret_cast<T*>(p.dq) + ((int64_t)b * Tn + myq) * p.ld_d + h * 64;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x16_t& a = acc[t2];
        uint32_t ax = pack_bf16x2(a[8 * m + 0] * p.scale, a[8 * m + 1] * p.scale), ay = pack_bf16x2(a[8 * m + 2] * p.scale, a[8 * m + 3] * p.scale);
        uint32_t bx = pack_bf16x2(a[8 * m + 4] * p.scale, a[8 * m + 5] * p.scale), by = pack_bf16x2(a[8 * m + 6] * p.scale, a[8 * m + 7] * p.scale);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        *reinterpret_cast<uint4*>(dst + 32 * t2 + 16 * m + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
      }
  }
}

}  // namespace

bool attn_bwd_pipe_supported(const AttnParams& p) {
  static const bool on = [] { const char* e = getenv("PENEO_ATTN_BWD_PIPE"); return !e || atoi(e) != 0; }();
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return on && p.d == 64 && p.bias != nullptr && p.key_bias == nullptr && p.ds_out != nullptr && p.g_bias == nullptr &&
         al16(p.q) && al16(p.k) && al16(p.v) && al16(p.d_out) && al16(p.dk) && al16(p.dv) && al16(p.bias) && al16(p.ds_out) &&
         (p.ld * 2) % 16 == 0 && (p.ld_out * 2) % 16 == 0 && (p.ld_d * 2) % 16 == 0 && (p.bias_ld * 2) % 16 == 0 &&
         p.bias_ld * 2 >= 256 && (int64_t)p.ld * 2 * 32 < (1ll << 31) && (int64_t)p.bias_ld * 2 * 32 < (1ll << 31);
}

int launch_attn_bwd_pipe(const AttnParams& p, hipStream_t st) {
  const dim3 grid((unsigned)((int64_t)((p.T + WK - 1) / WK) * p.nh * p.B));
  auto go = [&](auto kern, int lds) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("peneo_attn_bwd: cannot raise dynamic LDS to %d bytes", lds);
      return PENEO_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    return check_launch("peneo_attn_bwd(pipe)");
  };
  int rc = p.drop_p > 0.f ? go(attn_bwd_pipe_kernel<true>, LDS_BYTES) : go(attn_bwd_pipe_kernel<false>, LDS_BYTES);
  if (rc) return rc;
  // dQ from the slab just written
  static const bool dq_pipe = [] { const char* e = getenv("PENEO_ATTN_DQ_PIPE"); return !e || atoi(e) != 0; }();
  if (!dq_pipe || ((reinterpret_cast<uintptr_t>(p.dq) & 15) != 0)) return 1;   // 1: the caller launches attn_dq_from_ds_kernel
  const dim3 qgrid((unsigned)((int64_t)((p.T + 127) / 128) * p.nh * p.B));
  hipLaunchKernelGGL(attn_dq_pipe_kernel, qgrid, dim3(256), 3 * DQ_BUF, st, p);
  return check_launch("peneo_attn_bwd(dq pipe)");
}

}  // namespace peneo
