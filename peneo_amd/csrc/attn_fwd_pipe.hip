// K6 forward, pipelined (round 5; bf16, head dim 64, bias tensor, V row-major).  Same arithmetic, in the same order, as
// attn_fwd_kernel (attention.hip; reference: modeling_layoutlmv3.py:308-321,365-404) - online softmax over 32-key blocks in the exp2
// domain with the lazily moved maximum - and the same outputs bit for bit; the stream runs like the backward's (attn_bwd_pipe.hip):
//
//   * workgroup = 128 queries (lane = query, Q fragments and the O^T accumulator in registers) streaming 32-key tiles; K, V, the bias
//     block [128 q][32 keys] and the keep words of a tile arrive by LDS-DMA into a ring of three buffers (request for tile t + 2 at
//     the top of iteration t), ONE s_barrier per tile (attn_fwd_kernel: register-staged tiles, two barriers per tile);
//   * S^T of tile t + 1 is computed at the end of iteration t;
//   * lane-linear DMA images, conflicts removed by permuting the SOURCE 16-byte slots of a row: K / V rows (128 B) use
//     slot ^ bitrev3(row >> 1) (b128 fragment reads of S^T and transpose reads of V^T both conflict-free), the bias rows (64 B: a
//     lane reads 8 bytes of ITS query's row) use slot ^ ((row >> 2) & 3) (two-way: the 8-byte halves cannot be permuted by a 16-byte DMA);
//   * keys past T: K / V rows are clamped to T - 1 and the ragged last tile masks its scores itself (round 6: one wave-uniform branch
//     per launch; until then the kernel relied on -1e30 in the bias tensor's padding columns, which peneo_attn_fwd does not promise);
//   * O rows leave as 16-byte pieces straight from the accumulator layout (v_permlane32_swap pairs), no LDS round trip.
#include <cstdlib>
#include "common.h"
#include "attention.h"

namespace peneo {
namespace {

constexpr int TK = 32;     // keys per tile
constexpr int WQ = 128;    // queries per workgroup (4 waves x 32)
// a tile's buffer: K [32][128 B], V [32][128 B], bias [128 q][64 B], keep words [4 query blocks][64 slots]
constexpr int O_K = 0, O_V = 4096, O_BIAS = 8192, O_WORDS = 16384, BUF = 17408;
constexpr int NBUF = 3;
constexpr int LDS_BYTES = NBUF * BUF;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kMasked = -1.0e30f;
constexpr float kRescaleTau = 4.0f;   // (attention.hip: RESCALE_TAU)

__device__ __forceinline__ int bitrev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
__device__ __forceinline__ int kslot_swz(int row) { return bitrev3((row >> 1) & 7); }

typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr64(const char* p) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_s4p;
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)p));
}
__device__ __forceinline__ void dma4_s(uint32_t voff_lane, const char* base_uniform, uint32_t lds_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_lane), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
// x where the lane's bit of the 64-bit mask is set, else 0 (attention.hip: mask_keep; the s_nop pads the VALU-written-SGPR hazard)
__device__ __forceinline__ float mask_keep(float x, uint64_t m) {
  float r;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
  return r;
}
__device__ __forceinline__ uint64_t lane_words_mask(uint32_t w, int word /* even, compile-time */) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)w, word), hi = (uint32_t)__builtin_amdgcn_readlane((int)w, word + 1);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(AttnParams p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tn = p.T;
  // unit order: the query blocks of one (document, head) run on ONE XCD (they stream the same K / V rows through its L2)
  const int nqb = (Tn + WQ - 1) / WQ;
  int u;
  {
    const int nwg = gridDim.x, L = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, x = L & 7, i = L >> 3;
    u = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
  }
  const int qb = u % nqb, bh = u / nqb, h = bh % p.nh, b = bh / p.nh;
  const int q0 = qb * WQ;
  const int myq = q0 + wave * 32 + l31;
  const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + h * 64;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + h * 64;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + h * 64;
  const T* bias = reinterpret_cast<const T*>(p.bias) + (int64_t)bh * Tn * p.bias_ld;
  const float keep_scale = DROP ? p.keep_scale : 1.0f;
  const int nt = (Tn + TK - 1) / TK;

  // ---- Q fragments of this lane's query (B operands of S^T) ----
  Frag<T> qf[4];
  {
    const bool ok = myq < Tn;
    const T* qr = Q + (int64_t)(ok ? myq : 0) * p.ld + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks].v = *reinterpret_cast<const uint4*>(qr + 16 * ks);
      if (!ok) qf[ks].v = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the vm counter holds the DMA pieces only
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks].v.x), "+v"(qf[ks].v.y), "+v"(qf[ks].v.z), "+v"(qf[ks].v.w));

  // ---- DMA: wave w sends K piece w, V piece w (rows 8 w .. 8 w + 7 of the tile), the bias rows of ITS 32 queries (two pieces of
  //      16 rows x 64 B) and, with dropout, the 64 keep-word slots that start at the tile's first key for its query block ----
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t ldk2 = (uint32_t)(p.ld * 2), ldb2 = (uint32_t)(p.bias_ld * 2);
  const int krow = 8 * wave + (lane >> 3);
  const uint32_t kcol = (uint32_t)(((lane & 7) ^ kslot_swz(krow)) << 4);
  uint32_t bo0, bo1;                                 // bias pieces: row = 32 wave + 16 j + lane / 4, slot lane & 3 (source slot permuted)
  {
    const int r0 = 32 * wave + (lane >> 2), r1 = r0 + 16;
    const int lim = Tn - 1 - q0;                     // (query rows past T: clamped)
    bo0 = (uint32_t)min(r0, lim) * ldb2 + (uint32_t)(((lane & 3) ^ ((r0 >> 2) & 3)) << 4);
    bo1 = (uint32_t)min(r1, lim) * ldb2 + (uint32_t)(((lane & 3) ^ ((r1 >> 2) & 3)) << 4);
  }
  const char* nk = reinterpret_cast<const char*>(K);
  const char* nv = reinterpret_cast<const char*>(V);
  const char* nb = reinterpret_cast<const char*>(bias + (int64_t)q0 * p.bias_ld);
  const char* nw = DROP ? reinterpret_cast<const char*>(p.words + ((int64_t)bh * p.nqb + (qb * 4 + wave)) * (int64_t)p.Tk) : nullptr;
  int nk0 = 0;                                       // first key of the next tile to request
  auto dma_tile = [&](auto buf_c) {
    const int buf = buf_c;
    const uint32_t dst = lds0 + buf * BUF;
    const uint32_t ko = (uint32_t)min(krow, Tn - 1 - nk0) * ldk2 + kcol;     // (key rows past T: clamped; their bias is -1e30)
    lds_dma_1k_s<0>(ko, nk, dst + O_K + wave * 1024);
    lds_dma_1k_s<0>(ko, nv, dst + O_V + wave * 1024);
    lds_dma_1k_s<0>(bo0, nb, dst + O_BIAS + wave * 2048);
    lds_dma_1k_s<0>(bo1, nb, dst + O_BIAS + wave * 2048 + 1024);
    if (DROP) dma4_s((uint32_t)min(nk0 + lane, p.Tk - 1) * 4u, nw, dst + O_WORDS + wave * 256);
    nk += (int64_t)TK * ldk2; nv += (int64_t)TK * ldk2; nb += TK * 2; nk0 += TK;
  };

  // ---- LDS read addresses (lane constants relative to a buffer) ----
  const int aS0 = l31 * 128 + ((half ^ kslot_swz(l31)) << 4);           // K fragment of k-step ks: ^ (ks << 5)
  const int li = lane & 15, lj = (lane >> 4) & 1;
  int aT[2][2];                                                          // transpose reads of the V tile: [d tile][rows +0 / +8]; + 2048 kk
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int w8 = 0; w8 < 2; ++w8) {
      const int row = 4 * half + (li >> 2) + 8 * w8;
      const int slot = 4 * t2 + 2 * lj + ((li & 3) >> 1);
      aT[t2][w8] = row * 128 + ((slot ^ kslot_swz(row)) << 4) + ((li & 1) << 3);
    }
  const int qrow = wave * 32 + l31;
  const int aB0 = O_BIAS + qrow * 64 + 8 * half;                         // + ((g ^ swz) << 4)
  const int bswz = (qrow >> 2) & 3;
  const int aW = O_WORDS + wave * 256 + lane * 4;

  f32x16_t o[2], s;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; s[r] = 0.f; }
  float m_run = kMasked, l_run = 0.f;

  auto s_tile = [&](const char* buf) {               // S^T[key, q] of a tile: A = K rows, B = Q fragments
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag<T> a;
      a.v = *reinterpret_cast<const uint4*>(buf + O_K + (aS0 ^ (ks << 5)));
      mma_step(a, qf[ks], acc);
    }
    return acc;
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  dma_tile(I0{});
  if (nt > 1) dma_tile(I1{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  s = s_tile(smem);

  auto tile = [&](auto cur_c, int t) {
    const int cur = cur_c, nxt = cur + 1 == NBUF ? 0 : cur + 1, nn = nxt + 1 == NBUF ? 0 : nxt + 1;
    if (t > 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next tile has landed
      __builtin_amdgcn_s_barrier();
    }
    if (t + 2 < nt) dma_tile(nn);
    const char* buf = smem + cur * BUF;
    uint32_t cw = 0u;
    if constexpr (DROP) cw = *reinterpret_cast<const uint32_t*>(buf + aW);   // lane L: keep word of key slot L of this tile
    // scores (natural units) and the block's row maximum  (attention.hip: the `block` lambda of attn_fwd_kernel, same order)
    float mt = kMasked;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2 bu = *reinterpret_cast<const uint2*>(buf + aB0 + ((g ^ bswz) << 4));
      const float bb[4] = {__uint_as_float(bu.x << 16), __uint_as_float(bu.x & 0xffff0000u), __uint_as_float(bu.y << 16),
                           __uint_as_float(bu.y & 0xffff0000u)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(s[4 * g + e], p.scale, bb[e]);
        s[4 * g + e] = v;
        mt = fmaxf(mt, v);
      }
    }
    // Keys past T (the ragged last tile only; a wave-uniform branch taken once per launch).  Round 6: masked HERE, as attn_fwd_kernel
    // does -- until then the kernel relied on the caller's bias holding -1e30 in its padding columns (true for peneo_relpos_bias_fwd's
    // output, not part of peneo_attn_fwd's contract: a zero or uninitialised padding put the clamped key rows into the softmax).
    if (t + 1 == nt && (Tn & (TK - 1)) != 0) {
      mt = kMasked;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (t * TK + 8 * g + 4 * half + e >= Tn) s[4 * g + e] = kMasked;
          mt = fmaxf(mt, s[4 * g + e]);
        }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = (mt > m_run + kRescaleTau) ? mt : m_run;
    if (__builtin_amdgcn_ballot_w64(m_new != m_run)) {   // rare after the first tiles (wave-uniform branch)
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t2][r] *= alpha;
    }
    const float nm = -m_run * kLog2e;
    float ls = 0.f;
    auto soft = [&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      float e = __builtin_amdgcn_exp2f(fmaf(s[r], kLog2e, nm));
      ls += e;
      if constexpr (DROP) e = mask_keep(e, lane_words_mask(cw, 2 * r));
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
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = s[8 * kh + e];
      const Frag<T> pf = pack_frag8<T>(pv);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const uint2 a0 = tr64(buf + O_V + 2048 * kh + aT[t2][0]), a1 = tr64(buf + O_V + 2048 * kh + aT[t2][1]);
        Frag<T> vf;
        vf.v = make_uint4(a0.x, a0.y, a1.x, a1.y);
        mma_step(vf, pf, o[t2]);
      }
    }
    if (t + 1 < nt) s = s_tile(smem + nxt * BUF);
  };
  {
    int t = 0;
    for (; t + 3 <= nt; t += 3) { tile(I0{}, t); tile(I1{}, t + 1); tile(I2{}, t + 2); }
    if (t < nt) tile(I0{}, t);
    if (t + 1 < nt) tile(I1{}, t + 1);
  }

  // ---- normalise; O rows: accumulator = [d rows (registers)][query (lane)], two groups + a v_permlane32_swap = 16 bytes per lane ----
  const bool any = m_run > 0.5f * kMasked;
  const float inv = (any && l_run > 0.f) ? keep_scale / l_run : 0.f;
  if (half == 0 && myq < Tn && p.lse) p.lse[(int64_t)bh * Tn + myq] = any ? fmaf(m_run, kLog2e, log2f(l_run)) : kMasked;   // log2 units
  if (myq < Tn) {
    T* dst = reinterpret_cast<T*>(p.out) + ((int64_t)b * Tn + myq) * p.ld_out + h * 64;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x16_t& a = o[t2];
        uint32_t ax = pack_bf16x2(a[8 * m + 0] * inv, a[8 * m + 1] * inv), ay = pack_bf16x2(a[8 * m + 2] * inv, a[8 * m + 3] * inv);
        uint32_t bx = pack_bf16x2(a[8 * m + 4] * inv, a[8 * m + 5] * inv), by = pack_bf16x2(a[8 * m + 6] * inv, a[8 * m + 7] * inv);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        *reinterpret_cast<uint4*>(dst + 32 * t2 + 16 * m + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
      }
  }
}

}  // namespace

bool attn_fwd_pipe_supported(const AttnParams& p) {
  static const bool on = [] { const char* e = getenv("PENEO_ATTN_FWD_PIPE"); return !e || atoi(e) != 0; }();
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return on && p.d == 64 && p.bias != nullptr && p.key_bias == nullptr && p.vt == nullptr && p.v != nullptr &&
         al16(p.q) && al16(p.k) && al16(p.v) && al16(p.out) && al16(p.bias) &&
         (p.ld * 2) % 16 == 0 && (p.ld_out * 2) % 16 == 0 && (p.bias_ld * 2) % 16 == 0 && p.bias_ld >= 32 &&
         (int64_t)p.ld * 2 * 32 < (1ll << 31) && (int64_t)p.bias_ld * 2 * 128 < (1ll << 31) &&
         (p.T + TK - 1) / TK * TK <= (int)p.bias_ld;       // (whole 32-key tiles inside the padded bias row)
}

int launch_attn_fwd_pipe(const AttnParams& p, hipStream_t st) {
  const dim3 grid((unsigned)((int64_t)((p.T + WQ - 1) / WQ) * p.nh * p.B));
  auto go = [&](auto kern) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      set_error("peneo_attn_fwd: cannot raise dynamic LDS to %d bytes", LDS_BYTES);
      return PENEO_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, p);
    return check_launch("peneo_attn_fwd(pipe)");
  };
  return p.drop_p > 0.f ? go(attn_fwd_pipe_kernel<true>) : go(attn_fwd_pipe_kernel<false>);
}

}  // namespace peneo
