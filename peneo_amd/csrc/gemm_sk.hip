// bf16 GEMM as ONE persistent launch: at most one 8-wave workgroup per CU, each walking a contiguous range of
// (tile, k-tile) units -- "stream-k" (gfx950).
//
// Why (profiles/r03_gemm_fixed_cost_and_host_time.txt, r05_vendor_gemm_configs.txt): the tiled kernels of gemm.hip / gemm_big.hip
// multiply at ~920 TFLOP/s per ADDED k-tile, but a QKV-shaped launch costs 17 us before its first and after its last k-tile
// (cold instruction fetch, first-touch load latency, epilogue with nothing to overlap, 1.6 rounds of tiles on the CUs), and the
// M = 5672 problems quantise badly on any tile grid (270 / 414 / 552 / 810 tiles on 256 CUs).  Here
//   * the launch has G <= #CUs workgroups; workgroup v owns units [v U / G, (v + 1) U / G) of the U = tiles x k-tiles units, in
//     tile-major order: every CU does the same number of k-tiles (+- 1) whatever the tile count;
//   * the LDS ring is ONE stream over the units: the LDS-DMA of the next tile's first k-tiles is in flight while the last k-tiles
//     of this tile are multiplied and while its epilogue runs (the epilogue never touches LDS, below), so a tile boundary costs
//     no prologue;
//   * a tile cut by a range boundary is finished by the workgroup that holds its k = 0 piece: every other piece is the FIRST
//     thing its workgroup computes, leaves as an fp32 slab in accumulator-register order (16-byte write-through stores, 1 KiB
//     per wave instruction; every wave drains its vm counter, barrier, one lane stores the flag) and is added by the finisher --
//     at the END of its own range -- behind one relaxed poll + one agent-scope acquire (cdna guide, Guideline 16 recipe R1).  Flags are reset by
//     their single consumer, so a captured launch replays.  Pieces of one tile sit on consecutive v = the same XCD (v is
//     XCD-major), except at the seven XCD seams.
//   * progress: a finisher waits only for workgroups v + 1 .. v + 3, which publish before anything else; workgroups of an XCD
//     are dispatched in id order, so whatever subset is resident, its lowest members can finish (the vendor library's stream-k
//     kernels rely on the same).
// Operands / LDS images / ring are gemm_big.hip's (A k-major [M][K]; B k-major [N][K] or mn-major [K][N]; rows of 128 B with
// the 16-byte slot swizzled on the DMA source address; one s_barrier per k-tile; LDS-DMA NSTAGE - 1 units ahead, issued from
// inline asm in four groups between the MFMA steps).
// Epilogue: the MFMA runs with the operands SWAPPED (D = B_tile x A_tile^T), so a lane holds one output ROW (m = lane & 31) and
// 4 consecutive columns per register group; one v_permlane32_swap per register pair makes that 8 consecutive columns = the
// 16-byte granule of gemm_common.h's fused epilogue, straight from the accumulators: no LDS patch, nothing to fence against
// the ring.  The loads of the epilogue's inputs are issued in batches ahead of its stores (gemm_common.h: epilogue_apply8).
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include "common.h"
#include "gemm_common.h"

namespace peneo {

typedef short sk_s16x4 __attribute__((ext_vector_type(4)));

struct SkPlan {
  int tiles_n, tiles, ktiles, G;
  float* ws;            // [G] slabs of BM x BN floats (a workgroup's first piece when it is not the tile's k = 0 piece)
  uint32_t* flags;      // [G] 0 = empty, 1 = slab published; reset by the finisher
  uint64_t* prof;       // tools only (peneo_gemm_sk_set_prof): [G][16] s_memrealtime stamps of one lane, or null
};
#define SK_STAMP(k) do { if (pl.prof && tid == 0) pl.prof[v * 16 + (k)] = wall_clock64(); } while (0)

template <bool BK_, int WGM_, int WGN_, int FM_, int FN_, int NSTAGE_>
struct SkCfg {
  static constexpr bool BK = BK_;
  static constexpr int WGM = WGM_, WGN = WGN_, FM = FM_, FN = FN_, NSTAGE = NSTAGE_;
  static constexpr int BM = WGM * FM * 32, BN = WGN * FN * 32;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int APW = BM / 64, BPW = BN / 64, PPW = APW + BPW;   // 1 KiB pieces per wave and k-tile
  static constexpr int LDS_BYTES = NSTAGE * STAGE;
  static constexpr int SLAB_FLOATS = BM * BN;
  static_assert(WGM * WGN == 8, "eight waves");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "pieces of 8 rows are dealt to 8 waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <typename C>
__global__ __launch_bounds__(512) void gemm_sk_kernel(GemmParams p, SkPlan pl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FM = C::FM, FN = C::FN, NSTAGE = C::NSTAGE;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  // ---- this workgroup's range of units.  v is XCD-major: workgroup ids go round-robin to the 8 XCDs, so the 32 workgroups of an
  //      XCD hold consecutive ranges (neighbouring tiles share their A panel in that XCD's L2; a cut tile's pieces meet there) ----
  const int G = pl.G, lin = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = lin & 7, slot = lin >> 3;
  const int v = xcd * q8 + min(xcd, r8) + slot;
  const int64_t U = (int64_t)pl.tiles * pl.ktiles;
  const int ktiles = pl.ktiles;
  const int64_t u_begin = v * U / G, u_end = (v + 1) * U / G;
  const int n_units = (int)(u_end - u_begin);
  if (n_units <= 0) return;
  SK_STAMP(0);
  const int tile_begin = (int)(u_begin / ktiles), kt_begin = (int)(u_begin % ktiles);

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  // ---- LDS-DMA side: runs NSTAGE - 1 units ahead of the multiply.  Per tile: per-lane 32-bit byte offsets of this wave's pieces
  //      inside the tile (piece g = wave + 8 u) + one uniform 64-bit base per operand that walks k with scalar adds ----
  uint32_t offA[C::APW], offB[C::BPW];
  const char* bA = nullptr;
  const char* bB = nullptr;
  const int64_t stepB = C::BK ? 128 : (int64_t)64 * p.ldb * 2;
  auto uniform_ptr = [](const void* q) -> const char* {
    const uint64_t w = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)w), hi = __builtin_amdgcn_readfirstlane((uint32_t)(w >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto dma_set_tile = [&](int tile, int kt) {
    const int m0 = (tile / pl.tiles_n) * C::BM, n0 = (tile % pl.tiles_n) * C::BN;
#pragma unroll
    for (int u = 0; u < C::APW; ++u) {
      const int row = (wave + 8 * u) * 8 + (lane >> 3);
      const int sg = (lane & 7) ^ ((row >> 1) & 7);
      offA[u] = (uint32_t)(((int64_t)(min(m0 + row, p.M - 1) - m0) * p.lda + sg * 8) * 2);
    }
#pragma unroll
    for (int u = 0; u < C::BPW; ++u) {
      const int g = wave + 8 * u;
      if constexpr (C::BK) {
        const int row = g * 8 + (lane >> 3);
        const int sg = (lane & 7) ^ ((row >> 1) & 7);
        offB[u] = (uint32_t)(((int64_t)(min(n0 + row, p.N - 1) - n0) * p.ldb + sg * 8) * 2);
      } else {
        constexpr int NQ = C::BN / 64;
        const int kb = g / NQ, nq = g % NQ, kr = lane >> 3;
        const int cg = (lane & 7) ^ (((kr >> 1) & 1) << 2);
        offB[u] = (uint32_t)(((int64_t)(kb * 8 + kr) * p.ldb + (min(n0 + nq * 64 + cg * 8, p.N - 8) - n0)) * 2);
      }
    }
    bA = uniform_ptr(reinterpret_cast<const char*>(A + (int64_t)m0 * p.lda) + (int64_t)kt * 128);
    bB = uniform_ptr(reinterpret_cast<const char*>(C::BK ? B + (int64_t)n0 * p.ldb : B + n0) + kt * stepB);
  };
  int d_tile = tile_begin, d_kt = kt_begin;      // the next unit to issue
  dma_set_tile(d_tile, d_kt);
  const uint32_t lds0 = lds_addr(smem);
  uint32_t dbase = 0;
  // The pieces of a k-tile are issued in four groups, one between the MFMA clusters of each k-step (gemm_big.hip: a wave that
  // issues all of its pieces back to back stalls on the memory pipeline)
  auto issue_group = [&](auto gc) {
    constexpr int GI = decltype(gc)::value;
#pragma unroll
    for (int u = 0; u < C::PPW; ++u) {
      if (u * 4 / C::PPW != GI) continue;
      if (u < C::APW) lds_dma_1k_s<0>(offA[u], bA, dbase + u * 8192);
      else lds_dma_1k_s<0>(offB[u - C::APW], bB, dbase + C::A_BYTES + (u - C::APW) * 8192);
    }
    if constexpr (GI == 3) {      // the unit is complete: the bases move on, into the next tile when this one is through
      if (++d_kt == ktiles) {
        d_kt = 0;
        // (the tile after the last one of the problem is never issued: its units are beyond u_end; the offsets computed for it
        //  here are clamped addresses of the last rows and are not used)
        if (++d_tile < pl.tiles) dma_set_tile(d_tile, 0);
      } else {
        bA += 128; bB += stepB;
      }
    }
  };
  auto issue = [&](int stage) {
    dbase = __builtin_amdgcn_readfirstlane(lds0 + stage * C::STAGE + wave * 1024);
    issue_group(std::integral_constant<int, 0>{}); issue_group(std::integral_constant<int, 1>{});
    issue_group(std::integral_constant<int, 2>{}); issue_group(std::integral_constant<int, 3>{});
  };

  // ---- fragment offsets inside a stage ----
  int aoff[4], boff[C::BK ? 4 : FN];
  {
    const int row = wm * FM * 32 + (lane & 31), swz = (row >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aoff[ks] = row * 128 + (((2 * ks + half) ^ swz) << 4);
  }
  if constexpr (C::BK) {
    const int row = wn * FN * 32 + (lane & 31), swz = (row >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[ks] = C::A_BYTES + row * 128 + (((2 * ks + half) ^ swz) << 4);
  } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = wn * FN * 32 + j * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
      boff[j] = C::A_BYTES + (half * (C::BN / 64) + (n >> 6)) * 1024 + ((lane & 15) >> 2) * 128 +
                (((n & 63) * 2) ^ (((lane >> 3) & 1) << 6));
    }
  }

  f32x16_t acc[FM][FN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  auto load_a = [&](const char* st, int ks, uint4 (&fa)[FM]) {
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const uint4*>(st + aoff[ks] + i * 4096);
  };
  auto load_b = [&](const char* st, int ks, uint4 (&fb)[FN]) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if constexpr (C::BK) {
        fb[j] = *reinterpret_cast<const uint4*>(st + boff[ks] + j * 4096);
      } else {
        typedef __attribute__((address_space(3))) sk_s16x4* lds_s4p;
        const char* q = st + boff[j] + ks * (2 * (C::BN / 64) * 1024);
        const sk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(q));
        const sk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(q + 512));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
    }
  };
  // operands swapped: the 32 x 32 result is [n (registers, 4 consecutive per group)][m (lane & 31)]
  auto mma = [&](const uint4 (&fa)[FM], const uint4 (&fb)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[j]), __builtin_bit_cast(bf16x8_t, fa[i]),
                                                            acc[i][j], 0, 0, 0);
  };

  // ---- the end of a piece ----
  float* const my_slab = pl.ws + (int64_t)v * C::SLAB_FLOATS;
  // write-through (sc1) 16-byte stores: the slab is in memory when the vm counter says so, no release fence (a buffer_wbl2
  // here writes back everything the XCD's L2 holds dirty -- the other workgroups' slabs and C tiles: measured 25 us per launch)
  auto slab_store = [&]() {            // accumulator-register order: [(i, j, q)][thread] float4
    const char* base = uniform_ptr(my_slab);
    const uint32_t voff = tid * 16;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t x = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(voff), "v"(x), "s"(base) : "memory");
          base += 8192;
        }
  };
  auto slab_add = [&](int w) {
    const float4* src = reinterpret_cast<const float4*>(pl.ws + (int64_t)w * C::SLAB_FLOATS) + tid;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 x = src[((i * FN + j) * 4 + q) * 512];
          acc[i][j][4 * q] += x.x; acc[i][j][4 * q + 1] += x.y; acc[i][j][4 * q + 2] += x.z; acc[i][j][4 * q + 3] += x.w;
          // (every wait for a batch of loads is a memory round trip: as many in flight as the registers hold)
          if constexpr (FM * FN > 4) { if (q == 3 && (j & 1)) __builtin_amdgcn_sched_barrier(0); }
        }
  };
  // Batched epilogue: the bias of the tile's (j, pp) column granules once, then per 32-row block i the primary matrix input of
  // all its granules (block i + 1's before block i's stores), then arithmetic + stores.
  auto epilogue = [&](int tile) {
    const int m0 = (tile / pl.tiles_n) * C::BM, n0 = (tile % pl.tiles_n) * C::BN;
    const int prim = ep_primary(p);
    const int nbase = n0 + wn * FN * 32 + 8 * half;
    float bias[FN][2][8];
    if (p.ep.bias) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int n = nbase + j * 32 + 16 * pp;
          load8_any(p.ep.bias, PENEO_F32, min(n, p.N - 8), bias[j][pp]);
        }
    }
    EpIn8 in[2][FN][2];
    auto load_block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int m = m0 + (wm * FM + i) * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int n = nbase + j * 32 + 16 * pp;
          if (m < p.M && n + 8 <= p.N) ep_load_primary(p, prim, m, n, in[i & 1][j][pp]);
        }
    };
    auto store_block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int m = m0 + (wm * FM + i) * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          float val[8];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * pp + t]), __float_as_uint(acc[i][j][8 * pp + 4 + t]),
                                                             false, false);
            val[t] = __uint_as_float(sw[0]); val[4 + t] = __uint_as_float(sw[1]);
          }
          const int n = nbase + j * 32 + 16 * pp;
          if (m < p.M && n + 8 <= p.N) epilogue_apply8(p, m, n, val, p.ep.bias ? bias[j][pp] : nullptr, prim, in[i & 1][j][pp]);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    load_block(I0{});
    if constexpr (FM > 1) load_block(I1{});
    store_block(I0{});
    if constexpr (FM > 2) { __builtin_amdgcn_sched_barrier(0); load_block(I2{}); }
    if constexpr (FM > 1) store_block(I1{});
    if constexpr (FM > 3) { __builtin_amdgcn_sched_barrier(0); load_block(I3{}); }
    if constexpr (FM > 2) store_block(I2{});
    if constexpr (FM > 3) store_block(I3{});
    static_assert(FM <= 4, "row blocks");
  };

  // ---- prologue of the stream: units 0 .. NSTAGE - 2 ----
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < n_units) issue(s);

  SK_STAMP(1);
  int tile = tile_begin, kt = kt_begin;
  int kt_first = kt_begin;             // first k-tile of the piece being accumulated
  // Waiting.  A counted s_waitcnt vmcnt(N) says "at most N vector-memory operations of this wave are outstanding"; LDS-DMA loads
  // complete in order among themselves, so N = the pieces of the younger units is a correct wait for unit t whatever stores are
  // mixed in (they only make it wait longer).  At the end of a piece every wave first EMPTIES its counter (the NSTAGE - 1 units
  // in flight have landed: `landed` tops need no wait, and the compiler's own counts for the epilogue's loads are exact), then
  // issues its epilogue / slab traffic; the slab flag is published at the first top that waits again, behind a full drain.
  int landed = 0;
  bool publish = false;
  for (int t = 0; t < n_units; ++t) {
    bool drained = false;
    if (landed > 0) {
      --landed;
    } else {
      // this wave's pieces of unit t have landed (younger units may still be in flight) ...
      if (!publish && NSTAGE > 2 && t + NSTAGE - 2 < n_units) wait_vm<(NSTAGE - 2) * C::PPW>(); else { wait_vm<0>(); drained = true; }
    }
    // ... and everybody's: the barrier also says that every wave is done reading unit t - 1, whose stage is refilled now
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (publish && drained) {
      // every wave has drained its write-through slab stores in front of this barrier
      if (tid == 0) __hip_atomic_store(pl.flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      publish = false;
      SK_STAMP(4);
    }
    if (t == 0) SK_STAMP(2);
    const bool more = t + NSTAGE - 1 < n_units;
    dbase = __builtin_amdgcn_readfirstlane(lds0 + ((t + NSTAGE - 1) % NSTAGE) * C::STAGE + wave * 1024);
    const char* st = smem + (t % NSTAGE) * C::STAGE;
    uint4 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
    load_a(st, 0, fa0); load_b(st, 0, fb0);
    if (more) issue_group(std::integral_constant<int, 0>{});
    load_a(st, 1, fa1); load_b(st, 1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_group(std::integral_constant<int, 1>{});
    load_a(st, 2, fa0); load_b(st, 2, fb0);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_group(std::integral_constant<int, 2>{});
    load_a(st, 3, fa1); load_b(st, 3, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_group(std::integral_constant<int, 3>{});
    mma(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);

    const bool tile_done = kt + 1 == ktiles;
    if (tile_done || t + 1 == n_units) {
      wait_vm<0>();
      landed = min(NSTAGE - 1, n_units - 1 - t);
      if (kt_first != 0) {
        // not the k = 0 piece (only ever the first piece of a range): leave it for the finisher
        SK_STAMP(3);
        slab_store();
        publish = true;
      } else {
        if (!tile_done) {
          // the k = 0 piece of a tile that other workgroups complete: theirs are the ranges that begin inside this tile
          SK_STAMP(5);
          const int64_t tile_end = (int64_t)(tile + 1) * ktiles;
          int w_last = v;
          while (w_last + 1 < G && (int64_t)(w_last + 1) * U / G < tile_end) ++w_last;
          if (tid == 0) {
            for (int w = v + 1; w <= w_last; ++w)
              while (__hip_atomic_load(pl.flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
            if (pl.prof) pl.prof[v * 16 + 6] = wall_clock64();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
          SK_STAMP(7);
          for (int w = v + 1; w <= w_last; ++w) slab_add(w);
          if (tid == 0)
            for (int w = v + 1; w <= w_last; ++w) __hip_atomic_store(pl.flags + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SK_STAMP(8 + (tile_done ? 0 : 2));
        epilogue(tile);
        SK_STAMP(9 + (tile_done ? 0 : 2));
      }
      zero_acc();
      kt_first = 0;
    }
    if (tile_done) { kt = 0; ++tile; } else ++kt;
  }
  if (publish) {                       // the range ended before another top drained the slab stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(pl.flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SK_STAMP(4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SK_STAMP(12);
}

// ---- host side ----
struct SkWorkspace { float* ws = nullptr; uint32_t* flags = nullptr; size_t slab_floats = 0; int G = 0; };
static std::mutex g_sk_mutex;
static std::unordered_map<uint64_t, SkWorkspace> g_sk_ws;     // per (device, stream): launches of one stream are ordered
static int g_sk_cus[64] = {};
static uint64_t* g_sk_prof = nullptr;

static int sk_cu_count(int dev) {
  if (dev < 0 || dev >= 64) return 256;
  if (g_sk_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_sk_cus[dev] = n;
  }
  return g_sk_cus[dev];
}

// 0 = no workspace (allocation failed or a capture is in progress: the caller runs another kernel)
static bool sk_workspace(int dev, hipStream_t st, int G, size_t slab_floats, SkWorkspace& out) {
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  const uint64_t key = (reinterpret_cast<uint64_t>(st) << 6) ^ (uint64_t)dev;
  SkWorkspace& w = g_sk_ws[key];
  if (w.G < G || w.slab_floats < slab_floats) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    // the old slabs may still be read by a launch in flight on this stream
    if (w.ws) { (void)hipStreamSynchronize(st); (void)hipFree(w.ws); (void)hipFree(w.flags); w = SkWorkspace{}; }
    const int g = G > w.G ? G : w.G;
    const size_t sf = slab_floats > w.slab_floats ? slab_floats : w.slab_floats;
    float* ws = nullptr; uint32_t* fl = nullptr;
    if (hipMalloc(&ws, (size_t)g * sf * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMalloc(&fl, 4096) != hipSuccess || hipMemset(fl, 0, 4096) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ws); return false; }
    w.ws = ws; w.flags = fl; w.G = g; w.slab_floats = sf;
  }
  out = w;
  return true;
}

template <typename C>
static int launch_sk(const GemmParams& p, hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  SkPlan pl;
  const int tm = (p.M + C::BM - 1) / C::BM;
  pl.tiles_n = (p.N + C::BN - 1) / C::BN;
  pl.tiles = tm * pl.tiles_n;
  pl.ktiles = p.K / 64;
  const int64_t U = (int64_t)pl.tiles * pl.ktiles;
  int G = sk_cu_count(dev) & ~7;
  if (G > 1024) G = 1024;
  // every range holds at least a third of a tile's k-tiles: a cut tile has at most three foreign pieces
  const int64_t min_units = (pl.ktiles + 2) / 3;
  while (G > 8 && U / G < min_units) G -= 8;
  if (U < G) return 0;
  pl.G = G;
  SkWorkspace w;
  if (!sk_workspace(dev, st, G, C::SLAB_FLOATS, w)) return 0;
  pl.ws = w.ws; pl.flags = w.flags; pl.prof = g_sk_prof;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sk_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
      set_error("peneo_gemm: cannot raise dynamic LDS to %d bytes", C::LDS_BYTES);
      return PENEO_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_sk_kernel<C>, dim3((unsigned)G), dim3(512), C::LDS_BYTES, st, p, pl);
  const int rc = check_launch("peneo_gemm (stream-k)");
  return rc == PENEO_OK ? 1 : rc;
}

// Tile shapes: (workgroup tile, wave grid, wave tile, stages)
//   256 x 128: 4 x 2 waves of  64 x 64, 3 stages of 48 KiB, slab 128 KiB
//   256 x 256: 2 x 4 waves of 128 x 64, 2 stages of 64 KiB, slab 256 KiB
template <bool BK> using Sk128 = SkCfg<BK, 4, 2, 2, 2, 3>;
template <bool BK> using Sk256 = SkCfg<BK, 2, 4, 4, 2, 2>;

static int g_sk_mode = -1;   // PENEO_GEMM_SK: 0 = off, 1 = auto (default), 128 / 256 = force that tile where the kernel applies

int launch_gemm_sk(const GemmParams& p, bool b_kmajor, hipStream_t st) {
  if (g_sk_mode < 0) { const char* e = getenv("PENEO_GEMM_SK"); g_sk_mode = e ? atoi(e) : 1; }
  if (g_sk_mode == 0) return 0;
  if (p.split_k > 1 || p.dz_on || p.K % 64 != 0 || p.K < 128 || p.N % 8 != 0) return 0;
  if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15) return 0;
  if ((p.lda * 2) % 16 != 0 || (p.ldb * 2) % 16 != 0) return 0;
  {
    const peneo_gemm_epilogue& e = p.ep;
    const int csz = p.c_dtype == PENEO_F32 ? 4 : 2;
    auto al = [](const void* ptr, int64_t ld, int esz) {
      return ptr == nullptr || (((reinterpret_cast<uintptr_t>(ptr) & 15) == 0) && ((ld * esz) % 16 == 0));
    };
    if (!(al(p.C, p.ldc, csz) && al(e.preact, e.ld_preact, csz) && al(e.grad_src, e.ld_grad, csz) && al(e.residual, e.ld_res, csz) &&
          al(e.bias, 0, 4)))
      return 0;
  }
  // per-lane 32-bit offsets inside a tile
  if ((int64_t)256 * p.lda * 2 >= ((int64_t)1 << 31) || (int64_t)256 * p.ldb * 2 >= ((int64_t)1 << 31)) return 0;
  int pick = g_sk_mode;
  if (pick == 1) {
    if ((int64_t)p.M * p.N < (int64_t)1 << 21 || p.M < 256 || p.N < 128) return 0;   // small problems: the 128 x 128 kernel
    pick = 128;
  }
  if (pick == 256) return b_kmajor ? launch_sk<Sk256<true>>(p, st) : launch_sk<Sk256<false>>(p, st);
  return b_kmajor ? launch_sk<Sk128<true>>(p, st) : launch_sk<Sk128<false>>(p, st);
}

}  // namespace peneo

/* tools/ and tests only (declared in the header next to peneo_gemm_set_big_mode): 0 = off, 1 = auto, 128 / 256 = force that tile */
extern "C" void peneo_gemm_set_sk_mode(int mode) { peneo::g_sk_mode = mode; }
/* tools only, not in the header: device buffer of [1024][16] uint64 that receives one lane's s_memrealtime stamps (100 MHz) at the
 * stations of every workgroup's range (0 start, 1 stream primed, 2 first unit landed, 3 / 4 slab publish, 5 / 6 / 7 flag wait,
 * acquire, 8 / 9 last whole-tile epilogue, 10 / 11 finisher epilogue, 12 stores drained); null = off */
extern "C" void peneo_gemm_sk_set_prof(void* buf) { peneo::g_sk_prof = reinterpret_cast<uint64_t*>(buf); }
