// bf16 GEMM as ONE persistent launch: at most one 8-wave workgroup per CU, each walking a contiguous range of
// (tile, k-stage) units -- whole tiles ("data-parallel") or ranges that cut tiles ("stream-k") (gfx950).
//
// What bounds a GEMM on this chip (profiles/r06_gemm_persistent.txt, (b)): a CU turns global -> LDS requests into LDS lines at ~22
// B/clk whatever the kernel (ours, the vendor library's, the guide's template), so a launch costs (bytes staged through LDS) /
// (22 B/clk x CUs) when everything else hides under that stream, and the tiled kernels of gemm.hip / gemm_big.hip are far from
// it on the model's M = 5672 problems: 15 us of a QKV-shaped launch sit before the first and after the last k-tile, the tile
// counts (270 / 414 / 552 / 810 on 256 CUs) quantise badly, and a two-stage ring lets the request queue run dry in every k-tile.
// Here
//   * the tile comes from a FAMILY -- M extent 32 F (F = 4..8: 128 .. 256 rows) x N extent 128, F = 4..6 x N extent 256 -- which
//     the 16 x 16 x 32 MFMA allows: 2 x 4 waves, wave tile 16 F x (N extent / 4);
//   * the ring holds 3 .. 6 stages of 64 k (every LDS-DMA instruction fetches 8 whole 128-byte lines: the fill rate is a REQUEST
//     rate -- the same kernel staging 16 rows x 64 bytes per instruction fills at 14.6 instead of 22 B/clk), NSTAGE - 1 stages always
//     in flight: the request queue never drains inside a tile, nor between tiles -- the ring is ONE stream over the workgroup's
//     units, the next tile's first stages land while this tile's last stages are multiplied and while its epilogue runs;
//   * the epilogue never touches LDS (so nothing fences it against the ring): the MFMA runs with the operands SWAPPED
//     (D = B_tile x A_tile^T), a lane holds one output row and 4 consecutive columns per accumulator; the B rows are dealt to the
//     MFMA in an order that puts columns c .. c + 3 and c + 4 .. c + 7 on lanes 32 apart, so one v_permlane32_swap per register
//     pair makes 8 consecutive columns = the 16-byte granule of gemm_common.h's fused epilogue (four lanes = 64 contiguous
//     bytes of a row); the epilogue's loads are issued in batches ahead of its stores (epilogue_apply8);
//   * stream-k, where no tile fits (few tiles x deep K): workgroup v owns units [v U / G, (v + 1) U / G).  A tile cut by a range
//     boundary is finished by the workgroup that holds its k = 0 piece; every other piece is the FIRST thing its workgroup
//     computes, leaves as an fp32 slab in accumulator order (16-byte write-through stores; every wave drains its vm counter,
//     barrier, one lane stores the flag) and is added by the finisher at the END of its range behind one relaxed poll + one
//     agent-scope acquire (cdna guide, Guideline 16 recipe R1).  Flags are reset by their single consumer (a captured launch
//     replays).  v is XCD-major, so the pieces of a tile sit on one XCD except at the seven seams.  Progress: a finisher waits
//     only for workgroups v + 1 .. v + 3, which publish before anything else, and an XCD dispatches its workgroups in id order.
// Layouts: A k-major [M][K]; B k-major [N][K] (forward, x W^T); K % 64 == 0, N % 8 == 0, 16-byte aligned rows.
// LDS image of a stage (gemm_big.hip's): rows of 128 B (64 k), A rows then B rows, pieces of 8 rows = one LDS-DMA instruction;
// the 16-byte slot s of row r sits at s ^ ((r >> 1) & 7) (applied to the DMA source address, the destination is lane-linear):
// every 16-lane group of a ds_read_b128 fragment read (16 rows x 32 k: lane = row + 16 x slot) covers all 16 bank slots.
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include "common.h"
#include "gemm_common.h"

namespace peneo {

struct SkPlan {
  int tiles_n, tiles, ktiles, G;
  int cut;              // 1 = ranges of units (stream-k), 0 = ranges of whole tiles
  float* ws;            // [G] slabs of BM x BN floats (a workgroup's first piece when it is not the tile's k = 0 piece)
  uint32_t* flags;      // [G] 0 = empty, 1 = slab published; reset by the finisher
  uint64_t* prof;       // tools only (peneo_gemm_sk_set_prof): [G][16] s_memrealtime stamps of one lane, or null
};
#define SK_STAMP(k) do { if (pl.prof && tid == 0) pl.prof[v * 16 + (k)] = wall_clock64(); } while (0)

template <int F_, int BN_, int NSTAGE_>
struct SkCfg {
  static constexpr int F = F_, BN = BN_, NSTAGE = NSTAGE_;
  static constexpr int NB = BN / 64;                   // 16-column fragments of a wave
  static constexpr int BM = 32 * F;
  static constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8, PIECES = A_PIECES + B_PIECES;   // 8 rows x 128 B each
  static constexpr int PPW = (PIECES + 7) / 8;         // LDS-DMA instructions per wave and stage (odd F: the last four waves have one fewer)
  static constexpr int STAGE = PIECES * 1024;
  static constexpr int LDS_BYTES = NSTAGE * STAGE;
  static constexpr int SLAB_FLOATS = BM * BN;
  static_assert(NB == 2 || NB == 4, "column fragments pair up in the epilogue");
  static_assert(NSTAGE >= 2, "ring");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

typedef __attribute__((ext_vector_type(4))) float sk_f32x4;

// EP: what the epilogue of a tile is compiled for.  The fused epilogue of gemm_common.h inlined once per granule is 56 - 142 KB of code
// per kernel, run once per tile straight from a cold instruction cache (a tile's epilogue measured 5 - 8 us that way, 1.4 - 1.6 us
// with 12 - 15 KB of code): the three epilogues the encoder's forward uses get their own small instantiations (bf16 C, alpha = 1, no
// gradient source, no accumulation); everything else runs EP_GENERIC.
enum { SK_EP_GENERIC = 0, SK_EP_BIAS = 1, SK_EP_GELU = 2, SK_EP_RES = 3 };

template <typename C, int EP>
__global__ __launch_bounds__(512) void gemm_sk_kernel(GemmParams p, SkPlan pl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int F = C::F, NB = C::NB, NSTAGE = C::NSTAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r16 = lane & 15, c4 = lane >> 4;

  // ---- this workgroup's range of units.  v is XCD-major: workgroup ids go round-robin to the 8 XCDs, so the workgroups of an
  //      XCD hold consecutive ranges (neighbouring tiles share their A panel in that XCD's L2; a cut tile's pieces meet there) ----
  const int G = pl.G, lin = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = lin & 7, slot = lin >> 3;
  const int v = xcd * q8 + min(xcd, r8) + slot;
  const int ktiles = pl.ktiles;
  const int64_t U = (int64_t)pl.tiles * ktiles;
  const int64_t u_begin = pl.cut ? v * U / G : ((int64_t)v * pl.tiles / G) * ktiles;
  const int64_t u_end = pl.cut ? (v + 1) * U / G : ((int64_t)(v + 1) * pl.tiles / G) * ktiles;
  const int n_units = (int)(u_end - u_begin);
  if (n_units <= 0) return;
  SK_STAMP(0);
  const int tile_begin = (int)(u_begin / ktiles), kt_begin = (int)(u_begin % ktiles);

  // ---- LDS-DMA side: runs NSTAGE - 1 units ahead of the multiply.  Piece g = wave + 8 u of the stage (A pieces, then B pieces):
  //      per-lane 32-bit byte offset inside the tile + one uniform 64-bit base per operand that walks k with scalar adds ----
  uint32_t off[C::PPW];
  const char* bA = nullptr;
  const char* bB = nullptr;
  auto uniform_ptr = [](const void* q) -> const char* {
    const uint64_t w = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)w), hi = __builtin_amdgcn_readfirstlane((uint32_t)(w >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto dma_set_tile = [&](int tile, int kt) {
    const int m0 = (tile / pl.tiles_n) * C::BM, n0 = (tile % pl.tiles_n) * C::BN;
#pragma unroll
    for (int u = 0; u < C::PPW; ++u) {
      const int g = min(wave + 8 * u, C::PIECES - 1);
      if (g < C::A_PIECES) {
        const int row = g * 8 + (lane >> 3), sg = (lane & 7) ^ ((row >> 1) & 7);
        off[u] = (uint32_t)(((int64_t)(min(m0 + row, p.M - 1) - m0) * p.lda + sg * 8) * 2);
      } else {
        const int row = (g - C::A_PIECES) * 8 + (lane >> 3), sg = (lane & 7) ^ ((row >> 1) & 7);
        off[u] = (uint32_t)(((int64_t)(min(n0 + row, p.N - 1) - n0) * p.ldb + sg * 8) * 2);
      }
    }
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    bA = uniform_ptr(reinterpret_cast<const char*>(A + (int64_t)m0 * p.lda) + (int64_t)kt * 128);
    bB = uniform_ptr(reinterpret_cast<const char*>(B + (int64_t)n0 * p.ldb) + (int64_t)kt * 128);
  };
  int d_tile = tile_begin, d_kt = kt_begin;      // the next unit to issue
  dma_set_tile(d_tile, d_kt);
  const uint32_t lds0 = lds_addr(smem);
  const bool full = wave + 8 * (C::PPW - 1) < C::PIECES;      // this wave has a piece in the last round (selects its wait count)
  auto issue = [&](int stage) {
    const uint32_t dbase = __builtin_amdgcn_readfirstlane(lds0 + stage * C::STAGE + wave * 1024);
#pragma unroll
    for (int u = 0; u < C::PPW; ++u) {
      if (u == C::PPW - 1 && !full) continue;
      lds_dma_1k_s<0>(off[u], (wave + 8 * u < C::A_PIECES) ? bA : bB, dbase + u * 8192);
    }
    // the unit is complete: the bases move on, into the next tile when this one is through (the tile after the last one of the
    // problem is never issued: its units are beyond u_end)
    if (++d_kt == ktiles) {
      d_kt = 0;
      if (++d_tile < pl.tiles) dma_set_tile(d_tile, 0);
    } else {
      bA += 128; bB += 128;
    }
  };

  // ---- fragment offsets inside a stage.  A fragment i: rows wm 16 F + 16 i + r16.  B fragment j: rows wn BN / 4 + 16 j + pi(r16),
  //      pi = swap bits 2 and 3: the MFMA's result rows 4 g .. 4 g + 3 of lane group g = lane >> 4 are then columns 0-3, 8-11, 4-7,
  //      12-15 for g = 0 .. 3 -- lanes 32 apart hold the two halves of 8 consecutive columns ----
  const int pr = (r16 & 3) | ((r16 & 4) << 1) | ((r16 & 8) >> 1);
  int aoff[2], boff[2];                // k-half 0 / 1 of the stage (the 16-byte slot of row r sits at slot ^ ((r >> 1) & 7))
  {
    const int ra = wm * 16 * F + r16, rb = wn * (C::BN / 4) + pr;      // (fragment i / j: + 16 rows = + 2048 bytes, same swizzle)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      aoff[kh] = ra * 128 + (((4 * kh + c4) ^ ((ra >> 1) & 7)) << 4);
      boff[kh] = C::A_PIECES * 1024 + rb * 128 + (((4 * kh + c4) ^ ((rb >> 1) & 7)) << 4);
    }
  }

  sk_f32x4 acc[F][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // ---- the end of a piece ----
  // write-through (sc1) 16-byte stores: the slab is in memory when the vm counter says so, no release fence (a buffer_wbl2
  // here writes back everything the XCD's L2 holds dirty -- the other workgroups' slabs and C tiles: measured 25 us per launch)
  auto slab_store = [&]() {            // accumulator order: [(i, j)][thread] float4
    const char* base = uniform_ptr(pl.ws + (int64_t)v * C::SLAB_FLOATS);
    const uint32_t voff = tid * 16;
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(voff), "v"(acc[i][j]), "s"(base) : "memory");
        base += 8192;
      }
  };
  auto slab_add = [&](int w) {
    const float4* src = reinterpret_cast<const float4*>(pl.ws + (int64_t)w * C::SLAB_FLOATS) + tid;
#pragma unroll
    for (int i = 0; i < F; ++i) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float4 x = src[(i * NB + j) * 512];
        acc[i][j][0] += x.x; acc[i][j][1] += x.y; acc[i][j][2] += x.z; acc[i][j][3] += x.w;
      }
      // (every wait for a batch of loads is a memory round trip: as many in flight as the registers hold)
      if constexpr (F * NB > 16) { if (i & 1) __builtin_amdgcn_sched_barrier(0); }
    }
  };
  // Batched epilogue: the bias of the tile's column granules once, then per pair of 16-row blocks the primary matrix input of all
  // their granules, then arithmetic + stores.
  auto epilogue = [&](int tile) {
    const int m0 = (tile / pl.tiles_n) * C::BM, n0 = (tile % pl.tiles_n) * C::BN;
    const int prim = ep_primary(p);
    const int nbase = n0 + wn * (C::BN / 4) + 16 * (c4 >> 1) + 8 * (c4 & 1);     // + 32 jj
    const int mbase = m0 + wm * (16 * F) + r16;                                   // + 16 i
    float bias[NB / 2][8];
    if (p.ep.bias) {
#pragma unroll
      for (int jj = 0; jj < NB / 2; ++jj) load8_any(p.ep.bias, PENEO_F32, min(nbase + 32 * jj, p.N - 8), bias[jj]);
    }
    auto load_block = [&](auto ic, EpIn8 (&in)[NB / 2]) {
      constexpr int i = decltype(ic)::value;
      if constexpr (EP == SK_EP_BIAS || EP == SK_EP_GELU) return;        // (no matrix-shaped input)
#pragma unroll
      for (int jj = 0; jj < NB / 2; ++jj) {
        const int m = mbase + 16 * i, n = nbase + 32 * jj;
        if (m < p.M && n + 8 <= p.N) ep_load_primary(p, prim, m, n, in[jj]);
      }
    };
    auto store_block = [&](auto ic, const EpIn8 (&in)[NB / 2]) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int jj = 0; jj < NB / 2; ++jj) {
        float val[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][2 * jj][t]), __float_as_uint(acc[i][2 * jj + 1][t]), false, false);
          val[t] = __uint_as_float(sw[0]); val[4 + t] = __uint_as_float(sw[1]);
        }
        const int m = mbase + 16 * i, n = nbase + 32 * jj;
        if (m < p.M && n + 8 <= p.N) {
          if constexpr (EP == SK_EP_GENERIC) {
            epilogue_apply8(p, m, n, val, p.ep.bias ? bias[jj] : nullptr, prim, in[jj]);
          } else {
            // the same arithmetic in the same order as epilogue_apply8 for the options this instantiation is launched with
            bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n;
            if (p.ep.bias) {
#pragma unroll
              for (int e = 0; e < 8; ++e) val[e] += bias[jj][e];
            }
            if constexpr (EP == SK_EP_GELU) {
              if (p.ep.preact) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.ep.preact) + (int64_t)m * p.ep.ld_preact + n) = pack16<bf16_t>(val);
#pragma unroll
              for (int e = 0; e < 8; ++e) val[e] = gelu_fast_f(val[e]);
            }
            if constexpr (EP == SK_EP_RES) {
              if (p.ep.drop_p > 0.f) {
                const uint32_t thresh = (uint32_t)fminf(p.ep.drop_p * 4294967296.0f, 4294967040.0f);
                const float ks = 1.0f / (1.0f - p.ep.drop_p);
                const uint32_t keep = dropout_keep8(p.ep.drop_seed, (uint64_t)m * (uint64_t)p.N + n, thresh);
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = ((keep >> e) & 1u) ? val[e] * ks : 0.f;
              }
              float r[8];
              unpack16<bf16_t>(in[jj].x0, r);
#pragma unroll
              for (int e = 0; e < 8; ++e) val[e] += r[e];
            }
            *reinterpret_cast<uint4*>(crow) = pack16<bf16_t>(val);
          }
        }
      }
    };
    auto batch = [&](auto i0c) {          // two 16-row blocks per batch
      constexpr int i0 = decltype(i0c)::value;
      EpIn8 in0[NB / 2], in1[NB / 2];
      load_block(std::integral_constant<int, i0>{}, in0);
      if constexpr (i0 + 1 < F) load_block(std::integral_constant<int, i0 + 1>{}, in1);
      store_block(std::integral_constant<int, i0>{}, in0);
      if constexpr (i0 + 1 < F) store_block(std::integral_constant<int, i0 + 1>{}, in1);
      __builtin_amdgcn_sched_barrier(0);
    };
    batch(std::integral_constant<int, 0>{});
    if constexpr (F > 2) batch(std::integral_constant<int, 2>{});
    if constexpr (F > 4) batch(std::integral_constant<int, 4>{});
    if constexpr (F > 6) batch(std::integral_constant<int, 6>{});
    static_assert(F <= 8, "row blocks");
  };

  // ---- prologue of the stream: units 0 .. NSTAGE - 2 ----
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < n_units) issue(s);
  SK_STAMP(1);

  int tile = tile_begin, kt = kt_begin;
  int kt_first = kt_begin;             // first k-stage of the piece being accumulated
  // Waiting.  A counted s_waitcnt vmcnt(N) says "at most N vector-memory operations of this wave are outstanding"; LDS-DMA loads
  // complete in order among themselves, so N = the pieces of the younger units is a correct wait for unit t whatever stores are
  // mixed in (they only make it wait longer).  At the end of a piece every wave first EMPTIES its counter (the NSTAGE - 1 units
  // in flight have landed: `landed` tops need no wait, and the compiler's own counts for the epilogue's loads are exact), then
  // issues its epilogue / slab traffic; the slab flag is published at the first top that drains again.
  int landed = 0;
  bool publish = false;
  int st_cur = 0, st_fill = NSTAGE - 1;             // stage of unit t / stage refilled while unit t is multiplied
  for (int t = 0; t < n_units; ++t) {
    bool drained = false;
    if (landed > 0) {
      --landed;
    } else {
      // this wave's pieces of unit t have landed (younger units may still be in flight) ...
      if (!publish && NSTAGE > 2 && t + NSTAGE - 2 < n_units) {
        if (full) wait_vm<(NSTAGE - 2) * C::PPW>(); else wait_vm<(NSTAGE - 2) * (C::PPW - 1)>();
      } else { wait_vm<0>(); drained = true; }
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
    const char* st = smem + st_cur * C::STAGE;
    uint4 fb0[NB], fa0[F], fb1[NB], fa1[F];
#pragma unroll
    for (int j = 0; j < NB; ++j) fb0[j] = *reinterpret_cast<const uint4*>(st + boff[0] + j * 2048);
#pragma unroll
    for (int i = 0; i < F; ++i) fa0[i] = *reinterpret_cast<const uint4*>(st + aoff[0] + i * 2048);
    __builtin_amdgcn_sched_barrier(0);
    if (t + NSTAGE - 1 < n_units) issue(st_fill);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NB; ++j) fb1[j] = *reinterpret_cast<const uint4*>(st + boff[1] + j * 2048);
#pragma unroll
    for (int i = 0; i < F; ++i) fa1[i] = *reinterpret_cast<const uint4*>(st + aoff[1] + i * 2048);
    __builtin_amdgcn_sched_barrier(0);
    // operands swapped: the 16 x 16 result is [column (registers, 4 consecutive)][row (lane & 15)]
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb0[j]), __builtin_bit_cast(bf16x8_t, fa0[i]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb1[j]), __builtin_bit_cast(bf16x8_t, fa1[i]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    st_cur = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
    st_fill = st_fill + 1 == NSTAGE ? 0 : st_fill + 1;

    const bool tile_done = kt + 1 == ktiles;
    if (tile_done || t + 1 == n_units) {
      wait_vm<0>();
      landed = min(NSTAGE - 1, n_units - 1 - t);
      if (publish) {
        // (a second piece end before any top drained: the flag must be out before this workgroup waits for anybody)
        __builtin_amdgcn_s_barrier();
        if (tid == 0) __hip_atomic_store(pl.flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish = false;
        SK_STAMP(4);
      }
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
static int g_sk_max_g = 0;        // tools: cap on the workgroups of a launch (0 = one per CU)

static int sk_cu_count(int dev) {
  if (dev < 0 || dev >= 64) return 256;
  if (g_sk_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_sk_cus[dev] = n;
  }
  return g_sk_cus[dev];
}

// false = no workspace (allocation failed or a capture is in progress: the caller runs another kernel)
static bool sk_workspace(int dev, hipStream_t st, int G, size_t slab_floats, SkWorkspace& out) {
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  const uint64_t key = (reinterpret_cast<uint64_t>(st) << 6) ^ (uint64_t)dev;
  SkWorkspace& w = g_sk_ws[key];
  if (w.G < G || w.slab_floats < slab_floats) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    // the old slabs may still be read by a launch in flight on this stream
    if (w.ws) { (void)hipStreamSynchronize(st); (void)hipFree(w.ws); (void)hipFree(w.flags); }
    const int g = G > w.G ? G : w.G;
    const size_t sf = slab_floats > w.slab_floats ? slab_floats : w.slab_floats;
    w = SkWorkspace{};
    float* ws = nullptr; uint32_t* fl = nullptr;
    if (hipMalloc(&ws, (size_t)g * sf * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMalloc(&fl, 4096) != hipSuccess || hipMemset(fl, 0, 4096) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ws); return false; }
    w.ws = ws; w.flags = fl; w.G = g; w.slab_floats = sf;
  }
  out = w;
  return true;
}

template <typename C, int EP>
static int launch_sk_ep(const GemmParams& p, bool cut, hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  SkPlan pl;
  const int tm = (p.M + C::BM - 1) / C::BM;
  pl.tiles_n = (p.N + C::BN - 1) / C::BN;
  pl.tiles = tm * pl.tiles_n;
  pl.ktiles = p.K / 64;
  pl.cut = cut ? 1 : 0;
  const int64_t U = (int64_t)pl.tiles * pl.ktiles;
  int G = sk_cu_count(dev) & ~7;
  if (G > 1024) G = 1024;
  if (g_sk_max_g > 0 && G > g_sk_max_g) G = g_sk_max_g & ~7;
  pl.ws = nullptr; pl.flags = nullptr;
  if (cut) {
    // every range holds at least a third of a tile's k-stages: a cut tile has at most three foreign pieces
    const int64_t min_units = (pl.ktiles + 2) / 3;
    while (G > 8 && U / G < min_units) G -= 8;
    if (U < G) return 0;
    SkWorkspace w;
    if (!sk_workspace(dev, st, G, C::SLAB_FLOATS, w)) return 0;
    pl.ws = w.ws; pl.flags = w.flags;
  } else if (pl.tiles < G) {
    G = pl.tiles;
  }
  pl.G = G;
  pl.prof = g_sk_prof;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sk_kernel<C, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
      set_error("peneo_gemm: cannot raise dynamic LDS to %d bytes", C::LDS_BYTES);
      return PENEO_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_sk_kernel<C, EP>), dim3((unsigned)G), dim3(512), C::LDS_BYTES, st, p, pl);
  const int rc = check_launch("peneo_gemm (persistent)");
  return rc == PENEO_OK ? 1 : rc;
}

// which instantiation an epilogue runs: the small ones cover bf16 C with alpha = 1 and (bias) | (bias, GELU, pre-activation store) |
// (bias, dropout, bf16 residual); FAST = false keeps the tile shapes the rules never pick on the generic epilogue only (build time)
static int sk_ep_kind(const GemmParams& p) {
  const peneo_gemm_epilogue& e = p.ep;
  if (p.c_dtype != PENEO_BF16 || e.alpha != 1.f || e.grad_src || e.accumulate) return SK_EP_GENERIC;
  if (e.act == PENEO_ACT_GELU && !e.residual && e.drop_p == 0.f) return SK_EP_GELU;
  if (e.act != PENEO_ACT_NONE || e.preact) return SK_EP_GENERIC;
  if (e.residual) return SK_EP_RES;
  return e.drop_p == 0.f ? SK_EP_BIAS : SK_EP_GENERIC;
}
template <typename C, bool FAST>
static int launch_sk(const GemmParams& p, bool cut, hipStream_t st) {
  if constexpr (FAST) {
    switch (sk_ep_kind(p)) {
      case SK_EP_BIAS: return launch_sk_ep<C, SK_EP_BIAS>(p, cut, st);
      case SK_EP_GELU: return launch_sk_ep<C, SK_EP_GELU>(p, cut, st);
      case SK_EP_RES: return launch_sk_ep<C, SK_EP_RES>(p, cut, st);
      default: break;
    }
  }
  return launch_sk_ep<C, SK_EP_GENERIC>(p, cut, st);
}

// ring depth: as many stages as fit, at most 6
template <int F, int BN> struct SkPick {
  static constexpr int STAGE = (32 * F + BN) * 128;
  static constexpr int NS = (160 * 1024) / STAGE > 6 ? 6 : (160 * 1024) / STAGE;
  using type = SkCfg<F, BN, NS>;
};

static int g_sk_mode = -1;   // PENEO_GEMM_SK: 0 = off, 1 = auto (default), else F * 1000 + BN (+ 100000: stream-k ranges)

// F = 7, 8 at N extent 256 are not instantiated: 112 / 128 accumulators + two k-halves of fragments + the batched epilogue do not
// fit 256 registers (the compiler spills accumulators INSIDE the k loop: 230 us for a 20 GFLOP problem) and their 61 / 64 KiB
// stages leave a two-stage ring (the fill queue drains at every stage: DESIGN 8)
template <int BN>
static int launch_sk_f(const GemmParams& p, int F, bool cut, hipStream_t st) {
  switch (F) {
    case 4: return launch_sk<typename SkPick<4, BN>::type, BN == 256>(p, cut, st);
    case 5: return launch_sk<typename SkPick<5, BN>::type, true>(p, cut, st);
    case 6: return launch_sk<typename SkPick<6, BN>::type, false>(p, cut, st);
    case 7: if constexpr (BN == 128) return launch_sk<typename SkPick<7, BN>::type, true>(p, cut, st); else return 0;
    case 8: if constexpr (BN == 128) return launch_sk<typename SkPick<8, BN>::type, false>(p, cut, st); else return 0;
    default: return 0;
  }
}

int launch_gemm_sk(const GemmParams& p, bool b_kmajor, hipStream_t st) {
  if (g_sk_mode < 0) { const char* e = getenv("PENEO_GEMM_SK"); g_sk_mode = e ? atoi(e) : 1; }
  if (g_sk_mode == 0 || !b_kmajor) return 0;
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
  int mode = g_sk_mode;
  if (mode == 1) {
    // The choice per problem (profiles/r06_gemm_persistent.txt).  One workgroup per CU has nothing to overlap a tile's epilogue
    // with, so at M = 5672 (no tile of the family fills a whole number of rounds there) the launch only ties the tiled kernels
    // and they keep those shapes.  It is picked for
    //   (1) many tiles per workgroup (>= 3 rounds: the stream hides every prologue, epilogues drift apart);
    //   (2) a tile of the family that fills ONE or TWO rounds to >= 84 % (the large backbone: QKV, FFN1, FFN2, out-proj; the base
    //       backbone: QKV on 224 x 128 - 468 tiles, 33.5 against 36.5 us - and out-proj on 160 x 128 - 216 tiles, 15.5 against 17.6);
    //   (3) few tiles x deep K: stream-k ranges;
    // (2) and (3) only with one of the compact epilogues - with the generic one (56 - 142 KB of code) they lost inside the
    // large backbone's forward what they won alone.
    if ((int64_t)p.M * p.N < (int64_t)1 << 21 || p.M < 256 || p.N < 128) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int G = sk_cu_count(dev) & ~7;
    auto tiles = [&](int F, int BN) { return (int64_t)((p.M + 32 * F - 1) / (32 * F)) * ((p.N + BN - 1) / BN); };
    mode = 0;
    if (p.N >= 256 && tiles(4, 256) >= 3 * (int64_t)G) mode = 4256;
    else if (p.N < 256 && tiles(5, 128) >= 3 * (int64_t)G) mode = 5128;
    else if (sk_ep_kind(p) != SK_EP_GENERIC) {
      // (where one of gemm_big.hip's more intense tiles - 256 x 256, 384 x 192 - fills its rounds, that kernel stays: 4096^3 runs at
      //  1146 TFLOP/s on 256 tiles of 256 x 256 against 1000 here)
      auto fits = [&](int64_t t, int pct) { return (t <= G && t * 100 >= (int64_t)G * pct) || (t <= 2 * (int64_t)G && t * 100 >= 2 * (int64_t)G * pct); };
      const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256), t384 = (int64_t)((p.M + 383) / 384) * ((p.N + 191) / 192);
      const bool big_fits = fits(t256, 85) || fits(t384, 85);
      const int cand[4][2] = {{5, 256}, {4, 256}, {7, 128}, {5, 128}};            // most intense first
      for (int c = 0; c < 4 && !mode && !big_fits; ++c)
        for (int k = 1; k <= 2 && !mode; ++k) {
          const int64_t t = tiles(cand[c][0], cand[c][1]);
          // (a deep reduction wants the rounds fuller: FFN2 forward at 216 tiles = 0.84 ran 47 us here against 38 on the tiled kernel inside the layer)
          const int pct = p.K > 1024 ? 90 : 84;
          if (p.N >= cand[c][1] && t <= (int64_t)k * G && t * 100 >= (int64_t)k * G * pct) mode = cand[c][0] * 1000 + cand[c][1];
        }
      if (!mode && p.K >= 2048 && tiles(5, 128) * 10 <= (int64_t)G * 6) mode = 105128;
    }
    if (!mode) return 0;
  }
  const bool cut = mode >= 100000;
  mode %= 100000;
  const int F = mode / 1000, BN = mode % 1000;
  if (BN == 256) return launch_sk_f<256>(p, F, cut, st);
  if (BN == 128) return launch_sk_f<128>(p, F, cut, st);
  return 0;
}

}  // namespace peneo

/* tools/ and tests only (declared in the header next to peneo_gemm_set_big_mode) */
extern "C" void peneo_gemm_set_sk_mode(int mode) { peneo::g_sk_mode = mode; }
/* tools only (diagnostics block of the header): device buffer of [1024][16] uint64 that receives one lane's s_memrealtime stamps (100 MHz) at the
 * stations of every workgroup's range (0 start, 1 stream primed, 2 first unit landed, 3 / 4 slab publish, 5 / 6 / 7 flag wait,
 * acquire, 8 / 9 last whole-tile epilogue, 10 / 11 finisher epilogue, 12 stores drained); null = off */
extern "C" void peneo_gemm_sk_set_prof(void* buf) { peneo::g_sk_prof = reinterpret_cast<uint64_t*>(buf); }
/* tools only (diagnostics block of the header): at most this many workgroups per launch (a multiple of 8; 0 = one per CU) -- how a persistent launch on a
 * side stream shares the chip with the kernels of another stream (tools/run_sk_interference.py) */
extern "C" void peneo_gemm_sk_set_max_groups(int n) { peneo::g_sk_max_g = n; }
