// bf16 GEMM with ONE 8-wave workgroup per CU and workgroup tiles of 256 x 256 / 384 x 192 / 256 x 128 (gfx950).
//
// Why: the 128 x 128 kernel of gemm.hip moves 32 KiB global -> LDS per 2 MFLOP; a CU turns LDS-DMA requests into LDS lines
// at ~20-26 B/clk, so that kernel is fill-bound at ~40 % of the matrix cores, and at M = 5672 its 270 / 810 /
// 1080 tiles quantise badly on 512 resident workgroups.  A 256 x 256 tile moves 64 KiB per 8.4 MFLOP (half the bytes per
// FLOP), wave tiles of 128 x 64 (4 x 2 MFMA 32 x 32 accumulators) read 0.75 KiB of LDS per MFMA instead of 1, and the tile
// shape is chosen per problem so that the tiles fill ONE round of the 256 CUs (M = 5672: 207 tiles of 256 x 256 for
// N = 2304, 240 tiles of 384 x 192 for N = 3072, 138 tiles of 256 x 128 for N = 768).
//
// Layouts: A k-major [M][K]; B k-major [N][K] (forward, x W^T) or mn-major [K][N] (dgrad, dy W).  K % 64 == 0.
// LDS image of a k-major tile: rows of 128 B (64 k), 16-byte slot s of row r at slot s ^ ((r >> 1) & 7) (swizzle applied
// to the SOURCE address of each LDS-DMA lane, destination lane-linear: 1 KiB piece = 8 rows); fragments by ds_read_b128.
// mn-major tile: 1 KiB pieces of [8 k][64 columns], the two 64-byte halves of a line swapped when (k >> 1) & 1; fragments
// by ds_read_b64_tr_b16.  (Both are the images of gemm.hip's LDS-DMA kernels.)
// Pipeline: ring of NSTAGE k-tiles, LDS-DMA NSTAGE - 1 tiles ahead from inline asm (counted vmcnt, invisible to hipcc so
// that it does not drain the ring in front of every ds_read), one s_barrier per k-tile.
// Epilogue: each wave parks one 32-row block of its accumulators in its own LDS patch, reads it back row-major and runs
// the shared fused epilogue (gemm_common.h) on 8 consecutive columns per lane: 16-byte stores, whole 128-byte lines.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_common.h"

namespace peneo {

typedef short bg_s16x4 __attribute__((ext_vector_type(4)));

template <bool BK_, int WGM_, int WGN_, int FM_, int FN_, int NSTAGE_>
struct BigCfg {
  static constexpr bool BK = BK_;
  static constexpr int WGM = WGM_, WGN = WGN_, FM = FM_, FN = FN_, NSTAGE = NSTAGE_;
  static constexpr int BM = WGM * FM * 32, BN = WGN * FN * 32;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int APW = BM / 64, BPW = BN / 64, PPW = APW + BPW;   // 1 KiB pieces per wave and k-tile
  static constexpr int LDS_BYTES = NSTAGE * STAGE;
  static constexpr int EP_LD = FN * 32 + 4;                               // floats per row of a wave's epilogue patch
  static_assert(WGM * WGN == 8, "eight waves");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "pieces of 8 rows are dealt to 8 waves");
  static_assert(8 * 32 * EP_LD * 4 <= LDS_BYTES, "epilogue patches fit in the dead ring");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <typename C>
__global__ __launch_bounds__(512) void gemm_big_kernel(GemmParams p, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FM = C::FM, FN = C::FN, NSTAGE = C::NSTAGE;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  // XCD-aware tile order: workgroup ids go round-robin to the 8 XCDs; every XCD gets a contiguous band of tiles, n fastest
  const int total = gridDim.x, lin = blockIdx.x;
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, slot = lin >> 3;
  const int tile = xcd * q8 + min(xcd, r8) + slot;
  const int m0 = (tile / tiles_n) * C::BM, n0 = (tile % tiles_n) * C::BN;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const int ktiles = p.K / 64;

  // ---- LDS-DMA sources of this wave's pieces (piece g = wave + 8 u of the A tile / of the B tile): per-lane 32-bit byte
  //      offsets inside the tile (constant over k) + one uniform 64-bit base per operand that walks k with scalar adds ----
  uint32_t offA[C::APW], offB[C::BPW];
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
  auto uniform_ptr = [](const void* q) -> const char* {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  const char* bA = uniform_ptr(A + (int64_t)m0 * p.lda);
  const char* bB = uniform_ptr(C::BK ? B + (int64_t)n0 * p.ldb : B + n0);
  const int64_t stepB = C::BK ? 128 : (int64_t)64 * p.ldb * 2;
  const uint32_t lds0 = lds_addr(smem);
  // The pieces of a k-tile are issued in four groups, one between the MFMA clusters of each k-step: a wave that issues
  // all of its pieces back to back stalls on the memory pipeline (measured: fill and compute then run one after the other,
  // 2.1 us per 256 x 256 k-tile instead of ~1)
  uint32_t dbase = 0;
  auto issue_group = [&](auto gc) {
    constexpr int G = decltype(gc)::value;
#pragma unroll
    for (int u = 0; u < C::PPW; ++u) {
      if (u * 4 / C::PPW != G) continue;
      if (u < C::APW) lds_dma_1k_s<0>(offA[u], bA, dbase + u * 8192);
      else lds_dma_1k_s<0>(offB[u - C::APW], bB, dbase + C::A_BYTES + (u - C::APW) * 8192);
    }
    if constexpr (G == 3) { bA += 128; bB += stepB; }      // the k-tile is complete: the bases move on
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
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
        typedef __attribute__((address_space(3))) bg_s16x4* lds_s4p;
        const char* q = st + boff[j] + ks * (2 * (C::BN / 64) * 1024);
        const bg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(q));
        const bg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(q + 512));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
    }
  };
  auto mma = [&](const uint4 (&fa)[FM], const uint4 (&fb)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]),
                                                            acc[i][j], 0, 0, 0);
  };

  // ---- ring: tiles t .. t + NSTAGE - 2 in flight while tile t is multiplied ----
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < ktiles) issue(s);
  for (int t = 0; t < ktiles; ++t) {
    // this wave's pieces of tile t have landed (younger tiles may still be in flight) ...
    if (NSTAGE > 2 && t + NSTAGE - 2 < ktiles) wait_vm<(NSTAGE - 2) * C::PPW>(); else wait_vm<0>();
    // ... and everybody's: the barrier also says that every wave is done reading tile t - 1, whose stage is refilled now
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const bool more = t + NSTAGE - 1 < ktiles;
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
  }
  __syncthreads();     // the ring is dead: every wave has read its last fragments

  // ---- epilogue: 32-row blocks through a wave-private LDS patch ----
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * C::EP_LD);
  constexpr int CG = FN * 32 / 8;           // 8-column groups per row
  constexpr int RPP = 64 / CG;              // rows per pass (FN = 3: 5 rows, four lanes idle)
  constexpr int NPASS = (32 + RPP - 1) / RPP;
  const int cgi = lane % CG, rli = lane / CG;
  auto block = [&](auto ic) {
    constexpr int i = decltype(ic)::value;     // compile-time row block: a rolled loop would index acc[] dynamically (scratch)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[acc_row(r, lane) * C::EP_LD + j * 32 + acc_col(lane)] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    const int mb = m0 + (wm * FM + i) * 32, nb = n0 + wn * FN * 32 + cgi * 8;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      const int rl = pass * RPP + rli;
      if (rli < RPP && rl < 32 && mb + rl < p.M && nb + 8 <= p.N) {
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(patch + rl * C::EP_LD + cgi * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(patch + rl * C::EP_LD + cgi * 8 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        epilogue_store8(p, mb + rl, nb, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  block(std::integral_constant<int, 0>{});
  if constexpr (FM > 1) block(std::integral_constant<int, 1>{});
  if constexpr (FM > 2) block(std::integral_constant<int, 2>{});
  if constexpr (FM > 3) block(std::integral_constant<int, 3>{});
  static_assert(FM <= 4, "row blocks");
}

template <typename C>
static int launch_big(const GemmParams& p, hipStream_t st) {
  const int tm = (p.M + C::BM - 1) / C::BM, tn = (p.N + C::BN - 1) / C::BN;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
      set_error("peneo_gemm: cannot raise dynamic LDS to %d bytes", C::LDS_BYTES);
      return PENEO_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_big_kernel<C>, dim3((unsigned)(tm * tn)), dim3(512), C::LDS_BYTES, st, p, tn);
  const int rc = check_launch("peneo_gemm (big tiles)");
  return rc == PENEO_OK ? 1 : rc;
}

// Tile shapes: (workgroup tile, wave grid, wave tile, stages)
//   256 x 256: 2 x 4 waves of 128 x 64, 2 stages of 64 KiB
//   384 x 192: 4 x 2 waves of  96 x 96, 2 stages of 72 KiB
//   256 x 128: 4 x 2 waves of  64 x 64, 3 stages of 48 KiB
template <bool BK> using Big256 = BigCfg<BK, 2, 4, 4, 2, 2>;
template <bool BK> using Big384 = BigCfg<BK, 4, 2, 3, 3, 2>;
template <bool BK> using Big128 = BigCfg<BK, 4, 2, 2, 2, 3>;

// Cost model in cycles per k-tile of 64, calibrated on 4096^3 and the encoder shapes (tools/run_gemm_big.py): every kernel
// is bound by the CU's global -> LDS fill rate, so a round of tiles costs (bytes per k-tile) / (fill rate of that ring):
//   128 x 128, 2 workgroups per CU (gemm.hip)   3300 per round of 512 tiles (fractional rounds: the tails overlap)
//   256 x 256, 2 stages                         5080 per round of 256
//   384 x 192, 2 stages                         5550
//   256 x 128, 3 stages                         2760
static double big_cost(int M, int N, int bm, int bn, double per_round) {
  const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  return ((tiles + 255) / 256) * per_round;
}
static double small_cost(int M, int N) {
  const double r = ((M + 127) / 128) * (double)((N + 127) / 128) / 512.0;
  return (r > 1.0 ? r : 1.0) * 3300.0;
}

static int g_big_mode = -1;   // PENEO_GEMM_BIG: 0 = off, 1 = auto (default), 256 / 384 / 128 = force that tile where it applies

int launch_gemm_big(const GemmParams& p, bool b_kmajor, hipStream_t st) {
  if (g_big_mode < 0) { const char* e = getenv("PENEO_GEMM_BIG"); g_big_mode = e ? atoi(e) : 1; }
  if (g_big_mode == 0) return 0;
  // mn-major B = the dgrad GEMMs of the backward: in the step they run beside the weight-gradient stream, and a workgroup that
  // needs a CU's whole LDS cannot share it (measured: d_zi on 384 x 192 tiles 50 us alone, 166 us in the step; the step is
  // 0.2 ms FASTER with the 128 x 128 kernel there).  A forced tile (tools) still runs them.
  if (!b_kmajor && g_big_mode == 1) return 0;
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
  if ((int64_t)p.M * p.N < (int64_t)1 << 21 || p.M < 256 || p.N < 128) return 0;   // small problems: the 128 x 128 kernel fills the chip better
  int pick = g_big_mode;
  if (pick == 1) {
    const double c256 = big_cost(p.M, p.N, 256, 256, 5080.0), c384 = big_cost(p.M, p.N, 384, 192, 5550.0);
    const double c128 = b_kmajor ? big_cost(p.M, p.N, 256, 128, 2760.0) : 1e30;   // mn-major B: measured behind the 128 x 128 kernel
    pick = 256;
    double best = c256;
    if (c384 < best) { best = c384; pick = 384; }
    if (c128 < best) { best = c128; pick = 128; }
    if (best > 0.95 * small_cost(p.M, p.N)) return 0;      // how much better than the 128 x 128 kernel the model must predict
  }
  if (b_kmajor) {
    if (pick == 384) return launch_big<Big384<true>>(p, st);
    if (pick == 128) return launch_big<Big128<true>>(p, st);
    return launch_big<Big256<true>>(p, st);
  }
  if (pick == 384) return launch_big<Big384<false>>(p, st);
  if (pick == 128) return launch_big<Big128<false>>(p, st);
  return launch_big<Big256<false>>(p, st);
}

}  // namespace peneo

/* tools/ only (not in the header): 0 = off, 1 = choose by the cost model, 256 / 384 / 128 = force that tile shape */
extern "C" void peneo_gemm_set_big_mode(int mode) { peneo::g_big_mode = mode; }
