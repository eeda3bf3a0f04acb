// NOT BUILT INTO libpeneo_hip.so.  Round-4 experiment kept for reference (profiles/r04_gemm_p8_*.txt): a 256 x 256
// GEMM whose two wave groups run half a phase apart.  Correct on every layout / split at the first run and +6 .. +16 % over the
// lock-step 256 x 256 kernel at 4096^3, but behind the 128 x 128 kernel on every shape the model has (QKV forward 38.7 vs 35.3
// us): both 256 x 256 forms sit at the same ~2 us per k-tile, the rate at which ONE CU turns LDS-DMA requests into LDS lines
// while its matrix cores run (ablations: MFMA only 1.24, LDS-DMA only 1.07, both 1.91 us per k-tile).  To build it again:
// copy to peneo_amd/csrc/, declare launch_gemm_p8 in gemm_big.hip and call it where a 256 x 256 tile is picked.
// bf16 GEMM, 256 x 256 workgroup tile, ONE 8-wave workgroup per CU, two wave groups running half a phase apart (gfx950).
//
// Why: in gemm_big.hip all eight waves of the CU walk the k-tile in lock step behind one barrier -- they
// all read fragments, all issue LDS-DMA pieces, all multiply -- so the matrix cores idle whenever the waves are in their
// memory phase (2.1 us per 256 x 256 k-tile, 40 % of the MFMA rate; every vector-memory instruction cost ~100 cycles of
// matrix-core time although nothing waited for its data).  Here the two waves that share a SIMD (w and w + 4) are held HALF A
// PHASE APART by one extra s_barrier: while waves 0-3 issue their ds_reads and LDS-DMA pieces (the "L" half), waves 4-7 run
// their MFMA cluster (the "M" half), then they swap.  The matrix pipe of every SIMD always has one wave in its cluster
// (s_setprio 1 around it), the other wave's memory instructions issue underneath.
//
// Tile walk.  The 256 x 256 tile is four 128 x 128 quadrants; in a quadrant the 8 waves sit 2 x 4 on 64 x 32 sub-tiles
// (two 32 x 32 x 16 MFMA accumulators each).  A k-tile (64) is staged as four 16 KiB UNITS -- A rows 0-127 (A0), A rows
// 128-255 (A1), and the two halves of B (B0, B1) -- whole 128-byte lines each.  A phase multiplies ONE quadrant over the
// whole k-tile (8 MFMAs = 256 cycles per wave):
//     P0: read A0 (8 b128) + B0 (4)  -> quadrant (0,0)      P1: read B1 (4) -> (0,1)
//     P2: read A1 (8)                -> (1,1)               P3: nothing to read -> (1,0) from the kept B0
// so the units are consumed one phase apart (A0 B0 | B1 | A1 | -) and are re-staged in the same order, one unit per phase,
// a whole k-tile ahead (ring of 2 k-tiles = 8 units = 128 KiB): every piece has 3-4 phases (~0.7 us) to land, the counted
// s_waitcnt vmcnt(4) at the end of each L half never drains the queue (two younger units stay in flight across the
// barriers), and nothing is read in the phase whose wait retired it (the wait sits in front of the barrier that opens the
// reading phase for BOTH groups).
//
// Layouts: A k-major [M][K]; B k-major [N][K] (forward, x W^T) or mn-major [K][N] (dgrad, dy W); K % 64 == 0, N % 8 == 0.
// LDS images are gemm_big.hip's: k-major rows of 128 B with 16-byte slot s of row r at s ^ ((r >> 1) & 7) (swizzle on the
// DMA source address, destination lane-linear), fragments by ds_read_b128; mn-major 1 KiB pieces of [8 k][64 columns] with
// the two 64-byte halves swapped on odd k-pairs, fragments by ds_read_b64_tr_b16.  With a k-major B the rows of a unit are
// dealt so that a wave's two sub-tile columns are 64 CONSECUTIVE output columns (unit qn holds rows wc * 64 + qn * 32 + r);
// with an mn-major B a unit must be 128 consecutive columns (whole lines), so a wave owns two 32-column strips 128 apart.
// Split-k: blockIdx.y = slice; raw fp32 partials to ws[z][M][N], reduced + epilogue by gemm.hip's splitk_reduce8_kernel.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_common.h"

namespace peneo {

typedef short p8_s16x4 __attribute__((ext_vector_type(4)));

constexpr int P8_UNIT = 16384, P8_TILE = 4 * P8_UNIT, P8_LDS = 2 * P8_TILE;
constexpr int P8_A0 = 0, P8_B0 = P8_UNIT, P8_B1 = 2 * P8_UNIT, P8_A1 = 3 * P8_UNIT;   // issue order inside a ring slot
constexpr int P8_EP_LD = 64 + 4;                                                      // floats per row of an epilogue patch

int g_p8_flags = 1;     // bit 0: rotated k order (tools: peneo_gemm_set_p8_flags)

template <bool BK, bool PH2>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmParams p, int tiles_n, int flags) {
  const bool P8_ROTATE = flags & 1;
  const bool NO_DMA = flags & 2, NO_MMA = flags & 4;   // ablations (tools): wrong results
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware tile order (workgroup ids go round-robin to the 8 XCDs; every XCD gets a contiguous band of tiles, n fastest)
  const int total = gridDim.x, lin = blockIdx.x;
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, slot8 = lin >> 3;
  const int tile = xcd * q8 + min(xcd, r8) + slot8;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int ktiles = p.K / 64;
  const int kt_begin = p.split_k > 1 ? (int)blockIdx.y * p.kt_per_split : 0;
  const int nkt = (p.split_k > 1 ? min(ktiles, kt_begin + p.kt_per_split) : ktiles) - kt_begin;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  // ---- LDS-DMA sources: this wave's two pieces (g = wave, wave + 8) of each unit ----
  const char* sA0[2]; const char* sA1[2]; const char* sB0[2]; const char* sB1[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = wave + 8 * u;
    const int lr = g * 8 + (lane >> 3);
    const int sg = (lane & 7) ^ ((lr >> 1) & 7);
    sA0[u] = reinterpret_cast<const char*>(A + (int64_t)min(m0 + lr, p.M - 1) * p.lda + (int64_t)kt_begin * 64 + sg * 8);
    sA1[u] = reinterpret_cast<const char*>(A + (int64_t)min(m0 + 128 + lr, p.M - 1) * p.lda + (int64_t)kt_begin * 64 + sg * 8);
    if constexpr (BK) {
      const int nl = (lr >> 5) * 64 + (lr & 31);      // unit row lr -> tile column of unit 0 (unit 1: + 32)
      sB0[u] = reinterpret_cast<const char*>(B + (int64_t)min(n0 + nl, p.N - 1) * p.ldb + (int64_t)kt_begin * 64 + sg * 8);
      sB1[u] = reinterpret_cast<const char*>(B + (int64_t)min(n0 + nl + 32, p.N - 1) * p.ldb + (int64_t)kt_begin * 64 + sg * 8);
    } else {
      const int kb = g >> 1, nq = g & 1, kr = lane >> 3;
      const int cg = (lane & 7) ^ (((kr >> 1) & 1) << 2);
      const int64_t krow = (int64_t)kt_begin * 64 + kb * 8 + kr;
      sB0[u] = reinterpret_cast<const char*>(B + krow * p.ldb + min(n0 + nq * 64 + cg * 8, p.N - 8));
      sB1[u] = reinterpret_cast<const char*>(B + krow * p.ldb + min(n0 + 128 + nq * 64 + cg * 8, p.N - 8));
    }
  }
  const int64_t stepB = BK ? 128 : (int64_t)64 * p.ldb * 2;
  {
    const int kpos0 = P8_ROTATE ? ((tile / tiles_n) + (tile % tiles_n)) % nkt : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) { sA0[u] += (int64_t)kpos0 * 128; sA1[u] += (int64_t)kpos0 * 128; sB0[u] += kpos0 * stepB; sB1[u] += kpos0 * stepB; }
  }
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  auto issue2 = [&](const char* (&src)[2], uint32_t dst) {
    if (NO_DMA) return;
    lds_dma_1k<0>(src[0], dst);
    lds_dma_1k<0>(src[1], dst + 8192);
  };
  // k-tiles are walked in ROTATED order, start = (tile row + tile column) % nkt: workgroups that share an A panel (same tile
  // row) or a B panel (same tile column) run in lock step, and unrotated they all ask the XCD's L2 for the SAME lines at the
  // same moment (one channel serves a line); rotated, neighbours pull different k-tiles of the shared panel at any time
  int kpos = P8_ROTATE ? ((tile / tiles_n) + (tile % tiles_n)) % nkt : 0;
  auto advance = [&]() {
    ++kpos;
    int64_t da = 128, db = stepB;
    if (kpos == nkt) { kpos = 0; da -= (int64_t)nkt * 128; db -= (int64_t)nkt * stepB; }
#pragma unroll
    for (int u = 0; u < 2; ++u) { sA0[u] += da; sA1[u] += da; sB0[u] += db; sB1[u] += db; }
  };

  // ---- fragment offsets inside a unit ----
  int aoff[4], boff[4];
  {
    const int row = wr * 64 + (lane & 31), swz = (row >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aoff[ks] = row * 128 + (((2 * ks + half) ^ swz) << 4);
  }
  if constexpr (BK) {
    const int row = wc * 32 + (lane & 31), swz = (row >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[ks] = row * 128 + (((2 * ks + half) ^ swz) << 4);
  } else {
    const int n = wc * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int b0 = (half * 2 + (n >> 6)) * 1024 + ((lane & 15) >> 2) * 128 + (((n & 63) * 2) ^ (((lane >> 3) & 1) << 6));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[ks] = b0 + ks * 4096;
  }

  f32x16_t acc[2][2][2];      // [qm][qn][i]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.f;

  uint4 ra[4][2], rb0[4], rb1[4];
  auto load_a = [&](const char* unit) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ra[ks][0] = *reinterpret_cast<const uint4*>(unit + aoff[ks]);
      ra[ks][1] = *reinterpret_cast<const uint4*>(unit + aoff[ks] + 4096);
    }
  };
  auto load_b = [&](const char* unit, uint4 (&rb)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (BK) {
        rb[ks] = *reinterpret_cast<const uint4*>(unit + boff[ks]);
      } else {
        typedef __attribute__((address_space(3))) p8_s16x4* lds_s4p;
        const char* q = unit + boff[ks];
        const p8_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(q));
        const p8_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(q + 512));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        rb[ks] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
    }
  };
  auto mma = [&](f32x16_t (&c)[2], const uint4 (&rb)[4]) {
    if (NO_MMA) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        asm volatile("" :: "v"(ra[ks][0].x), "v"(ra[ks][0].w), "v"(ra[ks][1].x), "v"(ra[ks][1].w), "v"(rb[ks].x), "v"(rb[ks].w));
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ra[ks][i]), __builtin_bit_cast(bf16x8_t, rb[ks]),
                                                       c[i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: the four units of the first k-tile; A0 and B0 have landed (everybody's) behind the barrier ----
  issue2(sA0, wbase + P8_A0); issue2(sB0, wbase + P8_B0); issue2(sB1, wbase + P8_B1); issue2(sA1, wbase + P8_A1);
  advance();
  if constexpr (PH2) {
    // TWO phases per k-tile (16 MFMAs = 512 cycles per M half: half as many group swaps per FLOP):
    //   PA: read A0 B0 B1 (16 b128), issue A0 B0 B1 of the next tile -> quadrants (0,0) (0,1)
    //   PB: read A1 (8),            issue A1 of the next tile        -> quadrants (1,1) (1,0)
    wait_vm<2>();
    bar();
    if (wr == 1) bar();
    for (int t = 0; t < nkt; ++t) {
      const char* cur = smem + (t & 1) * P8_TILE;
      const uint32_t nxt = __builtin_amdgcn_readfirstlane(wbase + ((t + 1) & 1) * P8_TILE);
      const bool more = t + 1 < nkt;
      load_a(cur + P8_A0); load_b(cur + P8_B0, rb0); load_b(cur + P8_B1, rb1);
      if (more) { issue2(sA0, nxt + P8_A0); issue2(sB0, nxt + P8_B0); issue2(sB1, nxt + P8_B1); wait_vm<6>(); } else wait_vm<0>();   // A1 of this tile
      bar();
      mma(acc[0][0], rb0); mma(acc[0][1], rb1);
      bar();
      load_a(cur + P8_A1);
      if (more) { issue2(sA1, nxt + P8_A1); advance(); wait_vm<2>(); }                 // A0 B0 B1 of the next tile
      bar();
      mma(acc[1][1], rb1); mma(acc[1][0], rb0);
      bar();
    }
  } else {
  wait_vm<4>();
  bar();
  if (wr == 1) bar();          // waves 4-7 run half a phase behind waves 0-3 from here on (wave-uniform branch)

  for (int t = 0; t < nkt; ++t) {
    const char* cur = smem + (t & 1) * P8_TILE;
    const uint32_t nxt = __builtin_amdgcn_readfirstlane(wbase + ((t + 1) & 1) * P8_TILE);
    const bool more = t + 1 < nkt;
    // P0 -------------------------------------------------------------------------------------
    load_a(cur + P8_A0); load_b(cur + P8_B0, rb0);
    if (more) { issue2(sA0, nxt + P8_A0); wait_vm<4>(); } else wait_vm<2>();     // B1 of this tile has landed
    bar();
    mma(acc[0][0], rb0);
    bar();
    // P1 -------------------------------------------------------------------------------------
    load_b(cur + P8_B1, rb1);
    if (more) { issue2(sB0, nxt + P8_B0); wait_vm<4>(); } else wait_vm<0>();   // A1 of this tile has landed
    bar();
    mma(acc[0][1], rb1);
    bar();
    // P2 -------------------------------------------------------------------------------------
    load_a(cur + P8_A1);
    if (more) { issue2(sB1, nxt + P8_B1); wait_vm<4>(); }
    bar();
    mma(acc[1][1], rb1);
    bar();
    // P3 -------------------------------------------------------------------------------------
    if (more) { issue2(sA1, nxt + P8_A1); advance(); wait_vm<4>(); }                        // A0 and B0 of the next tile have landed
    bar();
    mma(acc[1][0], rb0);
    bar();
  }
  }
  if (wr == 0) bar();          // pairs with the last barrier of waves 4-7: the ring is dead behind it

  // ---- epilogue: 32-row blocks through a wave-private LDS patch, 8 consecutive columns per lane ----
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * P8_EP_LD);
  const int cgi = lane & 7, rli = lane >> 3;
  const int nb = BK ? n0 + wc * 64 + cgi * 8 : n0 + (cgi >> 2) * 128 + wc * 32 + (cgi & 3) * 8;
  auto block = [&](auto qmc, auto ic) {
    constexpr int qm = decltype(qmc)::value, i = decltype(ic)::value;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[acc_row(r, lane) * P8_EP_LD + qn * 32 + acc_col(lane)] = acc[qm][qn][i][r];
    __builtin_amdgcn_wave_barrier();
    const int mb = m0 + qm * 128 + wr * 64 + i * 32;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rl = pass * 8 + rli;
      if (mb + rl < p.M && nb + 8 <= p.N) {
        const float4 x0 = *reinterpret_cast<const float4*>(patch + rl * P8_EP_LD + cgi * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(patch + rl * P8_EP_LD + cgi * 8 + 4);
        if (p.split_k > 1) {
          float* w = p.ws + ((int64_t)blockIdx.y * p.M + (mb + rl)) * p.N + nb;
          *reinterpret_cast<float4*>(w) = x0;
          *reinterpret_cast<float4*>(w + 4) = x1;
        } else {
          float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          epilogue_store8(p, mb + rl, nb, v);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  block(I0{}, I0{}); block(I0{}, I1{}); block(I1{}, I0{}); block(I1{}, I1{});
}

template <bool BK, bool PH2>
static int launch_p8(const GemmParams& p, hipStream_t st) {
  const int tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_kernel<BK, PH2>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS) != hipSuccess) {
      set_error("peneo_gemm: cannot raise dynamic LDS to %d bytes", P8_LDS);
      return PENEO_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_p8_kernel<BK, PH2>), dim3((unsigned)(tm * tn), (unsigned)(p.split_k > 1 ? p.split_k : 1)), dim3(512), P8_LDS, st, p, tn, g_p8_flags);
  const int rc = check_launch("peneo_gemm (256 x 256, staggered)");
  return rc == PENEO_OK ? 1 : rc;
}

// 0 = shape / options not covered, 1 = launched, < 0 = error.  The caller has checked alignment (gemm_big.hip's gate).
int launch_gemm_p8(const GemmParams& p, bool b_kmajor, hipStream_t st) {
  if (p.dz_on || p.K % 64 != 0 || p.N % 8 != 0 || p.M < 128 || p.N < 128) return 0;
  if (p.split_k > 1 && (p.N % 4 != 0 || (reinterpret_cast<uintptr_t>(p.ws) & 15))) return 0;
  if (g_p8_flags & 8) return b_kmajor ? launch_p8<true, true>(p, st) : launch_p8<false, true>(p, st);
  return b_kmajor ? launch_p8<true, false>(p, st) : launch_p8<false, false>(p, st);
}

}  // namespace peneo

extern "C" void peneo_gemm_set_p8_flags(int flags) { peneo::g_p8_flags = flags; }
