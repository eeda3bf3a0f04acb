// K13, OHEM branch: CrossEntropyLossOHEM.forward with num_hard_positive / num_hard_negative != -1
// (model/custom_loss.py:204-288), exactly AS EXECUTED by the reference, on the device:
//
//   ce[p]     = w[y_p] * nll_p                                   (F.cross_entropy(..., reduction="none"), :204-210)
//   positives = ce[y != 0], negatives = ce[y == 0], both in flattened order               (:236-238)
//   sorted, idx = sort(descending)                                                         (:259-261, :269-271)
//   k = min(count, num_hard);  k <= 0: every element stays;  k < count: kept = sorted[idx[:k]]  (:262-267, :272-277)
//       NB that last line indexes the SORTED array with positions of the UNSORTED one: the kept VALUES are the
//       idx[t]-th largest losses, t < k — not the k largest.  Reproduced as is.
//   loss = (sum kept positives + sum kept negatives) / (k_pos + k_neg)      ("mean", :279-283; k may be <= 0 there too)
//
// One stable device radix sort (rocPRIM, header-only part of ROCm) over a 33-bit key {is_negative, ~ordered(ce)} puts the
// positives first, each class in descending loss, equal losses in flattened order; an exclusive scan of the positive
// flags gives every pair its position inside its own list (what `idx` holds).  No host round trip: the counts stay on
// the device, the caller reads the four result scalars when it needs them.
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace peneo {

struct OhemWs {
  float* ce; uint32_t* flag; uint32_t* rank; uint64_t* key_in; uint64_t* key_out; uint32_t* val_in; uint32_t* val_out;
  uint8_t* keep; int32_t* cnt; void* temp; size_t temp_bytes; size_t total;
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t prim_temp_bytes(int64_t n) {
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config>(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                                          (uint32_t*)nullptr, (size_t)n, 0u, 33u, (hipStream_t)0, false);
  (void)rocprim::exclusive_scan(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(),
                               (hipStream_t)0, false);
  return align256((a > b ? a : b) + 256);
}

static OhemWs carve(void* base, int64_t n) {
  OhemWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? (char*)base + off : nullptr; off += align256(bytes); return p; };
  w.ce = (float*)take(sizeof(float) * n);
  w.flag = (uint32_t*)take(sizeof(uint32_t) * n);
  w.rank = (uint32_t*)take(sizeof(uint32_t) * n);
  w.key_in = (uint64_t*)take(sizeof(uint64_t) * n);
  w.key_out = (uint64_t*)take(sizeof(uint64_t) * n);
  w.val_in = (uint32_t*)take(sizeof(uint32_t) * n);
  w.val_out = (uint32_t*)take(sizeof(uint32_t) * n);
  w.keep = (uint8_t*)take(n);
  w.cnt = (int32_t*)take(64);
  w.temp_bytes = prim_temp_bytes(n);
  w.temp = take(w.temp_bytes);
  w.total = off;
  return w;
}

// monotone float -> uint32 (larger float = larger integer), then inverted so that an ascending sort is descending in ce
__device__ __forceinline__ uint32_t desc_key(float f) {
  uint32_t u = __float_as_uint(f);
  u ^= (u & 0x80000000u) ? 0xffffffffu : 0x80000000u;
  return ~u;
}

__global__ __launch_bounds__(256) void ohem_ce_kernel(const float* logits, const int64_t* tags, const float* cw, int64_t n, int C,
                                                      float* ce, uint32_t* flag, uint64_t* key, uint32_t* val) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const float* l = logits + p * C;
    const int tag = (int)tags[p];
    float mx = l[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(l[c] - mx);      // full-precision exp / log: the ORDER of the losses is the result
    // ignore_index (-100, custom_loss.py:236-243: F.cross_entropy(..., ignore_index) gives such a pair zero loss while the
    // `target != 0` test still files it under the positives) and any other label outside [0, C) (memory safety): zero loss
    const bool valid = tag >= 0 && tag < C;
    const float w = !valid ? 0.f : (cw ? cw[tag] : 1.f);
    const float v = valid ? w * (logf(se) - (l[tag] - mx)) : 0.f;   // -log_softmax(l)[tag], grouped like torch's log_softmax
    const uint32_t pos = tag != 0;
    ce[p] = v;
    flag[p] = pos;
    key[p] = ((uint64_t)(pos ^ 1u) << 32) | desc_key(v);
    val[p] = (uint32_t)p;
  }
}

// cnt[0] = n_pos, cnt[1] = n_neg, cnt[2] = k_pos, cnt[3] = k_neg (k = min(count, num_hard), may be <= 0)
__global__ void ohem_counts_kernel(const uint32_t* flag, const uint32_t* rank, int64_t n, int num_pos, int num_neg, int32_t* cnt) {
  const int n_pos = (int)(rank[n - 1] + flag[n - 1]);
  const int n_neg = (int)(n - n_pos);
  cnt[0] = n_pos; cnt[1] = n_neg;
  cnt[2] = n_pos < num_pos ? n_pos : num_pos;
  cnt[3] = n_neg < num_neg ? n_neg : num_neg;
}

// t-th hardest of a class -> its position j in the class's own (flattened-order) list -> the element of sorted rank j stays
__global__ __launch_bounds__(256) void ohem_select_kernel(const uint32_t* sorted_idx, const uint32_t* rank, int64_t n,
                                                          const int32_t* cnt, uint8_t* keep) {
  const int n_pos = cnt[0], n_neg = cnt[1], k_pos = cnt[2], k_neg = cnt[3];
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    if (t < n_pos) {
      if (k_pos > 0 && k_pos < n_pos && t < k_pos) {
        const uint32_t p = sorted_idx[t];
        keep[sorted_idx[rank[p]]] = 1;
      }
    } else {
      const int64_t u = t - n_pos;
      if (k_neg > 0 && k_neg < n_neg && u < k_neg) {
        const uint32_t p = sorted_idx[t];
        keep[sorted_idx[n_pos + (p - rank[p])]] = 1;
      }
    }
  }
}

// sum of the kept losses, masked dlogits and their column sums; acc = [num, dl_sum[0..C)]
__global__ __launch_bounds__(256) void ohem_finish_kernel(const float* ce, const uint32_t* flag, const uint8_t* keep, int64_t n, int C,
                                                          const int32_t* cnt, float* dlogits, float* acc) {
  const int n_pos = cnt[0], n_neg = cnt[1], k_pos = cnt[2], k_neg = cnt[3];
  const bool all_pos = !(k_pos > 0 && k_pos < n_pos), all_neg = !(k_neg > 0 && k_neg < n_neg);
  float num = 0.f, dl[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) dl[c] = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const bool kept = flag[p] ? (all_pos || keep[p]) : (all_neg || keep[p]);
    if (kept) num += ce[p];
    if (dlogits) {
      if (kept) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < C) dl[c] += dlogits[p * C + c];
      } else {
        for (int c = 0; c < C; ++c) dlogits[p * C + c] = 0.f;
      }
    }
  }
  __shared__ float red[4][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 17; ++c) {
    float v = c == 0 ? num : dl[c - 1];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x <= C) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(acc + threadIdx.x, v);
  }
}

// out = [loss, num, den, n_pos, n_neg, k_pos, k_neg, 0]; dl_sum[C]
__global__ void ohem_out_kernel(const float* acc, const int32_t* cnt, int C, float* out, float* dl_sum) {
  const float den = (float)(cnt[2] + cnt[3]);
  out[0] = acc[0] / den; out[1] = acc[0]; out[2] = den;
  out[3] = (float)cnt[0]; out[4] = (float)cnt[1]; out[5] = (float)cnt[2]; out[6] = (float)cnt[3]; out[7] = 0.f;
  if (dl_sum)
    for (int c = 0; c < C; ++c) dl_sum[c] = acc[1 + c];
}

// per-head OHEM results -> what the decoder stage hands out: losses (+ weighted total) and the backward scales
__global__ void ohem_combine_kernel(const float* out8, const float* ratio, int nh, float* out, float* scale, float* inv_den) {
  float tot = 0.f;
  for (int h = 0; h < nh; ++h) {
    const float l = out8[h * 8], den = out8[h * 8 + 2];
    out[h] = l;
    tot += ratio[h] * l;
    if (scale) scale[h] = ratio[h] / den;
    if (inv_den) inv_den[h] = 1.0f / den;
  }
  out[nh] = tot;
}

}  // namespace peneo
using namespace peneo;

extern "C" int peneo_ohem_finish(const float* out8, const float* ratio, int num_heads, float* out, float* scale, float* inv_den,
                                 peneo_stream_t stream) {
  PENEO_REQUIRE(out8 && ratio && out && num_heads > 0 && num_heads <= PENEO_MAX_HEADS, "peneo_ohem_finish: bad arguments");
  hipLaunchKernelGGL(ohem_combine_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out8, ratio, num_heads, out, scale, inv_den);
  return check_launch("peneo_ohem_finish");
}

extern "C" size_t peneo_ohem_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return carve(nullptr, n).total;
}

extern "C" int peneo_ohem_ce(const float* logits, const int64_t* tags, const float* class_weight, int64_t n, int C,
                             int num_hard_positive, int num_hard_negative, float* dlogits, float* out8, float* dl_sum,
                             void* workspace, size_t workspace_bytes, peneo_stream_t stream) {
  PENEO_REQUIRE(logits && tags && out8 && workspace && n > 0 && n < ((int64_t)1 << 31) && C > 0 && C <= 16,
                "peneo_ohem_ce: bad arguments");
  PENEO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "peneo_ohem_ce: workspace must be 256-byte aligned");
  OhemWs w = carve(workspace, n);
  PENEO_REQUIRE(workspace_bytes >= w.total, "peneo_ohem_ce: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(ohem_ce_kernel, dim3(blocks), dim3(256), 0, st, logits, tags, class_weight, n, C, w.ce, w.flag, w.key_in, w.val_in);
  size_t tb = w.temp_bytes;
  if (rocprim::exclusive_scan(w.temp, tb, w.flag, w.rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st, false) != hipSuccess) {
    set_error("peneo_ohem_ce: scan failed"); return PENEO_ERR_LAUNCH;
  }
  tb = w.temp_bytes;
  if (rocprim::radix_sort_pairs<rocprim::default_config>(w.temp, tb, w.key_in, w.key_out, w.val_in, w.val_out, (size_t)n, 0u, 33u, st,
                                                        false) != hipSuccess) {
    set_error("peneo_ohem_ce: sort failed"); return PENEO_ERR_LAUNCH;
  }
  float* acc = reinterpret_cast<float*>(w.temp);      // [num, dl_sum[16]] — the scan / sort scratch is dead from here on
  if (hipMemsetAsync(w.keep, 0, n, st) != hipSuccess || hipMemsetAsync(acc, 0, sizeof(float) * 32, st) != hipSuccess) {
    set_error("peneo_ohem_ce: memset failed"); return PENEO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(ohem_counts_kernel, dim3(1), dim3(1), 0, st, w.flag, w.rank, n, num_hard_positive, num_hard_negative, w.cnt);
  hipLaunchKernelGGL(ohem_select_kernel, dim3(blocks), dim3(256), 0, st, w.val_out, w.rank, n, w.cnt, w.keep);
  hipLaunchKernelGGL(ohem_finish_kernel, dim3(blocks), dim3(256), 0, st, w.ce, w.flag, w.keep, n, C, w.cnt, dlogits, acc);
  hipLaunchKernelGGL(ohem_out_kernel, dim3(1), dim3(1), 0, st, acc, w.cnt, C, out8, dl_sum);
  return check_launch("peneo_ohem_ce");
}
