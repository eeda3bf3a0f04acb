// Memory-bound plumbing kernels: dtype casts, strided 2-D copy (+dropout), column sums.
#include <cmath>
#include "common.h"

namespace peneo {

__device__ __forceinline__ float ld_any(const void* p, int dt, int64_t i) {
  return dt == PENEO_F32 ? reinterpret_cast<const float*>(p)[i] : bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void st_any(void* p, int dt, int64_t i, float v) {
  if (dt == PENEO_F32) reinterpret_cast<float*>(p)[i] = v; else reinterpret_cast<bf16_t*>(p)[i] = f32_to_bf16(v);
}

// 8 elements per thread: 16/32-byte vector traffic per lane
__global__ void cast_kernel(const void* src, int sdt, void* dst, int ddt, int64_t n) {
  int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (; base < n; base += stride) {
    if (base + 8 <= n) {
      float f[8];
      if (sdt == PENEO_F32) {
        const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(src) + base);
        unpack16<float>(s[0], f); unpack16<float>(s[1], f + 4);
      } else {
        unpack16<bf16_t>(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(src) + base), f);
      }
      if (ddt == PENEO_F32) {
        uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<float*>(dst) + base);
        d[0] = pack16<float>(f); d[1] = pack16<float>(f + 4);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(dst) + base) = pack16<bf16_t>(f);
      }
    } else {
      for (int64_t i = base; i < n; ++i) st_any(dst, ddt, i, ld_any(src, sdt, i));
    }
  }
}

__global__ void copy2d_kernel(int dt, const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int64_t cols,
                              float drop_p, uint32_t seed) {
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / cols, c = i % cols;
    float v = ld_any(src, dt, r * lds_ + c);
    if (drop_p > 0.f) v = dropout_keep(seed, (uint64_t)i, thresh) ? v * keep_scale : 0.f;
    st_any(dst, dt, r * ldd + c, v);
  }
}

// rows of a batched, strided 2-D view: row r lives at (r / rpb) * bstride + (r % rpb) * ld  (rpb = 0: r * ld)
__global__ void copy_rows_kernel(int dt, const void* src, int64_t s_rpb, int64_t s_bs, int64_t s_ld, void* dst, int64_t d_rpb,
                                 int64_t d_bs, int64_t d_ld, int64_t rows, int64_t cols, float drop_p, uint32_t seed) {
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / cols, c = i % cols;
    int64_t so = (s_rpb > 0 ? (r / s_rpb) * s_bs + (r % s_rpb) * s_ld : r * s_ld) + c;
    int64_t dof = (d_rpb > 0 ? (r / d_rpb) * d_bs + (r % d_rpb) * d_ld : r * d_ld) + c;
    float v = ld_any(src, dt, so);
    if (drop_p > 0.f) v = dropout_keep(seed, (uint64_t)i, thresh) ? v * keep_scale : 0.f;
    st_any(dst, dt, dof, v);
  }
}

// per-head concat / split of two [rows, nh*d] matrices (LiLT's BiACM: one attention call over [text | layout] heads)
__global__ void head_concat_kernel(int dt, const void* a, int64_t lda, int da, float sa, const void* b, int64_t ldb, int db,
                                   float sb, void* out, int64_t ldo, int64_t rows, int nh) {
  const int dc = da + db;
  const int64_t total = rows * nh * dc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dc);
    const int h = (int)((i / dc) % nh);
    const int64_t r = i / ((int64_t)dc * nh);
    const float v = c < da ? ld_any(a, dt, r * lda + h * da + c) * sa : ld_any(b, dt, r * ldb + h * db + (c - da)) * sb;
    st_any(out, dt, r * ldo + h * dc + c, v);
  }
}
__global__ void head_split_kernel(int dt, const void* in, int64_t ldi, void* a, int64_t lda, int da, float sa, void* b,
                                  int64_t ldb, int db, float sb, int64_t rows, int nh) {
  const int dc = da + db;
  const int64_t total = rows * nh * dc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dc);
    const int h = (int)((i / dc) % nh);
    const int64_t r = i / ((int64_t)dc * nh);
    const float v = ld_any(in, dt, r * ldi + h * dc + c);
    if (c < da) st_any(a, dt, r * lda + h * da + c, v * sa);
    else st_any(b, dt, r * ldb + h * db + (c - da), v * sb);
  }
}

// 16-byte form of the two kernels above (da, db multiples of the vector width, 16-byte aligned rows): one vector of one
// head per thread, no per-element div / mod.  The scalar kernels moved 2 bytes per load and reached ~1.1 TB/s.
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void head_pack_vec_kernel(const T* a, int64_t lda, int da, float sa, const T* b, int64_t ldb, int db,
                                                           float sb, T* c, int64_t ldc, int64_t rows, int nh) {
  constexpr int VEC = Elem<T>::kVec;
  const int va = da / VEC, vc = (da + db) / VEC;
  const int64_t total = rows * nh * vc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % vc);
    const int h = (int)((i / vc) % nh);
    const int64_t r = i / ((int64_t)vc * nh);
    const bool first = v < va;
    T* pc = c + r * ldc + (int64_t)h * (da + db) + v * VEC;
    T* pab = first ? const_cast<T*>(a) + r * lda + (int64_t)h * da + v * VEC : const_cast<T*>(b) + r * ldb + (int64_t)h * db + (v - va) * VEC;
    const float sc = first ? sa : sb;
    float f[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(SPLIT ? pc : pab), f);
    if (sc != 1.0f) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] *= sc;
    }
    *reinterpret_cast<uint4*>(SPLIT ? pab : pc) = pack16<T>(f);
  }
}
static bool head_vec_ok(int dtype, const void* a, int64_t lda, int da, const void* b, int64_t ldb, int db, const void* c, int64_t ldc) {
  const int vec = dtype == PENEO_BF16 ? 8 : 4, esz = dtype == PENEO_BF16 ? 2 : 4;
  auto al = [&](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld * esz) % 16 == 0; };
  return da % vec == 0 && db % vec == 0 && al(a, lda) && al(b, ldb) && al(c, ldc);
}
template <bool SPLIT>
static void launch_head_pack_vec(int dtype, const void* a, int64_t lda, int da, float sa, const void* b, int64_t ldb, int db, float sb,
                                 void* c, int64_t ldc, int64_t rows, int nh, hipStream_t st) {
  const int vec = dtype == PENEO_BF16 ? 8 : 4;
  int64_t blocks = (rows * nh * ((da + db) / vec) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL((head_pack_vec_kernel<bf16_t, SPLIT>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)a, lda, da, sa,
                       (const bf16_t*)b, ldb, db, sb, (bf16_t*)c, ldc, rows, nh);
  else
    hipLaunchKernelGGL((head_pack_vec_kernel<float, SPLIT>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a, lda, da, sa,
                       (const float*)b, ldb, db, sb, (float*)c, ldc, rows, nh);
}

// block = 64 columns x 4 row lanes; each block reduces ROWS_PER_BLOCK rows and adds into out.
constexpr int CS_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_kernel(int dt, const void* x, int64_t ldx, int64_t M, int64_t N, float* out) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + cl;
  const int64_t m0 = (int64_t)blockIdx.y * CS_ROWS;
  const int64_t m1 = min(M, m0 + (int64_t)CS_ROWS);
  float s = 0.f;
  if (n < N)
    for (int64_t m = m0 + rl; m < m1; m += 4) s += ld_any(x, dt, m * ldx + n);
  part[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && n < N) atomicAdd(out + n, part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl]);
}

// vector variant: thread = 8 consecutive columns (16 B of bf16 / 2 x 16 B of fp32), 32 column groups x 8 row lanes per
// block, CSV_ROWS rows per block with every load of a thread issued before the first add (deep memory-level parallelism).
constexpr int CSV_ROWS = 128;
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* x, int64_t ldx, int64_t M, int64_t N, float* out) {
  __shared__ float part[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t n = ((int64_t)blockIdx.x * 32 + cg) * 8;
  const int64_t m0 = (int64_t)blockIdx.y * CSV_ROWS;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (n < N) {
    constexpr int NR = CSV_ROWS / 8;
    if constexpr (sizeof(T) == 2) {
      uint4 v[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int64_t m = m0 + rl + 8 * i;
        v[i] = m < M ? *reinterpret_cast<const uint4*>(x + m * ldx + n) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        float f[8];
        unpack16<T>(v[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    } else {
#pragma unroll 4
      for (int i = 0; i < NR; ++i) {
        const int64_t m = m0 + rl + 8 * i;
        if (m < M) {
          const float4 a = *reinterpret_cast<const float4*>(x + m * ldx + n);
          const float4 b = *reinterpret_cast<const float4*>(x + m * ldx + n + 4);
          acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
          acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int64_t nc = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (nc < N) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += part[r][threadIdx.x];
    atomicAdd(out + nc, sum);
  }
}

// fused multi-tensor AdamW (decoupled weight decay, torch.optim.AdamW arithmetic): one launch for every parameter of
// every group; block = one 4096-element chunk of one tensor, per-tensor lr / weight decay from the table
constexpr int ADAMW_CHUNK = 4096;
constexpr int GRAD_SQNORM_SLOTS = 64;     // partial sums of the gradient norm: 31 k blocks adding to ONE address took 0.39 ms (1.3 TB/s)
__global__ __launch_bounds__(256) void adamw_kernel(const peneo_adamw_tensor* tab, const int32_t* chunk_tensor,
                                                    const int32_t* chunk_index, float beta1, float beta2, float eps,
                                                    float bc1, float rsqrt_bc2, int step, const double* sqnorm, float max_norm) {
  const peneo_adamw_tensor t = tab[chunk_tensor[blockIdx.x]];
  // global gradient-norm clipping (torch.nn.utils.clip_grad_norm_, what HF Trainer does at max_grad_norm = 1.0 every step):
  // the coefficient comes from the device-side sum of squares of peneo_grad_sqnorm; the gradients themselves stay untouched
  float clip = 1.0f;
  if (sqnorm) {        // (sum of the GRAD_SQNORM_SLOTS partial sums peneo_grad_sqnorm left)
    double sq = 0.0;
    for (int i = 0; i < GRAD_SQNORM_SLOTS; ++i) sq += sqnorm[i];
    clip = fminf(1.0f, max_norm / ((float)sqrt(sq) + 1e-6f));
  }
  if (t.step_offset != 0) {   // this tensor's own step count (block-uniform branch; resumed / late-joining parameters only)
    const double s = (double)(step + t.step_offset);
    bc1 = (float)(1.0 - pow((double)beta1, s));
    rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, s)));
  }
  const int64_t base = (int64_t)chunk_index[blockIdx.x] * ADAMW_CHUNK;
  const int64_t end = min(t.numel, base + ADAMW_CHUNK);
  const float step_size = t.lr / bc1, decay = 1.0f - t.lr * t.weight_decay;
  const bool vec = ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.grad) |
                     reinterpret_cast<uintptr_t>(t.exp_avg) | reinterpret_cast<uintptr_t>(t.exp_avg_sq)) & 15) == 0;
  for (int64_t i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
    float p[4], g[4], m[4], v[4];
    const int cnt = (int)min((int64_t)4, end - i);
    if (vec && cnt == 4) {
      *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(t.param + i);
      *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(t.grad + i);
      *reinterpret_cast<float4*>(m) = *reinterpret_cast<const float4*>(t.exp_avg + i);
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(t.exp_avg_sq + i);
    } else {
      for (int e = 0; e < cnt; ++e) { p[e] = t.param[i + e]; g[e] = t.grad[i + e]; m[e] = t.exp_avg[i + e]; v[e] = t.exp_avg_sq[i + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] *= clip;
      p[e] *= decay;
      m[e] = m[e] + (1.0f - beta1) * (g[e] - m[e]);                 // lerp, as torch does
      v[e] = beta2 * v[e] + (1.0f - beta2) * g[e] * g[e];
      const float denom = sqrtf(v[e]) * rsqrt_bc2 + eps;
      p[e] -= step_size * (m[e] / denom);
    }
    if (vec && cnt == 4) {
      *reinterpret_cast<float4*>(t.param + i) = *reinterpret_cast<const float4*>(p);
      *reinterpret_cast<float4*>(t.exp_avg + i) = *reinterpret_cast<const float4*>(m);
      *reinterpret_cast<float4*>(t.exp_avg_sq + i) = *reinterpret_cast<const float4*>(v);
    } else {
      for (int e = 0; e < cnt; ++e) { t.param[i + e] = p[e]; t.exp_avg[i + e] = m[e]; t.exp_avg_sq[i + e] = v[e]; }
    }
  }
}

// sum of squares of every gradient of the table (same chunk maps): per-block fp32 tree, one fp64 atomic per block
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const peneo_adamw_tensor* tab, const int32_t* chunk_tensor,
                                                          const int32_t* chunk_index, double* out) {
  const peneo_adamw_tensor t = tab[chunk_tensor[blockIdx.x]];
  const int64_t base = (int64_t)chunk_index[blockIdx.x] * ADAMW_CHUNK;
  const int64_t end = min(t.numel, base + ADAMW_CHUNK);
  const bool vec = (reinterpret_cast<uintptr_t>(t.grad) & 15) == 0;
  float acc = 0.f;
  for (int64_t i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
    if (vec && i + 4 <= end) {
      const float4 g = *reinterpret_cast<const float4*>(t.grad + i);
      acc += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    } else {
      for (int64_t e = i; e < min(end, i + 4); ++e) acc += t.grad[e] * t.grad[e];
    }
  }
  __shared__ float part[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + (blockIdx.x % GRAD_SQNORM_SLOTS), (double)part[0] + (double)part[1] + (double)part[2] + (double)part[3]);
}

// Many fp32 -> bf16 casts in ONE launch (the working-precision copies of a model's weights after an optimizer step): block =
// one CAST_MULTI_CHUNK-element chunk of one tensor, table and chunk maps as for adamw_kernel
constexpr int CAST_MULTI_CHUNK = 16384;
__global__ __launch_bounds__(256) void cast_multi_kernel(const peneo_cast_item* tab, const int32_t* chunk_item, const int32_t* chunk_index) {
  const peneo_cast_item t = tab[chunk_item[blockIdx.x]];
  const int64_t base = (int64_t)chunk_index[blockIdx.x] * CAST_MULTI_CHUNK;
  const int64_t end = min(t.numel, base + CAST_MULTI_CHUNK);
  bf16_t* dst = reinterpret_cast<bf16_t*>(t.dst);
  const bool vec = ((reinterpret_cast<uintptr_t>(t.src) | reinterpret_cast<uintptr_t>(t.dst)) & 15) == 0;
  for (int64_t i = base + threadIdx.x * 8; i < end; i += 256 * 8) {
    if (vec && i + 8 <= end) {
      float f[8];
      *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(t.src + i);
      *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(t.src + i + 4);
      *reinterpret_cast<uint4*>(dst + i) = pack16<bf16_t>(f);
    } else {
      for (int64_t e = i; e < min(end, i + 8); ++e) dst[e] = f32_to_bf16(t.src[e]);
    }
  }
}

}  // namespace peneo
using namespace peneo;

extern "C" int peneo_cast_multi_chunk_elems(void) { return CAST_MULTI_CHUNK; }
extern "C" int peneo_cast_multi(const peneo_cast_item* table_dev, const int32_t* chunk_item_dev, const int32_t* chunk_index_dev,
                                int n_chunks, peneo_stream_t stream) {
  PENEO_REQUIRE(table_dev && chunk_item_dev && chunk_index_dev && n_chunks > 0, "peneo_cast_multi: bad arguments");
  hipLaunchKernelGGL(cast_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, chunk_item_dev, chunk_index_dev);
  return check_launch("peneo_cast_multi");
}

extern "C" int peneo_adamw_chunk_elems(void) { return ADAMW_CHUNK; }
extern "C" int peneo_grad_sqnorm_slots(void) { return GRAD_SQNORM_SLOTS; }

extern "C" int peneo_grad_sqnorm(const peneo_adamw_tensor* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                                 int n_chunks, double* sqnorm_dev, peneo_stream_t stream) {
  PENEO_REQUIRE(table_dev && chunk_tensor_dev && chunk_index_dev && n_chunks > 0 && sqnorm_dev, "peneo_grad_sqnorm: bad arguments");
  if (hipMemsetAsync(sqnorm_dev, 0, sizeof(double) * GRAD_SQNORM_SLOTS, (hipStream_t)stream) != hipSuccess) return check_launch("peneo_grad_sqnorm");
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, chunk_tensor_dev,
                     chunk_index_dev, sqnorm_dev);
  return check_launch("peneo_grad_sqnorm");
}

extern "C" int peneo_adamw_step_clip(const peneo_adamw_tensor* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                                     int n_chunks, float beta1, float beta2, float eps, int step, const double* sqnorm_dev,
                                     float max_grad_norm, peneo_stream_t stream) {
  PENEO_REQUIRE(table_dev && chunk_tensor_dev && chunk_index_dev && n_chunks > 0 && step >= 1, "peneo_adamw_step: bad arguments");
  PENEO_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "peneo_adamw_step: bad hyper-parameters");
  PENEO_REQUIRE(!sqnorm_dev || max_grad_norm > 0.f, "peneo_adamw_step_clip: max_grad_norm must be positive");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, chunk_tensor_dev, chunk_index_dev,
                     beta1, beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), step, sqnorm_dev, max_grad_norm);
  return check_launch("peneo_adamw_step");
}

extern "C" int peneo_adamw_step(const peneo_adamw_tensor* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                                int n_chunks, float beta1, float beta2, float eps, int step, peneo_stream_t stream) {
  return peneo_adamw_step_clip(table_dev, chunk_tensor_dev, chunk_index_dev, n_chunks, beta1, beta2, eps, step, nullptr, 0.f, stream);
}

static inline bool ok_dt(int d) { return d == PENEO_F32 || d == PENEO_BF16; }

extern "C" int peneo_cast(const void* src, int sdt, void* dst, int ddt, int64_t n, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(sdt) && ok_dt(ddt), "peneo_cast: bad dtype");
  if (n <= 0) return PENEO_OK;
  PENEO_REQUIRE(src && dst, "peneo_cast: null pointer");
  PENEO_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
                "peneo_cast: pointers must be 16-byte aligned");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, sdt, dst, ddt, n);
  return check_launch("peneo_cast");
}

extern "C" int peneo_copy2d(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows,
                            int64_t cols, float drop_p, uint32_t drop_seed, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype), "peneo_copy2d: bad dtype");
  if (rows <= 0 || cols <= 0) return PENEO_OK;
  PENEO_REQUIRE(src && dst && ld_src >= cols && ld_dst >= cols, "peneo_copy2d: bad arguments");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "peneo_copy2d: drop_p out of range");
  int64_t blocks = (rows * cols + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dtype, src, ld_src, dst,
                     ld_dst, rows, cols, drop_p, drop_seed);
  return check_launch("peneo_copy2d");
}

extern "C" int peneo_copy_rows(int dtype, const void* src, int64_t src_rpb, int64_t src_bstride, int64_t ld_src, void* dst,
                               int64_t dst_rpb, int64_t dst_bstride, int64_t ld_dst, int64_t rows, int64_t cols, float drop_p,
                               uint32_t drop_seed, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype), "peneo_copy_rows: bad dtype");
  if (rows <= 0 || cols <= 0) return PENEO_OK;
  PENEO_REQUIRE(src && dst && ld_src >= cols && ld_dst >= cols, "peneo_copy_rows: bad arguments");
  PENEO_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "peneo_copy_rows: drop_p out of range");
  int64_t blocks = (rows * cols + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dtype, src, src_rpb,
                     src_bstride, ld_src, dst, dst_rpb, dst_bstride, ld_dst, rows, cols, drop_p, drop_seed);
  return check_launch("peneo_copy_rows");
}

extern "C" int peneo_head_concat(int dtype, const void* a, int64_t lda, int da, float scale_a, const void* b, int64_t ldb, int db,
                                 float scale_b, void* out, int64_t ldo, int64_t rows, int nh, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && a && b && out && rows > 0 && nh > 0 && da > 0 && db > 0, "peneo_head_concat: bad arguments");
  PENEO_REQUIRE(lda >= (int64_t)nh * da && ldb >= (int64_t)nh * db && ldo >= (int64_t)nh * (da + db), "peneo_head_concat: leading dims too small");
  if (head_vec_ok(dtype, a, lda, da, b, ldb, db, out, ldo)) {
    launch_head_pack_vec<false>(dtype, a, lda, da, scale_a, b, ldb, db, scale_b, out, ldo, rows, nh, (hipStream_t)stream);
    return check_launch("peneo_head_concat");
  }
  int64_t blocks = (rows * nh * (da + db) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(head_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dtype, a, lda, da, scale_a, b,
                     ldb, db, scale_b, out, ldo, rows, nh);
  return check_launch("peneo_head_concat");
}

extern "C" int peneo_head_split(int dtype, const void* in, int64_t ldi, void* a, int64_t lda, int da, float scale_a, void* b,
                                int64_t ldb, int db, float scale_b, int64_t rows, int nh, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && a && b && in && rows > 0 && nh > 0 && da > 0 && db > 0, "peneo_head_split: bad arguments");
  PENEO_REQUIRE(lda >= (int64_t)nh * da && ldb >= (int64_t)nh * db && ldi >= (int64_t)nh * (da + db), "peneo_head_split: leading dims too small");
  if (head_vec_ok(dtype, a, lda, da, b, ldb, db, in, ldi)) {
    launch_head_pack_vec<true>(dtype, a, lda, da, scale_a, b, ldb, db, scale_b, const_cast<void*>(in), ldi, rows, nh, (hipStream_t)stream);
    return check_launch("peneo_head_split");
  }
  int64_t blocks = (rows * nh * (da + db) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(head_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dtype, in, ldi, a, lda, da,
                     scale_a, b, ldb, db, scale_b, rows, nh);
  return check_launch("peneo_head_split");
}

extern "C" int peneo_colsum(int dtype, const void* x, int64_t ldx, int64_t M, int64_t N, float* out, int accumulate,
                            peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype), "peneo_colsum: bad dtype");
  PENEO_REQUIRE(x && out && M > 0 && N > 0 && ldx >= N, "peneo_colsum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) { set_error("peneo_colsum: memset failed"); return PENEO_ERR_LAUNCH; }
  }
  const int esz = dtype == PENEO_BF16 ? 2 : 4;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ldx * esz) % 16 == 0 && N % 8 == 0) {
    dim3 vgrid((unsigned)((N + 255) / 256), (unsigned)((M + CSV_ROWS - 1) / CSV_ROWS));
    if (dtype == PENEO_BF16)
      hipLaunchKernelGGL(colsum_vec_kernel<bf16_t>, vgrid, dim3(256), 0, st, (const bf16_t*)x, ldx, M, N, out);
    else
      hipLaunchKernelGGL(colsum_vec_kernel<float>, vgrid, dim3(256), 0, st, (const float*)x, ldx, M, N, out);
    return check_launch("peneo_colsum");
  }
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + CS_ROWS - 1) / CS_ROWS));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, dtype, x, ldx, M, N, out);
  return check_launch("peneo_colsum");
}
