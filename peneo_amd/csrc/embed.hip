// K1 / K2: embedding-table gathers (and their scatter-add backward), RoBERTa position ids,
// 16x16 patch im2col and the cls/pos assembly of the visual tokens.  All HBM-bound: one wave per
// token row, lanes along the hidden dim so every table row is read as contiguous 256-byte segments.
#include "common.h"

namespace peneo {

__device__ __forceinline__ int64_t row_off(int64_t r, int64_t rpb, int64_t bstride, int H) {
  return rpb > 0 ? (r / rpb) * bstride + (r % rpb) * (int64_t)H : r * (int64_t)H;
}

// ---- position ids: cumsum(ids != pad) * (ids != pad) + pad, one block per document ----------
__global__ __launch_bounds__(256) void position_ids_kernel(const int64_t* ids, int S, int64_t pad, int32_t* out) {
  __shared__ int part[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int per = (S + 255) / 256;
  const int s0 = t * per, s1 = min(S, s0 + per);
  const int64_t* row = ids + (int64_t)b * S;
  int cnt = 0;
  for (int s = s0; s < s1; ++s) cnt += (row[s] != pad);
  part[t] = cnt;
  __syncthreads();
  // inclusive Hillis-Steele scan over 256 partials
  for (int o = 1; o < 256; o <<= 1) {
    int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - cnt;
  for (int s = s0; s < s1; ++s) {
    bool keep = row[s] != pad;
    run += keep;
    out[(int64_t)b * S + s] = keep ? (int32_t)(run + pad) : (int32_t)pad;
  }
}

struct SpatialIdx { int l, t, r, b, h, w; bool ok; };
__device__ __forceinline__ SpatialIdx spatial_idx(const int64_t* box, int max_2d, int clip_hw) {
  SpatialIdx s;
  int64_t l = box[0], t = box[1], r = box[2], b = box[3];
  int64_t h = b - t, w = r - l;
  if (clip_hw) {
    h = h < 0 ? 0 : (h > max_2d - 1 ? max_2d - 1 : h);
    w = w < 0 ? 0 : (w > max_2d - 1 ? max_2d - 1 : w);
  }
  s.ok = l >= 0 && l < max_2d && t >= 0 && t < max_2d && r >= 0 && r < max_2d && b >= 0 && b < max_2d &&
         h >= 0 && h < max_2d && w >= 0 && w < max_2d;
  s.l = (int)l; s.t = (int)t; s.r = (int)r; s.b = (int)b; s.h = (int)h; s.w = (int)w;
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void embed_text_fwd_kernel(const int64_t* ids, const int32_t* pos_ids, const int64_t* bbox,
                                                             peneo_embed_tables tab, int64_t rows, int H, int clip_hw,
                                                             T* out, int64_t rpb, int64_t bstride, int32_t* status) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  T* o = out + row_off(r, rpb, bstride, H);
  const bool has_text = tab.word != nullptr, has_sp = tab.x != nullptr;
  int64_t id = 0; int pid = 0;
  bool ok = true;
  if (has_text) {
    id = ids[r]; pid = pos_ids[r];
    ok = id >= 0 && id < tab.vocab && pid >= 0 && pid < tab.max_pos;
  }
  SpatialIdx sp{};
  if (has_sp) { sp = spatial_idx(bbox + r * 4, tab.max_2d, clip_hw); ok = ok && sp.ok; }
  if (!ok) {
    if (lane == 0 && status) atomicExch(status, 1);
    for (int c = lane; c < H; c += 64) Elem<T>::store(o + c, 0.f);
    return;
  }
  const int cs = tab.coord_size, ss = tab.shape_size;
  for (int c = lane; c < H; c += 64) {
    float v = 0.f;
    if (has_text) v = tab.word[id * (int64_t)H + c] + tab.type0[c] + tab.pos[(int64_t)pid * H + c];
    if (has_sp) {
      float s;
      if (c < cs) s = tab.x[(int64_t)sp.l * cs + c];
      else if (c < 2 * cs) s = tab.y[(int64_t)sp.t * cs + (c - cs)];
      else if (c < 3 * cs) s = tab.x[(int64_t)sp.r * cs + (c - 2 * cs)];
      else if (c < 4 * cs) s = tab.y[(int64_t)sp.b * cs + (c - 3 * cs)];
      else if (c < 4 * cs + ss) s = tab.h[(int64_t)sp.h * ss + (c - 4 * cs)];
      else s = tab.w[(int64_t)sp.w * ss + (c - 4 * cs - ss)];
      v += s;
    }
    Elem<T>::store(o + c, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_text_bwd_kernel(const T* d_out, int64_t rpb, int64_t bstride,
                                                             const int64_t* ids, const int32_t* pos_ids, const int64_t* bbox,
                                                             peneo_embed_grads g, int cs, int ss, int max_2d, int64_t rows,
                                                             int H, int clip_hw, int64_t pad) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T* d = d_out + row_off(r, rpb, bstride, H);
  const bool has_text = g.word != nullptr, has_sp = g.x != nullptr;
  int64_t id = 0; int pid = 0;
  if (has_text) { id = ids[r]; pid = pos_ids[r]; }
  SpatialIdx sp{};
  if (has_sp) sp = spatial_idx(bbox + r * 4, max_2d, clip_hw);
  for (int c = lane; c < H; c += 64) {
    const float v = Elem<T>::load(d + c);
    if (has_text) {
      if (id != pad) atomicAdd(g.word + id * (int64_t)H + c, v);
      if (pid != pad) atomicAdd(g.pos + (int64_t)pid * H + c, v);
    }
    if (has_sp) {
      if (c < cs) atomicAdd(g.x + (int64_t)sp.l * cs + c, v);
      else if (c < 2 * cs) atomicAdd(g.y + (int64_t)sp.t * cs + (c - cs), v);
      else if (c < 3 * cs) atomicAdd(g.x + (int64_t)sp.r * cs + (c - 2 * cs), v);
      else if (c < 4 * cs) atomicAdd(g.y + (int64_t)sp.b * cs + (c - 3 * cs), v);
      else if (c < 4 * cs + ss) atomicAdd(g.h + (int64_t)sp.h * ss + (c - 4 * cs), v);
      else atomicAdd(g.w + (int64_t)sp.w * ss + (c - 4 * cs - ss), v);
    }
  }
}

// Box-table gradients with the duplicates of a token chunk summed in LDS first.  A line's box is replicated over its tokens and
// heights / widths take a handful of values, so in the token-per-wave kernel above thousands of fp32 atomics queue on the same few
// hundred addresses (140 of its 160 us at 8 x 512 tokens).  Here a workgroup takes EBX_TOK consecutive tokens: per index kind
// (left, top, right, bottom, height, width) a token's slot is the FIRST token of the chunk with the same index; the chunk's
// d_out rows are added into an LDS image [slot][H] (a thread owns its columns: no LDS atomics), and only first occurrences go to the
// table with global atomics.
constexpr int EBX_TOK = 16;   // tokens per workgroup: 256 workgroups at 8 x 512 tokens (32: 128 workgroups = half the CUs, 57 us)
template <typename T>
__global__ __launch_bounds__(256) void embed_box_bwd_kernel(const T* d_out, int64_t rpb, int64_t bstride, const int64_t* bbox,
                                                            peneo_embed_grads g, int cs, int ss, int max_2d, int64_t rows, int H,
                                                            int clip_hw) {
  extern __shared__ float ebx_acc[];                       // [EBX_TOK][H]
  __shared__ int idx[6][EBX_TOK];
  __shared__ int slot[6][EBX_TOK];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * EBX_TOK;
  if (tid < EBX_TOK) {
    const int64_t r = r0 + tid;
    SpatialIdx sp{};
    bool ok = false;
    if (r < rows) { sp = spatial_idx(bbox + r * 4, max_2d, clip_hw); ok = sp.ok; }
    idx[0][tid] = ok ? sp.l : -1 - tid; idx[1][tid] = ok ? sp.t : -1 - tid; idx[2][tid] = ok ? sp.r : -1 - tid;
    idx[3][tid] = ok ? sp.b : -1 - tid; idx[4][tid] = ok ? sp.h : -1 - tid; idx[5][tid] = ok ? sp.w : -1 - tid;
  }
  for (int i = tid; i < EBX_TOK * H; i += 256) ebx_acc[i] = 0.f;
  __syncthreads();
  if (tid < 6 * EBX_TOK) {
    const int k = tid / EBX_TOK, i = tid % EBX_TOK, me = idx[k][i];
    int s = i;
    for (int j = 0; j < i; ++j)
      if (idx[k][j] == me) { s = j; break; }
    slot[k][i] = s;
  }
  __syncthreads();
  // a thread owns its columns: every cell (slot, c) of the image is touched by one thread only - plain LDS read-modify-write
  const int ntok = (int)min((int64_t)EBX_TOK, rows - r0);
  for (int c = tid; c < H; c += 256) {
    const int k = c < 4 * cs ? c / cs : (c < 4 * cs + ss ? 4 : 5);
    const T* col = d_out + c;
    float* acc = ebx_acc + c;
#pragma unroll 1
    for (int i0 = 0; i0 < ntok; i0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (i0 + u < ntok) ? Elem<T>::load(col + row_off(r0 + i0 + u, rpb, bstride, H)) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < ntok && idx[0][i0 + u] >= 0) acc[slot[k][i0 + u] * H] += v[u];
    }
    float* tab = (k == 0 || k == 2) ? g.x : (k == 1 || k == 3) ? g.y : (k == 4 ? g.h : g.w);
    const int wdt = k < 4 ? cs : ss, cc = k < 4 ? c - k * cs : (k == 4 ? c - 4 * cs : c - 4 * cs - ss);
    for (int i = 0; i < ntok; ++i)
      if (slot[k][i] == i && idx[k][i] >= 0) atomicAdd(tab + (int64_t)idx[k][i] * wdt + cc, acc[i * H]);   // first occurrences only
  }
}

// ---- patches: [B, C, Hi, Wi] -> [B * gh * gw, C*256], k = c*256 + py*16 + px (conv weight order) ----
template <typename T>
__global__ void im2col_kernel(const float* img, int B, int Cc, int Hi, int Wi, T* out) {
  const int gh = Hi / 16, gw = Wi / 16, K = Cc * 256;
  const int64_t total = (int64_t)B * gh * gw * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(i % K);
    int64_t pr = i / K;
    int px = k & 15, py = (k >> 4) & 15, c = k >> 8;
    int gx = (int)(pr % gw), gy = (int)((pr / gw) % gh), b = (int)(pr / ((int64_t)gw * gh));
    float v = img[(((int64_t)b * Cc + c) * Hi + gy * 16 + py) * Wi + gx * 16 + px];
    Elem<T>::store(out + i, v);
  }
}

template <typename T>
__global__ void visual_assemble_fwd_kernel(const T* proj, const float* cls, const float* pos, int B, int np, int H, T* vis) {
  const int64_t total = (int64_t)B * (np + 1) * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % H);
    int t = (int)((i / H) % (np + 1));
    int b = (int)(i / ((int64_t)H * (np + 1)));
    float v = pos[(int64_t)t * H + c];
    v += (t == 0) ? cls[c] : Elem<T>::load(proj + ((int64_t)b * np + (t - 1)) * H + c);
    Elem<T>::store(vis + i, v);
  }
}

// one thread per (t, c): loops over the batch so d_pos / d_cls need no atomics
template <typename T>
__global__ void visual_assemble_bwd_kernel(const T* d_vis, int B, int np, int H, T* d_proj, float* d_cls, float* d_pos) {
  const int64_t total = (int64_t)(np + 1) * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % H);
    int t = (int)(i / H);
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      T raw = d_vis[((int64_t)b * (np + 1) + t) * H + c];
      s += Elem<T>::load(&raw);
      if (t > 0) d_proj[((int64_t)b * np + (t - 1)) * H + c] = raw;
    }
    if (d_pos) d_pos[i] += s;
    if (t == 0 && d_cls) d_cls[c] += s;
  }
}

}  // namespace peneo
using namespace peneo;

static inline bool ok_dt(int d) { return d == PENEO_F32 || d == PENEO_BF16; }
static inline unsigned cap_blocks(int64_t n, int per_block = 256, int64_t cap = 8192) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int peneo_position_ids(const int64_t* input_ids, int B, int S, int64_t pad_id, int32_t* pos_ids,
                                  peneo_stream_t stream) {
  PENEO_REQUIRE(input_ids && pos_ids && B > 0 && S > 0, "peneo_position_ids: bad arguments");
  hipLaunchKernelGGL(position_ids_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, input_ids, S, pad_id, pos_ids);
  return check_launch("peneo_position_ids");
}

extern "C" int peneo_embed_text_fwd(int dtype, const int64_t* input_ids, const int32_t* pos_ids, const int64_t* bbox,
                                    const peneo_embed_tables* tab, int B, int S, int H, int clip_hw, void* out,
                                    int64_t out_rpb, int64_t out_bstride, int32_t* status, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && tab && out && B > 0 && S > 0 && H > 0, "peneo_embed_text_fwd: bad arguments");
  PENEO_REQUIRE(tab->word || tab->x, "peneo_embed_text_fwd: neither text nor spatial tables given");
  if (tab->word) PENEO_REQUIRE(input_ids && pos_ids && tab->type0 && tab->pos, "peneo_embed_text_fwd: text tables incomplete");
  if (tab->x) {
    PENEO_REQUIRE(bbox && tab->y && tab->h && tab->w, "peneo_embed_text_fwd: spatial tables incomplete");
    PENEO_REQUIRE(4 * tab->coord_size + 2 * tab->shape_size == H, "peneo_embed_text_fwd: 4*coord+2*shape != H");
  }
  int64_t rows = (int64_t)B * S;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(embed_text_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, input_ids, pos_ids, bbox, *tab,
                       rows, H, clip_hw, (bf16_t*)out, out_rpb, out_bstride, status);
  else
    hipLaunchKernelGGL(embed_text_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, input_ids, pos_ids, bbox, *tab,
                       rows, H, clip_hw, (float*)out, out_rpb, out_bstride, status);
  return check_launch("peneo_embed_text_fwd");
}

extern "C" int peneo_embed_text_bwd(int dtype, const void* d_out, int64_t rpb, int64_t bstride, const int64_t* input_ids,
                                    const int32_t* pos_ids, const int64_t* bbox, const peneo_embed_grads* g,
                                    int coord_size, int shape_size, int max_2d, int B, int S, int H, int clip_hw,
                                    int64_t pad_id, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && d_out && g && B > 0 && S > 0 && H > 0, "peneo_embed_text_bwd: bad arguments");
  if (g->word) PENEO_REQUIRE(input_ids && pos_ids && g->pos, "peneo_embed_text_bwd: text grads incomplete");
  if (g->x) PENEO_REQUIRE(bbox && g->y && g->h && g->w, "peneo_embed_text_bwd: spatial grads incomplete");
  int64_t rows = (int64_t)B * S;
  dim3 grid((unsigned)((rows + 3) / 4));
  peneo_embed_grads rest = *g;
  const size_t box_lds = (size_t)EBX_TOK * H * sizeof(float);
  if (g->x && H == 4 * coord_size + 2 * shape_size && coord_size > 0 && shape_size > 0 && box_lds <= 150 * 1024 && rows >= 4 * EBX_TOK) {
    // the box tables: duplicates of 32-token chunks summed in LDS, first occurrences to the tables (embed_box_bwd_kernel)
    dim3 bgrid((unsigned)((rows + EBX_TOK - 1) / EBX_TOK));
#define PENEO_EBX_LAUNCH(T_)                                                                                                      \
  {                                                                                                                               \
    if (box_lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(embed_box_bwd_kernel<T_>),                       \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)box_lds) != hipSuccess) {     \
      set_error("peneo_embed_text_bwd: cannot raise dynamic LDS to %zu bytes", box_lds);                                          \
      return PENEO_ERR_LAUNCH;                                                                                                    \
    }                                                                                                                             \
    hipLaunchKernelGGL(embed_box_bwd_kernel<T_>, bgrid, dim3(256), box_lds, (hipStream_t)stream, (const T_*)d_out, rpb, bstride,  \
                       bbox, *g, coord_size, shape_size, max_2d, rows, H, clip_hw);                                               \
  }
    if (dtype == PENEO_BF16) PENEO_EBX_LAUNCH(bf16_t) else PENEO_EBX_LAUNCH(float)
#undef PENEO_EBX_LAUNCH
    int rc = check_launch("peneo_embed_text_bwd (box tables)");
    if (rc) return rc;
    rest.x = rest.y = rest.h = rest.w = nullptr;
    if (!rest.word) return PENEO_OK;
  }
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(embed_text_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_out, rpb,
                       bstride, input_ids, pos_ids, bbox, rest, coord_size, shape_size, max_2d, rows, H, clip_hw, pad_id);
  else
    hipLaunchKernelGGL(embed_text_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_out, rpb,
                       bstride, input_ids, pos_ids, bbox, rest, coord_size, shape_size, max_2d, rows, H, clip_hw, pad_id);
  return check_launch("peneo_embed_text_bwd");
}

extern "C" int peneo_im2col_patch16(int dtype, const float* image, int B, int C, int Hi, int Wi, void* patches,
                                    peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && image && patches && B > 0 && C > 0, "peneo_im2col_patch16: bad arguments");
  PENEO_REQUIRE(Hi % 16 == 0 && Wi % 16 == 0 && Hi > 0 && Wi > 0, "peneo_im2col_patch16: image size must be a multiple of 16");
  int64_t total = (int64_t)B * C * Hi * Wi;
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream, image, B, C, Hi, Wi,
                       (bf16_t*)patches);
  else
    hipLaunchKernelGGL(im2col_kernel<float>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream, image, B, C, Hi, Wi,
                       (float*)patches);
  return check_launch("peneo_im2col_patch16");
}

extern "C" int peneo_visual_assemble_fwd(int dtype, const void* proj, const float* cls, const float* pos, int B, int np,
                                         int H, void* vis, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && proj && cls && pos && vis && B > 0 && np > 0 && H > 0, "peneo_visual_assemble_fwd: bad arguments");
  int64_t total = (int64_t)B * (np + 1) * H;
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(visual_assemble_fwd_kernel<bf16_t>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)proj, cls, pos, B, np, H, (bf16_t*)vis);
  else
    hipLaunchKernelGGL(visual_assemble_fwd_kernel<float>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)proj, cls, pos, B, np, H, (float*)vis);
  return check_launch("peneo_visual_assemble_fwd");
}

extern "C" int peneo_visual_assemble_bwd(int dtype, const void* d_vis, int B, int np, int H, void* d_proj, float* d_cls,
                                         float* d_pos, peneo_stream_t stream) {
  PENEO_REQUIRE(ok_dt(dtype) && d_vis && d_proj && B > 0 && np > 0 && H > 0, "peneo_visual_assemble_bwd: bad arguments");
  int64_t total = (int64_t)(np + 1) * H;
  if (dtype == PENEO_BF16)
    hipLaunchKernelGGL(visual_assemble_bwd_kernel<bf16_t>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)d_vis, B, np, H, (bf16_t*)d_proj, d_cls, d_pos);
  else
    hipLaunchKernelGGL(visual_assemble_bwd_kernel<float>, dim3(cap_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)d_vis, B, np, H, (float*)d_proj, d_cls, d_pos);
  return check_launch("peneo_visual_assemble_bwd");
}
