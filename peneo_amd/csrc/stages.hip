// Composite entry points: ONE C call enqueues every kernel of an encoder layer's forward or backward (bf16 path).
//
// Why: a training step is ~420 kernel launches; issued one by one through Python + ctypes the host needs ~12 ms per step
// against ~17.5 ms of device time (forward: 3.1-3.6 ms against 4.7 ms), so any further kernel work on the forward would make it
// host-bound.  The boundary contract already has the caller own every buffer, workspace and stream, so a whole
// stage is a pure function of a struct of device pointers: these functions are the per-kernel entry points of
// include/peneo_hip.h called back to back from C++, with the cross-stream ordering of the backward (weight-gradient work on a
// second stream behind HIP events) expressed in HIP directly.  The per-kernel entry points stay (tests, fp32 parity path).
//
// Reference: the layer is LayoutLMv3Layer.forward (modeling_layoutlmv3.py:482-529: self-attention :335-404, RobertaSelfOutput,
// RobertaIntermediate, RobertaOutput) and its autograd.
#include <cstdlib>
#include <mutex>
#include "common.h"

using namespace peneo;

namespace {

// choose_split_k of peneo_amd/ops.py: split the reduction only when the output grid cannot fill the 256 CUs
int auto_split(int M, int N, int K) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int kt = (K + 63) / 64;
  if (tiles >= 192 || kt < 8) return 1;
  int s = kt / 4;
  if (512 / tiles < s) s = 512 / tiles;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}
size_t ws_bytes(int M, int N, int K) { return peneo_gemm_workspace_bytes(M, N, K, auto_split(M, N, K)); }
size_t max2(size_t a, size_t b) { return a > b ? a : b; }

struct Ws { void* p; size_t n; };
int gemm(int ak, int bk, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
         int c_dtype, const peneo_gemm_epilogue& ep, Ws ws, hipStream_t st) {
  const int split = auto_split(M, N, K);
  return peneo_gemm(PENEO_BF16, ak, bk, M, N, K, A, lda, B, ldb, C, ldc, c_dtype, &ep, split, split > 1 ? ws.p : nullptr,
                    split > 1 ? ws.n : 0, st);
}

// two reusable events per (host thread, device) for the main -> side hand-offs of a backward (a wait captures the record that
// precedes it, so re-recording an event for the next layer does not disturb waits already enqueued).  Per THREAD: two host threads
// running backwards on one device with different stream pairs must not interleave record / wait on one event (round-4 review); a
// thread's two events per device are NOT destroyed when it exits: backwards run on autograd's worker threads, whose
// thread-locals are torn down at process exit, possibly after the HIP runtime has begun unloading -- two events per thread are a
// negligible leak, a hipEventDestroy into a half-unloaded runtime is not.
struct DevEvents { hipEvent_t e[2]; bool ok; };
struct ThreadEvents {
  DevEvents tab[64] = {};
};
DevEvents* events_of_current_device() {
  static thread_local ThreadEvents mine;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  DevEvents& d = mine.tab[dev];
  if (!d.ok) {
    if (hipEventCreateWithFlags(&d.e[0], hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&d.e[1], hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(d.e[0]); return nullptr; }
    d.ok = true;
  }
  return &d;
}

#define STAGE_TRY(call)                  \
  {                                      \
    const int rc_ = (call);              \
    if (rc_ != PENEO_OK) return rc_;     \
  }

bool layer_ok(const peneo_encoder_layer* L) {
  return L && L->B > 0 && L->T > 0 && L->H > 0 && L->nh > 0 && L->I > 0 && L->H % L->nh == 0 && L->H % 8 == 0 && L->I % 8 == 0 &&
         L->Wqkv && L->bqkv && L->Wo && L->bo && L->g1 && L->b1 && L->Wi && L->bi && L->Wo2 && L->bo2 && L->g2 && L->b2 && L->x && L->qkv &&
         L->att && L->lse && L->h1 && L->m1 && L->r1 && L->a && L->inter && L->h2 && L->m2 && L->r2;
}

}  // namespace

extern "C" size_t peneo_struct_bytes(int which) {
  return which == 0 ? sizeof(peneo_gemm_epilogue) : which == 1 ? sizeof(peneo_encoder_layer) : which == 2 ? sizeof(peneo_encoder_layer_grads) : 0;
}

extern "C" size_t peneo_encoder_layer_workspace_bytes(int rows, int H, int I, int which) {
  // which = 0: forward GEMMs (main stream); 1: backward dgrad GEMMs (main stream); 2: backward weight gradients (side stream)
  if (which == 0) return max2(max2(ws_bytes(rows, 3 * H, H), ws_bytes(rows, H, H)), max2(ws_bytes(rows, I, H), ws_bytes(rows, H, I)));
  if (which == 1) return max2(max2(ws_bytes(rows, I, H), ws_bytes(rows, H, I)), max2(ws_bytes(rows, H, H), ws_bytes(rows, H, 3 * H)));
  return max2(max2(ws_bytes(H, I, rows), ws_bytes(I, H, rows)), max2(ws_bytes(H, H, rows), ws_bytes(3 * H, H, rows)));
}

extern "C" int peneo_encoder_layer_fwd(const peneo_encoder_layer* L, void* out, void* workspace, size_t workspace_bytes,
                                       peneo_stream_t stream) {
  PENEO_REQUIRE(layer_ok(L) && out, "peneo_encoder_layer_fwd: incomplete layer description");
  PENEO_REQUIRE(workspace_bytes >= peneo_encoder_layer_workspace_bytes(L->B * L->T, L->H, L->I, 0) && (workspace || workspace_bytes == 0),
                "peneo_encoder_layer_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int R = L->B * L->T, H = L->H, I = L->I, d = H / L->nh;
  const Ws ws{workspace, workspace_bytes};
  const char* qkv = reinterpret_cast<const char*>(L->qkv);
  peneo_gemm_epilogue ep = {};
  // K5: fused QKV projection
  ep.bias = L->bqkv;
  STAGE_TRY(gemm(1, 1, R, 3 * H, H, L->x, H, L->Wqkv, H, L->qkv, 3 * H, PENEO_BF16, ep, ws, st));
  // K6: attention over the fused buffer (V read in place)
  STAGE_TRY(peneo_attn_fwd(PENEO_BF16, qkv, qkv + 2 * (size_t)H, qkv + 4 * (size_t)H, 3 * H, nullptr, L->B, L->nh, L->T, d, L->attn_scale,
                           L->bias, L->bias_ld, L->key_bias, L->att, H, L->lse, L->p_attn, L->p_attn > 0.f ? L->drop_words : nullptr, st));
  // K7: output projection + dropout + residual, LayerNorm
  ep = {};
  ep.bias = L->bo; ep.residual = L->x; ep.ld_res = H; ep.drop_p = L->p_hidden; ep.drop_seed = L->seed_o;
  STAGE_TRY(gemm(1, 1, R, H, H, L->att, H, L->Wo, H, L->h1, H, PENEO_BF16, ep, ws, st));
  STAGE_TRY(peneo_layernorm_fwd(PENEO_BF16, L->h1, 0, 0, L->a, 0, 0, L->g1, L->b1, L->eps, L->m1, L->r1, R, H, 0.f, 0u, st));
  // K8: intermediate (GELU; the pre-activation is stored only when a backward will need it), output + dropout + residual, LayerNorm
  ep = {};
  ep.bias = L->bi; ep.act = PENEO_ACT_GELU; ep.preact = L->zi; ep.ld_preact = I;
  STAGE_TRY(gemm(1, 1, R, I, H, L->a, H, L->Wi, H, L->inter, I, PENEO_BF16, ep, ws, st));
  ep = {};
  ep.bias = L->bo2; ep.residual = L->a; ep.ld_res = H; ep.drop_p = L->p_hidden; ep.drop_seed = L->seed_o2;
  STAGE_TRY(gemm(1, 1, R, H, I, L->inter, I, L->Wo2, I, L->h2, H, PENEO_BF16, ep, ws, st));
  STAGE_TRY(peneo_layernorm_fwd(PENEO_BF16, L->h2, 0, 0, out, 0, 0, L->g2, L->b2, L->eps, L->m2, L->r2, R, H, 0.f, 0u, st));
  return PENEO_OK;
}

extern "C" int peneo_encoder_layer_bwd(const peneo_encoder_layer* L, const peneo_encoder_layer_grads* G, void* ws_main,
                                       size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, peneo_stream_t stream,
                                       peneo_stream_t side_stream) {
  PENEO_REQUIRE(layer_ok(L) && L->zi && G, "peneo_encoder_layer_bwd: incomplete layer description");
  PENEO_REQUIRE(G->d_out && G->d_x && G->d_h2 && G->d_zi && G->d_a && G->d_h1 && G->d_att && G->dqkv && G->delta && G->ds_out &&
                    G->dwqkv && G->dbqkv && G->dwo && G->dbo && G->dg1 && G->db1 && G->dwi && G->dbi && G->dwo2 && G->dbo2 && G->dg2 && G->db2,
                "peneo_encoder_layer_bwd: incomplete gradient description");
  PENEO_REQUIRE(L->p_hidden == 0.f || (G->d_dense2 && G->d_dense1), "peneo_encoder_layer_bwd: dropout needs d_dense1 / d_dense2");
  const int R = L->B * L->T, H = L->H, I = L->I, d = H / L->nh;
  PENEO_REQUIRE(ws_main_bytes >= peneo_encoder_layer_workspace_bytes(R, H, I, 1) && ws_side_bytes >= peneo_encoder_layer_workspace_bytes(R, H, I, 2),
                "peneo_encoder_layer_bwd: workspace too small");
  hipStream_t main = (hipStream_t)stream, side = side_stream ? (hipStream_t)side_stream : main;
  DevEvents* ev = nullptr;
  if (side != main) {
    ev = events_of_current_device();
    PENEO_REQUIRE(ev, "peneo_encoder_layer_bwd: cannot create HIP events");
  }
  const Ws wm{ws_main, ws_main_bytes}, wsd{ws_side, ws_side_bytes};
  const void* d_dense2 = L->p_hidden > 0.f ? G->d_dense2 : G->d_h2;
  const void* d_dense1 = L->p_hidden > 0.f ? G->d_dense1 : G->d_h1;
  peneo_gemm_epilogue ep = {};
  // LayerNorm 2 backward: d_h2 (residual branch) and d_h2 through the dropout mask of the FFN2 output (its dgrad / wgrad input)
  STAGE_TRY(peneo_layernorm_bwd(PENEO_BF16, G->d_out, 0, 0, L->h2, 0, 0, G->d_h2, 0, 0, L->g2, L->m2, L->r2, G->dg2, G->db2, R, H, 0.f, 0u,
                                L->p_hidden > 0.f ? G->d_dense2 : nullptr, L->p_hidden, L->seed_o2, nullptr, main));
  // d_zi = (d_dense2 Wo2) * GELU'(zi);  d_a = d_zi Wi + d_h2
  ep.grad_src = L->zi; ep.ld_grad = I; ep.grad_act = PENEO_ACT_GELU;
  STAGE_TRY(gemm(1, 0, R, I, H, d_dense2, H, L->Wo2, I, G->d_zi, I, PENEO_BF16, ep, wm, main));
  ep = {};
  ep.residual = G->d_h2; ep.ld_res = H;
  STAGE_TRY(gemm(1, 0, R, H, I, G->d_zi, I, L->Wi, H, G->d_a, H, PENEO_BF16, ep, wm, main));
  STAGE_TRY(peneo_layernorm_bwd(PENEO_BF16, G->d_a, 0, 0, L->h1, 0, 0, G->d_h1, 0, 0, L->g1, L->m1, L->r1, G->dg1, G->db1, R, H, 0.f, 0u,
                                L->p_hidden > 0.f ? G->d_dense1 : nullptr, L->p_hidden, L->seed_o, nullptr, main));
  ep = {};
  STAGE_TRY(gemm(1, 0, R, H, H, d_dense1, H, L->Wo, H, G->d_att, H, PENEO_BF16, ep, wm, main));
  // The FFN / output-projection weight gradients and bias column sums start with the attention backward (its second, thin
  // round of workgroups leaves most CUs idle), on the side stream
  if (ev) {
    if (hipEventRecord(ev->e[0], main) != hipSuccess || hipStreamWaitEvent(side, ev->e[0], 0) != hipSuccess) return check_launch("peneo_encoder_layer_bwd (event)");
  }
  ep = {};
  // Round 5: with a side stream the four weight gradients of the layer are ONE grouped launch behind the attention backward
  // (peneo_gemm_group: 432 tiles with the full K = rows, no split-k partials, no reduce launches; 97 us alone against 160 + 40 for
  // the four GEMMs and their reductions) - measured 0.15-0.2 ms per step ahead of the per-GEMM launches, which started three of
  // them beside the attention backward (3 of 3 interleaved pairs, profiles/r05_ab_runs.txt; round 2 had measured the grouped
  // launch 0.3 ms BEHIND: the attention kernels it ran beside were latency-bound then and are bound by memory and issue now).
  // Serial mode (no side stream: profiling) and PENEO_WGRAD_GROUP=0 keep the per-GEMM launches.
  static const bool group_on = [] { const char* e = getenv("PENEO_WGRAD_GROUP"); return !e || atoi(e) != 0; }();
  const bool grouped = group_on && side != main;
  STAGE_TRY(peneo_colsum(PENEO_BF16, d_dense2, H, R, H, G->dbo2, 1, side));
  if (!grouped) STAGE_TRY(gemm(0, 0, H, I, R, d_dense2, H, L->inter, I, G->dwo2, I, PENEO_F32, ep, wsd, side));
  STAGE_TRY(peneo_colsum(PENEO_BF16, G->d_zi, I, R, I, G->dbi, 1, side));
  if (!grouped) STAGE_TRY(gemm(0, 0, I, H, R, G->d_zi, I, L->a, H, G->dwi, H, PENEO_F32, ep, wsd, side));
  STAGE_TRY(peneo_colsum(PENEO_BF16, d_dense1, H, R, H, G->dbo, 1, side));
  if (!grouped) STAGE_TRY(gemm(0, 0, H, H, R, d_dense1, H, L->att, H, G->dwo, H, PENEO_F32, ep, wsd, side));
  // attention backward (single pass: dS^T slab of this layer, dQ from the slab)
  const char* qkv = reinterpret_cast<const char*>(L->qkv);
  char* dqkv = reinterpret_cast<char*>(G->dqkv);
  STAGE_TRY(peneo_attn_bwd(PENEO_BF16, qkv, qkv + 2 * (size_t)H, qkv + 4 * (size_t)H, 3 * H, nullptr, nullptr, nullptr, L->att, G->d_att, H, L->lse,
                           L->B, L->nh, L->T, d, L->attn_scale, L->bias, L->bias_ld, L->key_bias, dqkv, dqkv + 2 * (size_t)H, dqkv + 4 * (size_t)H,
                           3 * H, nullptr, G->delta, nullptr, G->ds_out, L->p_attn, L->p_attn > 0.f ? L->drop_words : nullptr, main));
  if (ev) {
    if (hipEventRecord(ev->e[1], main) != hipSuccess || hipStreamWaitEvent(side, ev->e[1], 0) != hipSuccess) return check_launch("peneo_encoder_layer_bwd (event)");
  }
  STAGE_TRY(peneo_colsum(PENEO_BF16, G->dqkv, 3 * H, R, 3 * H, G->dbqkv, 1, side));
  if (!grouped) {
    STAGE_TRY(gemm(0, 0, 3 * H, H, R, G->dqkv, 3 * H, L->x, H, G->dwqkv, H, PENEO_F32, ep, wsd, side));
  } else {
    peneo_gemm_problem pr[4] = {};
    pr[0] = {3 * H, H, R, G->dqkv, 3 * H, L->x, H, G->dwqkv, H, 0, nullptr};
    pr[1] = {I, H, R, G->d_zi, I, L->a, H, G->dwi, H, 0, nullptr};
    pr[2] = {H, I, R, d_dense2, H, L->inter, I, G->dwo2, I, 0, nullptr};
    pr[3] = {H, H, R, d_dense1, H, L->att, H, G->dwo, H, 0, nullptr};
    STAGE_TRY(peneo_gemm_group(PENEO_BF16, 0, 0, PENEO_F32, pr, 4, side));
  }
  // d_x = dqkv Wqkv + d_h1
  ep = {};
  ep.residual = G->d_h1; ep.ld_res = H;
  STAGE_TRY(gemm(1, 0, R, H, 3 * H, G->dqkv, 3 * H, L->Wqkv, H, G->d_x, H, PENEO_BF16, ep, wm, main));
  return PENEO_OK;
}
