// Shared between attention.hip and attn_bwd_pipe.hip: the parameter block of the attention kernels.
#pragma once
#include "common.h"

namespace peneo {

struct AttnParams {
  const void* q; const void* k; const void* v; int64_t ld;
  const void* vt; const void* kt; const void* qt; const void* dot;  // [B, nh, DP, Tp] transposed copies
  int B, nh, T, d, Tp; float scale;
  const void* bias; int64_t bias_ld; const float* key_bias;       // bias [B, nh, T, bias_ld]; key_bias [B, Tp]
  void* out; int64_t ld_out; float* lse;
  float drop_p, keep_scale; const uint32_t* words; int nqb, Tk;   // dropout keep bits (peneo_attn_drop_words) or NULL
  const void* d_out; void* dq; void* dk; void* dv; int64_t ld_d; float* g_bias; float* delta;
  void* ds_out;   // single-pass backward only: bf16 dS^T [B, nh, T keys, Tp queries] of this layer (or NULL)
};

// slot of a key's dropout keep word inside its 32-key block (see the note on the keep words in attention.hip): slots 2r and 2r + 1
// hold the keys of forward accumulator register r for the two half-waves
__host__ __device__ __forceinline__ int attn_kslot(int key) {
  return (key & ~31) | (((key >> 3) & 3) << 3) | ((key & 3) << 1) | ((key >> 2) & 1);
}

// attn_bwd_pipe.hip: the pipelined single-pass backward (bf16, head dim 64, bias tensor, dS^T slab requested).
// attn_bwd_pipe_supported says whether a call qualifies; the launch writes dK, dV and the dS^T slab, then dQ from the slab
// (returns 0; or 1 when the caller still has to launch attn_dq_from_ds_kernel: PENEO_ATTN_DQ_PIPE=0).
bool attn_bwd_pipe_supported(const AttnParams& p);
int launch_attn_bwd_pipe(const AttnParams& p, hipStream_t st);
// attn_fwd_pipe.hip: the pipelined forward (bf16, head dim 64, bias tensor, V row-major), bit-identical to attn_fwd_kernel
bool attn_fwd_pipe_supported(const AttnParams& p);
int launch_attn_fwd_pipe(const AttnParams& p, hipStream_t st);

}  // namespace peneo
