// CTC head kernels: per-frame argmax / log-softmax over the vocabulary and the greedy "G1"
// collapse.  Reference: espnet2/asr/ctc.py:197-215 (log_softmax, argmax) and
// espnet2/bin/asr_inference.py:574-575 (`groupby` + drop blank / sos / eos).
#include "em_common.h"

namespace {

// One wave per row; ties resolve to the lowest index (torch.argmax on CPU returns the first
// maximal element).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int M, int V,
                                                          int* __restrict__ ids) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    float v = xr[c];
    if (v > best) {  // strictly greater keeps the earliest index within a lane
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) ids[row] = bi;
}

// Second half of the GEMM-fused arg-max: G (value, column) partials per row -> best column.
__global__ __launch_bounds__(256) void argmax_partials_kernel(const float2* __restrict__ part, int M,
                                                              int G, int* __restrict__ ids) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int g = lane; g < G; g += 64) {
    const float2 p = part[(size_t)row * G + g];
    const int c = __float_as_int(p.y);
    if (p.x > best || (p.x == best && c < bi)) {
      best = p.x;
      bi = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) ids[row] = bi;
}

// In-place log-softmax, one wave per row (two passes over an L2-resident row).
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(float* __restrict__ x, int M,
                                                               int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* xr = x + (size_t)row * V;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += expf(xr[c] - mx);
  s = wave_sum(s);
  const float lse = mx + logf(s);
  for (int c = lane; c < V; c += 64) xr[c] = xr[c] - lse;
}

// One wave per utterance: keep[t] = t < olen && ids[t] != ids[t-1] && ids[t] not in
// {blank, sos/eos}; compaction by ballot + popcount prefix.
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int* __restrict__ ids,
                                                          const int* __restrict__ olens, int Tn,
                                                          int blank, int sos_eos,
                                                          int* __restrict__ tokens,
                                                          int* __restrict__ out_lens) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int* row = ids + (size_t)b * Tn;
  int* out = tokens + (size_t)b * Tn;
  const int len = olens[b] < Tn ? olens[b] : Tn;
  int count = 0;
  for (int t0 = 0; t0 < len; t0 += 64) {
    int t = t0 + lane;
    int cur = t < len ? row[t] : -1;
    int prev = (t > 0 && t < len) ? row[t - 1] : -1;
    bool keep = t < len && cur != prev && cur != blank && cur != sos_eos;
    unsigned long long m = __ballot(keep);
    int pos = count + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) out[pos] = cur;
    count += __popcll(m);
  }
  for (int t = count + lane; t < Tn; t += 64) out[t] = -1;
  if (lane == 0) out_lens[b] = count;
}

}  // namespace

extern "C" int em_argmax_rows_f32(const float* logits, int32_t M, int32_t V, int32_t* ids,
                                  void* stream) {
  if (M <= 0 || V <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(em_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, M, V, ids);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_argmax_partials(const float* part, int32_t M, int32_t G, int32_t* ids,
                                  void* stream) {
  if (M <= 0 || G <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_partials_kernel, dim3(em_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)part, M, G, ids);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_log_softmax_rows_f32(float* logits, int32_t M, int32_t V, void* stream) {
  if (M <= 0 || V <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(em_cdiv(M, 4)), dim3(256), 0,
                     (hipStream_t)stream, logits, M, V);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_ctc_collapse(const int32_t* ids, const int32_t* olens, int32_t B, int32_t T,
                               int32_t blank, int32_t sos_eos, int32_t* tokens, int32_t* out_lens,
                               void* stream) {
  if (B <= 0 || T <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ids, olens, T,
                     blank, sos_eos, tokens, out_lens);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
