// Library-level entry points of the C ABI (include/espnet_amd.h).
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "em_common.h"
#include "switches.h"

namespace {
EmSwitches read_switches() {
  auto on = [](const char* n) { return getenv(n) != nullptr; };
  auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
  EmSwitches s = {};
  s.attn2_stamps = on("EM_ATTN2_STAMPS"); s.block_stamps = on("EM_BLOCK_STAMPS"); s.ffn_stamps = on("EM_FFN_STAMPS");
  s.sub2_stamps = on("EM_SUB2_STAMPS");
  s.block_no_helpers = on("ESPNET_AMD_BLOCK_NO_HELPERS");
  s.sa_split = num("ESPNET_AMD_SA_SPLIT", 0); s.sa_group = num("ESPNET_AMD_SA_GROUP", 0);
  s.no_sa_tree = on("ESPNET_AMD_NO_SA_TREE"); s.sa_tree_min_rows = num("ESPNET_AMD_SA_TREE_MIN_ROWS", 200);
  s.no_attn2_large = on("ESPNET_AMD_NO_ATTN2_LARGE"); s.no_ffn_rows = on("ESPNET_AMD_NO_FFN_ROWS");
  s.no_rows_ctc = on("ESPNET_AMD_NO_ROWS_CTC"); s.frontend_v1 = on("ESPNET_AMD_FRONTEND_V1");
  s.gemm_stages = num("ESPNET_AMD_GEMM_STAGES", 0); s.gemm_bm = num("ESPNET_AMD_GEMM_BM", 0); s.no_mid_gemm = on("ESPNET_AMD_NO_MID_GEMM");
  s.mid_tile = num("ESPNET_AMD_MID_TILE", 0); s.lng_rt = num("ESPNET_AMD_LNG_RT", 0); s.lng_wide = num("ESPNET_AMD_LNG_WIDE", 256);
  s.no_src_lnq = on("ESPNET_AMD_NO_SRC_LNQ"); s.no_tail_fusion = on("ESPNET_AMD_NO_TAIL_FUSION");
  s.stream_no_fused = on("ESPNET_AMD_STREAM_NO_FUSED"); s.stream_mha_v1 = on("ESPNET_AMD_STREAM_MHA_V1");
  s.stream_no_ctx_fold = on("ESPNET_AMD_STREAM_NO_CTX_FOLD"); s.stream_no_ln_gemm = on("ESPNET_AMD_STREAM_NO_LN_GEMM");
  s.stream_fused_min = num("ESPNET_AMD_STREAM_FUSED_MIN", 1); s.no_sub12 = on("ESPNET_AMD_NO_SUB12");
  s.stream_ffn_split = num("ESPNET_AMD_STREAM_FFN_SPLIT", 0); s.stream_split_att = on("ESPNET_AMD_STREAM_SPLIT_ATT");
  s.dec_ffn_split = num("ESPNET_AMD_DEC_FFN_SPLIT", 0); s.dec_ffn_rows = num("ESPNET_AMD_DEC_FFN_ROWS", 0); s.no_rows_qkv = on("ESPNET_AMD_NO_ROWS_QKV"); s.ffn_rows_min_fill = num("ESPNET_AMD_FFN_ROWS_MIN_FILL", 0);
  return s;
}
std::atomic<const EmSwitches*> g_sw{nullptr};
std::mutex g_sw_mu;
}  // namespace

// (a reload publishes a NEW block and leaves the old one allocated: a launcher on another thread may still be reading it)
const EmSwitches& em_sw() {
  const EmSwitches* p = g_sw.load(std::memory_order_acquire);
  if (!p) {
    std::lock_guard<std::mutex> lk(g_sw_mu);
    p = g_sw.load(std::memory_order_acquire);
    if (!p) {
      p = new EmSwitches(read_switches());
      g_sw.store(p, std::memory_order_release);
    }
  }
  return *p;
}
extern "C" void em_dev_switches_reload(void) {
  std::lock_guard<std::mutex> lk(g_sw_mu);
  g_sw.store(new EmSwitches(read_switches()), std::memory_order_release);
}

extern "C" int em_version(void) { return 1; }

extern "C" const char* em_error_string(int code) {
  switch (code) {
    case EM_OK: return "ok";
    case EM_ERR_UNSUPPORTED: return "shape/config outside the MI355X fast path";
    case EM_ERR_BAD_ARG: return "bad argument";
    case EM_ERR_TOO_SHORT: return "utterance too short for Conv2dSubsampling (needs >= 7 frames)";
    case EM_ERR_LAUNCH: return "HIP kernel launch failed";
    case EM_ERR_WORKSPACE: return "workspace too small";
    case EM_ERR_IO: return "file I/O failed";
  }
  return "unknown error";
}

// f32 -> act dtype copy (host API convenience: lets reference-shaped entry points such as
// CTC.argmax(hs_pad f32) feed the act-dtype GEMMs).
template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ src, size_t n, T* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = from_f32<T>(src[i]);
}

extern "C" int em_cast_f32(int dtype, const float* src, size_t n, void* dst, void* stream) {
  if (!src || !dst) return EM_ERR_BAD_ARG;
  if (n == 0) return EM_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == EM_F32)
    hipLaunchKernelGGL(cast_f32_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                       n, (float*)dst);
  else if (dtype == EM_BF16)
    hipLaunchKernelGGL(cast_f32_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                       n, (bf16*)dst);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}
