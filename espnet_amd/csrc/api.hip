// Library-level entry points of the C ABI (include/espnet_amd.h).
#include "em_common.h"

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
