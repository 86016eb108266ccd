// Split-K GEMM for the residual projections of the beam-search decoder step:
//   C[f32][M][N] += scale * (A[M][K] . W[N][K]^T + bias),   48 < M <= 192 rows (M = B * beam), N % 64 == 0.
//
// Why.  At M = 160 the tiled kernel (csrc/gemm.hip) has 12 workgroups for N = 512 and walks K sequentially: the
// decoder's output projections (K = 512) and second FFN matrices (K = 2048) took 7-17 us each, 18 of them per
// label step (profiles/r02f_bench_beam_large_b16.md: 10.6 us average, 30 % of the step).  These launches are pure
// latency, so the work is spread over (N / 64) x (K / 128) workgroups that each make ONE pass:
//   * every wave loads its A / W fragments straight from global memory into MFMA operand registers (16 bytes
//     per lane, no LDS staging: within a workgroup nothing is reused across waves except through L1) and owns up to
//     three 16-row tiles x 64 columns;
//   * the f32 partial tile goes to a per-stream scratch buffer; the workgroup that arrives LAST at the column
//     tile's ticket counter adds the K-slices in slice order (a fixed order: the result does not depend on which
//     workgroup happens to be last, so the search stays bit-reproducible), applies bias / scale / residual and
//     re-arms the counter -- which also makes the kernel safe to replay from a captured hipGraph.
#include <mutex>
#include <unordered_map>

#include "em_common.h"

namespace {

constexpr int SK_BN = 64;      // columns per workgroup
constexpr int SK_KS = 128;     // K slice per workgroup
constexpr int SK_MAXM = 192;   // 12 row tiles: 3 per wave
constexpr int SK_MAXN = 2048;
constexpr int SK_MAXS = 32;    // K <= 4096

template <typename T>
__global__ __launch_bounds__(256) void gemm_splitk_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          float* __restrict__ C, const float* __restrict__ bias,
                                                          int M, int N, int K, int lda, int ldc, float scale,
                                                          float* __restrict__ part, unsigned* __restrict__ tickets) {
  using MM = Mma<T>;
  constexpr int KSTEPS = SK_KS / MM::K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * SK_BN, s = blockIdx.y, S = gridDim.y;
  const int k0 = s * SK_KS;
  const int mt = (M + 15) >> 4;
  f32x4 acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // operand rows of this lane (clamped: tail rows recompute row M-1 and are never stored)
  const T* ap[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int m = (wave + 4 * i) * 16 + lr;
    m = m < M ? m : M - 1;
    ap[i] = A + (size_t)m * lda + k0 + lg * MM::EPL;
  }
  const T* wp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wp[j] = W + (size_t)(n0 + j * 16 + lr) * K + k0 + lg * MM::EPL;
#pragma unroll 4
  for (int ks = 0; ks < KSTEPS; ++ks) {
    typename MM::frag a[3], w[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = MM::load(ap[i] + ks * MM::K);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = MM::load(wp[j] + ks * MM::K);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (wave + 4 * i < mt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(a[i], w[j], acc[i][j]);
      }
  }
  // partial tile -> scratch [s][m][n]  (C/D layout: row = lg*4 + r, col = lr)
  float* pp = part + (size_t)s * SK_MAXM * N;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (wave + 4 * i >= mt) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = (wave + 4 * i) * 16 + lg * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) pp[(size_t)m * N + n0 + j * 16 + lr] = acc[i][j][r];
    }
  }
  __threadfence();  // this workgroup's partials are visible device-wide before its ticket is
  __syncthreads();
  __shared__ unsigned last;
  if (tid == 0) last = atomicAdd(&tickets[blockIdx.x], 1u);
  __syncthreads();
  if (last != (unsigned)(S - 1)) return;
  __threadfence();  // acquire: every other slice's partials
  // the last workgroup of this column tile reduces the slices in slice order
  for (int e = tid; e < M * (SK_BN / 4); e += 256) {
    const int m = e / (SK_BN / 4), c4 = (e - m * (SK_BN / 4)) * 4;
    float4 sum = *(const float4*)(part + (size_t)m * N + n0 + c4);
    for (int q = 1; q < S; ++q) {
      const float4 v = *(const float4*)(part + ((size_t)q * SK_MAXM + m) * N + n0 + c4);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    float4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = *(const float4*)(bias + n0 + c4);
    float4* cp = (float4*)(C + (size_t)m * ldc + n0 + c4);
    float4 c = *cp;
    c.x += scale * (sum.x + b4.x);
    c.y += scale * (sum.y + b4.y);
    c.z += scale * (sum.z + b4.z);
    c.w += scale * (sum.w + b4.w);
    *cp = c;
  }
  if (tid == 0) tickets[blockIdx.x] = 0;  // re-armed for the next launch on this stream
}

struct Scratch {
  float* part = nullptr;
  unsigned* tickets = nullptr;
};
std::mutex g_mu;
std::unordered_map<void*, Scratch> g_scratch;  // one per stream: launches on different streams may overlap

// nullptr while the stream is being captured and nothing has been allocated for it yet (hipMalloc is illegal
// inside a capture): the caller then keeps the tiled kernel
Scratch* scratch_for(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_scratch.find((void*)s);
  if (it != g_scratch.end()) return &it->second;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
  Scratch sc;
  if (hipMalloc((void**)&sc.part, (size_t)SK_MAXS * SK_MAXM * SK_MAXN * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMalloc((void**)&sc.tickets, (SK_MAXN / SK_BN) * sizeof(unsigned)) != hipSuccess) return nullptr;
  if (hipMemset(sc.tickets, 0, (SK_MAXN / SK_BN) * sizeof(unsigned)) != hipSuccess) return nullptr;
  return &(g_scratch[(void*)s] = sc);
}

}  // namespace

// EM_ERR_UNSUPPORTED = "not my shape": the caller falls through to the tiled kernel.
int em_gemm_splitk(int dtype, int epilogue, const EmGemmArgs* p, void* stream) {
  if (epilogue != EM_EPI_RESID_F32) return EM_ERR_UNSUPPORTED;
  if (p->M <= 48 || p->M > SK_MAXM || p->N % SK_BN != 0 || p->N > SK_MAXN) return EM_ERR_UNSUPPORTED;
  if (p->K < 512 || p->K % SK_KS != 0 || p->K / SK_KS > SK_MAXS || p->ldc % 4 != 0) return EM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  Scratch* sc = scratch_for(s);
  if (!sc) return EM_ERR_UNSUPPORTED;
  dim3 grid(p->N / SK_BN, p->K / SK_KS);
  if (dtype == EM_BF16)
    hipLaunchKernelGGL(gemm_splitk_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)p->A, (const bf16*)p->W,
                       (float*)p->C, p->bias, p->M, p->N, p->K, p->lda, p->ldc, p->scale, sc->part, sc->tickets);
  else if (dtype == EM_F32)
    hipLaunchKernelGGL(gemm_splitk_kernel<float>, grid, dim3(256), 0, s, (const float*)p->A, (const float*)p->W,
                       (float*)p->C, p->bias, p->M, p->N, p->K, p->lda, p->ldc, p->scale, sc->part, sc->tickets);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}
