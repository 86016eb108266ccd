// Dense contraction C = A[M,K] * W[N,K]^T (+ fused epilogue) on the gfx950 matrix cores.
//
// One kernel template covers every GEMM-shaped op of the path (SURVEY.md §8(a) A4, A7, A8, A9,
// A11, A14): torch.nn.Linear (positionwise_feed_forward.py:30-32, attention.py:77-119, ctc.py:39),
// Conv1d(k=1) (convolution.py:28-53) and Conv2d(d,d,3,2) as an implicit GEMM over a channel-last
// feature map (subsampling.py:402-403).
//
// Tile: 128x128 per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K-step = 128 bytes of K per row (64 bf16 / 32 f32), register-prefetched, LDS rows padded to
// 144 B.  f32 accumulate in both modes; EM_F32 uses v_mfma_f32_16x16x4_f32 (exact f32 products),
// EM_BF16 uses v_mfma_f32_16x16x32_bf16.
#include "em_common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;   // K bytes per row per K-step
constexpr int LDS_STRIDE = 144;  // bytes (16 B pad against ds_read bank conflicts)

struct ConvGeom {
  int T1, F1, T2, F2, d;
};

template <typename T, int EPI, int AMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const T* __restrict__ A,
                                                   const T* __restrict__ W, void* __restrict__ Cv,
                                                   const float* __restrict__ bias, int M, int N,
                                                   int K, int lda, int ldc, float scale,
                                                   ConvGeom g) {
  constexpr int BK = ROW_BYTES / (int)sizeof(T);  // elements of K per step
  constexpr int CH = 16 / (int)sizeof(T);         // elements per 16-byte chunk
  constexpr int KSUB = BK / Mma<T>::K;            // MFMA k-steps per K-step
  __shared__ __attribute__((aligned(16))) unsigned char s_a[BM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) unsigned char s_b[BN * LDS_STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // ---- staging assignment: chunk c = tid + 256 i  ->  row c/8, 16-byte piece c%8
  const int piece = tid & 7;
  size_t a_base[4], w_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (tid >> 3) + 32 * i;
    int m = m0 + row;
    m = m < M ? m : M - 1;
    if (AMODE == EM_A_CONV2) {
      int per_b = g.T2 * g.F2;
      int bb = m / per_b, rem = m - bb * per_b;
      int t2 = rem / g.F2, f2 = rem - t2 * g.F2;
      a_base[i] = ((size_t)(bb * g.T1 + 2 * t2) * g.F1 + 2 * f2) * g.d;
    } else {
      a_base[i] = (size_t)m * lda;
    }
    int n = n0 + row;
    n = n < N ? n : N - 1;
    w_base[i] = (size_t)n * K;
  }

  uint4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
    size_t koff = k0;
    if (AMODE == EM_A_CONV2) {
      int q = k0 / g.d, c0 = k0 - q * g.d;
      int kt = q / 3, kf = q - kt * 3;
      koff = (size_t)(kt * g.F1 + kf) * g.d + c0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *(const uint4*)(A + a_base[i] + koff + piece * CH);
      rb[i] = *(const uint4*)(W + w_base[i] + k0 + piece * CH);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = (tid >> 3) + 32 * i;
      *(uint4*)(s_a + row * LDS_STRIDE + piece * 16) = ra[i];
      *(uint4*)(s_b + row * LDS_STRIDE + piece * 16) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15;
  const int frag_koff = (lane >> 4) * Mma<T>::EPL * (int)sizeof(T);  // bytes
  const unsigned char* pa = s_a + (wr * 64 + frag_row) * LDS_STRIDE + frag_koff;
  const unsigned char* pb = s_b + (wc * 64 + frag_row) * LDS_STRIDE + frag_koff;

  const int nk = K / BK;
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      typename Mma<T>::frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = Mma<T>::load((const T*)(pa + i * 16 * LDS_STRIDE + ks * Mma<T>::K * (int)sizeof(T)));
        fb[i] = Mma<T>::load((const T*)(pb + i * 16 * LDS_STRIDE + ks * Mma<T>::K * (int)sizeof(T)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r.
  const int col_l = lane & 15, row_l = (lane >> 4) * 4;
  if (EPI == EM_EPI_GLU) {
    T* C = (T*)Cv;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      int ncol = n0 + wc * 64 + j * 16 + col_l;  // packed column of the value half
      if (ncol >= N) continue;
      float bv = bias ? bias[ncol] : 0.f, bg = bias ? bias[ncol + 16] : 0.f;
      int ocol = (n0 + wc * 64 + j * 16) / 2 + col_l;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = m0 + wr * 64 + i * 16 + row_l + r;
          if (m < M) {
            float v = acc[i][j][r] + bv, gt = acc[i][j + 1][r] + bg;
            C[(size_t)m * ldc + ocol] = from_f32<T>(v * sigmoidf_(gt));
          }
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = n0 + wc * 64 + j * 16 + col_l;
    if (n >= N) continue;
    float bj = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wr * 64 + i * 16 + row_l + r;
        if (m >= M) continue;
        float v = acc[i][j][r] + bj;
        size_t o = (size_t)m * ldc + n;
        if (EPI == EM_EPI_STORE) ((T*)Cv)[o] = from_f32<T>(v);
        else if (EPI == EM_EPI_SWISH) ((T*)Cv)[o] = from_f32<T>(swishf_(v));
        else if (EPI == EM_EPI_RELU) ((T*)Cv)[o] = from_f32<T>(fmaxf(v, 0.f));
        else if (EPI == EM_EPI_RESID_F32) ((float*)Cv)[o] += scale * v;
        else if (EPI == EM_EPI_SCALE_F32) ((float*)Cv)[o] = scale * v;
        else if (EPI == EM_EPI_STORE_F32) ((float*)Cv)[o] = v;
      }
  }
}

template <typename T, int EPI, int AMODE>
int launch(const EmGemmArgs* p, hipStream_t s) {
  dim3 grid(em_cdiv(p->N, BN), em_cdiv(p->M, BM));
  ConvGeom g{p->T1, p->F1, p->T2, p->F2, p->d};
  hipLaunchKernelGGL((gemm_kernel<T, EPI, AMODE>), grid, dim3(256), 0, s, (const T*)p->A,
                     (const T*)p->W, p->C, p->bias, p->M, p->N, p->K, p->lda, p->ldc, p->scale, g);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T>
int dispatch(int epi, int amode, const EmGemmArgs* p, hipStream_t s) {
  if (amode == EM_A_CONV2) {
    if (epi != EM_EPI_RELU) return EM_ERR_UNSUPPORTED;
    return launch<T, EM_EPI_RELU, EM_A_CONV2>(p, s);
  }
  switch (epi) {
    case EM_EPI_STORE: return launch<T, EM_EPI_STORE, EM_A_PLAIN>(p, s);
    case EM_EPI_SWISH: return launch<T, EM_EPI_SWISH, EM_A_PLAIN>(p, s);
    case EM_EPI_RELU: return launch<T, EM_EPI_RELU, EM_A_PLAIN>(p, s);
    case EM_EPI_RESID_F32: return launch<T, EM_EPI_RESID_F32, EM_A_PLAIN>(p, s);
    case EM_EPI_SCALE_F32: return launch<T, EM_EPI_SCALE_F32, EM_A_PLAIN>(p, s);
    case EM_EPI_GLU: return launch<T, EM_EPI_GLU, EM_A_PLAIN>(p, s);
    case EM_EPI_STORE_F32: return launch<T, EM_EPI_STORE_F32, EM_A_PLAIN>(p, s);
  }
  return EM_ERR_BAD_ARG;
}

}  // namespace

extern "C" int em_gemm(int dtype, int epilogue, int a_mode, const EmGemmArgs* p, void* stream) {
  if (!p || !p->A || !p->W || !p->C) return EM_ERR_BAD_ARG;
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return EM_ERR_BAD_ARG;
  const int bk = dtype == EM_BF16 ? 64 : 32;
  if (p->K % bk != 0) return EM_ERR_UNSUPPORTED;
  if (epilogue == EM_EPI_GLU && (p->N % 32 != 0)) return EM_ERR_UNSUPPORTED;
  if (a_mode == EM_A_CONV2) {
    if (p->d <= 0 || p->d % bk != 0 || p->K != 9 * p->d) return EM_ERR_UNSUPPORTED;
  } else if (a_mode != EM_A_PLAIN) {
    return EM_ERR_BAD_ARG;
  } else if (p->lda % (dtype == EM_BF16 ? 8 : 4) != 0) {
    return EM_ERR_UNSUPPORTED;  // 16-byte aligned rows
  }
  if (dtype == EM_F32) return dispatch<float>(epilogue, a_mode, p, (hipStream_t)stream);
  if (dtype == EM_BF16) return dispatch<bf16>(epilogue, a_mode, p, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
