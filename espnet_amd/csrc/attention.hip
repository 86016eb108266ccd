// Relative-position multi-head self-attention core for the Conformer encoder.
//
// Reference: RelPositionMultiHeadedAttention.forward
// (espnet2/legacy/nets/pytorch_backend/transformer/attention.py:416-459), rel_shift (:391-408),
// forward_attention (:121-151).  Computed per (utterance b, head, 64-query tile):
//     AC[i][j] = (q_i + u) . k_j
//     BD[i][j] = (q_i + v) . p[T-1-i+j]          (rel_shift folded into the index: never
//                                                 materialises the (T, 2T) padded view)
//     P = softmax_j((AC + BD) / sqrt(dk)) over keys j < klens[b]  (masked probabilities are 0)
//     ctx_i = sum_j P[i][j] v_j
// Flash-style: 64-key tiles, online softmax in f32, MFMA 16x16 tiles for AC, BD and P.V.
// Each wave owns 16 query rows; the BD window of a (16-query, 64-key) tile is the 80 consecutive
// rows p[c0 .. c0+79], c0 = T-1-(i_w+15)+j0, computed dense and re-read skewed through LDS.
#include "em_common.h"

namespace {

constexpr int QT = 64;  // queries per workgroup (4 waves x 16)
constexpr int KT = 64;  // keys per tile
constexpr int DK = 64;

template <typename T, bool LEGACY = false>
struct AttnLds {
  static constexpr int LDT = 64 + 16 / (int)sizeof(T);  // padded row (elements)
  static constexpr int LDB = 84;                         // padded BD row (floats)
  // LEGACY keeps a second BD window per wave (its position rows reuse the sP staging area)
  static constexpr size_t bytes = (size_t)(64 + 64 + 128 + 64) * LDT * sizeof(T) +
                                  (size_t)(LEGACY ? 2 : 1) * 4 * 16 * LDB * sizeof(float);
};

// LEGACY = LegacyRelPositionMultiHeadedAttention (attention.py:268-360, the class default
// rel_pos_type="legacy" of older checkpoints): p has T rows (LegacyRelPositionalEncoding: the sinusoid of
// position max_len-1-k in row k) and rel_shift (:296-316) is the square pad-and-reshape, which gives
//     BD[i][j] = (q_i + v) . p[T-1-i+j]          j <= i       (the same index as the current scheme)
//              = 0                                j == i+1
//              = (q_{i+1} + v) . p[j-i-2]         j >  i+1     (the wrapped rows of the reshape)
// The upper part is a second dense window, computed from the NEXT query row's fragments over the 80
// position rows p[j0-iw0-17 ..]; both windows are read with the same skew 15 - i_local + j_local.
template <typename T, bool LEGACY = false>
__global__ __launch_bounds__(256) void relpos_attn_kernel(
    const T* __restrict__ qkv, const T* __restrict__ p, int ldp, const float* __restrict__ pos_u,
    const float* __restrict__ pos_v, const int* __restrict__ klens, int Tn, int h,
    T* __restrict__ ctx) {
  using M = Mma<T>;
  constexpr int LDT = AttnLds<T, LEGACY>::LDT, LDB = AttnLds<T, LEGACY>::LDB;
  constexpr int CH = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int PPR = DK / CH;             // chunks per 64-element row
  constexpr int KS = DK / M::K;            // MFMA steps over a 64-deep contraction
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sK = (T*)smem;            // [64 keys][LDT]
  T* sVt = sK + 64 * LDT;      // [64 dk][LDT keys]
  T* sP = sVt + 64 * LDT;      // [128 pos rows][LDT]
  T* sPr = sP + 128 * LDT;     // [4 waves][16][LDT]
  float* sBD = (float*)(sPr + 64 * LDT);  // [4 waves][16][LDB] (LEGACY: x2)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = blockIdx.y, b = blockIdx.z;
  const int d = h * DK, ld = 3 * d;
  const int i0 = blockIdx.x * QT, iw0 = i0 + 16 * wave;
  const int klen = klens[b] < Tn ? klens[b] : Tn;
  const int lr = lane & 15, lg = lane >> 4;
  T* myPr = sPr + wave * 16 * LDT;
  float* myBD = sBD + wave * 16 * LDB;
  float* myBD2 = sBD + (4 + wave) * 16 * LDB;  // LEGACY only

  // ---- query fragments (A operand): row = iw0 + lr, k-slice lg
  typename M::frag qu[KS], qv[KS], qv2[LEGACY ? KS : 1];
  if (LEGACY) {  // (q_{i+1} + v) for the wrapped upper part
    int i = iw0 + lr + 1;
    i = i < Tn ? i : Tn - 1;
    const T* qrow = qkv + (size_t)(b * Tn + i) * ld + hh * DK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __attribute__((aligned(16))) T tv[M::EPL];
#pragma unroll
      for (int e = 0; e < M::EPL; ++e) {
        int c = ks * M::K + lg * M::EPL + e;
        tv[e] = from_f32<T>(to_f32(qrow[c]) + pos_v[hh * DK + c]);
      }
      qv2[ks] = M::load(tv);
    }
  }
  {
    int i = iw0 + lr;
    i = i < Tn ? i : Tn - 1;
    const T* qrow = qkv + (size_t)(b * Tn + i) * ld + hh * DK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __attribute__((aligned(16))) T tu[M::EPL];
      __attribute__((aligned(16))) T tv[M::EPL];
#pragma unroll
      for (int e = 0; e < M::EPL; ++e) {
        int c = ks * M::K + lg * M::EPL + e;
        float q = to_f32(qrow[c]);
        tu[e] = from_f32<T>(q + pos_u[hh * DK + c]);
        tv[e] = from_f32<T>(q + pos_v[hh * DK + c]);
      }
      qu[ks] = M::load(tu);
      qv[ks] = M::load(tv);
    }
  }

  f32x4 acc_o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc_o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float row_m[4], row_l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    row_m[r] = -INFINITY;
    row_l[r] = 0.f;
  }

  for (int j0 = 0; j0 < klen; j0 += KT) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K, V^T (keys j0..j0+63) and the 128 position rows of this (q-tile, k-tile)
    for (int c = tid; c < 64 * PPR; c += 256) {
      int row = c / PPR, piece = c - row * PPR;
      int j = j0 + row;
      j = j < Tn ? j : Tn - 1;
      const T* src = qkv + (size_t)(b * Tn + j) * ld + hh * DK + piece * CH;
      uint4 kv = *(const uint4*)(src + d);
      uint4 vv = *(const uint4*)(src + 2 * d);
      *(uint4*)(sK + row * LDT + piece * CH) = kv;
      const T* ve = (const T*)&vv;
#pragma unroll
      for (int e = 0; e < CH; ++e) sVt[(piece * CH + e) * LDT + row] = ve[e];
    }
    const int cbase = Tn - 1 - (i0 + 63) + j0;
    for (int c = tid; c < 128 * PPR; c += 256) {
      int row = c / PPR, piece = c - row * PPR;
      int pc = cbase + row;
      const int pmax = LEGACY ? Tn - 1 : 2 * Tn - 2;
      pc = pc < 0 ? 0 : (pc > pmax ? pmax : pc);
      *(uint4*)(sP + row * LDT + piece * CH) =
          *(const uint4*)(p + (size_t)pc * ldp + hh * DK + piece * CH);
    }
    __syncthreads();

    // ---- AC (16 x 64) and dense BD window (16 x 80)
    f32x4 acc_s[4], acc_bd[5];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc_s[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 5; ++n) acc_bd[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rbase = 48 - 16 * wave;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ko = ks * M::K + lg * M::EPL;
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc_s[n] = M::mma(qu[ks], M::load(sK + (n * 16 + lr) * LDT + ko), acc_s[n]);
#pragma unroll
      for (int n = 0; n < 5; ++n)
        acc_bd[n] = M::mma(qv[ks], M::load(sP + (rbase + n * 16 + lr) * LDT + ko), acc_bd[n]);
    }
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) myBD[(lg * 4 + r) * LDB + n * 16 + lr] = acc_bd[n][r];
    if (LEGACY) {
      __syncthreads();  // every wave is done with the first position window
      const int c2base = j0 - i0 - 65;
      for (int c = tid; c < 128 * PPR; c += 256) {
        int row = c / PPR, piece = c - row * PPR;
        int pc = c2base + row;
        pc = pc < 0 ? 0 : (pc > Tn - 1 ? Tn - 1 : pc);
        *(uint4*)(sP + row * LDT + piece * CH) = *(const uint4*)(p + (size_t)pc * ldp + hh * DK + piece * CH);
      }
      __syncthreads();
      f32x4 acc_b2[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) acc_b2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ko = ks * M::K + lg * M::EPL;
#pragma unroll
        for (int n = 0; n < 5; ++n)
          acc_b2[n] = M::mma(qv2[ks], M::load(sP + (rbase + n * 16 + lr) * LDT + ko), acc_b2[n]);
      }
#pragma unroll
      for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) myBD2[(lg * 4 + r) * LDB + n * 16 + lr] = acc_b2[n][r];
    }
    __syncthreads();

    // ---- scores, mask, online softmax
    float tile_m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tile_m[r] = -INFINITY;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ii = lg * 4 + r, jj = n * 16 + lr;
        float bd = myBD[ii * LDB + 15 - ii + jj];
        if (LEGACY) {
          const int i = iw0 + ii, j = j0 + jj;
          if (j == i + 1) bd = 0.f;
          else if (j > i + 1) bd = myBD2[ii * LDB + 15 - ii + jj];
        }
        float s = (acc_s[n][r] + bd) * 0.125f;  // 1/sqrt(64)
        s = (j0 + jj < klen) ? s : -INFINITY;
        acc_s[n][r] = s;
        tile_m[r] = fmaxf(tile_m[r], s);
      }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = tile_m[r];
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      m = fmaxf(m, __shfl_xor(m, 4, 64));
      m = fmaxf(m, __shfl_xor(m, 8, 64));
      float m_new = fmaxf(row_m[r], m);
      alpha[r] = expf(row_m[r] - m_new);
      row_m[r] = m_new;
    }
    float psum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = expf(acc_s[n][r] - row_m[r]);  // masked: exp(-inf) = 0
        T prt = from_f32<T>(pr);
        psum[r] += to_f32(prt);
        myPr[(lg * 4 + r) * LDT + n * 16 + lr] = prt;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = psum[r];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      row_l[r] = row_l[r] * alpha[r] + s;
#pragma unroll
      for (int n = 0; n < 4; ++n) acc_o[n][r] *= alpha[r];
    }
    __syncthreads();

    // ---- O += P . V   (A = probabilities [16 q][64 keys], B = V^T [dk][keys])
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ko = ks * M::K + lg * M::EPL;
      typename M::frag pa = M::load(myPr + lr * LDT + ko);
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc_o[n] = M::mma(pa, M::load(sVt + (n * 16 + lr) * LDT + ko), acc_o[n]);
    }
  }

  // ---- normalise and store ctx[b][i][hh*64 + col]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int i = iw0 + lg * 4 + r;
    if (i >= Tn) continue;
    float inv = row_l[r] > 0.f ? 1.0f / row_l[r] : 0.f;
    T* o = ctx + (size_t)(b * Tn + i) * d + hh * DK;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n * 16 + lr] = from_f32<T>(acc_o[n][r] * inv);
  }
}

template <typename T, bool LEGACY>
int launch_attn(const void* qkv, const void* p, int ldp, const float* pu, const float* pv,
                const int* klens, int B, int Tn, int h, void* ctx, hipStream_t s) {
  const size_t lds = AttnLds<T, LEGACY>::bytes;
  static EmLdsCap cap = {};
  if (em_raise_lds_cap((const void*)relpos_attn_kernel<T, LEGACY>, lds, &cap) != EM_OK) return EM_ERR_LAUNCH;
  dim3 grid(em_cdiv(Tn, QT), h, B);
  hipLaunchKernelGGL((relpos_attn_kernel<T, LEGACY>), grid, dim3(256), lds, s, (const T*)qkv, (const T*)p,
                     ldp, pu, pv, klens, Tn, h, (T*)ctx);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

}  // namespace

extern "C" int em_relpos_attention(int dtype, const void* qkv, const void* p, int32_t ldp,
                                   const float* pos_u, const float* pos_v, const int32_t* klens,
                                   int32_t B, int32_t T, int32_t h, int32_t dk, void* ctx,
                                   void* stream) {
  if (B <= 0 || T <= 0 || h <= 0) return EM_ERR_BAD_ARG;
  if (dk != DK) return EM_ERR_UNSUPPORTED;
  if (dtype == EM_F32)
    return launch_attn<float, false>(qkv, p, ldp, pos_u, pos_v, klens, B, T, h, ctx, (hipStream_t)stream);
  if (dtype == EM_BF16)
    return launch_attn<bf16, false>(qkv, p, ldp, pos_u, pos_v, klens, B, T, h, ctx, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}

extern "C" int em_legacy_relpos_attention(int dtype, const void* qkv, const void* p, int32_t ldp,
                                          const float* pos_u, const float* pos_v, const int32_t* klens,
                                          int32_t B, int32_t T, int32_t h, int32_t dk, void* ctx,
                                          void* stream) {
  if (B <= 0 || T <= 0 || h <= 0) return EM_ERR_BAD_ARG;
  if (dk != DK) return EM_ERR_UNSUPPORTED;
  if (dtype == EM_F32)
    return launch_attn<float, true>(qkv, p, ldp, pos_u, pos_v, klens, B, T, h, ctx, (hipStream_t)stream);
  if (dtype == EM_BF16)
    return launch_attn<bf16, true>(qkv, p, ldp, pos_u, pos_v, klens, B, T, h, ctx, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
