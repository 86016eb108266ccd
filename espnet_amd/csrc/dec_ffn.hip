// The feed-forward sublayer of a decoder label step on FRAGMENT-MAJOR operands (bf16), two launches:
//     H = relu(W1 . LN(x; g, b) + b1)          dec_ffn_hidden_kernel (this file): LayerNorm in the prologue, H fragment-major
//     x += W2 . H + b2                         mid_gemm_kernel<..., FRAG> (csrc/gemm_mid.hip): both operands fragment-major
// Reference: DecoderLayer.forward, pre-norm branch (espnet2/legacy/nets/pytorch_backend/transformer/decoder_layer.py:150-158:
// norm3 -> feed_forward -> residual), PositionwiseFeedForward.forward (transformer/positionwise_feed_forward.py:30-32).
//
// Why (round 6).  At configs[3]'s per-GPU shape (640 rows, d = 512, ff = 2048) the sublayer was three launches per layer -
// LayerNorm 5.0 us, the tiled GEMM with the ReLU epilogue 7.5 us, mid_gemm over K = 2048 14.4 us: 26.9 of a layer's 83 us,
// six layers per label step (profiles/r06s_search640_kernel_stats.csv).  The arithmetic is 2.7 GFLOP; what the launches
// wait for is operands, and they read them as MFMA fragments straight from ROW-MAJOR matrices: 16 rows x 64 bytes per
// wave-wide load, which a CU ingests at ~37 GB/s where contiguous KiB run at ~128 GB/s (block.hip's header; measured again
// here: the first form of this file, one launch with the hidden dimension dealt to S workgroups per row block and a
// ticketed meeting of their partial sums, ran 39 us per launch from row-major weights and 25.5 us from fragment-major
// ones - and every doubling of S cost more at the meeting point than the halved weight stream gave back,
// profiles/r06ab_dec_ffn_meeting_ab.txt.  A launch boundary is the cheaper meeting point on this chip.)
//   * fragment-major = a [R][K] matrix stored as [R / 16][K / 32][lane = 16 (k % 32 / 8) + r % 16][k % 8]: the 16 rows x 32 k
//     of one MFMA operand are 1 KiB contiguous (host: espnet_amd.lib.pack_frag16 for W1 / W2; H is WRITTEN that way by the
//     first launch, 8 bytes per lane);
//   * first launch: a workgroup owns 16 rows and ff / S hidden units (S chosen so that the launch is ~160 workgroups: 640
//     rows: 4, 160 rows: 16): it normalises its rows (the statistics of ln_gemm_kernel, bit for bit) into LDS, holds them
//     as MFMA B operands for the whole launch, and streams its W1 rows as A operands, a batch of 16 wave-wide loads ahead;
//   * second launch: mid_gemm's shape (a tile per workgroup, K split over the 8 waves, every operand fragment requested
//     before the first MFMA, LDS reduction in fixed order: deterministic) with 1 KiB loads for both operands.
#include "em_common.h"
#include "switches.h"

namespace {

constexpr int FR = 16;       // rows per workgroup: one MFMA column tile
constexpr int LOADS = 16;    // wave-wide 1 KiB loads per batch (one batch in flight behind the one being consumed)

__device__ __forceinline__ float ffn_row16_allsum(float v) {  // (ln_gemm.hip's reduction: the same order, the same bits)
  v += dpp_f32<DPP_XOR1>(v);
  v += dpp_f32<DPP_XOR2>(v);
  v += dpp_f32<DPP_HALF_MIRROR>(v);
  v += dpp_f32<DPP_MIRROR>(v);
  return v;
}

// out[n][N] = epilogue(LN(x[n][DT]; g, be) . W1^T + b1), W1 fragment-major with rows padded to a multiple of HS.
// DT: model dimension (256 | 512); HS: output columns of one workgroup (128 | 256 | 512); OUT: EM_LNF_RELU_FRAG - ReLU, bf16,
// written FRAGMENT-MAJOR (the feed-forward's hidden activation; `ff` = its width); EM_LNF_STORE - bf16 rows [n][ff];
// EM_LNF_STORE_F32 - f32 rows [n][ff] (the vocabulary logits).  Row-major outputs mask rows >= n and columns >= ff.
// FRN: 16-row fragments per workgroup (1 | 2).  Two from 320 rows on: a weight fragment then feeds two MFMAs and the launch streams
// half the weight bytes per row (640 rows: the feed-forward's first projection 9.8 us with 16-row workgroups).
template <int DT, int HS, int OUT, int FRN>
__global__ __launch_bounds__(256) void ln_frag_gemm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ be, float eps,
                                                           const bf16* __restrict__ W1, const float* __restrict__ b1, int n,
                                                           int ff, void* __restrict__ Hv) {
  constexpr int XP = DT + 8;  // LDS row pitch (elements): 16-byte aligned rows, a 4-bank step per row
  constexpr int NV = DT / 64;              // float4 chunks per lane of a row's LayerNorm (16 lanes per row)
  // per wave NF1 fragments of 16 hidden units x NK1 k-steps, in batches of GB1 fragments = 16 loads
  constexpr int NK1 = DT / 32, NF1 = HS / 64, GB1 = (LOADS / NK1) < 1 ? 1 : (LOADS / NK1), NB1 = NF1 / GB1;
  static_assert(GB1 * NK1 == LOADS && NB1 >= 1 && NB1 * GB1 == NF1, "batches of 16 loads");
  constexpr int FRT = FR * FRN;  // rows per workgroup
  __shared__ __attribute__((aligned(16))) bf16 sX[FRT * XP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int sp = blockIdx.x, rb = blockIdx.y, r0 = rb * FRT;
  const int h0 = sp * HS;

  // W1 fragment-major: the wave's stream is NF1 x NK1 consecutive KiB
  const bf16* const w1f = W1 + ((size_t)((h0 >> 4) + wave * NF1) * NK1 * 64 + lane) * 8;  // + (f * NK1 + ks) * 512
  bf16x8 w[2][LOADS];
  auto load1 = [&](bf16x8 (&dst)[LOADS], int bt) __attribute__((always_inline)) {
#pragma unroll
    for (int gi = 0; gi < GB1; ++gi)
#pragma unroll
      for (int ks = 0; ks < NK1; ++ks) dst[gi * NK1 + ks] = *(const bf16x8*)(w1f + (size_t)(((bt * GB1 + gi) * NK1 + ks) * 512));
  };
  // the first batch of W1 does not depend on the LayerNorm: requested before anything else
  load1(w[0], 0);
  float4 bias1[NF1];
#pragma unroll
  for (int f = 0; f < NF1; ++f) {  // (columns past the matrix' real width - zero weight rows of the padded packing - have no bias)
    const int c0 = h0 + (wave * NF1 + f) * 16 + lg * 4;
    bias1[f] = (OUT == EM_LNF_RELU_FRAG || c0 + 3 < ff) ? *(const float4*)(b1 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- LayerNorm of the 16 rows into LDS (16 lanes per row; rows past n repeat row n - 1 and are never stored)
  {
    const int li = tid & 15;
    float4 v[FRN][NV], g4[NV], b4[NV];
#pragma unroll
    for (int ps = 0; ps < FRN; ++ps) {
      int m = r0 + ps * FR + (tid >> 4);
      m = m < n ? m : n - 1;
      const float4* xr = (const float4*)(x + (size_t)m * DT);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[ps][j] = xr[j * 16 + li];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      g4[j] = *(const float4*)(g + (j * 16 + li) * 4);
      b4[j] = *(const float4*)(be + (j * 16 + li) * 4);
    }
    asm volatile("" ::: "memory");  // every request above is out before the first reduction
#pragma unroll
    for (int ps = 0; ps < FRN; ++ps) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) s += (v[ps][j].x + v[ps][j].y) + (v[ps][j].z + v[ps][j].w);
      s = ffn_row16_allsum(s);
      const float mean = s / (float)DT;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float a = v[ps][j].x - mean, b = v[ps][j].y - mean, c = v[ps][j].z - mean, d = v[ps][j].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
      q = ffn_row16_allsum(q);
      const float rstd = 1.0f / sqrtf(q / (float)DT + eps);
      bf16* dst = sX + (ps * FR + (tid >> 4)) * XP;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        bf16x4 o;
        o[0] = (bf16)((v[ps][j].x - mean) * rstd * g4[j].x + b4[j].x);
        o[1] = (bf16)((v[ps][j].y - mean) * rstd * g4[j].y + b4[j].y);
        o[2] = (bf16)((v[ps][j].z - mean) * rstd * g4[j].z + b4[j].z);
        o[3] = (bf16)((v[ps][j].w - mean) * rstd * g4[j].w + b4[j].w);
        *(bf16x4*)(dst + (j * 16 + li) * 4) = o;
      }
    }
  }
  __syncthreads();

  // ---- H[row][hidden] = relu(W1 . LN(x) + b1), this wave's NF1 fragments of 16 hidden units
  {
    bf16x8 xb[FRN][NK1];  // B operands: column = row mi * 16 + lr, k-slice lg
#pragma unroll
    for (int mi = 0; mi < FRN; ++mi)
#pragma unroll
      for (int ks = 0; ks < NK1; ++ks) xb[mi][ks] = *(const bf16x8*)(sX + (mi * FR + lr) * XP + ks * 32 + lg * 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bt = 0; bt < NB1; ++bt) {
      // (sched_barrier: left alone hipcc interleaves the fully unrolled batches - a load, a wait, an MFMA - and the batch
      // ahead becomes a chain of dependent round trips; tools/isa_waits.py)
      if (bt + 1 < NB1) load1(w[(bt + 1) & 1], bt + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gi = 0; gi < GB1; ++gi) {
        f32x4 acc[FRN], acc1[FRN];  // two chains per row fragment: even / odd k-steps
#pragma unroll
        for (int mi = 0; mi < FRN; ++mi) acc[mi] = acc1[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK1; ks += 2)
#pragma unroll
          for (int mi = 0; mi < FRN; ++mi) {
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[bt & 1][gi * NK1 + ks], xb[mi][ks], acc[mi], 0, 0, 0);
            acc1[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[bt & 1][gi * NK1 + ks + 1], xb[mi][ks + 1], acc1[mi], 0, 0, 0);
          }
        // lane (lr, lg): row mi * 16 + lr, output columns 16 f + 4 lg .. + 3 of this workgroup's slice
        const int f = bt * GB1 + gi;
        const float4 bb = bias1[f];
#pragma unroll
        for (int mi = 0; mi < FRN; ++mi) {
          const f32x4 a = acc[mi] + acc1[mi];
          const int row = r0 + mi * FR + lr;
          if constexpr (OUT == EM_LNF_RELU_FRAG) {
            bf16x4 o;
            o[0] = (bf16)fmaxf(a[0] + bb.x, 0.f);
            o[1] = (bf16)fmaxf(a[1] + bb.y, 0.f);
            o[2] = (bf16)fmaxf(a[2] + bb.z, 0.f);
            o[3] = (bf16)fmaxf(a[3] + bb.w, 0.f);
            // H fragment-major: hidden fragment fg is the (fg & 1) half of k-step fg >> 1 of row fragment rb * FRN + mi; this
            // lane's four values are elements (lg & 1) * 4 .. + 3 of operand lane ((fg & 1) * 2 + (lg >> 1)) * 16 + lr.
            // (a row fragment wholly past n - the second one of the last workgroup when n / 16 is odd - is not in H)
            const int fg = (h0 >> 4) + wave * NF1 + f;
            if (r0 + mi * FR < n)
              *(bf16x4*)((bf16*)Hv + (((size_t)(rb * FRN + mi) * (ff >> 5) + (fg >> 1)) * 64 + ((fg & 1) * 2 + (lg >> 1)) * 16 + lr) * 8 +
                         (lg & 1) * 4) = o;
          } else {
            const int c0 = h0 + (wave * NF1 + f) * 16 + lg * 4;
            if (row < n && c0 + 3 < ff) {
              const size_t o0 = (size_t)row * ff + c0;
              if constexpr (OUT == EM_LNF_STORE) {
                bf16x4 o;
                o[0] = (bf16)(a[0] + bb.x);
                o[1] = (bf16)(a[1] + bb.y);
                o[2] = (bf16)(a[2] + bb.z);
                o[3] = (bf16)(a[3] + bb.w);
                *(bf16x4*)((bf16*)Hv + o0) = o;
              } else {
                *(float4*)((float*)Hv + o0) = make_float4(a[0] + bb.x, a[1] + bb.y, a[2] + bb.z, a[3] + bb.w);
              }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int DT, int HS, int OUT, int FRN>
int launch_lnf(const float* x, const float* g, const float* be, float eps, const void* w1, const float* b1, int n, int N,
               void* out, hipStream_t s) {
  hipLaunchKernelGGL((ln_frag_gemm_kernel<DT, HS, OUT, FRN>), dim3(em_cdiv(N, HS), em_cdiv(n, FR * FRN)), dim3(256), 0, s, x, g, be,
                     eps, (const bf16*)w1, b1, n, N, out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <int OUT>
int dispatch_lnf(const float* x, const float* g, const float* be, float eps, const void* wf, const float* bias, int n, int N,
                 int d, int hs, int frn, void* out, hipStream_t st) {
#define EM_LNF_CASE(DD, HH)                                                                          \
  if (d == DD && hs == HH)                                                                           \
  return frn == 2 ? launch_lnf<DD, HH, OUT, 2>(x, g, be, eps, wf, bias, n, N, out, st)               \
                  : launch_lnf<DD, HH, OUT, 1>(x, g, be, eps, wf, bias, n, N, out, st)
  EM_LNF_CASE(512, 512);
  EM_LNF_CASE(512, 256);
  EM_LNF_CASE(512, 128);
  EM_LNF_CASE(256, 512);
  EM_LNF_CASE(256, 256);
  EM_LNF_CASE(256, 128);
#undef EM_LNF_CASE
  return EM_ERR_UNSUPPORTED;
}

// rows per workgroup (16 | 32: two row fragments from 320 rows on; ESPNET_AMD_DEC_FFN_ROWS=16 | 32: developer switch)
int lnf_row_frags(int n) {
  const int forced = em_sw().dec_ffn_rows;
  if (forced == 16 || forced == 32) return forced / 16;
  return n >= 320 ? 2 : 1;
}
// output columns per workgroup: ~160 workgroups per launch (fewer columns for fewer rows), 128 at least
int lnf_columns(int n, int N) {
  const int forced = em_sw().dec_ffn_split;  // ESPNET_AMD_DEC_FFN_SPLIT: 0 automatic, 1 off, 128 | 256 | 512 forced
  if (forced == 128 || forced == 256 || forced == 512) return forced;
  const int rbs = em_cdiv(n, FR * lnf_row_frags(n));
  int hs = 512;
  while (hs > 128 && rbs * em_cdiv(N, hs) < 160) hs >>= 1;
  return hs;
}

}  // namespace

// C[n][N] = epilogue(LN(x[n][d]; g, be, eps) . W^T + bias) with W FRAGMENT-MAJOR, its N rows zero-padded to a multiple of
// 512 (espnet_amd.lib.pack_frag16(w, pad_rows=512)): the LayerNorm + projection launches of a label step with 1 KiB operand loads.
// out_mode EM_LNF_RELU_FRAG (ReLU, bf16, fragment-major: n % 16 == 0, N % 128 == 0) | EM_LNF_STORE (bf16 rows) |
// EM_LNF_STORE_F32 (f32 rows); N % 4 == 0; d = 256 | 512; bf16 only.
extern "C" int em_ln_gemm_frag(int out_mode, const float* x, const float* ln_g, const float* ln_b, float eps, const void* wf,
                               const float* bias, void* out, int32_t n, int32_t N, int32_t d, void* stream) {
  if (!x || !ln_g || !ln_b || !wf || !bias || !out || n <= 0 || N <= 0) return EM_ERR_BAD_ARG;
  if ((d != 256 && d != 512) || N % 4 != 0) return EM_ERR_UNSUPPORTED;
  if (em_sw().dec_ffn_split == 1) return EM_ERR_UNSUPPORTED;
  const int hs = lnf_columns(n, N), frn = lnf_row_frags(n);
  hipStream_t st = (hipStream_t)stream;
  switch (out_mode) {
    case EM_LNF_RELU_FRAG:
      if (n % FR != 0 || N % hs != 0) return EM_ERR_UNSUPPORTED;
      return dispatch_lnf<EM_LNF_RELU_FRAG>(x, ln_g, ln_b, eps, wf, bias, n, N, d, hs, frn, out, st);
    case EM_LNF_STORE: return dispatch_lnf<EM_LNF_STORE>(x, ln_g, ln_b, eps, wf, bias, n, N, d, hs, frn, out, st);
    case EM_LNF_STORE_F32: return dispatch_lnf<EM_LNF_STORE_F32>(x, ln_g, ln_b, eps, wf, bias, n, N, d, hs, frn, out, st);
  }
  return EM_ERR_BAD_ARG;
}

// Output columns per workgroup of the feed-forward's first launch for (n, d, ff); 0 = the shape is not covered (the caller
// keeps LayerNorm + two row-major projections).  Covered: d = 256 | 512, ff a multiple of 128, n a multiple of 16 (H is
// written in whole row fragments; the search's n = B x beam rows are).
extern "C" int em_dec_ffn_split(int32_t n, int32_t d, int32_t ff) {
  if (n <= 0 || n % FR != 0 || (d != 256 && d != 512) || ff <= 0 || ff % 128 != 0) return 0;
  if (em_sw().dec_ffn_split == 1) return 0;
  const int hs = lnf_columns(n, ff);
  return ff % hs == 0 ? hs : 0;
}

extern "C" int em_dec_ffn(int dtype, float* x, const float* ln_g, const float* ln_b, float eps, const void* w1f,
                          const float* b1, const void* w2f, const float* b2, int32_t n, int32_t d, int32_t ff, void* hbuf,
                          void* stream) {
  if (dtype != EM_BF16) return EM_ERR_UNSUPPORTED;
  if (!x || !ln_g || !ln_b || !w1f || !b1 || !w2f || !b2 || !hbuf || n <= 0) return EM_ERR_BAD_ARG;
  if (em_dec_ffn_split(n, d, ff) == 0) return EM_ERR_UNSUPPORTED;
  const int rc = em_ln_gemm_frag(EM_LNF_RELU_FRAG, x, ln_g, ln_b, eps, w1f, b1, hbuf, n, ff, d, stream);
  if (rc != EM_OK) return rc;
  EmGemmArgs a = {};
  a.A = hbuf; a.W = w2f; a.C = x; a.bias = b2;
  a.M = n; a.N = d; a.K = ff; a.lda = ff; a.ldc = d; a.scale = 1.f;
  return em_gemm_mid_frag(EM_EPI_RESID_F32, 1, &a, stream);
}
