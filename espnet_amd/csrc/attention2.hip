// Relative-position multi-head self-attention, LDS-resident form (bf16 MFMA, d_k = 64) for the fused
// Conformer path (csrc/block.hip writes its operands per head).
//
// Reference: RelPositionMultiHeadedAttention.forward
// (espnet2/legacy/nets/pytorch_backend/transformer/attention.py:416-459), rel_shift (:391-408),
// forward_attention (:121-151):
//     AC[i][j] = (q_i + u) . k_j        BD[i][j] = (q_i + v) . p[T-1-i+j]
//     P = softmax_j((AC + BD) / sqrt(d_k)) over keys j < klens[b],   ctx_i = sum_j P[i][j] v_j
//
// One workgroup = (utterance b, head, 128 queries), 8 waves x 16 queries.  For up to 256 keys at a time
// everything a workgroup touches sits in LDS, loaded ONCE with global_load_lds (K 32 KiB, V^T 32 KiB, the
// 383 position rows p[T-1-i+j] the (128 query, 256 key) rectangle can reach: 48 KiB): at T = 249 (10 s
// of audio) that is the whole problem, the key loop has no barrier in it and no operand is fetched twice
// (csrc/attention.hip re-stages K, V and 128 position rows per 64 x 64 tile behind four barriers each).
// Longer inputs walk 256-key super-tiles with the online softmax carried in registers.
//
// Orientation: scores are computed TRANSPOSED, S^T[key][query] = K . (Q+u)^T, so that in the MFMA C/D
// layout a lane holds 4 consecutive keys of ONE query: the softmax statistics are per-lane scalars
// (reduced over the 4 lane groups with two shuffles), and the probabilities are already in the
// B-operand layout of O^T[dk][query] = V^T . P^T -- they never go through LDS.  (The 8 contraction
// slots of lane group g then carry keys {4g..4g+3} and {16+4g..16+4g+3} of a 32-key step; V^T is read to
// match with two 8-byte LDS loads.)  rel_shift is index arithmetic: the dense window
// D[c][i] = p[c] . (q_i + v) over the 80 position rows a (16 query, 64 key) tile can reach is written to
// a per-wave scratch and read back skewed, BD^T[j][i] = D[15 - i + j][i].
#include "em_common.h"

namespace {

constexpr int QB = 128;                       // queries per workgroup
constexpr int KSUP = 256;                     // keys per LDS super-tile
constexpr int SK_OFF = 0;                     // [256 keys][128 B], chunks XOR-swizzled by (row & 7)
constexpr int SV_OFF = 32768;                 // [64 dk][512 B],   chunks XOR-swizzled by (row & 15)
constexpr int SP_OFF = 65536;                 // [384 position rows][128 B]
constexpr int SBD_OFF = SP_OFF + 384 * 128;   // [8 waves][16 queries][84] f32
constexpr int LDB = 84;
constexpr int SMEM_BYTES = SBD_OFF + 8 * 16 * LDB * 4;  // 157 696 B

// LDS-DMA from inline asm (see csrc/block.hip: hidden from hipcc's waitcnt pass on purpose)
__device__ __forceinline__ void glds16(const unsigned char* sbase, int voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(512) void relpos_attn2_kernel(
    const bf16* __restrict__ qh, const bf16* __restrict__ kh, const bf16* __restrict__ vt,
    const bf16* __restrict__ p, int ldp, const float* __restrict__ pos_u, const float* __restrict__ pos_v,
    const int* __restrict__ klens, int T, int Tpad, int H, bf16* __restrict__ ctx) {
  using MM = Mma<bf16>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4, swz = lr & 7;
  const int hh = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * QB, iw0 = i0 + 16 * wave;
  const int klen = klens[b] < T ? klens[b] : T;
  const size_t bh = (size_t)b * H + hh;

  const unsigned char* k_base = uniform_ptr((const unsigned char*)(kh + bh * Tpad * 64));
  const unsigned char* v_base = uniform_ptr((const unsigned char*)(vt + bh * 64 * Tpad));
  const unsigned char* p_base = uniform_ptr((const unsigned char*)(p + hh * 64));
  auto stage = [&](int js) {
    const int gc8 = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // K rows js .. js + 255
      const int j = 4 * wave + i;
      int row = js + 8 * j + (lane >> 3);
      row = row < Tpad ? row : Tpad - 1;
      glds16(k_base, row * 128 + gc8 * 16, SK_OFF + j * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // V^T: 64 rows of 256 keys
      const int j = 4 * wave + i, row = 2 * j + (lane >> 5);
      const int gc = (lane & 31) ^ (row & 15);
      glds16(v_base, (row * Tpad + js) * 2 + gc * 16, SV_OFF + j * 1024);
    }
    const int cbase = T - 1 - (i0 + QB - 1) + js;
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // position rows cbase .. cbase + 383 (clamped: only masked entries see the clamp)
      const int j = 6 * wave + i;
      int c = cbase + 8 * j + (lane >> 3);
      c = c < 0 ? 0 : (c > 2 * T - 2 ? 2 * T - 2 : c);
      glds16(p_base, c * (ldp * 2) + gc8 * 16, SP_OFF + j * 1024);
    }
  };
  stage(0);

  // ---- query fragments (B operand: column = query iw0 + lr, k-slice lg), with the two position biases
  bf16x8 qu[2], qv[2];
  {
    const bf16* qrow = qh + (bh * Tpad + iw0 + lr) * 64 + lg * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 raw = *(const bf16x8*)(qrow + ks * 32);
      const float* pu = pos_u + hh * 64 + ks * 32 + lg * 8;
      const float* pv = pos_v + hh * 64 + ks * 32 + lg * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float q = (float)raw[e];
        qu[ks][e] = (bf16)(q + pu[e]);
        qv[ks][e] = (bf16)(q + pv[e]);
      }
    }
  }

  f32x4 acc_o[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc_o[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float row_m = -INFINITY, row_l = 0.f;
  float* const bd = (float*)(smem + SBD_OFF) + wave * 16 * LDB;

  for (int js = 0; js < klen; js += KSUP) {
    if (js > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done with the previous super-tile
      stage(js);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < KSUP / 64; ++kt) {
      const int jl = kt * 64, j0 = js + jl;
      if (j0 >= klen) break;
      // ---- S^T (64 keys x 16 queries) and the dense position window D (80 rows x 16 queries)
      f32x4 sc[4], dd[5];
#pragma unroll
      for (int n = 0; n < 4; ++n) sc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < 5; ++n) dd[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char* sk = smem + SK_OFF + (jl + lr) * 128;
      const unsigned char* sp = smem + SP_OFF + (112 - 16 * wave + jl + lr) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + lg) ^ swz) << 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) sc[n] = MM::mma(*(const bf16x8*)(sk + n * 2048 + coff), qu[ks], sc[n]);
#pragma unroll
        for (int n = 0; n < 5; ++n) dd[n] = MM::mma(*(const bf16x8*)(sp + n * 2048 + coff), qv[ks], dd[n]);
      }
      // ---- rel_shift: D[c][i] -> scratch[i][c], read back at c = 15 - i + j
#pragma unroll
      for (int n = 0; n < 5; ++n)
        *(float4*)(bd + lr * LDB + 16 * n + 4 * lg) = make_float4(dd[n][0], dd[n][1], dd[n][2], dd[n][3]);
      float tm = -INFINITY;
      const float* bdr = bd + lr * LDB + 15 - lr + 4 * lg;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = (sc[n][r] + bdr[16 * n + r]) * 0.125f;  // 1 / sqrt(64)
          s = (j0 + 16 * n + 4 * lg + r < klen) ? s : -INFINITY;
          sc[n][r] = s;
          tm = fmaxf(tm, s);
        }
      // ---- online softmax; this lane's query is iw0 + lr, its keys the 16 (n, r) of lane group lg
      tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(row_m, tm);
      const float alpha = __expf(row_m - mn);
      row_m = mn;
      float ps = 0.f;
      bf16x8 pb[2];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16 pr = (bf16)__expf(sc[n][r] - mn);  // masked: exp(-inf) = 0
          ps += (float)pr;
          pb[n >> 1][(n & 1) * 4 + r] = pr;
        }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      row_l = row_l * alpha + ps;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        acc_o[f][0] *= alpha; acc_o[f][1] *= alpha; acc_o[f][2] *= alpha; acc_o[f][3] *= alpha;
      }
      // ---- O^T += V^T . P^T
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int c0 = ((jl + 32 * jp) >> 3) + (lg >> 1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const unsigned char* sv = smem + SV_OFF + (16 * f + lr) * 512 + (lg & 1) * 8;
          const bf16x4 a0 = *(const bf16x4*)(sv + ((c0 ^ lr) << 4));
          const bf16x4 a1 = *(const bf16x4*)(sv + (((c0 + 2) ^ lr) << 4));
          const bf16x8 vf = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          acc_o[f] = MM::mma(vf, pb[jp], acc_o[f]);
        }
      }
    }
  }

  // ---- normalise and store ctx[b*T + i][hh*64 + 16 f + 4 lg + r]
  const int i = iw0 + lr;
  if (i < T) {
    const float inv = row_l > 0.f ? 1.0f / row_l : 0.f;
    bf16* o = ctx + ((size_t)b * T + i) * (H * 64) + hh * 64 + 4 * lg;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bf16x4 pk = {(bf16)(acc_o[f][0] * inv), (bf16)(acc_o[f][1] * inv), (bf16)(acc_o[f][2] * inv),
                   (bf16)(acc_o[f][3] * inv)};
      *(bf16x4*)(o + 16 * f) = pk;
    }
  }
}

}  // namespace

extern "C" int em_relpos_attention2_bf16(const void* qh, const void* kh, const void* vt, const void* p,
                                         int32_t ldp, const float* pos_u, const float* pos_v,
                                         const int32_t* klens, int32_t B, int32_t T, int32_t Tpad, int32_t h,
                                         void* ctx, void* stream) {
  if (!qh || !kh || !vt || !p || !pos_u || !pos_v || !klens || !ctx) return EM_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || h <= 0) return EM_ERR_BAD_ARG;
  if (Tpad % KSUP != 0 || Tpad < T || ldp % 8 != 0) return EM_ERR_UNSUPPORTED;
  static EmLdsCap cap = {};
  if (em_raise_lds_cap((const void*)relpos_attn2_kernel, SMEM_BYTES, &cap) != EM_OK) return EM_ERR_LAUNCH;
  dim3 grid(em_cdiv(T, QB), h, B);
  const bool rec = em_prof_begin(stream);
  hipLaunchKernelGGL(relpos_attn2_kernel, grid, dim3(512), SMEM_BYTES, (hipStream_t)stream, (const bf16*)qh,
                     (const bf16*)kh, (const bf16*)vt, (const bf16*)p, ldp, pos_u, pos_v, klens, T, Tpad, h,
                     (bf16*)ctx);
  if (rec) em_prof_end(stream, 6.0 * B * h * (double)T * T * 64, EM_PROF_ATTN);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
