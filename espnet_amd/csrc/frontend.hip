// Frontend kernels: framed STFT(512) -> power -> mel -> log, utterance-MVN partial sums, and the
// first subsampling conv.  HBM-bound stage (960 320 algorithmic bytes / 10 s utt).
//
// Reference behaviour being reproduced (paths relative to espnet/espnet):
//   espnet2/layers/stft.py:48-120      torch.stft(center=True, reflect pad, onesided, hann window)
//   espnet2/asr/frontend/default.py:110 power = re^2 + im^2
//   espnet2/layers/log_mel.py:57-84    matmul(melmat) -> clamp(1e-10) -> log -> zero the padding
//   espnet2/layers/utterance_mvn.py:45-88
//   espnet2/legacy/nets/pytorch_backend/transformer/subsampling.py:400-403 (first Conv2d + ReLU)
#include <stdlib.h>

#include "em_common.h"
#include <type_traits>
#include "switches.h"

__device__ const float2 EM_TW512[384] = {
#include "twiddle512.inc"
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// radix-4 DFT of u[0..3] (forward, e^{-2 pi i m q / 4})
__device__ __forceinline__ void bfly4(float2 u[4]) {
  float2 a = make_float2(u[0].x + u[2].x, u[0].y + u[2].y);
  float2 b = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
  float2 c = make_float2(u[1].x + u[3].x, u[1].y + u[3].y);
  float2 d = make_float2(u[1].x - u[3].x, u[1].y - u[3].y);
  u[0] = make_float2(a.x + c.x, a.y + c.y);
  u[2] = make_float2(a.x - c.x, a.y - c.y);
  // -i*d = (d.y, -d.x)
  u[1] = make_float2(b.x + d.y, b.y - d.x);
  u[3] = make_float2(b.x - d.y, b.y + d.x);
}

__device__ __forceinline__ int reflect_idx(int p, int N) {
  if (p < 0) p = -p;
  if (p >= N) p = 2 * (N - 1) - p;
  // degenerate N < n_fft/2 inputs are rejected on the host (torch.stft raises as well)
  return p;
}

// One wave per frame, four frames per 256-thread block.  Each lane owns 4 points of the packed
// 256-point complex FFT (Stockham radix-4, 4 stages, ping-pong in LDS), then the real-FFT
// untangling, |X|^2, and the banded mel contraction.
__global__ __launch_bounds__(256) void frontend_logmel_kernel(
    const float* __restrict__ wav, int N, int hop, const float* __restrict__ window,
    const float* __restrict__ melw, const int* __restrict__ mel_lo, int mel_maxlen, int n_mels,
    const int* __restrict__ flens, const int* __restrict__ wlens, int T_f,
    float* __restrict__ feats) {
  __shared__ float s_re[4][2][256];
  __shared__ float s_im[4][2][256];
  __shared__ float s_pow[4][264];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int t_raw = blockIdx.x * 4 + wave;
  const int t = t_raw < T_f ? t_raw : T_f - 1;
  const float* x = wav + (size_t)b * N;
  // reflect boundary: the padded batch length (what torch.stft sees for a padded batch) or, with
  // wlens, this utterance's own length (what it sees when the utterance is decoded alone)
  const int Nb = wlens ? (wlens[b] < N ? wlens[b] : N) : N;
  float(*re)[256] = s_re[wave];
  float(*im)[256] = s_im[wave];

  // ---- load + window: z[n] = x[2n] + i x[2n+1], n = lane + 64 m
  float2 u[4];
  const int start = t * hop - 256;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int n = lane + 64 * m;
    int i0 = reflect_idx(start + 2 * n, Nb), i1 = reflect_idx(start + 2 * n + 1, Nb);
    i0 = i0 < 0 ? 0 : (i0 < N ? i0 : N - 1);  // frames past flens[b] are zeroed below: keep them in range
    i1 = i1 < 0 ? 0 : (i1 < N ? i1 : N - 1);
    u[m] = make_float2(x[i0] * window[2 * n], x[i1] * window[2 * n + 1]);
  }
  // ---- stage p = 1 (no twiddles): out[4*lane + q]
  bfly4(u);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    re[0][4 * lane + q] = u[q].x;
    im[0][4 * lane + q] = u[q].y;
  }
  __syncthreads();
  // ---- stages p = 4, 16, 64
  int cur = 0;
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    const int p = 1 << (2 * s);
    const int k = lane & (p - 1);
    const int j = ((lane - k) << 2) + k;
    const int twstep = 2 * k * (64 / p);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float2 v = make_float2(re[cur][lane + 64 * m], im[cur][lane + 64 * m]);
      u[m] = (m == 0) ? v : cmul(v, EM_TW512[m * twstep]);
    }
    bfly4(u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      re[cur ^ 1][j + q * p] = u[q].x;
      im[cur ^ 1][j + q * p] = u[q].y;
    }
    cur ^= 1;
    __syncthreads();
  }
  // ---- real-FFT untangle + power: k = lane + 64 m (m < 4) and k = 256 on lane 0
  float* pw = s_pow[wave];
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    int k = lane + 64 * m;
    if (m == 4 && lane != 0) break;
    int ka = k & 255, kb = (256 - k) & 255;
    float2 zk = make_float2(re[cur][ka], im[cur][ka]);
    float2 zc = make_float2(re[cur][kb], -im[cur][kb]);
    float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
    // o = -i/2 (zk - zc)
    float2 dlt = make_float2(zk.x - zc.x, zk.y - zc.y);
    float2 o = make_float2(0.5f * dlt.y, -0.5f * dlt.x);
    float2 w = (k < 384) ? EM_TW512[k] : make_float2(0.f, 0.f);
    float2 xo = cmul(w, o);
    float xr = e.x + xo.x, xi = e.y + xo.y;
    pw[k] = xr * xr + xi * xi;
  }
  __syncthreads();
  // ---- banded mel + log
  const bool valid_frame = t < flens[b];
  if (t_raw < T_f) {
    float* out = feats + ((size_t)b * T_f + t) * n_mels;
    auto band = [&](int m, int s0, int s1) {
      float acc = 0.f;
      const int lo = mel_lo[m];
      for (int s = s0; s < s1; ++s) {
        int k = lo + s;
        k = k < 256 ? k : 256;
        acc = fmaf(pw[k], melw[s * n_mels + m], acc);
      }
      return acc;
    };
    // whole rounds of 64 filters: a lane per filter
    const int nfull = n_mels & ~63;
    for (int m = lane; m < nfull; m += 64) {
      const float acc = fmaxf(band(m, 0, mel_maxlen), 1e-10f);
      out[m] = valid_frame ? logf(acc) : 0.f;
    }
    // the rest (80 filters: 16): with at most 16 filters left, FOUR lanes per filter, a quarter of the band each, summed
    // over the lane groups with two register swaps - 9 trips instead of 33 with three quarters of the wave idle
    const int rest = n_mels - nfull;
    if (rest > 16) {
      const int m = nfull + lane;
      if (m < n_mels) {
        const float acc = fmaxf(band(m, 0, mel_maxlen), 1e-10f);
        out[m] = valid_frame ? logf(acc) : 0.f;
      }
    } else if (rest > 0) {
      const int q = lane >> 4, mm = nfull + (lane & 15);
      const int m = mm < n_mels ? mm : n_mels - 1;
      const int per = (mel_maxlen + 3) >> 2;
      const int s0 = q * per, s1 = (s0 + per < mel_maxlen) ? s0 + per : mel_maxlen;
      float acc = band(m, s0, s1);
      {
        const Pair2 p16 = wave_xor16_pair(acc);
        acc = p16.a + p16.b;
        const Pair2 p32 = wave_xor32_pair(acc);
        acc = p32.a + p32.b;
      }
      acc = fmaxf(acc, 1e-10f);
      if (q == 0 && mm < n_mels) out[mm] = valid_frame ? logf(acc) : 0.f;
    }
  }
}


// ---- round 5: the same arithmetic, organised around what round 4's counters said (39 % of LDS-active cycles in bank
// conflicts, ~900 instructions per frame and wave, 0.07 of the HBM roofline).  A wave now walks FR frames instead of one,
// so everything that does not depend on the frame lives in registers for the walk: the lane's 8 window values, its 9
// stage twiddles + 5 untangling twiddles, and - a lane per mel filter - the filter's band of up to ML weights (the old
// kernel fetched every tap of every filter from global memory inside a serial fma chain, per frame).  The Stockham
// exchanges go through WAVE-PRIVATE LDS tiles without workgroup barriers (LDS operations of one wave execute in order;
// the four waves never read each other's tiles), and the tiles are padded by 4 words per 32 (P(i) = i + 4 (i >> 5)):
// the scattered stores of the p = 4 / p = 16 stages (lane 4 g + k -> 16 g + k + 4 q, lane 16 g + k -> 64 g + k + 16 q)
// hit 8 resp. 16 of the 32 banks in the dense layout (8- / 4-way conflicts) and every bank exactly twice - the floor for
// 64 lanes - in the padded one; stage 1 stores four consecutive points as one 16-byte write.  Results are bit-identical
// to frontend_logmel_kernel (same operations in the same order; tests/test_gpu_kernels.py::test_frontend_v2_equals_v1).
__device__ __forceinline__ int fpad(int i) { return i + ((i >> 5) << 2); }
constexpr int FE_TILE = 288;  // 256 + 4 * 8 words
constexpr int FE_POW = 320;   // 257 power bins + room for a band that runs past them (zero weights there, zeroed once)

template <int ML, int FR>
__global__ __launch_bounds__(256) void frontend_logmel_kernel2(
    const float* __restrict__ wav, int N, int hop, const float* __restrict__ window,
    const float* __restrict__ melw, const int* __restrict__ mel_lo, int mel_maxlen, int n_mels,
    const int* __restrict__ flens, const int* __restrict__ wlens, int T_f,
    float* __restrict__ feats) {
  __shared__ __attribute__((aligned(16))) float s_re[4][2][FE_TILE];
  __shared__ __attribute__((aligned(16))) float s_im[4][2][FE_TILE];
  __shared__ float s_pow[4][FE_POW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const float* x = wav + (size_t)b * N;
  const int Nb = wlens ? (wlens[b] < N ? wlens[b] : N) : N;
  float(*re)[FE_TILE] = s_re[wave];
  float(*im)[FE_TILE] = s_im[wave];
  float* pw = s_pow[wave];
  const int flen = flens[b];

  // ---- frame-independent state of this lane
  float2 win[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) win[m] = *(const float2*)(window + 2 * (lane + 64 * m));
  float2 tw[3][3];
  int jst[3];  // PADDED index of this lane's first output of stages p = 4, 16, 64
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    const int p = 1 << (2 * s);
    const int k = lane & (p - 1);
    jst[s - 1] = fpad(((lane - k) << 2) + k);
    const int twstep = 2 * k * (64 / p);
#pragma unroll
    for (int m = 1; m < 4; ++m) tw[s - 1][m - 1] = EM_TW512[m * twstep];
  }
  // Padded addresses as ONE lane-dependent base + compile-time offsets (immediate offsets of the ds instructions):
  //   fpad(lane + 64 m)          = rd0 + 72 m
  //   fpad(j + q p), p = 4       = jst + 4 q               (16 (g & 1) + k + 4 q < 32: no carry into the pad)
  //                  p = 16      = jst + 16 q + 4 (q >> 1)
  //                  p = 64      = jst + 72 q
  //   fpad(256 - lane - 64 m)    = kb0 - 72 m              (the mirrored bin of the untangling step; lane 0, m = 0: bin 0)
  const int rd0 = fpad(lane);
  const int kb0 = fpad(256 - lane);
  const int st0 = fpad(4 * lane);
  float2 twu[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) twu[m] = (lane + 64 * m < 384) ? EM_TW512[lane + 64 * m] : make_float2(0.f, 0.f);
  // mel: lanes 0 .. 63 own filters 0 .. 63 (n_mels <= 64: filter `lane` if it exists); 64 < n_mels <= 80: the rest as four
  // lanes per filter, a quarter of the band each
  const int nfull = n_mels > 64 ? 64 : 0;
  const int mA = lane < n_mels ? lane : n_mels - 1;
  const int loA = mel_lo[mA];
  // (taps past the widest band carry zero weights and read zeroed words behind the 257 bins: no per-tap predicate or index
  // clamp in the frame loop; fma(p, 0, acc) = acc exactly)
  float wA[ML];
#pragma unroll
  for (int s = 0; s < ML; ++s) wA[s] = (s < mel_maxlen && loA + s <= 256) ? melw[s * n_mels + mA] : 0.f;
  const int rest = n_mels - nfull;
  (void)rest;
  const int qB = lane >> 4, mmB = nfull + (lane & 15);
  const int mB = mmB < n_mels ? mmB : n_mels - 1;
  const int per = (mel_maxlen + 3) >> 2;
  const int s0B = qB * per, s1B = (s0B + per < mel_maxlen) ? s0B + per : mel_maxlen;
  const int loB = mel_lo[mB] + s0B;
  constexpr int MLB = (ML + 3) / 4;
  float wB[MLB];
#pragma unroll
  for (int s = 0; s < MLB; ++s) wB[s] = (nfull && s0B + s < s1B && loB + s <= 256) ? melw[(s0B + s) * n_mels + mB] : 0.f;
  for (int i = 257 + lane; i < FE_POW; i += 64) pw[i] = 0.f;
  const float* const pwA = pw + loA;
  const float* const pwB = pw + (loB < FE_POW - MLB ? loB : FE_POW - MLB);
  for (int fi = 0; fi < FR; ++fi) {
    const int t = (blockIdx.x * FR + fi) * 4 + wave;
    if (t >= T_f) break;  // (wave-uniform)
    // ---- load + window: z[n] = x[2n] + i x[2n+1], n = lane + 64 m
    float2 u[4];
    const int start = t * hop - 256;
    if (start >= 0 && start + 512 <= Nb && !(((size_t)b * N + start) & 1)) {  // (wave-uniform) no reflection, 8-byte aligned
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float2 v = *(const float2*)(x + start + 2 * (lane + 64 * m));
        u[m] = make_float2(v.x * win[m].x, v.y * win[m].y);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int n = lane + 64 * m;
        int i0 = reflect_idx(start + 2 * n, Nb), i1 = reflect_idx(start + 2 * n + 1, Nb);
        i0 = i0 < 0 ? 0 : (i0 < N ? i0 : N - 1);
        i1 = i1 < 0 ? 0 : (i1 < N ? i1 : N - 1);
        u[m] = make_float2(x[i0] * win[m].x, x[i1] * win[m].y);
      }
    }
    // ---- stage p = 1 (no twiddles): out[4 lane + q], one 16-byte store per component
    bfly4(u);
    *(float4*)(re[0] + st0) = make_float4(u[0].x, u[1].x, u[2].x, u[3].x);
    *(float4*)(im[0] + st0) = make_float4(u[0].y, u[1].y, u[2].y, u[3].y);
    __builtin_amdgcn_wave_barrier();
    // ---- stages p = 4, 16, 64
    int cur = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s) {
      const int p = 1 << (2 * s);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float2 v = make_float2(re[cur][rd0 + 72 * m], im[cur][rd0 + 72 * m]);
        u[m] = (m == 0) ? v : cmul(v, tw[s - 1][m - 1]);
      }
      bfly4(u);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int off = p == 4 ? 4 * q : (p == 16 ? 16 * q + 4 * (q >> 1) : 72 * q);
        re[cur ^ 1][jst[s - 1] + off] = u[q].x;
        im[cur ^ 1][jst[s - 1] + off] = u[q].y;
      }
      cur ^= 1;
      __builtin_amdgcn_wave_barrier();
    }
    // ---- real-FFT untangle + power: k = lane + 64 m (m < 4) and k = 256 on lane 0
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const int k = lane + 64 * m;
      if (m == 4 && lane != 0) break;
      // bins ka = k & 255 and kb = (256 - k) & 255, padded
      const int pa = m < 4 ? rd0 + 72 * m : 0;
      const int pb = m == 0 ? (lane == 0 ? 0 : kb0) : (m < 4 ? kb0 - 72 * m : 0);
      const float2 zk = make_float2(re[cur][pa], im[cur][pa]);
      const float2 zc = make_float2(re[cur][pb], -im[cur][pb]);
      const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      const float2 dlt = make_float2(zk.x - zc.x, zk.y - zc.y);
      const float2 o = make_float2(0.5f * dlt.y, -0.5f * dlt.x);
      const float2 w = m < 4 ? twu[m] : EM_TW512[256];
      const float2 xo = cmul(w, o);
      const float xr = e.x + xo.x, xi = e.y + xo.y;
      pw[k] = xr * xr + xi * xi;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- banded mel + log
    const bool valid_frame = t < flen;
    float* out = feats + ((size_t)b * T_f + t) * n_mels;
    {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < ML; ++s) acc = fmaf(pwA[s], wA[s], acc);
      acc = fmaxf(acc, 1e-10f);
      if (lane < n_mels) out[lane] = valid_frame ? logf(acc) : 0.f;
    }
    if (nfull) {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < MLB; ++s) acc = fmaf(pwB[s], wB[s], acc);
      {
        const Pair2 p16 = wave_xor16_pair(acc);
        acc = p16.a + p16.b;
        const Pair2 p32 = wave_xor32_pair(acc);
        acc = p32.a + p32.b;
      }
      acc = fmaxf(acc, 1e-10f);
      if (qB == 0 && mmB < n_mels) out[mmB] = valid_frame ? logf(acc) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();  // (the next frame's stage-1 stores come after this frame's reads of the tiles)
  }
}

extern "C" int em_frontend_logmel_f32(const float* wav, int32_t B, int32_t N, int32_t hop,
                                      const float* window, const float* mel_packed,
                                      const int32_t* mel_lo, int32_t mel_maxlen, int32_t n_mels,
                                      const int32_t* flens, const int32_t* wlens, int32_t T_f,
                                      float* feats, void* stream) {
  if (B <= 0 || T_f <= 0) return EM_ERR_BAD_ARG;
  if (N <= 256 || hop <= 0 || T_f != 1 + N / hop) return EM_ERR_BAD_ARG;
  if (mel_maxlen < 1 || mel_maxlen > 257 || n_mels < 1) return EM_ERR_BAD_ARG;
  // ESPNET_AMD_FRONTEND_V1=1: developer A/B switch (one frame per wave, the kernel of rounds 1-4)
  const bool v1 = em_sw().frontend_v1;  // (the equality test toggles it in-process: em_dev_switches_reload)
  if (!v1 && mel_maxlen <= 32 && n_mels <= 80) {
    // a wave walks FR frames with its frame-independent state in registers: 8 for a batch of utterances; 2 where that would
    // leave the chip mostly idle (a streaming chunk is 67 frames: 3 workgroups per stream, 18 - 21 us of a serial walk -
    // round 6).  Frames are independent: the same bits either way.
    auto launch = [&](auto ml_c, auto fr_c) {
      constexpr int ML = decltype(ml_c)::value, FR = decltype(fr_c)::value;
      dim3 grid(em_cdiv(T_f, 4 * FR), B);
      hipLaunchKernelGGL((frontend_logmel_kernel2<ML, FR>), grid, dim3(256), 0, (hipStream_t)stream, wav, N, hop, window,
                         mel_packed, mel_lo, mel_maxlen, n_mels, flens, wlens, T_f, feats);
    };
    const bool few = (long)em_cdiv(T_f, 32) * B < 256;
    using I2 = std::integral_constant<int, 2>;
    using I8 = std::integral_constant<int, 8>;
    using I20 = std::integral_constant<int, 20>;  // (80 mel filters over 257 bins: 18)
    using I32 = std::integral_constant<int, 32>;
    if (mel_maxlen <= 20) {
      if (few) launch(I20{}, I2{});
      else launch(I20{}, I8{});
    } else {
      if (few) launch(I32{}, I2{});
      else launch(I32{}, I8{});
    }
    EM_CHECK_LAUNCH();
    return EM_OK;
  }
  dim3 grid(em_cdiv(T_f, 4), B);
  hipLaunchKernelGGL(frontend_logmel_kernel, grid, dim3(256), 0, (hipStream_t)stream, wav, N, hop,
                     window, mel_packed, mel_lo, mel_maxlen, n_mels, flens, wlens, T_f, feats);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// ---- utterance MVN: 8 partial column sums per utterance over the valid frames --------------
// grid (8, B).  The 256 threads of a block are FL = 256 / n_mels frame lanes x n_mels columns: lane fl
// walks frames t0 + fl, t0 + fl + FL, ... (coalesced over the columns), the FL lane sums meet in LDS.
// (One lane per column and a serial walk over the block's ~125 frames was pure load latency: 33 us.)
__global__ __launch_bounds__(256) void utt_mvn_partial_kernel(const float* __restrict__ feats,
                                                              const int* __restrict__ flens,
                                                              int T_f, int n_mels,
                                                              float* __restrict__ partial) {
  __shared__ float s_acc[256];
  const int b = blockIdx.y, part = blockIdx.x;
  const int len = flens[b] < T_f ? flens[b] : T_f;
  const int chunk = (len + 7) / 8;
  const int t0 = part * chunk;
  const int t1 = (t0 + chunk) < len ? (t0 + chunk) : len;
  if (n_mels > 256) {  // wide features: one lane per column group, serial over frames
    for (int m = threadIdx.x; m < n_mels; m += blockDim.x) {
      float acc = 0.f;
      for (int t = t0; t < t1; ++t) acc += feats[((size_t)b * T_f + t) * n_mels + m];
      partial[((size_t)b * 8 + part) * n_mels + m] = acc;
    }
    return;
  }
  if (n_mels % 4 == 0 && n_mels <= 256) {
    // 16-byte loads, four frames per lane in flight (round 3: the 4-byte walk below - ~42 dependent-looking loads per
    // lane - took 12 us for 10 MB; the frame order inside a lane and the lane order of the final sum are fixed, so
    // the result is deterministic)
    const int C4 = n_mels >> 2;                   // float4 columns
    const int FL4 = 256 / C4;                     // frame lanes
    const int fl = threadIdx.x / C4, c = threadIdx.x - fl * C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fl < FL4) {
      const float4* base = (const float4*)(feats + (size_t)b * T_f * n_mels) + c;
      for (int t = t0 + fl; t < t1; t += 4 * FL4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tt = t + u * FL4;
          v[u] = base[(size_t)(tt < t1 ? tt : t1 - 1) * C4];  // unconditional; frames past the end are masked below
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (t + u * FL4 < t1) {
            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
          }
      }
    }
    __shared__ float4 s_acc4[256];
    s_acc4[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < C4) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = 0; q < FL4; ++q) {
        const float4 a = s_acc4[q * C4 + threadIdx.x];
        tot.x += a.x; tot.y += a.y; tot.z += a.z; tot.w += a.w;
      }
      *(float4*)(partial + ((size_t)b * 8 + part) * n_mels + 4 * threadIdx.x) = tot;
    }
    return;
  }
  const int FL = 256 / n_mels;
  const int fl = threadIdx.x / n_mels, m = threadIdx.x - fl * n_mels;
  float acc = 0.f;
  if (fl < FL)
    for (int t = t0 + fl; t < t1; t += FL) acc += feats[((size_t)b * T_f + t) * n_mels + m];
  s_acc[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < n_mels) {
    float tot = 0.f;
    for (int q = 0; q < FL; ++q) tot += s_acc[q * n_mels + threadIdx.x];
    partial[((size_t)b * 8 + part) * n_mels + threadIdx.x] = tot;
  }
}

extern "C" int em_utt_mvn_partial_f32(const float* feats, const int32_t* flens, int32_t B,
                                      int32_t T_f, int32_t n_mels, float* partial, void* stream) {
  if (B <= 0 || T_f <= 0 || n_mels <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(utt_mvn_partial_kernel, dim3(8, B), dim3(256), 0, (hipStream_t)stream, feats,
                     flens, T_f, n_mels, partial);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// ---- first subsampling conv: Conv2d(1, d, 3, stride 2) + ReLU, MVN subtraction fused ---------
// grid (T1, B); the block stages the three (mean-subtracted) input rows in LDS and every thread
// produces all F1 outputs of its channel(s); stores are coalesced over channels (channel-last).
template <typename T>
__global__ __launch_bounds__(256) void conv2d_sub1_kernel(
    const float* __restrict__ feats, const float* __restrict__ partial,
    const int* __restrict__ flens, int T_f, int n_mels, const float* __restrict__ w1,
    const float* __restrict__ b1, int d, int T1, int F1, T* __restrict__ out) {
  __shared__ float s_in[3][128];
  const int t1 = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * n_mels; i += blockDim.x) {
    int r = i / n_mels, f = i - r * n_mels;
    float mean = 0.f;
    if (partial) {
      const float* pp = partial + (size_t)b * 8 * n_mels + f;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += pp[q * n_mels];
      mean = s / (float)flens[b];
    }
    s_in[r][f] = feats[((size_t)b * T_f + 2 * t1 + r) * n_mels + f] - mean;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w1[c * 9 + i];
    const float bias = b1[c];
    T* o = out + ((size_t)(b * T1 + t1) * F1) * d + c;
    for (int f1 = 0; f1 < F1; ++f1) {
      float acc = bias;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = fmaf(w[i * 3 + j], s_in[i][2 * f1 + j], acc);
      o[(size_t)f1 * d] = from_f32<T>(fmaxf(acc, 0.f));
    }
  }
}

extern "C" int em_conv2d_sub1(int dtype, const float* feats, const float* partial,
                              const int32_t* flens, int32_t B, int32_t T_f, int32_t n_mels,
                              const float* w1, const float* b1, int32_t d, void* out,
                              void* stream) {
  if (B <= 0 || n_mels < 3 || n_mels > 128 || d <= 0) return EM_ERR_BAD_ARG;
  if (T_f < 7) return EM_ERR_TOO_SHORT;
  const int T1 = (T_f - 3) / 2 + 1, F1 = (n_mels - 3) / 2 + 1;
  dim3 grid(T1, B);
  if (dtype == EM_F32)
    hipLaunchKernelGGL(conv2d_sub1_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, feats,
                       partial, flens, T_f, n_mels, w1, b1, d, T1, F1, (float*)out);
  else if (dtype == EM_BF16)
    hipLaunchKernelGGL(conv2d_sub1_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, feats,
                       partial, flens, T_f, n_mels, w1, b1, d, T1, F1, (bf16*)out);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// ---- GlobalMVN.forward (espnet2/layers/global_mvn.py:71-100): x <- (x - mean) / std on the valid
// frames, padded frames forced to zero; in place.  mean / stdv may be NULL (norm_means / norm_vars off).
__global__ __launch_bounds__(256) void global_mvn_kernel(float* __restrict__ x,
                                                         const int* __restrict__ flens,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, int T_f,
                                                         int n_mels) {
  const int b = blockIdx.y;
  const size_t n = (size_t)T_f * n_mels;
  float* xb = x + (size_t)b * n;
  const int len = flens ? flens[b] : T_f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / n_mels), c = (int)(i - (size_t)t * n_mels);
    float v = xb[i];
    if (mean) v -= mean[c];
    if (t >= len) v = 0.f;
    if (stdv) v /= stdv[c];
    xb[i] = v;
  }
}

extern "C" int em_global_mvn_f32(float* feats, const int32_t* flens, const float* mean,
                                 const float* stdv, int32_t B, int32_t T_f, int32_t n_mels,
                                 void* stream) {
  if (!feats || B <= 0 || T_f <= 0 || n_mels <= 0) return EM_ERR_BAD_ARG;
  int gx = (int)(((size_t)T_f * n_mels + 255) / 256);
  gx = gx > 256 ? 256 : gx;
  hipLaunchKernelGGL(global_mvn_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, feats, flens,
                     mean, stdv, T_f, n_mels);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// ---- utterance_mvn applied stand-alone (espnet2/layers/utterance_mvn.py:45-88, norm_means only):
// padded frames are zeroed first, then the mean over the valid frames is subtracted from EVERY
// frame (so padded frames become -mean, :73), in place.  partial from em_utt_mvn_partial_f32.
__global__ __launch_bounds__(256) void utt_mvn_apply_kernel(float* __restrict__ x,
                                                            const float* __restrict__ partial,
                                                            const int* __restrict__ flens, int T_f,
                                                            int n_mels) {
  const int b = blockIdx.y;
  const size_t n = (size_t)T_f * n_mels;
  float* xb = x + (size_t)b * n;
  const int len = flens[b];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / n_mels), c = (int)(i - (size_t)t * n_mels);
    const float* pp = partial + (size_t)b * 8 * n_mels + c;
    float sm = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sm += pp[q * n_mels];
    xb[i] = (t < len ? xb[i] : 0.f) - sm / (float)len;
  }
}

extern "C" int em_utt_mvn_apply_f32(float* feats, const float* partial, const int32_t* flens,
                                    int32_t B, int32_t T_f, int32_t n_mels, void* stream) {
  if (!feats || !partial || !flens || B <= 0 || T_f <= 0 || n_mels <= 0) return EM_ERR_BAD_ARG;
  int gx = (int)(((size_t)T_f * n_mels + 255) / 256);
  gx = gx > 256 ? 256 : gx;
  hipLaunchKernelGGL(utt_mvn_apply_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, feats,
                     partial, flens, T_f, n_mels);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
