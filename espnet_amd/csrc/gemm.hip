// Dense contraction C = A[M,K] * W[N,K]^T (+ fused epilogue) on the gfx950 matrix cores.
//
// One kernel template covers every GEMM-shaped op of the path (SURVEY.md §8(a) A4, A7, A8, A9,
// A11, A14): torch.nn.Linear (positionwise_feed_forward.py:30-32, attention.py:77-119, ctc.py:39),
// Conv1d(k=1) (convolution.py:28-53) and Conv2d(d,d,3,2) as an implicit GEMM over a channel-last
// feature map (subsampling.py:402-403).
//
// Structure (256 threads = 2x2 waves, BM x 128 output tile, BM in {128, 64}):
//   * operands stream HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging), double
//     buffered; loads of K-step t+1 stay in flight across the barriers of step t (counted
//     s_waitcnt vmcnt + raw s_barrier, never a draining __syncthreads);
//   * LDS rows are 128 B of K (64 bf16 / 32 f32), stored linearly as the DMA requires and
//     XOR-swizzled on the SOURCE address (16-byte chunk c of row r lands at chunk c ^ (r & 7)),
//     so ds_read_b128 fragment reads are bank-conflict free;
//   * f32 accumulate in both modes: EM_BF16 -> v_mfma_f32_16x16x32_bf16, EM_F32 ->
//     v_mfma_f32_16x16x4_f32 (exact f32 products; the parity mode);
//   * epilogue: accumulators go through a per-wave LDS transpose so global stores are 8/16-byte
//     per lane over 128-256 B contiguous row segments (bias / activation / residual fused).
#include <stdlib.h>

#include <type_traits>

#include "em_common.h"
#include "switches.h"

struct EmProfile {
  int capacity, count;
  hipEvent_t *start, *stop;
  double* flops;
  int* tag;
};
static thread_local EmProfile* tl_profile = nullptr;

// Launch bracketing shared by every MFMA kernel family (em_common.h): a start event when a profile is
// attached and has room, then the stop event with the launch's algorithmic flops and its family tag.
bool em_prof_begin(void* stream) {
  EmProfile* prof = tl_profile;
  if (!prof || prof->count >= prof->capacity) return false;
  hipEventRecord(prof->start[prof->count], (hipStream_t)stream);
  return true;
}
void em_prof_end(void* stream, double flops, int tag) {
  EmProfile* prof = tl_profile;
  hipEventRecord(prof->stop[prof->count], (hipStream_t)stream);
  prof->flops[prof->count] = flops;
  prof->tag[prof->count] = tag;
  prof->count++;
}

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS row per K-step

struct ConvGeom {
  int T1, F1, T2, F2, d;
  int kw, st;  // square kernel width and stride (3, 2: Conv2dSubsampling; 5, 3: second conv of Conv2dSubsampling6)
};

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <typename T, int EPI, int AMODE, int BM, int BN, int NS>
__global__ __launch_bounds__(256) void gemm_kernel(const T* __restrict__ A,
                                                   const T* __restrict__ W, void* __restrict__ Cv,
                                                   const float* __restrict__ bias, int M, int N,
                                                   int K, int lda, int ldc, float scale,
                                                   ConvGeom g) {
  using MM = Mma<T>;
  constexpr int BK = ROWB / (int)sizeof(T);  // K elements per step
  constexpr int KSUB = BK / MM::K;           // MFMA k-steps per K-step
  constexpr int WAVES_N = BN / 64;           // 4 waves as WAVES_M x WAVES_N, wave tile WM x 64
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M;
  constexpr int MI = WM / 16;                // 16-row fragments per wave along M
  constexpr int A_LD = BM / 32;              // glds instructions per wave for the A tile (8 rows each)
  constexpr int W_LD = BN / 32;
  constexpr int NLOADS = A_LD + W_LD;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, BUF = A_BYTES + W_BYTES;
  static_assert(NS >= 2 && (NS - 2) * NLOADS <= 63 && NS * BUF >= 4 * (BM / WAVES_M) * 64 * 4, "stages");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // NS stages of BUF bytes (the epilogue's staging after them)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WAVES_N, wc = wave % WAVES_N;
  // XCD-aware tile order.  Workgroup L runs on XCD L % 8 (each XCD has its own 4 MiB L2); tiles are
  // numbered M-major (q = m_tile * n_tiles + n_tile) and XCD x takes the contiguous chunk
  // [x * cpx, (x+1) * cpx), so the n_tiles workgroups that share an A row-panel run on the same
  // XCD at about the same time (A is fetched into one L2 once instead of into all eight).
  const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
  const int cpx = (n_tiles * m_tiles + 7) >> 3;
  const int q = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  if (q >= n_tiles * m_tiles) return;
  const int m0 = (q / n_tiles) * BM, n0 = (q % n_tiles) * BN;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- per-lane global source addresses.  Load instruction i of this wave fills LDS rows
  // (wave*LD + i)*8 .. +7; lane l supplies row (l >> 3) of those, LDS chunk (l & 7), i.e. global
  // chunk (l & 7) ^ (row & 7) = (l & 7) ^ (l >> 3).
  const int ld_row = lane >> 3;
  const int ld_chunk = (lane & 7) ^ ld_row;
  const unsigned char* a_src[A_LD];
  const unsigned char* w_src[W_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    int m = m0 + (wave * A_LD + i) * 8 + ld_row;
    m = m < M ? m : M - 1;
    size_t base;
    if (AMODE == EM_A_CONV2) {
      int per_b = g.T2 * g.F2;
      int bb = m / per_b, rem = m - bb * per_b;
      int t2 = rem / g.F2, f2 = rem - t2 * g.F2;
      base = ((size_t)(bb * g.T1 + g.st * t2) * g.F1 + g.st * f2) * g.d;
    } else {
      base = (size_t)m * lda;
    }
    a_src[i] = (const unsigned char*)(A + base) + ld_chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < W_LD; ++i) {
    int n = n0 + (wave * W_LD + i) * 8 + ld_row;
    n = n < N ? n : N - 1;
    w_src[i] = (const unsigned char*)(W + (size_t)n * K) + ld_chunk * 16;
  }

  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
    size_t koff = (size_t)k0;
    if (AMODE == EM_A_CONV2) {
      int q = k0 / g.d, c0 = k0 - q * g.d;
      int t3 = q / g.kw, f3 = q - t3 * g.kw;
      koff = (size_t)(t3 * g.F1 + f3) * g.d + c0;
    }
    unsigned char* sa = smem + buf * BUF;
    unsigned char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      glds16(a_src[i] + koff * sizeof(T), sa + (wave * A_LD + i) * 1024);
#pragma unroll
    for (int i = 0; i < W_LD; ++i)
      glds16(w_src[i] + (size_t)k0 * sizeof(T), sw + (wave * W_LD + i) * 1024);
  };

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = tile_row + lr (tile_row % 16 == 0 -> row & 7 == lr & 7)
  const int swz = lr & 7;
  const int a_row_off = (wr * WM + lr) * ROWB;
  const int w_row_off = (wc * 64 + lr) * ROWB;

  // RESID epilogue: the f32 residual tile this lane will update is requested now, so its HBM/L2
  // latency overlaps the K loop (loads retire in order: it has landed by the first tile wait).
  // Loading it inside the store loop serialises MI*4 dependent round trips (each load may alias
  // the previous store): -1.5 us per launch at M = 7968.
  constexpr bool PREFETCH_C = (EPI == EM_EPI_RESID_F32);
  float4 cpre[PREFETCH_C ? MI * 4 : 1];
  float4 bpre = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PREFETCH_C) {
    // UNCONDITIONAL (row and column clamped; em_gemm admits this epilogue only with N % 4 == 0 and ldc % 4 == 0): under a
    // branch hipcc's wait counting gives up and every later use of cpre - one per store below - waits for EVERYTHING
    // outstanding, the previous row's store included
    int ncol_p = n0 + wc * 64 + (lane & 15) * 4;
    ncol_p = ncol_p + 3 < N ? ncol_p : N - 4;
#pragma unroll
    for (int q = 0; q < MI * 4; ++q) {
      int m = m0 + wr * WM + (q >> 2) * 16 + (q & 3) * 4 + lg;
      m = m < M ? m : M - 1;
      cpre[q] = *(const float4*)((const float*)Cv + (size_t)m * ldc + ncol_p);
    }
    if (bias) bpre = *(const float4*)(bias + ncol_p);  // with the residual rows: not a round trip of its own in the epilogue
  }

  const int nk = K / BK;
  // one tile's MFMAs from stage `cur`
  auto compute = [&](int cur) {
    const unsigned char* sa = smem + cur * BUF;
    const unsigned char* sw = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      int coff;  // byte offset inside the 128-byte row
      if (sizeof(T) == 2) coff = (((ks * 4 + lg) ^ swz) << 4);
      else coff = ((ks ^ swz) << 4) + lg * 4;
      typename MM::frag fa[MI], fb[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = MM::load((const T*)(sa + a_row_off + i * 16 * ROWB + coff));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = MM::load((const T*)(sw + w_row_off + j * 16 * ROWB + coff));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fa[i], fb[j], acc[i][j]);
    }
  };
  if constexpr (NS == 2) {
    // two stages (rounds 1 and 2): the next tile is requested BEFORE the wait for the current one, so two tiles are on
    // their way while the wave waits; two barriers per tile.  (The one-barrier ring below with two stages can only
    // request a tile after the one before it has landed: measured slower on the label step's vocabulary GEMM.)
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) {
        issue(kt + 1, cur ^ 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // tile kt has landed for every wave
      compute(cur);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done reading buf[cur]
    }
  } else {
    // a ring of NS >= 3 LDS stages, NS - 1 tiles requested ahead, ONE barrier per tile: after it every wave's pieces of
    // tile kt have landed and every wave is done with tile kt - 1, whose stage is then refilled with tile kt + NS - 1.
    // (With two stages a tile's loads had one tile's MFMAs, ~270 cycles, to cross a memory system whose latency is 900:
    // the embedding GEMM, K = 4 864, spent 1 140 cycles per tile.)  The requests are unconditional - past the end the
    // last tile is requested again into a dead stage - so the wait is one constant.
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s < nk ? s : nk - 1, s);
    int cur = 0, fill = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
      // lgkmcnt(0): this wave's fragment reads of tile kt - 1 have RETURNED before it passes the barrier that lets
      // another wave's LDS-DMA refill that stage (gfx950 barriers do not wait for outstanding LDS reads themselves)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NLOADS) : "memory");
      __builtin_amdgcn_s_barrier();
      {
        const int kn = kt + NS - 1;
        issue(kn < nk ? kn : nk - 1, fill);
      }
      compute(cur);
      cur = cur + 1 == NS ? 0 : cur + 1;
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the surplus requests too: the epilogue reuses the stages)
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue.  C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r.
  const int wm0 = m0 + wr * WM, wn0 = n0 + wc * 64;
  if (EPI == EM_EPI_GLU) {
    // value/gate column pairs sit in neighbouring fragments of the same lane: direct stores.
    T* C = (T*)Cv;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      int ncol = wn0 + j * 16 + lr;  // packed column of the value half
      if (ncol >= N) continue;
      float bv = bias ? bias[ncol] : 0.f, bg = bias ? bias[ncol + 16] : 0.f;
      int ocol = (wn0 + j * 16) / 2 + lr;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = wm0 + i * 16 + lg * 4 + r;
          if (m < M) {
            float v = acc[i][j][r] + bv, gt = acc[i][j + 1][r] + bg;
            C[(size_t)m * ldc + ocol] = from_f32<T>(v * sigmoidf_(gt));
          }
        }
    }
    return;
  }
  if constexpr (EPI == EM_EPI_ARGMAX_PART) {
    // per-row (max, argmax) over this wave's 64 columns, straight from the accumulators (ties -> lowest
    // column, = torch.argmax).  Cv: [M][ldc] pairs (f32 value, i32 column); group = n0/64 + wc.
    // Each lane first reduces its 4 fragments (columns j*16 + lr), giving NVAL = MI*4 candidates per lane
    // (one per row it touches); the 16 lanes of a row group then run a reduce-SCATTER butterfly: at every
    // xor step a lane keeps half of its candidates and receives the partner's copy of exactly those, so
    // the exchange volume halves each step (8+4+2+1 pairs instead of 16 x 4) and lane lr ends up owning
    // the finished result of candidate lr.
    float2* part = (float2*)Cv;
    const int grp = (wn0 >> 6);
    constexpr int NVAL = MI * 4;
    float bv[NVAL];
    int bi[NVAL];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float best = -INFINITY;
        int idx = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = wn0 + j * 16 + lr;
          const float v = col < N ? acc[i][j][r] + (bias ? bias[col] : 0.f) : -INFINITY;
          if (v > best) {  // ascending columns: strict > keeps the lowest on ties
            best = v;
            idx = col;
          }
        }
        bv[i * 4 + r] = best;
        bi[i * 4 + r] = idx;
      }
    int owned = 0;  // candidate index this lane will own
    auto better = [](float av, int ai, float ov, int oi) { return ov > av || (ov == av && oi < ai); };
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const int off = 8 >> step;
      const bool hi = (lr & off) != 0;
      const int cur = NVAL >> step;  // candidates still held before this step (compile-time after unrolling)
      if (cur >= 2) {
        const int h = cur >> 1;
#pragma unroll
        for (int k = 0; k < NVAL / 2; ++k) {
          if (k < h) {
            const float sv = hi ? bv[k] : bv[k + h];
            const int si = hi ? bi[k] : bi[k + h];
            const float ov = __shfl_xor(sv, off, 64);
            const int oi = __shfl_xor(si, off, 64);
            float mv = hi ? bv[k + h] : bv[k];
            int mi = hi ? bi[k + h] : bi[k];
            if (better(mv, mi, ov, oi)) {
              mv = ov;
              mi = oi;
            }
            bv[k] = mv;
            bi[k] = mi;
          }
        }
        owned += hi ? h : 0;
      } else {  // a single candidate left but lanes to go: plain exchange
        const float ov = __shfl_xor(bv[0], off, 64);
        const int oi = __shfl_xor(bi[0], off, 64);
        if (better(bv[0], bi[0], ov, oi)) {
          bv[0] = ov;
          bi[0] = oi;
        }
      }
    }
    // NVAL == 16: every lane owns one row; NVAL == 8: lanes lr and lr^1 hold the same row, the even one writes
    const bool writer = NVAL >= 16 || (lr & 1) == 0;
    const int m = wm0 + (owned >> 2) * 16 + lg * 4 + (owned & 3);
    if (writer && m < M) part[(size_t)m * ldc + grp] = make_float2(bv[0], __int_as_float(bi[0]));
    return;
  }
  // per-wave LDS transpose of the whole wave tile: [BM/2 rows][64 cols] f32 (16 KiB per wave at
  // BM = 128; exactly the 64 KiB of staging LDS), 16-column groups XOR-swizzled by (row >> 2) & 3.
  // Three phases (all LDS writes -> all LDS reads -> math + global stores): hipcc orders every
  // ds access after an LDS-DMA behind vmcnt(0), so no LDS access may follow the first store.
  constexpr int WROWS = WM;
  float* ep = (float*)smem + wave * (WROWS * 64);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ep[(i * 16 + lg * 4 + r) * 64 + ((j ^ lg) << 4) + lr] = acc[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int jg = (lane & 15) >> 2, cin = (lane & 3) * 4;
  float4 vals[MI * 4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jr = 0; jr < 4; ++jr)
      vals[i * 4 + jr] = *(const float4*)(ep + (i * 16 + jr * 4 + lg) * 64 + ((jg ^ jr) << 4) + cin);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const int c4 = (lane & 15) * 4;  // 4 consecutive output columns handled by this lane
  const int ncol = wn0 + c4;
  if (ncol >= N) return;
  if constexpr (EPI == EM_EPI_QK_HEADS) {
    // q | k projections written PER HEAD for csrc/attention2.hip: column n = which * d + 64 * head + c, row m = (b, t) ->
    // C[which][b][head][t][c] with [Tpad][64] per (b, head)  (g.T1 = T, g.T2 = Tpad, g.F1 = heads, g.d = d; N = 2 d).
    // A lane's 4 columns stay inside one head (64 | head boundary): 8-byte stores through a buffer resource.
    if constexpr (sizeof(T) == 2) {
      const int Tn = g.T1, Tpad = g.T2, H = g.F1, dm = g.d;
      const size_t per_head = (size_t)(M / Tn) * dm * Tpad;
      const int which = ncol / dm, c = ncol - which * dm, head = c >> 6, cc = c & 63;
      const float4 b4 = bias ? *(const float4*)(bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)(2 * per_head * 2), 0x00020000);
      const int b0 = wm0 / Tn, r0 = wm0 - b0 * Tn;
#pragma unroll
      for (int q = 0; q < MI * 4; ++q) {
        const int dm_ = (q >> 2) * 16 + (q & 3) * 4 + lg;
        const int m = wm0 + dm_;
        int rem = r0 + dm_, bb = b0;
        while (rem >= Tn) {
          rem -= Tn;
          ++bb;
        }
        const bf16x4 pk = {(bf16)(vals[q].x + b4.x), (bf16)(vals[q].y + b4.y), (bf16)(vals[q].z + b4.z), (bf16)(vals[q].w + b4.w)};
        const size_t el = (size_t)which * per_head + ((size_t)(bb * H + head) * Tpad + rem) * 64 + cc;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rs, m < M ? (unsigned)(el * 2) : 0xffffffffu, 0, 0);
      }
    }
    return;
  }
  if constexpr (EPI == EM_EPI_VT_HEADS) {
    // V^T for csrc/attention2.hip from the SWAPPED product: A = W_v [d][K] (row m = channel), W = activations [B*T][K]
    // (column n = (b, t)) -> C[b][m][t] with Tpad frames per row (= [B][heads][64][Tpad]); bias per ROW.  T is odd in
    // general, so a lane's 4 frames are not 8-byte aligned and may straddle an utterance: 2-byte stores.
    if constexpr (sizeof(T) == 2) {
      const int Tn = g.T1, Tpad = g.T2, dm = g.d;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)((size_t)(N / Tn) * dm * Tpad * 2), 0x00020000);
      size_t colbase[4];
      bool colok[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = ncol + e;
        colok[e] = n < N;
        const int bb = (n < N ? n : N - 1) / Tn, t = (n < N ? n : N - 1) - bb * Tn;
        colbase[e] = (size_t)bb * dm * Tpad + t;
      }
#pragma unroll
      for (int q = 0; q < MI * 4; ++q) {
        const int m = wm0 + (q >> 2) * 16 + (q & 3) * 4 + lg;
        const float bm = (bias && m < M) ? bias[m] : 0.f;
        const float vv[4] = {vals[q].x, vals[q].y, vals[q].z, vals[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16 hv = (bf16)(vv[e] + bm);
          const unsigned off = (m < M && colok[e]) ? (unsigned)((colbase[e] + (size_t)m * Tpad) * 2) : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rs, off, 0, 0);
        }
      }
    }
    return;
  }
  if constexpr (EPI == EM_EPI_RESID_F32) {
    // Residual rows leave through a raw buffer resource: a row past M gets an out-of-range offset and the hardware
    // drops its store - no per-row branch.  (With `if (m < M)` around each store hipcc put an s_waitcnt vmcnt(0)
    // in front of every one of them: eight serialized store round trips in the epilogue of the decoder step's output
    // projections, 18 launches per label step; tools/isa_waits.py, round 3.)  em_gemm admits this epilogue only
    // with N % 4 == 0, ldc % 4 == 0 and M * ldc * 4 < 4 GiB.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)((size_t)M * ldc * 4), 0x00020000);
    const float4 b4 = bpre;
#pragma unroll
    for (int q = 0; q < MI * 4; ++q) {
      const int m = wm0 + (q >> 2) * 16 + (q & 3) * 4 + lg;
      float4 x = cpre[q];
      x.x += scale * (vals[q].x + b4.x); x.y += scale * (vals[q].y + b4.y);
      x.z += scale * (vals[q].z + b4.z); x.w += scale * (vals[q].w + b4.w);
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const u32x4 raw = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
      const unsigned off = m < M ? (unsigned)(((size_t)m * ldc + ncol) * 4) : 0xffffffffu;
      __builtin_amdgcn_raw_buffer_store_b128(raw, rs, off, 0, 0);
    }
    return;
  }
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    b4.x = bias[ncol];
    b4.y = ncol + 1 < N ? bias[ncol + 1] : 0.f;
    b4.z = ncol + 2 < N ? bias[ncol + 2] : 0.f;
    b4.w = ncol + 3 < N ? bias[ncol + 3] : 0.f;
  }
  const bool full4 = (ncol + 3 < N) && ((ldc & 3) == 0);
  auto activate = [&](float4 v) {
    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    if (EPI == EM_EPI_SWISH) {
      v.x = swishf_(v.x); v.y = swishf_(v.y); v.z = swishf_(v.z); v.w = swishf_(v.w);
    } else if (EPI == EM_EPI_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (EPI == EM_EPI_GELU) {
      v.x = geluf_(v.x); v.y = geluf_(v.y); v.z = geluf_(v.z); v.w = geluf_(v.w);
    } else if (EPI == EM_EPI_SCALE_F32) {
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    }
    return v;
  };
  constexpr bool OUT_ACT = EPI == EM_EPI_STORE || EPI == EM_EPI_SWISH || EPI == EM_EPI_RELU || EPI == EM_EPI_GELU;
  constexpr size_t ES = OUT_ACT ? sizeof(T) : 4;
  // Vector path (the rule: whole 4-column groups, 16-byte aligned rows, the matrix behind one 4 GiB buffer resource):
  // the MI * 4 row stores of a lane leave back to back through raw buffer stores, a row past M carrying an
  // out-of-range offset that the hardware drops.  With `if (m < M)` around every store hipcc put an
  // s_waitcnt vmcnt(0) in front of EACH of them - 16 serialized store round trips per 128-row tile, about as long
  // as the K loop of the conv2 implicit GEMM itself (tools/isa_waits.py; round 3).
  if (full4 && (size_t)M * ldc * ES < ((size_t)1 << 32) - 64) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)((size_t)M * ldc * ES), 0x00020000);
#pragma unroll
    for (int q = 0; q < MI * 4; ++q) {
      const int m = wm0 + (q >> 2) * 16 + (q & 3) * 4 + lg;
      const float4 v = activate(vals[q]);
      const unsigned off = m < M ? (unsigned)(((size_t)m * ldc + ncol) * ES) : 0xffffffffu;
      if constexpr (OUT_ACT && sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const bf16x4 pk = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rs, off, 0, 0);
      } else {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 raw = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(raw, rs, off, 0, 0);
      }
    }
    return;
  }
  // edge columns / unaligned rows / matrices beyond 4 GiB: element by element
#pragma unroll
  for (int q = 0; q < MI * 4; ++q) {
    const int m = wm0 + (q >> 2) * 16 + (q & 3) * 4 + lg;
    if (m >= M) continue;
    const float4 v = activate(vals[q]);
    const size_t o = (size_t)m * ldc + ncol;
    const float vv[4] = {v.x, v.y, v.z, v.w};
    for (int e = 0; e < 4; ++e)
      if (ncol + e < N) {
        if constexpr (OUT_ACT) ((T*)Cv)[o + e] = from_f32<T>(vv[e]);
        else ((float*)Cv)[o + e] = vv[e];
      }
  }
}

template <typename T, int EPI, int AMODE, int BM, int NS>
int launch_tile(const EmGemmArgs* p, hipStream_t s, const ConvGeom& g) {
  constexpr int BN = 128;
  constexpr int lds = NS * (BM + BN) * ROWB;
  auto* fn = gemm_kernel<T, EPI, AMODE, BM, BN, NS>;
  static EmLdsCap cap = {};
  if (lds > 64 * 1024 && em_raise_lds_cap((const void*)fn, lds, &cap) != EM_OK) return EM_ERR_LAUNCH;
  dim3 grid(8 * em_cdiv(em_cdiv(p->N, BN) * em_cdiv(p->M, BM), 8));
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, (const T*)p->A, (const T*)p->W, p->C, p->bias, p->M, p->N, p->K,
                     p->lda, p->ldc, p->scale, g);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T, int EPI, int AMODE>
int launch(const EmGemmArgs* p, hipStream_t s) {
  ConvGeom g{p->T1, p->F1, p->T2, p->F2, p->d, p->conv_k > 0 ? p->conv_k : 3, p->conv_s > 0 ? p->conv_s : 2};
  const int nb = em_cdiv(p->N, 128);
  // fewer than ~1.5 workgroups per CU at BM=128 -> halve the M tile to fill the 256 CUs
  const int force_bm = em_sw().gemm_bm;
  const bool small = force_bm ? force_bm == 64 : (long)nb * em_cdiv(p->M, 128) < 384;
  // Stages of the LDS ring.  A long K walked by at most ~two workgroups per CU has nothing else to hide a tile's memory
  // latency behind: four stages (96 / 128 KiB of LDS, one workgroup per CU) - the embedding Linear, K = 19 d: 46.0 ->
  // 40.7 us at d = 256, 82.8 -> 72.3 at d = 512 (profiles/r03t_gemm_bench.txt).  Short K (the d x d and d x 4 d
  // projections) and larger grids keep two stages and two or three co-resident workgroups per CU, which cover each
  // other's waits and cost no three-tile prologue: four stages are 25-50 % slower there.  ESPNET_AMD_GEMM_STAGES = 2 | 4:
  // developer A/B.
  const int forced = em_sw().gemm_stages;
  const long wgs = (long)nb * em_cdiv(p->M, small ? 64 : 128);
  // (round 4: K >= 4096, was 2048 - the large model's second FFN matrix, K = 2048 on 500 workgroups, runs 65.6 us with
  // four stages and ~40 with two: the whole B = 64 encoder step 8.24 -> 7.64 ms, profiles/r04b_gemm_stages_large_b64.txt)
  const bool deep = forced ? forced >= 4 : wgs <= 2 * 256 && p->K >= 4096;
  if (small) return deep ? launch_tile<T, EPI, AMODE, 64, 4>(p, s, g) : launch_tile<T, EPI, AMODE, 64, 2>(p, s, g);
  return deep ? launch_tile<T, EPI, AMODE, 128, 4>(p, s, g) : launch_tile<T, EPI, AMODE, 128, 2>(p, s, g);
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
    case EM_EPI_GELU: return launch<T, EM_EPI_GELU, EM_A_PLAIN>(p, s);
    case EM_EPI_RESID_F32: return launch<T, EM_EPI_RESID_F32, EM_A_PLAIN>(p, s);
    case EM_EPI_SCALE_F32: return launch<T, EM_EPI_SCALE_F32, EM_A_PLAIN>(p, s);
    case EM_EPI_GLU: return launch<T, EM_EPI_GLU, EM_A_PLAIN>(p, s);
    case EM_EPI_STORE_F32: return launch<T, EM_EPI_STORE_F32, EM_A_PLAIN>(p, s);
    case EM_EPI_ARGMAX_PART: return launch<T, EM_EPI_ARGMAX_PART, EM_A_PLAIN>(p, s);
    case EM_EPI_QK_HEADS:
      if (sizeof(T) != 2 || p->T1 <= 0 || p->T2 < p->T1 || p->F1 <= 0 || p->d != p->F1 * 64 || p->N != 2 * p->d ||
          p->M % p->T1 != 0 || !p->bias || (size_t)(p->M / p->T1) * p->d * p->T2 * 4 >= ((size_t)1 << 32) - 64)
        return EM_ERR_UNSUPPORTED;
      return launch<T, EM_EPI_QK_HEADS, EM_A_PLAIN>(p, s);
    case EM_EPI_VT_HEADS:
      if (sizeof(T) != 2 || p->T1 <= 0 || p->T2 < p->T1 || p->d != p->M || p->N % p->T1 != 0 ||
          (size_t)(p->N / p->T1) * p->d * p->T2 * 2 >= ((size_t)1 << 32) - 64)
        return EM_ERR_UNSUPPORTED;
      return launch<T, EM_EPI_VT_HEADS, EM_A_PLAIN>(p, s);
  }
  return EM_ERR_BAD_ARG;
}

}  // namespace

int em_gemm_skinny(int dtype, int epilogue, const EmGemmArgs* p, void* stream);  // gemm_skinny.hip
int em_gemm_mid(int dtype, int epilogue, const EmGemmArgs* p, void* stream);     // gemm_mid.hip

extern "C" int em_gemm(int dtype, int epilogue, int a_mode, const EmGemmArgs* p, void* stream) {
  if (!p || !p->A || !p->W || !p->C) return EM_ERR_BAD_ARG;
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return EM_ERR_BAD_ARG;
  const int bk = dtype == EM_BF16 ? 64 : 32;
  if (p->K % bk != 0) return EM_ERR_UNSUPPORTED;
  if (epilogue == EM_EPI_GLU && (p->N % 32 != 0)) return EM_ERR_UNSUPPORTED;
  if (epilogue == EM_EPI_RESID_F32 && (p->N % 4 != 0 || p->ldc % 4 != 0)) return EM_ERR_UNSUPPORTED;  // 16-byte residual rows
  if (epilogue == EM_EPI_RESID_F32 && a_mode == EM_A_PLAIN && (size_t)p->M * p->ldc * 4 >= ((size_t)1 << 31)) {
    // the residual rows sit behind ONE buffer resource (31-bit offsets): a taller matrix runs as row slabs, each a
    // launch of its own over its own resource (rows are independent; a very large batch must not turn into an error)
    const size_t esz = dtype == EM_BF16 ? 2 : 4;
    const int slab = (int)((((size_t)1 << 31) - 64) / ((size_t)p->ldc * 4));
    if (slab <= 0) return EM_ERR_UNSUPPORTED;
    for (int m0 = 0; m0 < p->M; m0 += slab) {
      EmGemmArgs q = *p;
      q.M = p->M - m0 < slab ? p->M - m0 : slab;
      q.A = (const char*)p->A + (size_t)m0 * p->lda * esz;
      q.C = (char*)p->C + (size_t)m0 * p->ldc * 4;
      const int rc = em_gemm(dtype, epilogue, a_mode, &q, stream);
      if (rc != EM_OK) return rc;
    }
    return EM_OK;
  }
  if (epilogue == EM_EPI_RESID_F32 && (size_t)p->M * p->ldc * 4 >= ((size_t)1 << 31)) return EM_ERR_UNSUPPORTED;
  if (a_mode == EM_A_CONV2) {
    const int kw = p->conv_k > 0 ? p->conv_k : 3;
    if (p->d <= 0 || p->d % bk != 0 || p->K != kw * kw * p->d) return EM_ERR_UNSUPPORTED;
  } else if (a_mode != EM_A_PLAIN) {
    return EM_ERR_BAD_ARG;
  } else if (p->lda % (dtype == EM_BF16 ? 8 : 4) != 0) {
    return EM_ERR_UNSUPPORTED;  // 16-byte aligned rows
  }
  const bool rec = em_prof_begin(stream);
  int rc = EM_ERR_UNSUPPORTED;
  const bool no_mid = em_sw().no_mid_gemm;  // developer A/B switch (tools/gemm_bench.py)
  // very few rows (streaming encoder step, single-utterance beam): latency-bound weight streaming
  if (a_mode == EM_A_PLAIN && p->M <= 48) rc = em_gemm_skinny(dtype, epilogue, p, stream);
  // a few hundred rows x a few hundred columns (the residual projections of a beam-search label step): the tiled grid
  // would be a dozen workgroups walking K as a chain of dependent round trips; gemm_mid.hip splits K inside the workgroup
  else if (a_mode == EM_A_PLAIN && !no_mid && (long)em_cdiv(p->M, 64) * em_cdiv(p->N, 128) < 96 &&
           (long)em_cdiv(p->M, 32) * em_cdiv(p->N, 32) <= 2048)
    rc = em_gemm_mid(dtype, epilogue, p, stream);
  if (rc == EM_ERR_UNSUPPORTED) {
    rc = EM_ERR_BAD_ARG;
    if (dtype == EM_F32) rc = dispatch<float>(epilogue, a_mode, p, (hipStream_t)stream);
    else if (dtype == EM_BF16) rc = dispatch<bf16>(epilogue, a_mode, p, (hipStream_t)stream);
  }
  if (rec) em_prof_end(stream, 2.0 * (double)p->M * (double)p->N * (double)p->K, EM_PROF_GEMM);
  return rc;
}

// ---- optional per-launch timing of the GEMM family (bench.py's roofline leg) -----------------
extern "C" EmProfile* em_profile_create(int32_t capacity) {
  if (capacity <= 0) return nullptr;
  EmProfile* pr = new EmProfile();
  pr->capacity = capacity;
  pr->count = 0;
  pr->start = new hipEvent_t[capacity];
  pr->stop = new hipEvent_t[capacity];
  pr->flops = new double[capacity];
  pr->tag = new int[capacity];
  for (int i = 0; i < capacity; ++i) {
    hipEventCreate(&pr->start[i]);
    hipEventCreate(&pr->stop[i]);
  }
  return pr;
}

extern "C" void em_profile_destroy(EmProfile* pr) {
  if (!pr) return;
  if (tl_profile == pr) tl_profile = nullptr;
  for (int i = 0; i < pr->capacity; ++i) {
    hipEventDestroy(pr->start[i]);
    hipEventDestroy(pr->stop[i]);
  }
  delete[] pr->start;
  delete[] pr->stop;
  delete[] pr->flops;
  delete[] pr->tag;
  delete pr;
}

extern "C" void em_profile_attach(EmProfile* pr) { tl_profile = pr; }

extern "C" int em_profile_read2(EmProfile* pr, float* ms, double* flops, int32_t* tags, int32_t max_n,
                                int32_t* count) {
  if (!pr || !ms || !flops || !count) return EM_ERR_BAD_ARG;
  int n = pr->count < max_n ? pr->count : max_n;
  if (n > 0) hipEventSynchronize(pr->stop[n - 1]);
  for (int i = 0; i < n; ++i) {
    if (hipEventElapsedTime(&ms[i], pr->start[i], pr->stop[i]) != hipSuccess) return EM_ERR_LAUNCH;
    flops[i] = pr->flops[i];
    if (tags) tags[i] = pr->tag[i];
  }
  *count = n;
  pr->count = 0;
  return EM_OK;
}

extern "C" int em_profile_read(EmProfile* pr, float* ms, double* flops, int32_t max_n, int32_t* count) {
  return em_profile_read2(pr, ms, flops, nullptr, max_n, count);
}
