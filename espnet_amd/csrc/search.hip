// Label-synchronous joint CTC/attention beam search, batched over utterances, device resident.
//
// Reference semantics being reproduced (paths relative to espnet/espnet):
//   BeamSearch.__init__ / forward        espnet2/legacy/nets/beam_search.py:36-126, 385-498
//   BatchBeamSearch.search / batch_beam / post_process
//                                        espnet2/legacy/nets/batch_beam_search.py:98-122, 253-423
//   CTCPrefixScoreTH.__call__            espnet2/legacy/nets/ctc_prefix_score.py:71-191
//   CTCPrefixScorer.select_state         espnet2/legacy/nets/scorers/ctc.py:40-63
//   LengthBonus.batch_score              espnet2/legacy/nets/scorers/length_bonus.py:39-58
//   end_detect                           espnet2/legacy/nets/e2e_asr_common.py:14-44
//
// Layout.  B utterances x W beam slots = n rows (row = b*W + k).  Hypotheses are paths in a token
// tree: tok[pos][row] / parent[pos][row]; `anc[row][pos]` lists the slot of every prefix position
// (what the decoder's self-attention reads).  A row is alive or not; ended hypotheses are
// appended to a per-utterance list (node position/slot + scores) and rebuilt on the host by
// walking `parent`.  One step = pre-beam top-S on the full scorers -> CTC prefix scores of the
// S candidates (+ <eos>) -> per-utterance top-W -> state update (tree, anc, scores, CTC forward
// variables of the winners, ended list, end detection).  No host synchronisation inside a step.
//
// CTC prefix scoring: a candidate's log psi is a reduction over frames of terms built from the
// PREFIX's forward variables, so it is computed by one wave per (row, candidate) with lanes over
// frames; the forward-variable recurrence itself (sequential in t) runs only for the W winners
// per utterance, in the update kernel.  CTC log-probs are kept transposed ([V][B*T]).
#include <math.h>
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

constexpr float LOGZERO = -10000000000.0f;  // ctc_prefix_score.py:34
constexpr float D_END = -10.0f;             // log(1 * exp(-10)), e2e_asr_common.py:14

__device__ __forceinline__ float logaddexp_(float a, float b) {
  // torch.logsumexp over two elements: max + log(exp(a-max) + exp(b-max)); one of the two
  // exponentials is exactly 1.  The forward variables reach magnitudes of 1e2..1e3, where an f32
  // ulp (6e-5 at 1e3) dwarfs the error of the hardware exp/log approximations used here.
  const float m = fmaxf(a, b);
  return m + __logf(1.0f + __expf(-fabsf(a - b)));
}

// CTC log-probs are held TRANSPOSED, lpT[v][b*T + t] (em_search_init), so that the T frames of
// one (utterance, label) pair are contiguous: a wave reads them with one coalesced load per lane
// and a thread walking them sequentially stays inside a few cache lines / one page.
//
// log psi of a candidate needs NO recurrence: log_psi = logsumexp_t(log_phi[t-1] + x[t]) (+)
// r[start-1, 0] where log_phi comes from the PREFIX's forward variables (ctc_prefix_score.py
// :135-144, :166-181).  It is a pure reduction over t -> one wave per (row, candidate), lanes over
// frames (`candidate_kernel`).  Only the W winners per utterance need the sequential recurrence
// r[t] (:158-164), which yields the next step's r_prev (`ctc_state_scan`).

struct Ctx {
  EmSearchParams p;
  EmSearchBuffers b;
};

// frame stride of the CTC tables (ctc_lpT rows, r_a / r_b rows): the visible length T offline, the
// fixed capacity ldT of a stream whose visible length grows block by block
__host__ __device__ inline int ldt(const EmSearchParams& p) { return p.ldT > 0 ? p.ldT : p.T; }

// ---- init ---------------------------------------------------------------------------------
__global__ void search_init_rows_kernel(Ctx c) {
  const int n = c.p.B * c.p.W;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const bool first = (r % c.p.W) == 0;
  c.b.alive[r] = first ? 1 : 0;
  c.b.run_score[r] = first ? 0.f : -INFINITY;
  c.b.run_sdec[r] = 0.f;
  c.b.run_sctc[r] = 0.f;
  c.b.run_slen[r] = 0.f;
  if (c.b.run_slm) c.b.run_slm[r] = 0.f;
  c.b.s_prev[r] = 0.f;
  c.b.tok[r] = c.p.sos;
  c.b.parent[r] = -1;
  for (int j = 0; j < c.p.Lmax; ++j) {
    c.b.anc_a[(size_t)r * c.p.Lmax + j] = r;
    c.b.anc_b[(size_t)r * c.p.Lmax + j] = r;
  }
}

// r_prev of the empty prefix: (logzero, cumsum_t logp[t][blank])  (ctc_prefix_score.py:89-96)
__global__ void search_init_utt_kernel(Ctx c) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= c.p.B) return;
  c.b.end_count[b] = 0;
  c.b.done[b] = 0;
  c.b.best_all[b] = -INFINITY;
  for (int l = 0; l < c.p.Lmax + 2; ++l) c.b.best_by_len[(size_t)b * (c.p.Lmax + 2) + l] = -INFINITY;
  if (c.p.w_ctc != 0.f) {
    const int LT = ldt(c.p);
    const float* xb = c.b.ctc_lpT + (size_t)c.p.blank * c.p.B * LT + (size_t)b * LT;
    float2* r0 = (float2*)c.b.r_a + (size_t)(b * c.p.W) * LT;
    float cum = 0.f;
    for (int t = 0; t < c.p.T; ++t) {
      cum += xb[t];
      r0[t] = make_float2(LOGZERO, cum);
    }
  }
}

// ---- steps 1 + 2: decoder log-softmax + pre-beam + candidate totals, one 256-thread workgroup per row -----------
// logits -> log-probs in place (transformer_decoder.py:233), then the top-S token ids of the
// weighted full scores w_dec*logp + w_len (batch_beam_search.py:289-302), S rounds of block
// arg-max over register-resident values (ties -> lowest id).  NV values per thread: V <= 256*NV.
//
// Round 3.  (a) The arg-max rounds reduce on DPP (wave_allmax_dpp / wave_allmin_dpp, em_common.h): with __shfl_xor
// every round was a chain of twelve dependent ds_bpermute round trips, S rounds per wave and S more for the merge -
// 30 us per label step for 5 000 numbers (profiles/r03f_search_kernel_stats.csv).  (b) The candidate totals of the
// row's S + 1 slots (candidate_kernel below: CTC log psi of every pre-beam candidate, batch_beam_search.py:289-314)
// are computed HERE, by the workgroup that has just chosen the candidates: one launch less per label step, the phi
// terms of the row's prefix are built once in LDS for all slots instead of once per slot, and a slot is worked by
// one 16-lane row (four slots per wave in lockstep, their log-prob columns requested UNT frames deep).
constexpr int PREBEAM_SMAX = 128;  // pre-beam width of the fused row kernel (beam_size <= 85)
constexpr int CTC_TMAX = 2048;

template <int NV>
__global__ __launch_bounds__(256) void logsoftmax_prebeam_kernel(Ctx c, int i_host) {
  __shared__ float s_v[4];
  __shared__ float m_v[4 * PREBEAM_SMAX];
  __shared__ int m_i[4 * PREBEAM_SMAX];
  __shared__ float s_cf[PREBEAM_SMAX];  // the row's pre-beam: weighted full score
  __shared__ int s_ct[PREBEAM_SMAX];    //                     token id
  __shared__ float s_eosfull;
  __shared__ float s_phi_same[CTC_TMAX], s_phi_diff[CTC_TMAX];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int V = c.p.V, S = c.p.S, NC = c.p.NC, W = c.p.W;
  const int i = c.b.step ? *c.b.step : i_host;
  const bool with_cand = S < V;  // pre-beam mode: this kernel also produces the candidate totals
  if (with_cand && i >= c.p.Lmax - 1) return;
  const int b = r / W;
  if (!c.b.alive[r] || (with_cand && c.b.done[b])) {
    if (with_cand)
      for (int s = tid; s < NC; s += 256) c.b.cand_total[(size_t)r * NC + s] = -INFINITY;
    return;
  }
  // log-softmax of one logits row in place (block-wide), returning this thread's NV log-probs
  auto row_log_softmax = [&](float* lp, float (&out)[NV]) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = tid + 256 * k;
      out[k] = v < V ? lp[v] : -INFINITY;
      mx = fmaxf(mx, out[k]);
    }
    mx = wave_allmax_dpp(mx);
    if (lane == 0) s_v[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) sum += (tid + 256 * k < V) ? __expf(out[k] - mx) : 0.f;  // (v_exp_f32: ~1 ulp; the sum of 5 000 terms rounds more)
    sum = wave_allsum_dpp(sum);
    if (lane == 0) s_v[wave] = sum;
    __syncthreads();
    const float lse = mx + logf(s_v[0] + s_v[1] + s_v[2] + s_v[3]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = tid + 256 * k;
      if (v < V) {
        out[k] -= lse;
        lp[v] = out[k];
      }
    }
  };
  // weighted full scores in the scorer order of the reference (decoder, length_bonus, lm;
  // batch_beam_search.py:289-290)
  float w[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) w[k] = (tid + 256 * k < V) ? 0.f : -INFINITY;
  if (c.p.w_dec != 0.f) {
    float l[NV];
    row_log_softmax(c.b.dec_logp + (size_t)r * V, l);
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (tid + 256 * k < V) w[k] = c.p.w_dec * l[k];
  }
  if (c.p.w_len != 0.f) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (tid + 256 * k < V) w[k] += c.p.w_len * 1.0f;
  }
  if (c.p.w_lm != 0.f) {
    float l[NV];
    row_log_softmax(c.b.lm_logp + (size_t)r * V, l);
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (tid + 256 * k < V) w[k] += c.p.w_lm * l[k];
  }
  if (S >= V) return;
  // the <eos> slot's full score (always scored, :186-187), before the rounds knock values out
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (tid + 256 * k == c.p.eos) s_eosfull = w[k];
  // each wave extracts the top-S of its own 64*NV values with wave-level ops only (no block
  // barrier inside the rounds); the 4*S survivors are merged by wave 0.
  // Fast path: a thread first sorts out its own three largest values (one scan of its NV registers); a round then
  // compares ONE value per lane (the head of that list) and pops the winner's list - ~45 instructions per round
  // instead of ~190 for the rounds that rescan all NV registers.  A thread whose list runs empty might still own a
  // value the wave needs: the wave then redoes its rounds the exact way (wave-uniform, rare: three of a wave's top S
  // in one thread's NV strided values).  Same order as the exact rounds: largest value, lowest id on ties (ids ascend
  // within a thread and a later equal value never displaces an earlier one).
  const int SS = S < PREBEAM_SMAX ? S : PREBEAM_SMAX;
  {
    float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
    int i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const float x = w[q];
      const int ix = tid + 256 * q;
      const bool g1 = x > v1, g2 = x > v2, g3 = x > v3;
      v3 = g2 ? v2 : (g3 ? x : v3);
      i3 = g2 ? i2 : (g3 ? ix : i3);
      v2 = g1 ? v1 : (g2 ? x : v2);
      i2 = g1 ? i1 : (g2 ? ix : i2);
      v1 = g1 ? x : v1;
      i1 = g1 ? ix : i1;
    }
    int pops = 0;
    bool exact = false;
    for (int k = 0; k < SS; ++k) {
      const float wb = wave_allmax_dpp(v1);
      const int wi = wave_allmin_dpp(v1 == wb ? i1 : 0x7fffffff);  // lowest id among the holders of the maximum
      if (lane == 0) {
        m_v[wave * SS + k] = wb;
        m_i[wave * SS + k] = wi;
      }
      const bool mine = (wi == i1) && (wi != 0x7fffffff);
      if (mine) {
        v1 = v2; i1 = i2;
        v2 = v3; i2 = i3;
        v3 = -INFINITY; i3 = 0x7fffffff;
        ++pops;
      }
      if (__any(pops == 3 && k + 1 < SS)) {  // (its fourth value is unknown: the list cannot answer the next round)
        exact = true;
        break;
      }
    }
    if (exact) {
      for (int k = 0; k < SS; ++k) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < NV; ++q)
          if (w[q] > best) {  // ascending ids within a thread: strict > keeps the lowest
            best = w[q];
            bi = tid + 256 * q;
          }
        const float wb = wave_allmax_dpp(best);
        const int wi = wave_allmin_dpp(best == wb ? bi : 0x7fffffff);
        if (lane == 0) {
          m_v[wave * SS + k] = wb;
          m_i[wave * SS + k] = wi;
        }
#pragma unroll
        for (int q = 0; q < NV; ++q)
          if (tid + 256 * q == wi) w[q] = -INFINITY;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    auto emit = [&](int k, float wb, int wi) {
      if (lane == 0) {
        c.b.cand_tok[(size_t)r * NC + k] = wi;
        c.b.cand_full[(size_t)r * NC + k] = wb;
        s_ct[k] = wi;
        s_cf[k] = wb;
      }
    };
    if (4 * SS <= 64) {  // one survivor per lane (beam <= 10): a round is two reductions and a select
      float cv = lane < 4 * SS ? m_v[lane] : -INFINITY;
      const int ci = lane < 4 * SS ? m_i[lane] : 0x7fffffff;
      for (int k = 0; k < SS; ++k) {
        const float wb = wave_allmax_dpp(cv);
        const int wi = wave_allmin_dpp(cv == wb ? ci : 0x7fffffff);
        emit(k, wb, wi);
        if (ci == wi) cv = -INFINITY;
      }
    } else {
      constexpr int NQ = 4 * PREBEAM_SMAX / 64;
      float cv[NQ];
      int ci[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j = lane + 64 * q;
        cv[q] = j < 4 * SS ? m_v[j] : -INFINITY;
        ci[q] = j < 4 * SS ? m_i[j] : 0x7fffffff;
      }
      for (int k = 0; k < SS; ++k) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (cv[q] > best || (cv[q] == best && ci[q] < bi)) {
            best = cv[q];
            bi = ci[q];
          }
        const float wb = wave_allmax_dpp(best);
        const int wi = wave_allmin_dpp(best == wb ? bi : 0x7fffffff);
        emit(k, wb, wi);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (ci[q] == wi) cv[q] = -INFINITY;
      }
    }
  }
  // ---- candidate totals of the row's NC = S + 1 slots: slots 0..S-1 = the pre-beam, slot S = <eos>
  //      total = (w_dec*dec + w_len + w_lm*lm) + w_ctc*(psi - s_prev) + running score   (batch_beam_search.py:289-314)
  const bool ctc = c.p.w_ctc != 0.f;
  const int xlen = ctc ? c.b.xlens[b] : 0;
  const int LT = ldt(c.p);
  const float2* rprev = (const float2*)((i & 1) ? c.b.r_b : c.b.r_a) + (size_t)r * LT;
  if (ctc) {  // phi of the prefix, both forms (ctc_prefix_score.py:135-144), once for all slots
    for (int t = tid; t < xlen; t += 256) {
      const float2 rp = rprev[t];
      s_phi_same[t] = rp.y;
      s_phi_diff[t] = logaddexp_(rp.x, rp.y);
    }
  }
  __syncthreads();
  const int lr = lane & 15, q4 = lane >> 4;
  const int last_tok = c.b.tok[(size_t)i * c.p.B * W + r];
  const float base = c.b.run_score[r], sprev = ctc ? c.b.s_prev[r] : 0.f;
  constexpr int UNT = 8;
  for (int s0 = 0; s0 < NC; s0 += 16) {
    const int s = s0 + wave * 4 + q4;       // this 16-lane row's slot
    const bool slot_ok = s < NC;
    const bool eos_slot = s >= S;
    const int tokc = (slot_ok && !eos_slot) ? s_ct[s] : c.p.eos;
    float full = (slot_ok && !eos_slot) ? s_cf[s] : s_eosfull;
    bool dup = false;  // <eos> already among the pre-beam candidates: that slot carries it
    if (eos_slot)
      for (int k = 0; k < S; ++k) dup |= (s_ct[k] == tokc);
    float total = full;
    float psi = 0.f;
    if (ctc) {
      // log psi = logsumexp_t(phi[t-1] + x[t]) (+) r[start-1, 0]  (:166-181), lanes of the row over frames
      const bool same = tokc == last_tok;
      const float* ph = same ? s_phi_same : s_phi_diff;
      const float* xc = c.b.ctc_lpT + (size_t)tokc * c.p.B * LT + (size_t)b * LT;
      const int start = i > 1 ? i : 1;
      float m = -INFINITY, sm = 0.f;
      if (lr == 0) {  // the r[start-1, 0] term (:176-178): x[0] for the empty prefix, else logzero
        m = (i == 0) ? xc[0] : LOGZERO;
        sm = 1.f;
      }
      for (int tb = start; tb < xlen; tb += 16 * UNT) {  // (wave-uniform trip count)
        float xv[UNT];
#pragma unroll
        for (int u = 0; u < UNT; ++u) {  // unconditional requests from clamped frames
          const int t = tb + lr + 16 * u;
          xv[u] = xc[t < xlen ? t : xlen - 1];
        }
#pragma unroll
        for (int u = 0; u < UNT; ++u) {
          const int t = tb + lr + 16 * u;
          if (t < xlen) {
            const float term = ph[t - 1] + xv[u];
            const float mm = fmaxf(m, term);
            sm = sm * __expf(m - mm) + __expf(term - mm);
            m = mm;
          }
        }
      }
      // the 16 lanes of the row combine (DPP, all-reduce inside the row)
      float M = m;
      M = fmaxf(M, dpp_f32<DPP_XOR1>(M));
      M = fmaxf(M, dpp_f32<DPP_XOR2>(M));
      M = fmaxf(M, dpp_f32<DPP_HALF_MIRROR>(M));
      M = fmaxf(M, dpp_f32<DPP_MIRROR>(M));
      sm = (m > -INFINITY) ? sm * __expf(m - M) : 0.f;
      sm += dpp_f32<DPP_XOR1>(sm);
      sm += dpp_f32<DPP_XOR2>(sm);
      sm += dpp_f32<DPP_HALF_MIRROR>(sm);
      sm += dpp_f32<DPP_MIRROR>(sm);
      psi = M + logf(sm);
      if (tokc == c.p.blank && c.p.eos != c.p.blank) psi = LOGZERO;        // :188-190
      else if (tokc == c.p.eos) psi = s_phi_diff[xlen - 1];                // :184-186
      total = total + c.p.w_ctc * (psi - sprev);
    }
    total = dup ? -INFINITY : total + base;
    if (slot_ok && lr == 0) {
      if (eos_slot) c.b.cand_tok[(size_t)r * NC + s] = tokc;
      if (ctc && !dup) c.b.cand_psi[(size_t)r * NC + s] = psi;
      c.b.cand_total[(size_t)r * NC + s] = total;
    }
  }
}

// generic-vocabulary fallback of the pre-beam (after em_log_softmax_rows_f32): one wave per row,
// row values staged in LDS.
__global__ __launch_bounds__(64) void prebeam_kernel(Ctx c) {
  extern __shared__ float vals[];
  const int r = blockIdx.x, lane = threadIdx.x;
  const int V = c.p.V, S = c.p.S, NC = c.p.NC;
  if (!c.b.alive[r]) return;
  const float* lp = c.b.dec_logp + (size_t)r * V;
  for (int v = lane; v < V; v += 64) vals[v] = c.p.w_dec * lp[v] + c.p.w_len * 1.0f;
  __syncthreads();
  for (int k = 0; k < S; ++k) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      const float x = vals[v];
      if (x > best) {
        best = x;
        bi = v;
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
    if (lane == 0) {
      c.b.cand_tok[(size_t)r * NC + k] = bi;
      c.b.cand_full[(size_t)r * NC + k] = best;
      vals[bi] = -INFINITY;
    }
    __syncthreads();
  }
}

// log psi of one (prefix, label) pair by one wave, lanes over frames (ctc_prefix_score.py:166-181):
// logsumexp_t(log_phi[t-1] + x[t]) (+) r[start-1, 0].  rprev = the prefix's forward variables, xc = the
// label's log-prob column, same = label equals the prefix's last label (:135-144), i = prefix length.
__device__ __forceinline__ float ctc_psi_wave(const float2* __restrict__ rprev, const float* __restrict__ xc,
                                              bool same, int i, int xlen, int lane) {
  const int start = i > 1 ? i : 1;
  // lane-local online logsumexp over its frames, then a wave combine
  float m = -INFINITY, sm = 0.f;
  if (lane == 0) {  // the r[start-1, 0] term (:176-178): x[0] for the empty prefix, else logzero
    m = (i == 0) ? xc[0] : LOGZERO;
    sm = 1.f;
  }
  for (int t = start + lane; t < xlen; t += 64) {
    const float2 rp = rprev[t - 1];
    const float phi = same ? rp.y : logaddexp_(rp.x, rp.y);  // :135-144
    const float term = phi + xc[t];
    const float mm = fmaxf(m, term);
    sm = sm * expf(m - mm) + expf(term - mm);
    m = mm;
  }
  const float M = wave_max(m);
  sm = (m > -INFINITY) ? sm * expf(m - M) : 0.f;
  sm = wave_sum(sm);
  return M + logf(sm);
}

// ---- step 2: candidate totals.  One wave per (row, candidate slot), 4 waves per block ----------
//   pre-beam mode  : slots 0..S-1 = cand_tok, slot S = <eos> (always scored, :186-187)
//   all-vocab mode : slot s = token s (S == V, NC == V)
// total = (w_dec*dec + w_len) + w_ctc*(psi - s_prev) + running score   (batch_beam_search.py:289-314)
__global__ __launch_bounds__(256) void candidate_kernel(Ctx c, int i_host) {
  const int i = c.b.step ? *c.b.step : i_host;
  if (i >= c.p.Lmax - 1) return;
  const int NC = c.p.NC, V = c.p.V, S = c.p.S;
  const int lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= (long)c.p.B * c.p.W * NC) return;
  const int r = (int)(idx / NC), s = (int)(idx - (long)r * NC);
  const int b = r / c.p.W;
  float* out_total = c.b.cand_total + (size_t)r * NC + s;
  if (!c.b.alive[r] || c.b.done[b]) {
    if (lane == 0) *out_total = -INFINITY;
    return;
  }
  const bool allv = (S >= V);
  int tokc;
  float full;
  // weighted sum of the full scorers for token t, in the reference's scorer order
  auto full_score = [&](int t) {
    float f = 0.f;
    if (c.p.w_dec != 0.f) f = c.p.w_dec * c.b.dec_logp[(size_t)r * V + t];
    if (c.p.w_len != 0.f) f += c.p.w_len * 1.0f;
    if (c.p.w_lm != 0.f) f += c.p.w_lm * c.b.lm_logp[(size_t)r * V + t];
    return f;
  };
  if (allv) {
    tokc = s;
    full = full_score(s);
    if (lane == 0) c.b.cand_tok[(size_t)r * NC + s] = s;
  } else if (s < S) {
    tokc = c.b.cand_tok[(size_t)r * NC + s];
    full = c.b.cand_full[(size_t)r * NC + s];
  } else {  // the extra <eos> slot
    tokc = c.p.eos;
    full = full_score(tokc);
    if (lane == 0) c.b.cand_tok[(size_t)r * NC + s] = tokc;
    // already among the pre-beam candidates: that slot carries it
    bool dup = false;
    for (int k = lane; k < S; k += 64) dup |= (c.b.cand_tok[(size_t)r * NC + k] == tokc);
    if (__any(dup)) {
      if (lane == 0) *out_total = -INFINITY;
      return;
    }
  }
  float total = full;
  if (c.p.w_ctc != 0.f) {
    const int xlen = c.b.xlens[b];
    const int LT = ldt(c.p);
    const float2* rprev = (const float2*)(i & 1 ? c.b.r_b : c.b.r_a) + (size_t)r * LT;
    float psi;
    if (tokc == c.p.blank && c.p.eos != c.p.blank) {
      psi = LOGZERO;  // :188-190
    } else if (tokc == c.p.eos) {
      const float2 re = rprev[xlen - 1];
      psi = logaddexp_(re.x, re.y);  // :184-186
    } else {
      const bool same = (tokc == c.b.tok[(size_t)i * c.p.B * c.p.W + r]);
      const float* xc = c.b.ctc_lpT + (size_t)tokc * c.p.B * LT + (size_t)b * LT;
      psi = ctc_psi_wave(rprev, xc, same, i, xlen, lane);
    }
    if (lane == 0) c.b.cand_psi[(size_t)r * NC + s] = psi;
    total = total + c.p.w_ctc * (psi - c.b.s_prev[r]);
  }
  if (lane == 0) *out_total = total + c.b.run_score[r];
}

// ---- step 3: per-utterance top-W over the W*NC candidate totals (batch_beam :98-122) -----------
// one wave per utterance; W rounds of arg-max, ties -> lowest flat index (row-major slot order)
// The register-resident form (one wave, up to 64 * SEL_KM candidates: beam 10 x 16 = 160, beam 20 x 31 = 620 ...): ONE
// global round trip, W rounds of arg-max on DPP (until round 3 the rounds were __shfl_xor chains - twelve dependent
// ds_bpermute round trips per round, 9.7 us per label step).  Largest total, lowest flat index on ties.  s_sel / s_tot
// (LDS, may be NULL) receive the winners for a caller that goes on in the same workgroup.
constexpr int SEL_KM = 16;
// WITH_REC: the winners' token and log psi are handed over too (s_tok / s_psi), read with the totals in the same
// round trip: the caller's per-row records then need no cand_tok -> dec_logp chain of dependent loads.
template <int KM, bool WITH_REC>
__device__ __forceinline__ void select_wave(const Ctx& c, int b, int lane, int* s_sel, float* s_tot, int* s_tok,
                                            float* s_psi) {
  const int W = c.p.W, total = W * c.p.NC;
  const float* tot = c.b.cand_total + (size_t)b * total;
  float x[KM], xp[KM];
  int xt[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    const int v = lane + 64 * k;
    const int vc = v < total ? v : total - 1;  // unconditional loads, masked afterwards
    const float t = tot[vc];
    if constexpr (WITH_REC) {
      xt[k] = c.b.cand_tok[(size_t)b * total + vc];
      xp[k] = c.b.cand_psi ? c.b.cand_psi[(size_t)b * total + vc] : 0.f;
    }
    x[k] = v < total ? t : -INFINITY;
  }
  for (int r = 0; r < W; ++r) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (x[k] > best) {
        best = x[k];
        bi = lane + 64 * k;
      }
    const float wb = wave_allmax_dpp(best);
    const int wi = wave_allmin_dpp(best == wb ? bi : 0x7fffffff);
    if (lane == 0) {
      const int sel = (wb > -INFINITY) ? wi : -1;
      c.b.sel_idx[b * W + r] = sel;
      c.b.sel_total[b * W + r] = wb;
      if (s_sel) {
        s_sel[r] = sel;
        s_tot[r] = wb;
      }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (wi == lane + 64 * k) {  // the winner leaves its owner's registers
        x[k] = -INFINITY;
        if constexpr (WITH_REC) {
          s_tok[r] = xt[k];
          s_psi[r] = xp[k];
        }
      }
  }
}
__global__ __launch_bounds__(64) void select_kernel(Ctx c) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int W = c.p.W, NC = c.p.NC;
  const int total = W * NC;
  float* tot = c.b.cand_total + (size_t)b * total;
  if (total <= 64 * 4) {
    select_wave<4, false>(c, b, lane, nullptr, nullptr, nullptr, nullptr);
    return;
  }
  if (total <= 64 * SEL_KM) {
    select_wave<SEL_KM, false>(c, b, lane, nullptr, nullptr, nullptr, nullptr);
    return;
  }
  for (int k = 0; k < W; ++k) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < total; v += 64) {
      const float x = tot[v];
      if (x > best) {
        best = x;
        bi = v;
      }
    }
    const float wb = wave_allmax_dpp(best);
    const int wi = wave_allmin_dpp(best == wb ? bi : 0x7fffffff);
    best = wb;
    bi = wi;
    if (lane == 0) {
      c.b.sel_idx[b * W + k] = (best > -INFINITY) ? bi : -1;
      c.b.sel_total[b * W + k] = best;
      if (best > -INFINITY) tot[bi] = -INFINITY;
    }
    __syncthreads();
  }
}

// ---- step 3b: CTC forward variables of the winners (scorers/ctc.py:54-62) ---------------------
// One 64-thread workgroup per new row (prefix of row `prow` + label tk).  The recurrence
// r[t] = logsumexp(..) (ctc_prefix_score.py:158-164) is sequential in t, so everything that is
// not on the chain is taken off it: all lanes stage x[t][tk], x[t][blank] and the phi terms
// (:135-144, computed in parallel from the prefix's r_prev) in LDS with coalesced reads, lane 0
// walks the chain LDS -> LDS, all lanes write the result back coalesced.  Only t >= max(i,1)-1
// is produced: a later step's recurrence starts at t = i+1 and reads r_prev[t-1].
// Forward variables r[t] = (r^n, r^b) of prefix + label (ctc_prefix_score.py:131-132, 158-164) by one
// 64-thread workgroup: all lanes stage x[t][label], x[t][blank] and the phi terms in LDS with coalesced
// reads, lane 0 walks the sequential chain LDS -> LDS, all lanes write the result back coalesced.  Only
// t >= max(i,1)-1 is produced: a later step's recurrence starts at t = i+1 and reads r_prev[t-1].
//
// Round 3: the chain is no longer walked by one lane.  In the log semiring the step t is LINEAR in (r^n, r^b, 1):
//     r^n[t] = logsum(x_n[t] + r^n[t-1], (x_n[t] + phi[t]) + 0)        r^b[t] = logsum(x_b[t] + r^n[t-1], x_b[t] + r^b[t-1])
// i.e. a matrix M_t = [[a, -, c], [d, e, f], [-, -, 0]] with a = x_n, c = x_n + phi, d = e = x_b, f = "zero", and
// products of such matrices keep that shape (five numbers).  So: every lane composes the steps of its own chunk of
// ceil(T / 64) frames, an inclusive scan over the 64 lanes composes the chunks (6 shuffle rounds), and every lane
// re-walks its chunk from the state the scan hands it, writing its frames.  ~4 + 6 + 4 dependent compositions
// instead of ~250 dependent steps on one lane (23 us per label step before).  The result differs from the
// sequential walk only by f32 reassociation of log-sums (the tests' score tolerance is 2e-3 + 2e-5 |score|).
struct CtcMap {
  float a, c, d, e, f;
};
__device__ __forceinline__ CtcMap ctc_compose(const CtcMap& hi, const CtcMap& lo) {  // hi after lo
  CtcMap m;
  m.a = hi.a + lo.a;
  m.c = logaddexp_(hi.a + lo.c, hi.c);
  m.d = logaddexp_(hi.d + lo.a, hi.e + lo.d);
  m.e = hi.e + lo.e;
  m.f = logaddexp_(logaddexp_(hi.d + lo.c, hi.e + lo.f), hi.f);
  return m;
}
// WG_SYNC: the 64 threads are a workgroup of their own (__syncthreads); otherwise one wave of a larger workgroup working
// in LDS arrays private to it: DS operations of a wave execute in issue order, so only the compiler has to be kept from
// reordering across the hand-over points.
template <bool WG_SYNC>
__device__ __forceinline__ void ctc_sync() {
  if constexpr (WG_SYNC) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
template <bool WG_SYNC>
__device__ __forceinline__ void ctc_chain_wg(const float2* __restrict__ rprev, const float* __restrict__ xc,
                                             const float* __restrict__ xb, bool same, int i, int xlen,
                                             float2* __restrict__ rout, int lane, float* s_xn, float* s_xb,
                                             float* s_phi, float2* s_out) {
  const int start = i > 1 ? i : 1;
  for (int t = start + lane; t < xlen; t += 64) {
    s_xn[t] = xc[t];
    s_xb[t] = xb[t];
    const float2 rp = rprev[t - 1];
    s_phi[t] = same ? rp.y : logaddexp_(rp.x, rp.y);
  }
  ctc_sync<WG_SYNC>();
  const int steps = xlen - start;
  const int chunk = (steps + 63) >> 6;
  const int t_lo = start + lane * chunk;
  const int t_hi = t_lo + chunk < xlen ? t_lo + chunk : xlen;
  CtcMap m = {0.f, LOGZERO, LOGZERO, 0.f, LOGZERO};  // identity
  for (int t = t_lo; t < t_hi; ++t) {
    const CtcMap st = {s_xn[t], s_xn[t] + s_phi[t], s_xb[t], s_xb[t], LOGZERO};
    m = ctc_compose(st, m);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    CtcMap lo;
    lo.a = __shfl_up(m.a, o, 64);
    lo.c = __shfl_up(m.c, o, 64);
    lo.d = __shfl_up(m.d, o, 64);
    lo.e = __shfl_up(m.e, o, 64);
    lo.f = __shfl_up(m.f, o, 64);
    if (lane >= o) m = ctc_compose(m, lo);
  }
  // exclusive prefix = what happened before this lane's chunk
  CtcMap pre;
  pre.a = __shfl_up(m.a, 1, 64);
  pre.c = __shfl_up(m.c, 1, 64);
  pre.d = __shfl_up(m.d, 1, 64);
  pre.e = __shfl_up(m.e, 1, 64);
  pre.f = __shfl_up(m.f, 1, 64);
  const float rn0 = (i == 0) ? xc[0] : LOGZERO, rb0 = LOGZERO;  // r[0,0] = x[0] (:131-132)
  float rn = rn0, rb = rb0;
  if (lane > 0) {
    rn = logaddexp_(pre.a + rn0, pre.c);
    rb = logaddexp_(logaddexp_(pre.d + rn0, pre.e + rb0), pre.f);
  }
  if (lane == 0 && start - 1 < xlen) s_out[start - 1] = make_float2(rn0, rb0);  // (i >= xlen: empty chain, slot out of range)
  for (int t = t_lo; t < t_hi; ++t) {
    const float nn = logaddexp_(rn, s_phi[t]) + s_xn[t];
    const float nb = logaddexp_(rn, rb) + s_xb[t];
    rn = nn;
    rb = nb;
    s_out[t] = make_float2(rn, rb);
  }
  ctc_sync<WG_SYNC>();
  for (int t = start - 1 + lane; t < xlen; t += 64) rout[t] = s_out[t];
}
__global__ __launch_bounds__(64) void ctc_state_kernel(Ctx c, int i_host) {
  const int i = c.b.step ? *c.b.step : i_host;
  if (i >= c.p.Lmax - 1) return;
  __shared__ float s_xn[CTC_TMAX], s_xb[CTC_TMAX], s_phi[CTC_TMAX];
  __shared__ float2 s_out[CTC_TMAX];
  const int rnew = blockIdx.x, lane = threadIdx.x;
  const int W = c.p.W, NC = c.p.NC, n = c.p.B * c.p.W;
  const int b = rnew / W;
  if (c.b.done[b]) return;
  const int sel = c.b.sel_idx[rnew];
  if (sel < 0) return;
  const int pk = sel / NC, slot = sel - pk * NC;
  const int prow = b * W + pk;
  const int tk = c.b.cand_tok[(size_t)prow * NC + slot];
  if (tk == c.p.eos || i == c.b.maxlens[b] - 1) return;  // ended: no state needed
  const int xlen = c.b.xlens[b];
  const int LT = ldt(c.p);
  const size_t BT = (size_t)c.p.B * LT;
  const float* xc = c.b.ctc_lpT + (size_t)tk * BT + (size_t)b * LT;
  const float* xb = c.b.ctc_lpT + (size_t)c.p.blank * BT + (size_t)b * LT;
  const float2* rprev = (const float2*)((i & 1) ? c.b.r_b : c.b.r_a) + (size_t)prow * LT;
  float2* rout = (float2*)((i & 1) ? c.b.r_a : c.b.r_b) + (size_t)rnew * LT;
  const bool same = (tk == c.b.tok[(size_t)i * n + prow]);
  ctc_chain_wg<true>(rprev, xc, xb, same, i, xlen, rout, lane, s_xn, s_xb, s_phi, s_out);
}

// ---- step 4: build the new rows (batch_beam_search.py:317-357 + post_process :359-423) ---------
// one workgroup (64 threads) per utterance
// Graph mode (c.b.step != NULL): this is the LAST kernel of a label step and it also advances the device step counter:
// every workgroup has read the step index by the time it takes a ticket at step[1], and the workgroup that takes
// the last one resets the ticket and adds one to step[0] (a launch of its own until round 3: 4.5 us per step).
__device__ __forceinline__ void step_advance(const Ctx& c, int lane) {
  if (!c.b.step || lane != 0) return;
  const unsigned t = __hip_atomic_fetch_add((unsigned*)c.b.step + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (t == gridDim.x - 1) {
    __hip_atomic_store((unsigned*)c.b.step + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add((unsigned*)c.b.step, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(64) void update_kernel(Ctx c, int i_host) {
  const int i = c.b.step ? *c.b.step : i_host;
  if (i >= c.p.Lmax - 1) {
    step_advance(c, threadIdx.x);
    return;
  }
  const int b = blockIdx.x, lane = threadIdx.x;
  const int W = c.p.W, NC = c.p.NC, V = c.p.V, Lmax = c.p.Lmax, n = c.p.B * c.p.W;
  __shared__ int s_prev_row[64], s_tok[64], s_valid[64], s_end[64];
  const int* anc_old = (i & 1) ? c.b.anc_b : c.b.anc_a;
  int* anc_new = (i & 1) ? c.b.anc_a : c.b.anc_b;
  const float2* r_old = (const float2*)((i & 1) ? c.b.r_b : c.b.r_a);
  float2* r_new = (float2*)((i & 1) ? c.b.r_a : c.b.r_b);
  const bool was_done = c.b.done[b] != 0;
  const int maxlen = c.b.maxlens[b], minlen = c.b.minlens[b];

  float n_score = -INFINITY, n_sdec = 0.f, n_sctc = 0.f, n_slen = 0.f, n_sprev = 0.f, n_slm = 0.f;
  if (lane < W) {
    const int rnew = b * W + lane;
    const int sel = was_done ? -1 : c.b.sel_idx[rnew];
    int valid = sel >= 0, prow = b * W, tk = c.p.eos;
    if (valid) {
      const int pk = sel / NC, s = sel - pk * NC;
      prow = b * W + pk;
      tk = c.b.cand_tok[(size_t)prow * NC + s];
      n_score = c.b.sel_total[rnew];
      if (c.p.w_dec != 0.f) n_sdec = c.b.run_sdec[prow] + c.b.dec_logp[(size_t)prow * V + tk];
      if (c.p.w_len != 0.f) n_slen = c.b.run_slen[prow] + 1.0f;
      if (c.p.w_lm != 0.f) n_slm = c.b.run_slm[prow] + c.b.lm_logp[(size_t)prow * V + tk];
      if (c.p.w_ctc != 0.f) {
        const float psi = c.b.cand_psi[(size_t)prow * NC + s];
        n_sctc = c.b.run_sctc[prow] + (psi - c.b.s_prev[prow]);
        n_sprev = psi;  // select_state: s = log_psi[i, new_id] (scorers/ctc.py:56)
      }
    }
    s_prev_row[lane] = prow;
    s_tok[lane] = tk;
    s_valid[lane] = valid;
    // ended: the new token is <eos>, or <eos> is forced at the last position (:393-410)
    s_end[lane] = valid && (tk == c.p.eos || i == maxlen - 1);
  }
  __syncthreads();
  // (a) ancestor tables of the new rows
  for (int k = 0; k < W; ++k) {
    if (!s_valid[k]) continue;
    const int rnew = b * W + k;
    const int* src = anc_old + (size_t)s_prev_row[k] * Lmax;
    int* dst = anc_new + (size_t)rnew * Lmax;
    for (int j = lane; j <= i; j += 64) dst[j] = src[j];
    if (lane == 0 && i + 1 < Lmax) dst[i + 1] = rnew;
  }
  __syncthreads();  // every read of the old per-row state is done
  if (lane < W) {
    const int rnew = b * W + lane;
    if (s_valid[lane]) {
      c.b.tok[(size_t)(i + 1) * n + rnew] = s_tok[lane];
      c.b.parent[(size_t)(i + 1) * n + rnew] = s_prev_row[lane];
    }
    const int alive = s_valid[lane] && !s_end[lane];
    c.b.alive[rnew] = alive;
    c.b.run_score[rnew] = alive ? n_score : -INFINITY;
    c.b.run_sdec[rnew] = n_sdec;
    c.b.run_sctc[rnew] = n_sctc;
    c.b.run_slen[rnew] = n_slen;
    if (c.b.run_slm) c.b.run_slm[rnew] = n_slm;
    c.b.s_prev[rnew] = n_sprev;
    // stash for the serial ended-list pass
    c.b.sel_total[rnew] = n_score;
  }
  __syncthreads();
  if (lane == 0 && !was_done) {
    // (c) ended list, in row order (the reference appends in batch order)
    int cnt = c.b.end_count[b];
    const int cap = c.p.end_cap;
    int n_alive = 0;
    for (int k = 0; k < W; ++k) {
      const int rnew = b * W + k;
      if (!s_valid[k]) continue;
      if (!s_end[k]) {
        ++n_alive;
        continue;
      }
      if (i < minlen) continue;  // :417-418
      const int forced = (i == maxlen - 1);
      const int ylen = i + 2 + forced;  // <sos> + (i+1) tokens [+ forced <eos>]
      const float sc = c.b.sel_total[rnew];
      if (cnt < cap) {
        const size_t e = (size_t)b * cap + cnt;
        c.b.end_pos[e] = i + 1;
        c.b.end_slot[e] = rnew;
        c.b.end_forced[e] = forced;
        c.b.end_score[e] = sc;
        c.b.end_sdec[e] = c.b.run_sdec[rnew];
        c.b.end_sctc[e] = c.b.run_sctc[rnew];
        c.b.end_slen[e] = c.b.run_slen[rnew];
        if (c.b.end_slm) c.b.end_slm[e] = c.b.run_slm[rnew];
        ++cnt;
      }
      if (sc > c.b.best_all[b]) c.b.best_all[b] = sc;
      float* bl = c.b.best_by_len + (size_t)b * (Lmax + 2) + ylen;
      if (sc > *bl) *bl = sc;
    }
    c.b.end_count[b] = cnt;
    // (d) end detection (e2e_asr_common.py:14-44) and "no hypothesis" (beam_search.py:446-448)
    int done = (n_alive == 0);
    if (!done && c.p.use_end_detect && cnt > 0) {
      int count = 0;
      for (int m = 0; m < 3; ++m) {
        const int len = i - m;
        if (len < 0) continue;
        const float v = c.b.best_by_len[(size_t)b * (Lmax + 2) + len];
        if (v > -INFINITY && v - c.b.best_all[b] < D_END) ++count;
      }
      if (count == 3) done = 1;
    }
    if (done) c.b.done[b] = 1;
  }
  step_advance(c, lane);
}

// ---- steps 3 + 3b + 4 in ONE launch (round 3): per-utterance selection, the winners' CTC forward variables and the
// new rows.  One workgroup per utterance, one wave per beam slot.  Until round 3 these were three dependent launches
// (select 9.7 us, ctc_state 6.7 us, update 12 us per label step): the selection's rounds were ds_bpermute chains (now
// DPP, select_wave), and update copied the W ancestor rows one after the other in one wave - W x ceil(i / 64) dependent
// load -> store round trips - where here wave k copies row k while it also walks row k's CTC chain.
// Same arithmetic, same order of the ended list as the three kernels (which stay for W > 16, long memories whose
// chains do not fit the LDS W at a time, and the streaming search, which commits rows under the host's control).
__global__ __launch_bounds__(1024) void tail_kernel(Ctx c, int i_host, int lt_cap) {
  extern __shared__ float tail_lds[];  // per wave: s_xn, s_xb, s_phi [lt_cap] + s_out [lt_cap] float2
  __shared__ int s_prev_row[64], s_tok[64], s_valid[64], s_end[64], s_sel[64];
  __shared__ float s_seltot[64], s_rec[4][64], s_wpsi[64];
  __shared__ int s_wtok[64];
  const int i = c.b.step ? *c.b.step : i_host;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (i >= c.p.Lmax - 1) {
    step_advance(c, tid);
    return;
  }
  const int b = blockIdx.x;
  const int W = c.p.W, NC = c.p.NC, V = c.p.V, Lmax = c.p.Lmax, n = c.p.B * c.p.W;
  const int* anc_old = (i & 1) ? c.b.anc_b : c.b.anc_a;
  int* anc_new = (i & 1) ? c.b.anc_a : c.b.anc_b;
  const bool was_done = c.b.done[b] != 0;
  const int maxlen = c.b.maxlens[b], minlen = c.b.minlens[b];
  const int cnt0 = c.b.end_count[b];  // (read at entry: not one more dependent round trip at the end)
  // ---- selection (batch_beam :98-122): wave 0
  if (wave == 0) {
    if (W * NC <= 64 * 4) select_wave<4, true>(c, b, lane, s_sel, s_seltot, s_wtok, s_wpsi);
    else select_wave<SEL_KM, true>(c, b, lane, s_sel, s_seltot, s_wtok, s_wpsi);
  }
  __syncthreads();
  // ---- the new rows' records (batch_beam_search.py:317-357), wave 0, a lane per row
  float n_score = -INFINITY, n_sdec = 0.f, n_sctc = 0.f, n_slen = 0.f, n_sprev = 0.f, n_slm = 0.f;
  if (wave == 0 && lane < W) {
    const int sel = was_done ? -1 : s_sel[lane];
    int valid = sel >= 0, prow = b * W, tk = c.p.eos;
    if (valid) {
      const int pk = sel / NC;
      prow = b * W + pk;
      tk = s_wtok[lane];  // (= cand_tok[prow][slot], handed over by the selection with the totals)
      n_score = s_seltot[lane];
      if (c.p.w_dec != 0.f) n_sdec = c.b.run_sdec[prow] + c.b.dec_logp[(size_t)prow * V + tk];
      if (c.p.w_len != 0.f) n_slen = c.b.run_slen[prow] + 1.0f;
      if (c.p.w_lm != 0.f) n_slm = c.b.run_slm[prow] + c.b.lm_logp[(size_t)prow * V + tk];
      if (c.p.w_ctc != 0.f) {
        const float psi = s_wpsi[lane];
        n_sctc = c.b.run_sctc[prow] + (psi - c.b.s_prev[prow]);
        n_sprev = psi;  // select_state: s = log_psi[i, new_id] (scorers/ctc.py:56)
      }
    }
    s_prev_row[lane] = prow;
    s_tok[lane] = tk;
    s_valid[lane] = valid;
    // ended: the new token is <eos>, or <eos> is forced at the last position (:393-410)
    s_end[lane] = valid && (tk == c.p.eos || i == maxlen - 1);
  }
  __syncthreads();  // every read of the old per-row scalars is done (wave 0 made them all)
  // ---- wave k: row k's ancestor table and CTC forward variables (scorers/ctc.py:54-62)
  if (wave < W && s_valid[wave]) {
    const int k = wave, rnew = b * W + k, prow = s_prev_row[k], tk = s_tok[k];
    const int* src = anc_old + (size_t)prow * Lmax;
    int* dst = anc_new + (size_t)rnew * Lmax;
    for (int j = lane; j <= i; j += 64) dst[j] = src[j];
    if (lane == 0 && i + 1 < Lmax) dst[i + 1] = rnew;
    if (c.p.w_ctc != 0.f && !s_end[k]) {  // ended rows need no state
      const int xlen = c.b.xlens[b];
      const int LT = ldt(c.p);
      const size_t BT = (size_t)c.p.B * LT;
      const float* xc = c.b.ctc_lpT + (size_t)tk * BT + (size_t)b * LT;
      const float* xb = c.b.ctc_lpT + (size_t)c.p.blank * BT + (size_t)b * LT;
      const float2* rprev = (const float2*)((i & 1) ? c.b.r_b : c.b.r_a) + (size_t)prow * LT;
      float2* rout = (float2*)((i & 1) ? c.b.r_a : c.b.r_b) + (size_t)rnew * LT;
      const bool same = (tk == c.b.tok[(size_t)i * n + prow]);
      float* base = tail_lds + (size_t)wave * 5 * lt_cap;
      ctc_chain_wg<false>(rprev, xc, xb, same, i, xlen, rout, lane, base, base + lt_cap, base + 2 * lt_cap,
                          (float2*)(base + 3 * lt_cap));
    }
  }
  // ---- wave 0: the rows' scalars, the ended list, end detection (as update_kernel)
  if (wave == 0) {
    if (lane < W) {
      const int rnew = b * W + lane;
      if (s_valid[lane]) {
        c.b.tok[(size_t)(i + 1) * n + rnew] = s_tok[lane];
        c.b.parent[(size_t)(i + 1) * n + rnew] = s_prev_row[lane];
      }
      const int alive = s_valid[lane] && !s_end[lane];
      c.b.alive[rnew] = alive;
      c.b.run_score[rnew] = alive ? n_score : -INFINITY;
      c.b.run_sdec[rnew] = n_sdec;
      c.b.run_sctc[rnew] = n_sctc;
      c.b.run_slen[rnew] = n_slen;
      if (c.b.run_slm) c.b.run_slm[rnew] = n_slm;
      c.b.s_prev[rnew] = n_sprev;
      c.b.sel_total[rnew] = n_score;  // (what update_kernel leaves there)
      s_seltot[lane] = n_score;
      s_rec[0][lane] = n_sdec;
      s_rec[1][lane] = n_sctc;
      s_rec[2][lane] = n_slen;
      s_rec[3][lane] = n_slm;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // The serial pass below is needed only when a row ended, when nobody is left alive, or when end detection has
    // something to compare (an ended list): on every other step (all of them, for a beam that runs to maxlen) it is
    // skipped - it cost ~10 us per label step when it ran unconditionally (profiles/r03i: timing attribution).
    const bool lanerow = lane < W;
    const unsigned long long endm = __ballot(lanerow && s_end[lanerow ? lane : 0]);
    const unsigned long long alivem = __ballot(lanerow && s_valid[lanerow ? lane : 0] && !s_end[lanerow ? lane : 0]);
    const bool need_tail = endm != 0ull || alivem == 0ull || (c.p.use_end_detect && cnt0 > 0);
    if (lane == 0 && !was_done && need_tail) {
      // ended list, in row order (the reference appends in batch order)
      int cnt = cnt0;
      const int cap = c.p.end_cap;
      int n_alive = 0;
      for (int k = 0; k < W; ++k) {
        const int rnew = b * W + k;
        if (!s_valid[k]) continue;
        if (!s_end[k]) {
          ++n_alive;
          continue;
        }
        if (i < minlen) continue;  // :417-418
        const int forced = (i == maxlen - 1);
        const int ylen = i + 2 + forced;  // <sos> + (i+1) tokens [+ forced <eos>]
        const float sc = s_seltot[k];
        if (cnt < cap) {
          const size_t e = (size_t)b * cap + cnt;
          c.b.end_pos[e] = i + 1;
          c.b.end_slot[e] = rnew;
          c.b.end_forced[e] = forced;
          c.b.end_score[e] = sc;
          c.b.end_sdec[e] = s_rec[0][k];
          c.b.end_sctc[e] = s_rec[1][k];
          c.b.end_slen[e] = s_rec[2][k];
          if (c.b.end_slm) c.b.end_slm[e] = s_rec[3][k];
          ++cnt;
        }
        if (sc > c.b.best_all[b]) c.b.best_all[b] = sc;
        float* bl = c.b.best_by_len + (size_t)b * (Lmax + 2) + ylen;
        if (sc > *bl) *bl = sc;
      }
      c.b.end_count[b] = cnt;
      // end detection (e2e_asr_common.py:14-44) and "no hypothesis" (beam_search.py:446-448)
      int done = (n_alive == 0);
      if (!done && c.p.use_end_detect && cnt > 0) {
        int count = 0;
        for (int m = 0; m < 3; ++m) {
          const int len = i - m;
          if (len < 0) continue;
          const float v = c.b.best_by_len[(size_t)b * (Lmax + 2) + len];
          if (v > -INFINITY && v - c.b.best_all[b] < D_END) ++count;
        }
        if (count == 3) done = 1;
      }
      if (done) c.b.done[b] = 1;
    }
  }
  step_advance(c, tid);
}

// x[v][col] <- log_softmax over v of (x[v][col] + bias[v]), columns = (utterance, frame) pairs.
// 64 columns x 4 vocabulary slices per block; coalesced over columns.
__global__ __launch_bounds__(256) void col_logsoftmax_kernel(float* __restrict__ x,
                                                             const float* __restrict__ bias, int V,
                                                             int ncol, int ld) {
  __shared__ float s_m[4][64], s_s[4][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float m = -INFINITY, sm = 0.f;
  if (col < ncol)
    for (int v = g; v < V; v += 4) {
      const float val = x[(size_t)v * ld + col] + (bias ? bias[v] : 0.f);
      const float mm = fmaxf(m, val);
      sm = sm * expf(m - mm) + expf(val - mm);
      m = mm;
    }
  s_m[g][cl] = m;
  s_s[g][cl] = sm;
  __syncthreads();
  float M = fmaxf(fmaxf(s_m[0][cl], s_m[1][cl]), fmaxf(s_m[2][cl], s_m[3][cl]));
  float S = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) S += s_m[q][cl] > -INFINITY ? s_s[q][cl] * expf(s_m[q][cl] - M) : 0.f;
  const float lse = M + logf(S);
  if (col < ncol)
    for (int v = g; v < V; v += 4) {
      const size_t o = (size_t)v * ld + col;
      x[o] = x[o] + (bias ? bias[v] : 0.f) - lse;
    }
}


inline int gemm(int dtype, int epi, const void* A, const void* W, void* C, const float* bias, int M,
                int N, int K, int lda, int ldc, float scale, void* stream) {
  EmGemmArgs a = {};
  a.A = A; a.W = W; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.scale = scale;
  a.T1 = a.F1 = a.T2 = a.F2 = a.d = 0;
  return em_gemm(dtype, epi, EM_A_PLAIN, &a, stream);
}

#define EM_TRY(expr)                \
  do {                              \
    int rc__ = (expr);              \
    if (rc__ != EM_OK) return rc__; \
  } while (0)

constexpr float LN_EPS = 1e-12f;

// C = epi(LN(x) W^T + bias) for the n rows of a search step.  Measured at K = 512 (tools/ln_gemm_bench.py):
// the fused kernel takes 8-9 us against 9.4-9.7 us for LayerNorm + tiled GEMM (two launches at the ~4.7 us
// floor) while its grid stays within one wave of workgroups; for wide outputs (vocabulary) or many rows
// the pair wins (10.0 vs 11.8 us at N = 5000), so those keep it.
inline int ln_proj(int dtype, int epi, const float* x, const float* g, const float* be, const void* W,
                   const float* bias, void* C, void* xn_scratch, int n, int N, int K, void* stream) {
  const long wgs = (long)em_cdiv(N, 64) * em_cdiv(n, 32);
  // n <= 48 (one stream): LayerNorm + the skinny GEMM is 6.6 us, the fused kernel 7.5
  const int wide = em_sw().lng_wide;  // developer A/B switch
  if (n > 48 && wgs <= wide && K % 64 == 0 && K <= 1024)
    return em_ln_gemm(dtype, epi, x, g, be, LN_EPS, W, bias, C, n, N, K, N, stream);
  EM_TRY(em_layernorm(dtype, x, g, be, n, K, LN_EPS, xn_scratch, nullptr, stream));
  return gemm(dtype, epi, xn_scratch, W, C, bias, n, N, K, K, N, 1.f, stream);
}

// ---- SequentialRNNLM (LSTM) step pieces (espnet2/lm/seq_rnn_lm.py:140-177 batch_score) -------
// States of the rows of step i live in ring slot i % 3: a step reads its parents' states from slot (i-1) % 3.
// Three slots, not two: the streaming search computes step i, may then rewind to step i-1
// (batch_beam_search_online.py:484-487) and recompute it, which needs the states of step i-2 intact.
// hin[l][r] = h of row r's PARENT after the previous step (zero state at step 0, :155-156)
template <typename T>
__global__ void rnn_gather_kernel(Ctx c, int i_host, int layers, int ld) {
  const int i = c.b.step ? *c.b.step : i_host;
  if (i >= c.p.Lmax - 1) return;
  const int n = c.p.B * c.p.W;
  const int r = blockIdx.x, l = blockIdx.y;
  int p = i > 0 ? c.b.parent[(size_t)i * n + r] : r;
  p = (p < 0 || p >= n) ? r : p;  // rows that never lived carry no parent
  const T* src = (const T*)c.b.rnn_hs + (((size_t)((i + 2) % 3) * layers + l) * n + p) * ld;
  T* dst = (T*)c.b.rnn_hin + ((size_t)l * n + r) * ld;
  for (int ch = threadIdx.x; ch < ld; ch += blockDim.x) dst[ch] = i > 0 ? src[ch] : from_f32<T>(0.f);
}

// Recurrent cell of torch.nn.{LSTM,GRU,RNN} on the summed gate pre-activations `gates` (f32, W_ih x + W_hh h
// + b from the two GEMMs).  The f32 master state lives in rnn_cs (c for the LSTM, h for the GRU); the
// activation-type copy of h in rnn_hs feeds the next step's GEMM.
//   LSTM      gates [n][4*nhid] = i | f | g | o:    c' = s(f) c + s(i) tanh(g),  h' = s(o) tanh(c')
//   GRU       gates [n][4*nhid] = r | z | n_x | n_h (the candidate's input and hidden parts kept apart by
//             zero weight blocks, see EmRnnLayer):  n = tanh(n_x + s(r) n_h),  h' = (1 - s(z)) n + s(z) h
//   RNN       gates [n][nhid]:                        h' = tanh(g) | relu(g)
template <typename T, int KIND>
__global__ void rnn_cell_kernel(Ctx c, int i_host, int layers, int l, int nhid, int ld) {
  const int i = c.b.step ? *c.b.step : i_host;
  if (i >= c.p.Lmax - 1) return;
  const int n = c.p.B * c.p.W;
  const int r = blockIdx.x;
  int p = i > 0 ? c.b.parent[(size_t)i * n + r] : r;
  p = (p < 0 || p >= n) ? r : p;
  constexpr int G = (KIND == EM_LM_LSTM || KIND == EM_LM_GRU) ? 4 : 1;
  const float* g = c.b.rnn_gates + (size_t)r * G * nhid;
  const float* sprev = c.b.rnn_cs + (((size_t)((i + 2) % 3) * layers + l) * n + p) * ld;
  float* snew = c.b.rnn_cs + (((size_t)(i % 3) * layers + l) * n + r) * ld;
  T* hnew = (T*)c.b.rnn_hs + (((size_t)(i % 3) * layers + l) * n + r) * ld;
  T* hout = (T*)c.b.rnn_hin + ((size_t)l * n + r) * ld;  // A operand of the next GEMM (fixed address)
  for (int ch = threadIdx.x; ch < nhid; ch += blockDim.x) {
    float hh;
    if constexpr (KIND == EM_LM_LSTM) {
      const float gi = 1.f / (1.f + expf(-g[ch])), gf = 1.f / (1.f + expf(-g[nhid + ch]));
      const float gg = tanhf(g[2 * nhid + ch]), go = 1.f / (1.f + expf(-g[3 * nhid + ch]));
      const float cc = gf * (i > 0 ? sprev[ch] : 0.f) + gi * gg;
      hh = go * tanhf(cc);
      snew[ch] = cc;
    } else if constexpr (KIND == EM_LM_GRU) {
      const float gr = 1.f / (1.f + expf(-g[ch])), gz = 1.f / (1.f + expf(-g[nhid + ch]));
      const float nn = tanhf(g[2 * nhid + ch] + gr * g[3 * nhid + ch]);
      hh = (1.f - gz) * nn + gz * (i > 0 ? sprev[ch] : 0.f);
      snew[ch] = hh;
    } else if constexpr (KIND == EM_LM_RNN_TANH) {
      hh = tanhf(g[ch]);
    } else {
      hh = fmaxf(g[ch], 0.f);
    }
    hnew[ch] = from_f32<T>(hh);
    hout[ch] = from_f32<T>(hh);
  }
}

template <typename T>
void rnn_cell_launch(int kind, hipStream_t s, const Ctx& c, int i, int Ln, int l, int nh, int d) {
  const int n = c.p.B * c.p.W;
  switch (kind) {
    case EM_LM_LSTM: hipLaunchKernelGGL((rnn_cell_kernel<T, EM_LM_LSTM>), dim3(n), dim3(128), 0, s, c, i, Ln, l, nh, d); break;
    case EM_LM_GRU: hipLaunchKernelGGL((rnn_cell_kernel<T, EM_LM_GRU>), dim3(n), dim3(128), 0, s, c, i, Ln, l, nh, d); break;
    case EM_LM_RNN_TANH: hipLaunchKernelGGL((rnn_cell_kernel<T, EM_LM_RNN_TANH>), dim3(n), dim3(128), 0, s, c, i, Ln, l, nh, d); break;
    default: hipLaunchKernelGGL((rnn_cell_kernel<T, EM_LM_RNN_RELU>), dim3(n), dim3(128), 0, s, c, i, Ln, l, nh, d); break;
  }
}

int check(const EmSearchParams* p, const EmSearchBuffers* b) {
  if (!p || !b) return EM_ERR_BAD_ARG;
  if (p->B <= 0 || p->W <= 0 || p->W > 64 || p->V <= 1 || p->T <= 0 || p->Lmax < 2) return EM_ERR_BAD_ARG;
  if (p->w_ctc != 0.f && p->T > CTC_TMAX) return EM_ERR_UNSUPPORTED;
  if (p->S <= 0 || p->NC <= 0 || p->end_cap <= 0) return EM_ERR_BAD_ARG;
  if (p->S >= p->V ? p->NC != p->V : p->NC != p->S + 1) return EM_ERR_BAD_ARG;
  if (p->w_dec == 0.f && p->w_ctc == 0.f && p->w_lm == 0.f) return EM_ERR_BAD_ARG;
  if (p->w_dec == 0.f && p->w_lm == 0.f && p->w_len == 0.f && p->S < p->V)
    return EM_ERR_BAD_ARG;  // no full scorer -> no pre-beam
  if (p->w_lm != 0.f && (!b->lm_logp || !b->run_slm || !b->end_slm || !b->lm)) return EM_ERR_BAD_ARG;
  return EM_OK;
}

// One TransformerLM step for the n rows (espnet2/lm/transformer_lm.py:103-137 batch_score ->
// Encoder.forward_one_step): embedding -> input Linear + LayerNorm(1e-5) + ReLU (+ pos-enc) ->
// pre-norm self-attention / ReLU feed-forward layers over the token-tree K/V cache -> after_norm ->
// vocabulary projection (logits; the log-softmax is fused into the pre-beam kernel).
int rnn_lm_step(int dtype, const EmSearchParams* p, const EmSearchBuffers* b, int i, void* stream) {
  const EmLmWeights* lm = b->lm;
  const int n = p->B * p->W, V = p->V, d = lm->d, nh = lm->nhid, eu = lm->embed_unit, Ln = lm->num_blocks;
  const int gw = (lm->kind == EM_LM_LSTM || lm->kind == EM_LM_GRU ? 4 : 1) * nh;  // gate row width
  if (!lm->rnn || !b->rnn_hs || !b->rnn_cs || !b->rnn_hin || !b->rnn_gates || !b->lm_e) return EM_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  Ctx c{*p, *b};
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  const int32_t* step = b->step;
  // embedding of the last token (seq_rnn_lm.py:92) and the parents' hidden states
  EM_TRY(em_lm_embed(dtype, lm->embed, step ? b->tok : b->tok + (size_t)i * n, n, V, eu, step, p->Lmax,
                     b->lm_e, stream));
  if (dtype == EM_BF16)
    hipLaunchKernelGGL(rnn_gather_kernel<bf16>, dim3(n, Ln), dim3(128), 0, s, c, i, Ln, d);
  else
    hipLaunchKernelGGL(rnn_gather_kernel<float>, dim3(n, Ln), dim3(128), 0, s, c, i, Ln, d);
  for (int l = 0; l < Ln; ++l) {
    const EmRnnLayer& q = lm->rnn[l];
    const void* x = l == 0 ? b->lm_e : (const void*)((const unsigned char*)b->rnn_hin + (size_t)(l - 1) * n * d * es);
    const int kin = l == 0 ? eu : d;
    const void* hin = (const unsigned char*)b->rnn_hin + (size_t)l * n * d * es;
    EM_TRY(gemm(dtype, EM_EPI_SCALE_F32, x, q.w_ih, b->rnn_gates, q.bias, n, gw, kin, kin, gw, 1.f, stream));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, hin, q.w_hh, b->rnn_gates, nullptr, n, gw, d, d, gw, 1.f, stream));
    if (dtype == EM_BF16) rnn_cell_launch<bf16>(lm->kind, s, c, i, Ln, l, nh, d);
    else rnn_cell_launch<float>(lm->kind, s, c, i, Ln, l, nh, d);
  }
  const void* top = (const unsigned char*)b->rnn_hin + (size_t)(Ln - 1) * n * d * es;
  EM_TRY(gemm(dtype, EM_EPI_STORE_F32, top, lm->out_w, b->lm_logp, lm->out_b, n, V, d, d, V, 1.f, stream));
  EM_CHECK_LAUNCH();
  return EM_OK;
}

int lm_step(int dtype, const EmSearchParams* p, const EmSearchBuffers* b, int i, void* stream) {
  const EmLmWeights* lm = b->lm;
  if (lm->kind != EM_LM_TRANSFORMER) return rnn_lm_step(dtype, p, b, i, stream);
  const int n = p->B * p->W, V = p->V, d = lm->d, ff = lm->ff, eu = lm->embed_unit;
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  const int* anc = (i & 1) ? b->anc_b : b->anc_a;
  const int32_t* step = b->step;
  EM_TRY(em_lm_embed(dtype, lm->embed, step ? b->tok : b->tok + (size_t)i * n, n, V, eu, step, p->Lmax,
                     b->lm_e, stream));
  EM_TRY(gemm(dtype, EM_EPI_SCALE_F32, b->lm_e, lm->in_w, b->lm_x, lm->in_b, n, d, eu, eu, d, 1.f, stream));
  EM_TRY(em_lm_input_norm_f32(b->lm_x, lm->in_ln_g, lm->in_ln_b, lm->pe, n, d, i, step, p->Lmax, stream));
  for (int l = 0; l < lm->num_blocks; ++l) {
    const EmLmLayer& q = lm->layers[l];
    unsigned char* kc = (unsigned char*)b->lm_k + (size_t)l * p->Lmax * n * d * es;
    unsigned char* vc = (unsigned char*)b->lm_v + (size_t)l * p->Lmax * n * d * es;
    EM_TRY(ln_proj(dtype, EM_EPI_STORE, b->lm_x, q.norm1_g, q.norm1_b, q.wqkv, q.bqkv, b->lm_qkv, b->lm_xn, n, 3 * d,
                   d, stream));
    if (step)
      EM_TRY(em_dec_self_attention(dtype, b->lm_qkv, kc, vc, b->anc_a, b->anc_b, n, d, lm->heads, p->Lmax,
                                   0, step, (p->W + 1) / 2, b->tok, b->lm_ctx, stream));
    else
      EM_TRY(em_dec_self_attention(dtype, b->lm_qkv, kc, vc, anc, anc, n, d, lm->heads, p->Lmax, i,
                                   nullptr, (p->W + 1) / 2, b->tok, b->lm_ctx, stream));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, b->lm_ctx, q.wout, b->lm_x, q.bout, n, d, d, d, d, 1.f, stream));
    EM_TRY(ln_proj(dtype, EM_EPI_RELU, b->lm_x, q.norm2_g, q.norm2_b, q.w1, q.b1, b->lm_h, b->lm_xn, n, ff, d, stream));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, b->lm_h, q.w2, b->lm_x, q.b2, n, d, ff, ff, d, 1.f, stream));
  }
  EM_TRY(ln_proj(dtype, EM_EPI_STORE_F32, b->lm_x, lm->after_norm_g, lm->after_norm_b, lm->out_w, lm->out_b,
                 b->lm_logp, b->lm_xn, n, V, d, stream));
  return EM_OK;
}



// Source-attention memory of every decoder layer for the p->T visible frames: K | V by one GEMM per
// layer over the whole batch, then V^T (strides follow p->T / p->Tpad).
int project_memory(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw, const EmSearchBuffers* b,
                   const void* enc_act, void* stream) {
  const int d = dw->d;
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  for (int l = 0; l < dw->num_blocks; ++l) {
    unsigned char* kv = (unsigned char*)b->mem_kv + (size_t)l * p->B * p->T * 2 * d * es;
    unsigned char* vT = (unsigned char*)b->mem_vT + (size_t)l * p->B * d * p->Tpad * es;
    EM_TRY(gemm(dtype, EM_EPI_STORE, enc_act, dw->layers[l].src_wkv, kv, dw->layers[l].src_bkv,
                p->B * p->T, 2 * d, d, d, 2 * d, 1.f, stream));
    EM_TRY(em_dec_transpose_v(dtype, kv, p->B, p->T, d, p->Tpad, vT, stream));
    // round 6: ... and fragment-major where the caller provides the buffers (bf16, d_k = 64; the label step's source attention)
    if (dtype == EM_BF16 && b->mem_kf && b->mem_vf && d == 64 * dw->heads) {
      const size_t per = (size_t)p->B * d * p->Tpad * es;
      EM_TRY(em_dec_pack_memory_frag_bf16(kv, p->B, p->T, p->Tpad, d, dw->heads, (unsigned char*)b->mem_kf + l * per,
                                          (unsigned char*)b->mem_vf + l * per, stream));
    }
  }
  return EM_OK;
}

// logits^T [V][B*ldT] = W_ctc [V][d] x enc^T: the transposed layout falls out of swapping the GEMM
// operands; bias + log-softmax over V (asr/ctc.py:197-205) run column-wise in place
int ctc_log_probs(int dtype, const EmSearchParams* p, const EmSearchBuffers* b, const void* enc_act,
                  int d_model, const void* ctc_w, const float* ctc_b, void* stream) {
  if (!ctc_w || !enc_act || !b->ctc_lpT) return EM_ERR_BAD_ARG;
  const int BT = p->B * p->T, ld = p->B * ldt(*p);
  if (ld != BT && p->B != 1) return EM_ERR_BAD_ARG;  // a frame capacity only makes sense per stream
  EM_TRY(gemm(dtype, EM_EPI_STORE_F32, ctc_w, enc_act, b->ctc_lpT, nullptr, p->V, BT, d_model, d_model,
              ld, 1.f, stream));
  hipLaunchKernelGGL(col_logsoftmax_kernel, dim3(em_cdiv(BT, 64)), dim3(256), 0, (hipStream_t)stream,
                     b->ctc_lpT, ctc_b, p->V, BT, ld);
  return EM_OK;
}

}  // namespace

extern "C" int em_search_init(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                              const EmSearchBuffers* b, const void* enc_act, int32_t d_model,
                              const void* ctc_w, const float* ctc_b, void* stream) {
  EM_TRY(check(p, b));
  hipStream_t s = (hipStream_t)stream;
  Ctx c{*p, *b};
  const int n = p->B * p->W;
  if (p->w_dec != 0.f) {
    if (!dw || !enc_act) return EM_ERR_BAD_ARG;
    EM_TRY(project_memory(dtype, p, dw, b, enc_act, stream));
  }
  if (p->w_ctc != 0.f) EM_TRY(ctc_log_probs(dtype, p, b, enc_act, d_model, ctc_w, ctc_b, stream));
  hipLaunchKernelGGL(search_init_rows_kernel, dim3(em_cdiv(n, 64)), dim3(64), 0, s, c);
  hipLaunchKernelGGL(search_init_utt_kernel, dim3(em_cdiv(p->B, 64)), dim3(64), 0, s, c);
  if (b->step) hipMemsetAsync(b->step, 0, 2 * sizeof(int32_t), s);  // [0] step index, [1] arrival ticket of update_kernel
  EM_CHECK_LAUNCH();
  return EM_OK;
}

namespace {

// One TransformerDecoder.forward_one_step (espnet2/asr/decoder/transformer_decoder.py:191-247) for the
// n = B*W rows of a search step: embedding + position, the pre-norm decoder layers over the token-tree
// K/V cache and the per-utterance memory K / V^T, after_norm + output_layer -> logits [n][V] (the
// log-softmax is fused into the pre-beam kernel, or em_log_softmax_rows_f32 for em_decoder_step callers).
struct DecStep {
  int B, W, T, Tpad, Lmax, pos;
  const int32_t* pos_dev;           // graph mode: position read from device memory (then tok = table base)
  const int32_t* tok;               // [Lmax][n] token table
  const int32_t *anc_a, *anc_b;     // ancestor tables by step parity (the same table twice when not double buffered)
  const int32_t* xlens;
  void *self_k, *self_v;
  const void *mem_kv, *mem_vT;
  float* x;
  void *xn, *qkv, *qs, *ctx, *hbuf;
  float* logits;
  const void *mem_kf = nullptr, *mem_vf = nullptr;  // the memory fragment-major (EmSearchBuffers.mem_kf / mem_vf), or NULL
};

int decoder_step(int dtype, const EmDecoderWeights* dw, const DecStep& a, void* stream) {
  const int n = a.B * a.W, V = dw->vocab, i = a.pos;
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  const int d = dw->d, ff = dw->ff, h = dw->heads;
  const int* anc = (i & 1) ? a.anc_b : a.anc_a;
  // round 6: the step's projections on FRAGMENT-MAJOR copies of the weights where the host packed them (EmDecoderLayer.*_frag:
  // 1 KiB operand loads instead of 16 rows x 64 bytes; csrc/dec_ffn.hip, mid_gemm<FRAG>); bf16, d = 256 | 512, from 96 rows
  // (fewer rows: ln_gemm + mid_gemm sit at the latency floor).  ESPNET_AMD_DEC_FFN_SPLIT=1: developer A/B switch (off).
  const bool frag = dtype == EM_BF16 && n >= 96 && (d == 256 || d == 512) && em_sw().dec_ffn_split != 1;
  const int ffn_frag = frag ? em_dec_ffn_split(n, d, ff) : 0;
  auto resid_proj = [&](const void* ctx, const void* w, const void* wfrag, const float* bias) -> int {
    if (frag && wfrag) {
      EmGemmArgs g = {};
      g.A = ctx; g.W = wfrag; g.C = a.x; g.bias = bias;
      g.M = n; g.N = d; g.K = d; g.lda = d; g.ldc = d; g.scale = 1.f;
      const int rc = em_gemm_mid_frag(EM_EPI_RESID_F32, 2, &g, stream);
      if (rc != EM_ERR_UNSUPPORTED) return rc;
    }
    return gemm(dtype, EM_EPI_RESID_F32, ctx, w, a.x, bias, n, d, d, d, d, 1.f, stream);
  };
  if (a.pos_dev)
    EM_TRY(em_dec_embed_f32(dw->embed, dw->pe, a.tok, n, V, d, 0, a.pos_dev, a.Lmax, a.x, stream));
  else
    EM_TRY(em_dec_embed_f32(dw->embed, dw->pe, a.tok + (size_t)i * n, n, V, d, i, nullptr, dw->pe_len, a.x,
                            stream));
  for (int l = 0; l < dw->num_blocks; ++l) {
    const EmDecoderLayer& q = dw->layers[l];
    unsigned char* kc = (unsigned char*)a.self_k + (size_t)l * a.Lmax * n * d * es;
    unsigned char* vc = (unsigned char*)a.self_v + (size_t)l * a.Lmax * n * d * es;
    const unsigned char* kv = (const unsigned char*)a.mem_kv + (size_t)l * a.B * a.T * 2 * d * es;
    const unsigned char* vT = (const unsigned char*)a.mem_vT + (size_t)l * a.B * d * a.Tpad * es;
    // every pre-norm LayerNorm rides in the prologue of the projection that consumes it (ln_gemm.hip)
    if (frag && q.self_wqkv_frag)
      EM_TRY(em_ln_gemm_frag(EM_LNF_STORE, a.x, q.norm1_g, q.norm1_b, LN_EPS, q.self_wqkv_frag, q.self_bqkv, a.qkv, n, 3 * d, d, stream));
    else
      EM_TRY(ln_proj(dtype, EM_EPI_STORE, a.x, q.norm1_g, q.norm1_b, q.self_wqkv, q.self_bqkv, a.qkv, a.xn, n,
                     3 * d, d, stream));
    // round 5: over the union of the beam's ancestors where the shape allows (bf16, d_k = 64, beams of <= 16, Lmax <= 512)
    if (a.pos_dev)
      EM_TRY(em_dec_self_attention_beam(dtype, a.qkv, kc, vc, a.anc_a, a.anc_b, n, d, h, a.Lmax, 0, a.pos_dev, a.W, a.ctx, stream));
    else
      EM_TRY(em_dec_self_attention_beam(dtype, a.qkv, kc, vc, anc, anc, n, d, h, a.Lmax, i, nullptr, a.W, a.ctx, stream));
    EM_TRY(resid_proj(a.ctx, q.self_wout, q.self_wout_frag, q.self_bout));
    // round 4: norm2 + the source attention's query projection ride in the attention kernel's prologue (bf16, d_k = 64,
    // d = 256 | 512: one launch less per layer; ESPNET_AMD_NO_SRC_LNQ=1: developer A/B switch)
    const bool no_lnq = em_sw().no_src_lnq;
    // (the fused kernel adds its LN(x) rows and query tile to the score / probability tiles of the plain kernel: long
    // memories - Tpad > 1 472 at d = 512, > 1 568 at d = 256 - fit only the two-launch form)
    const size_t lnq_lds = (size_t)16 * (a.Tpad + 4) * 4 + 64 + (size_t)16 * (a.Tpad + 8) * 2 + (size_t)16 * (d + 8) * 2 + 16 * (64 + 8) * 2;
    if (dtype == EM_BF16 && !no_lnq && d == 64 * h && (d == 256 || d == 512) && lnq_lds <= 160 * 1024) {
      if (frag && q.src_wq_frag && a.mem_kf && a.mem_vf) {
        const size_t per = (size_t)a.B * d * a.Tpad * es;
        EM_TRY(em_dec_src_attention_lnq_memfrag_bf16(a.x, q.norm2_g, q.norm2_b, LN_EPS, q.src_wq_frag, q.src_bq,
                                                     (const unsigned char*)a.mem_kf + l * per, (const unsigned char*)a.mem_vf + l * per,
                                                     a.xlens, a.B, a.W, d, h, a.T, a.Tpad, a.ctx, stream));
      } else if (frag && q.src_wq_frag)
        EM_TRY(em_dec_src_attention_lnq_frag(dtype, a.x, q.norm2_g, q.norm2_b, LN_EPS, q.src_wq_frag, q.src_bq, kv, 2 * d, vT, a.xlens,
                                             a.B, a.W, d, h, a.T, a.Tpad, a.ctx, stream));
      else
        EM_TRY(em_dec_src_attention_lnq(dtype, a.x, q.norm2_g, q.norm2_b, LN_EPS, q.src_wq, q.src_bq, kv, 2 * d, vT, a.xlens,
                                        a.B, a.W, d, h, a.T, a.Tpad, a.ctx, stream));
    } else {
      EM_TRY(ln_proj(dtype, EM_EPI_STORE, a.x, q.norm2_g, q.norm2_b, q.src_wq, q.src_bq, a.qs, a.xn, n, d, d,
                     stream));
      EM_TRY(em_dec_src_attention(dtype, a.qs, kv, 2 * d, vT, a.xlens, a.B, a.W, d, h, a.T, a.Tpad, a.ctx, stream));
    }
    EM_TRY(resid_proj(a.ctx, q.src_wout, q.src_wout_frag, q.src_bout));
    // round 6: norm3 + feed_forward + residual on fragment-major operands (csrc/dec_ffn.hip: LayerNorm in the first
    // projection's prologue, 1 KiB operand loads in both; bf16, d = 256 | 512, rows in whole fragments of 16)
    if (ffn_frag > 0 && q.w1_frag && q.w2_frag) {
      EM_TRY(em_dec_ffn(dtype, a.x, q.norm3_g, q.norm3_b, LN_EPS, q.w1_frag, q.b1, q.w2_frag, q.b2, n, d, ff, a.hbuf, stream));
    } else {
      EM_TRY(ln_proj(dtype, EM_EPI_RELU, a.x, q.norm3_g, q.norm3_b, q.w1, q.b1, a.hbuf, a.xn, n, ff, d, stream));
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, a.hbuf, q.w2, a.x, q.b2, n, d, ff, ff, d, 1.f, stream));
    }
  }
  // (the vocabulary projection: up to 320 rows - at 640 rows x 5 000 labels it is a GEMM proper, and 16-row workgroups that
  // each stream their 512 labels' weights lose to the tiled kernel's 64-row tiles: 20.3 against 11.3 + 5.0 us, profiles/r06ad)
  if (frag && n <= 320 && dw->out_w_frag && V % 4 == 0)
    return em_ln_gemm_frag(EM_LNF_STORE_F32, a.x, dw->after_norm_g, dw->after_norm_b, LN_EPS, dw->out_w_frag, dw->out_b, a.logits,
                           n, V, d, stream);
  return ln_proj(dtype, EM_EPI_STORE_F32, a.x, dw->after_norm_g, dw->after_norm_b, dw->out_w, dw->out_b,
                 a.logits, a.xn, n, V, d, stream);
}

// One label step up to the per-utterance top-W selection: decoder (+ LM) step for the n rows,
// log-softmax + pre-beam, CTC prefix scores of the candidates, selection.  Nothing of the search
// state (tree, ancestor tables, running scores, r) is modified; the decoder / LM K/V caches get
// their position-i entries.
int search_core(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw, const EmSearchBuffers* b,
                int i, void* stream, bool with_select = true) {
  hipStream_t s = (hipStream_t)stream;
  Ctx c{*p, *b};
  const int n = p->B * p->W, V = p->V;
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  if (p->w_dec != 0.f) {
    DecStep a{p->B, p->W, p->T, p->Tpad, p->Lmax, i, b->step, b->tok, b->anc_a, b->anc_b, b->xlens, b->self_k,
              b->self_v, b->mem_kv, b->mem_vT, b->x, b->xn, b->qkv, b->qs, b->ctx, b->hbuf, b->dec_logp, b->mem_kf, b->mem_vf};
    EM_TRY(decoder_step(dtype, dw, a, stream));
  }
  if (p->w_lm != 0.f) EM_TRY(lm_step(dtype, p, b, i, stream));
  bool cand_done = false;  // the fused row kernel also produced the candidate totals
  if (p->w_dec != 0.f || p->w_lm != 0.f || p->w_len != 0.f) {
    if ((p->S > PREBEAM_SMAX && p->S < V) || V > 256 * 40) {
      if (p->w_lm != 0.f || p->w_dec == 0.f) return EM_ERR_UNSUPPORTED;  // needs the fused row kernel
      EM_TRY(em_log_softmax_rows_f32(b->dec_logp, n, V, stream));
      if (p->S < V)
        hipLaunchKernelGGL(prebeam_kernel, dim3(n), dim3(64), (size_t)V * sizeof(float), s, c);
    } else if (V <= 256 * 8) {
      hipLaunchKernelGGL(logsoftmax_prebeam_kernel<8>, dim3(n), dim3(256), 0, s, c, i);
      cand_done = p->S < V;
    } else if (V <= 256 * 20) {
      hipLaunchKernelGGL(logsoftmax_prebeam_kernel<20>, dim3(n), dim3(256), 0, s, c, i);
      cand_done = p->S < V;
    } else {
      hipLaunchKernelGGL(logsoftmax_prebeam_kernel<40>, dim3(n), dim3(256), 0, s, c, i);
      cand_done = p->S < V;
    }
  }
  if (!cand_done) {  // all-vocabulary mode (no pre-beam) and the generic pre-beam: candidate totals as a launch of their own
    const long waves = (long)n * p->NC;
    hipLaunchKernelGGL(candidate_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, c, i);
  }
  if (with_select) hipLaunchKernelGGL(select_kernel, dim3(p->B), dim3(64), 0, s, c);
  return EM_OK;
}

// ---- streaming (block-synchronous) search: device ops under the host's control flow -----------
// Reference: BatchBeamSearchOnline (espnet2/legacy/nets/batch_beam_search_online.py:155-534).  The
// host mirrors its loop (repetition / local-<eos> breaks, rewind, ended lists); the device keeps the
// numeric state.  Row layout, token tree, ancestor tables and the r ping-pong are the offline ones;
// the visible memory length (p->T, xlens[0]) grows block by block under a fixed frame stride ldT.

// CTCPrefixScoreTH.extend_state (ctc_prefix_score.py:248-270, Eq. 14 of arXiv:2006.14941): over the
// new frames [t_old, T) the prefix continues along the blank path only:
// r[t] = (logzero, r[t-1][1] + logp[t][blank]).  One thread per alive row (a block brings ~16 frames).
__global__ void online_extend_r_kernel(Ctx c, int i, int t_old) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= c.p.B * c.p.W || !c.b.alive[r]) return;
  const int LT = ldt(c.p), b = r / c.p.W;
  float2* rr = (float2*)((i & 1) ? c.b.r_b : c.b.r_a) + (size_t)r * LT;
  const float* xb = c.b.ctc_lpT + (size_t)c.p.blank * c.p.B * LT + (size_t)b * LT;
  const int start = t_old > 1 ? t_old : 1, T = c.b.xlens[b];
  float cum = rr[start - 1].y;
  for (int t = start; t < T; ++t) {
    cum += xb[t];
    rr[t] = make_float2(LOGZERO, cum);
  }
}

// `best` of BatchBeamSearch.search (batch_beam_search.py:317-357) as one record per new row k:
// [valid, parent slot, token, total, decoder, ctc, length_bonus, lm]; the new s_prev goes to online_psi.
__global__ __launch_bounds__(64) void online_best_kernel(Ctx c) {
  const int b = blockIdx.x, k = threadIdx.x;
  const int W = c.p.W, NC = c.p.NC, V = c.p.V;
  if (k >= W) return;
  const int rnew = b * W + k;
  float* o = c.b.online_best + (size_t)rnew * 8;
  const int sel = c.b.sel_idx[rnew];
  if (sel < 0) {
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    c.b.online_psi[rnew] = 0.f;
    return;
  }
  const int pk = sel / NC, sl = sel - pk * NC, prow = b * W + pk;
  const int tk = c.b.cand_tok[(size_t)prow * NC + sl];
  float sdec = 0.f, sctc = 0.f, slen = 0.f, slm = 0.f, psi = 0.f;
  if (c.p.w_dec != 0.f) sdec = c.b.run_sdec[prow] + c.b.dec_logp[(size_t)prow * V + tk];
  if (c.p.w_len != 0.f) slen = c.b.run_slen[prow] + 1.0f;
  if (c.p.w_lm != 0.f) slm = c.b.run_slm[prow] + c.b.lm_logp[(size_t)prow * V + tk];
  if (c.p.w_ctc != 0.f) {
    psi = c.b.cand_psi[(size_t)prow * NC + sl];
    sctc = c.b.run_sctc[prow] + (psi - c.b.s_prev[prow]);
  }
  o[0] = 1.f; o[1] = (float)pk; o[2] = (float)tk; o[3] = c.b.sel_total[rnew];
  o[4] = sdec; o[5] = sctc; o[6] = slen; o[7] = slm;
  c.b.online_psi[rnew] = psi;
}

// prev_hyps = running_hyps (:459): the per-row scalars are copied; r / anc of the state before step i
// stay where they are (parity i & 1), the commit writes the other parity.
__global__ void online_snapshot_kernel(Ctx c, int restore) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= c.p.B * c.p.W) return;
  float* sn = c.b.online_snap + (size_t)r * 8;
  if (!restore) {
    sn[0] = (float)c.b.alive[r]; sn[1] = c.b.run_score[r]; sn[2] = c.b.run_sdec[r];
    sn[3] = c.b.run_sctc[r]; sn[4] = c.b.run_slen[r]; sn[5] = c.b.run_slm ? c.b.run_slm[r] : 0.f;
    sn[6] = c.b.s_prev[r];
  } else {
    c.b.alive[r] = (int)sn[0]; c.b.run_score[r] = sn[1]; c.b.run_sdec[r] = sn[2];
    c.b.run_sctc[r] = sn[3]; c.b.run_slen[r] = sn[4];
    if (c.b.run_slm) c.b.run_slm[r] = sn[5];
    c.b.s_prev[r] = sn[6];
  }
}

// running_hyps = post_process(best) (:460-462 with batch_beam_search.py:412-423): the rows of `best`
// that did not end with <eos> become the running rows; tree, ancestor tables and scores as offline.
__global__ __launch_bounds__(64) void online_commit_kernel(Ctx c, int i_host) {
  const int i = c.b.step ? *c.b.step : i_host;  // (graph mode: the host writes the step index before the replay)
  const int b = blockIdx.x, lane = threadIdx.x;
  const int W = c.p.W, Lmax = c.p.Lmax, n = c.p.B * c.p.W;
  __shared__ int s_prow[64], s_tok[64], s_valid[64];
  const int* anc_old = (i & 1) ? c.b.anc_b : c.b.anc_a;
  int* anc_new = (i & 1) ? c.b.anc_a : c.b.anc_b;
  float rec[8];
  if (lane < W) {
    const float* o = c.b.online_best + (size_t)(b * W + lane) * 8;
    for (int j = 0; j < 8; ++j) rec[j] = o[j];
    s_valid[lane] = rec[0] != 0.f;
    s_prow[lane] = b * W + (int)rec[1];
    s_tok[lane] = (int)rec[2];
  }
  __syncthreads();
  for (int k = 0; k < W; ++k) {
    if (!s_valid[k]) continue;
    const int rnew = b * W + k;
    const int* src = anc_old + (size_t)s_prow[k] * Lmax;
    int* dst = anc_new + (size_t)rnew * Lmax;
    for (int j = lane; j <= i; j += 64) dst[j] = src[j];
    if (lane == 0 && i + 1 < Lmax) dst[i + 1] = rnew;
  }
  __syncthreads();
  if (lane < W) {
    const int rnew = b * W + lane;
    if (s_valid[lane]) {
      c.b.tok[(size_t)(i + 1) * n + rnew] = s_tok[lane];
      c.b.parent[(size_t)(i + 1) * n + rnew] = s_prow[lane];
    }
    const int alive = s_valid[lane] && s_tok[lane] != c.p.eos;
    c.b.alive[rnew] = alive;
    c.b.run_score[rnew] = alive ? rec[3] : -INFINITY;
    c.b.run_sdec[rnew] = rec[4];
    c.b.run_sctc[rnew] = rec[5];
    c.b.run_slen[rnew] = rec[6];
    if (c.b.run_slm) c.b.run_slm[rnew] = rec[7];
    c.b.s_prev[rnew] = c.b.online_psi[rnew];
  }
}

int check_online(const EmSearchParams* p, const EmSearchBuffers* b) {
  EM_TRY(check(p, b));
  // one stream, host-driven steps.  b->step != NULL (graph mode): every kernel of a step reads the step index from
  // step[0], which the HOST writes before each em_search_online_core / _commit - nothing advances it on the device (the
  // select / online kernels have no step_advance) - so one captured launch sequence serves every step of a block.
  if (p->B != 1) return EM_ERR_BAD_ARG;
  if (!b->online_best || !b->online_psi || !b->online_snap) return EM_ERR_BAD_ARG;
  if (p->ldT < p->T) return EM_ERR_BAD_ARG;
  return EM_OK;
}

}  // namespace

// Steps i0 .. i1-1 (i = number of tokens after <sos> already in every running hypothesis).
extern "C" int em_search_steps(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                               const EmSearchBuffers* b, int32_t i0, int32_t i1, void* stream) {
  EM_TRY(check(p, b));
  if (i0 < 0 || (!b->step && i1 > p->Lmax - 1)) return EM_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  Ctx c{*p, *b};
  const int n = p->B * p->W;
  // selection + the winners' CTC state + the new rows in one launch where a beam's chains fit the LDS side by side
  const int lt_cap = ldt(*p);
  const size_t tail_lds = p->w_ctc != 0.f ? (size_t)p->W * 5 * lt_cap * sizeof(float) : 0;
  const bool no_tail = em_sw().no_tail_fusion;  // developer A/B switch
  const bool fused_tail = !no_tail && p->W <= 16 && p->W * p->NC <= 64 * SEL_KM && tail_lds <= 144 * 1024;
  static EmLdsCap cap = {};
  if (fused_tail && tail_lds > 64 * 1024 && em_raise_lds_cap((const void*)tail_kernel, tail_lds, &cap) != EM_OK) return EM_ERR_LAUNCH;
  for (int i = i0; i < i1; ++i) {
    EM_TRY(search_core(dtype, p, dw, b, i, stream, !fused_tail));
    if (fused_tail) {
      hipLaunchKernelGGL(tail_kernel, dim3(p->B), dim3(64 * p->W), tail_lds, s, c, i, lt_cap);
    } else {
      if (p->w_ctc != 0.f) hipLaunchKernelGGL(ctc_state_kernel, dim3(n), dim3(64), 0, s, c, i);
      hipLaunchKernelGGL(update_kernel, dim3(p->B), dim3(64), 0, s, c, i);
    }
    EM_CHECK_LAUNCH();
  }
  return EM_OK;
}

// extend (batch_beam_search_online.py:518-534) for the p->T frames now visible: decoder memory
// re-projected (strides follow p->T), CTC log-probs of all visible frames (scorers/ctc.py:128-139;
// frames already present keep their values: the encoder output of a frame never changes), and r of the
// running rows (state before step i) continued from t_old.  xlens[0] must already hold p->T.
extern "C" int em_search_online_extend(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                                       const EmSearchBuffers* b, const void* enc_act, int32_t d_model,
                                       const void* ctc_w, const float* ctc_b, int32_t i, int32_t t_old,
                                       void* stream) {
  EM_TRY(check_online(p, b));
  if (t_old < 1 || t_old > p->T || i < 0) return EM_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  Ctx c{*p, *b};
  if (p->w_dec != 0.f) {
    if (!dw || !enc_act) return EM_ERR_BAD_ARG;
    EM_TRY(project_memory(dtype, p, dw, b, enc_act, stream));
  }
  if (p->w_ctc != 0.f) {
    EM_TRY(ctc_log_probs(dtype, p, b, enc_act, d_model, ctc_w, ctc_b, stream));
    hipLaunchKernelGGL(online_extend_r_kernel, dim3(1), dim3(64), 0, s, c, i, t_old);
  }
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// best = search(running_hyps, h) (:400): fills online_best [W][8] for the host to read.
extern "C" int em_search_online_core(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                                     const EmSearchBuffers* b, int32_t i, void* stream) {
  EM_TRY(check_online(p, b));
  if (i < 0 || i >= p->Lmax - 1) return EM_ERR_BAD_ARG;
  EM_TRY(search_core(dtype, p, dw, b, i, stream));
  Ctx c{*p, *b};
  hipLaunchKernelGGL(online_best_kernel, dim3(p->B), dim3(64), 0, (hipStream_t)stream, c);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// prev_hyps = running_hyps; running_hyps = post_process(best) (:459-462) after em_search_online_core(i).
extern "C" int em_search_online_commit(int dtype, const EmSearchParams* p, const EmSearchBuffers* b,
                                       int32_t i, void* stream) {
  (void)dtype;
  EM_TRY(check_online(p, b));
  if (i < 0 || i >= p->Lmax - 1) return EM_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  Ctx c{*p, *b};
  const int n = p->B * p->W;
  hipLaunchKernelGGL(online_snapshot_kernel, dim3(em_cdiv(n, 64)), dim3(64), 0, s, c, 0);
  if (p->w_ctc != 0.f) hipLaunchKernelGGL(ctc_state_kernel, dim3(n), dim3(64), 0, s, c, i);
  hipLaunchKernelGGL(online_commit_kernel, dim3(p->B), dim3(64), 0, s, c, i);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// running_hyps = prev_hyps (:484-487): the scalars come back from the snapshot; the caller steps
// its position back by one, which re-selects the r / anc parity of that state.
extern "C" int em_search_online_rewind(const EmSearchParams* p, const EmSearchBuffers* b, void* stream) {
  EM_TRY(check_online(p, b));
  Ctx c{*p, *b};
  hipLaunchKernelGGL(online_snapshot_kernel, dim3(em_cdiv(p->B * p->W, 64)), dim3(64), 0,
                     (hipStream_t)stream, c, 1);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// ---- per-call scorer entry points (the reference's scorer interface, one call per search step) -----
// The fused search above never leaves the device; these expose its stages one by one so that a
// label-synchronous search written against `BatchScorerInterface` / `BatchPartialScorerInterface`
// (espnet2/legacy/nets/scorer_interface.py:29-190) -- the reference's own BatchBeamSearch -- can be
// driven by the same kernels (espnet_amd TransformerDecoder.batch_score, CTCPrefixScorer.batch_score_partial,
// TransformerLM / SequentialRNNLM.batch_score).
namespace {

// log psi (and score = log psi - s_prev) of the S candidates (+ <eos> in slot S when cand != NULL) of every
// row: one wave per (row, slot); CTCPrefixScoreTH.__call__ ctc_prefix_score.py:71-191 without the recurrence.
__global__ __launch_bounds__(256) void ctc_prefix_score_kernel(
    const float* __restrict__ lpT, const int* __restrict__ xlens, const float2* __restrict__ r_prev,
    const float* __restrict__ s_prev, const int* __restrict__ last, const int* __restrict__ cand, int B, int W,
    int S, int NC, int LT, int i, int eos, int blank, float* __restrict__ log_psi, float* __restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= (long)B * W * NC) return;
  const int r = (int)(idx / NC), s = (int)(idx - (long)r * NC);
  const int b = r / W;
  const int tk = cand ? (s < S ? cand[(size_t)r * S + s] : eos) : s;
  const int xlen = xlens[b];
  const float2* rp = r_prev + (size_t)r * LT;
  float psi;
  if (tk == blank && eos != blank) {
    psi = LOGZERO;  // :188-190
  } else if (tk == eos) {
    const float2 re = rp[xlen - 1];
    psi = logaddexp_(re.x, re.y);  // :184-186
  } else {
    psi = ctc_psi_wave(rp, lpT + (size_t)tk * B * LT + (size_t)b * LT, tk == last[r], i, xlen, lane);
  }
  if (lane == 0) {
    log_psi[(size_t)r * NC + s] = psi;
    if (scores) scores[(size_t)r * NC + s] = psi - s_prev[r];
  }
}

// forward variables of prefix(row) + label for m (row, label) pairs: one workgroup per pair
__global__ __launch_bounds__(64) void ctc_prefix_state_kernel(
    const float* __restrict__ lpT, const int* __restrict__ xlens, const float2* __restrict__ r_prev,
    const int* __restrict__ last, const int* __restrict__ rows, const int* __restrict__ toks, int B, int W,
    int LT, int i, int blank, float2* __restrict__ r_out) {
  __shared__ float s_xn[CTC_TMAX], s_xb[CTC_TMAX], s_phi[CTC_TMAX];
  __shared__ float2 s_out[CTC_TMAX];
  const int k = blockIdx.x, lane = threadIdx.x;
  const int r = rows[k], tk = toks[k];
  const int b = r / W, xlen = xlens[b];
  const size_t BT = (size_t)B * LT;
  ctc_chain_wg<true>(r_prev + (size_t)r * LT, lpT + (size_t)tk * BT + (size_t)b * LT,
               lpT + (size_t)blank * BT + (size_t)b * LT, tk == last[r], i, xlen, r_out + (size_t)k * LT, lane,
               s_xn, s_xb, s_phi, s_out);
}

// r_prev of the empty prefix (ctc_prefix_score.py:87-99): r^n = logzero, r^b[t] = cumsum of blank log-probs
__global__ __launch_bounds__(64) void ctc_prefix_init_kernel(const float* __restrict__ lpT,
                                                            const int* __restrict__ xlens, int B, int LT,
                                                            int blank, float2* __restrict__ r0) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  const float* xb = lpT + (size_t)blank * B * LT + (size_t)b * LT;
  float acc = 0.f;
  for (int t = 0; t < LT; ++t) {
    if (t < xlens[b]) acc += xb[t];
    r0[(size_t)b * LT + t] = make_float2(LOGZERO, t < xlens[b] ? acc : LOGZERO);
  }
}

// CTCPrefixScoreTH.extend_state (ctc_prefix_score.py:248-270) on stand-alone scorer states: r_old [n][T_old][2] ->
// r_new [n][T_new][2]; frames below max(T_old, 1) are copied, the new ones continue along the blank path only
// (r^n = logzero, r^b[t] = r^b[t-1] + logp[t][blank]), in the reference's own left-to-right order.  One wave per state.
__global__ __launch_bounds__(64) void ctc_prefix_extend_kernel(const float* __restrict__ lp_blank, const float2* __restrict__ r_old,
                                                              int T_old, int T_new, float2* __restrict__ r_new) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const float2* ro = r_old + (size_t)k * T_old;
  float2* rn = r_new + (size_t)k * T_new;
  const int start = T_old > 1 ? T_old : 1;
  for (int t = lane; t < start; t += 64) rn[t] = t < T_old ? ro[t] : make_float2(LOGZERO, LOGZERO);
  if (lane == 0) {
    float cum = T_old > 0 ? ro[start - 1].y : LOGZERO;
    for (int t = start; t < T_new; ++t) {
      cum += lp_blank[t];
      rn[t] = make_float2(LOGZERO, cum);
    }
  }
}

}  // namespace

extern "C" int em_ctc_prefix_extend(const float* lpT, int32_t T_new, int32_t blank, const float* r_old, int32_t n,
                                    int32_t T_old, float* r_new, void* stream) {
  if (!lpT || !r_old || !r_new || n <= 0 || T_old < 0 || T_new < T_old || T_new <= 0 || blank < 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(ctc_prefix_extend_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, lpT + (size_t)blank * T_new,
                     (const float2*)r_old, T_old, T_new, (float2*)r_new);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_decoder_memory(int dtype, const EmDecoderWeights* dw, const void* enc_act, int32_t B,
                                 int32_t T, int32_t Tpad, void* mem_kv, void* mem_vT, void* stream) {
  if (!dw || !enc_act || !mem_kv || !mem_vT || B <= 0 || T <= 0 || Tpad < T) return EM_ERR_BAD_ARG;
  EmSearchParams p = {};
  p.B = B; p.T = T; p.Tpad = Tpad;
  EmSearchBuffers b = {};
  b.mem_kv = mem_kv; b.mem_vT = mem_vT;
  return project_memory(dtype, &p, dw, &b, enc_act, stream);
}

extern "C" int em_decoder_step(int dtype, const EmDecoderWeights* dw, const EmDecoderStepArgs* a, void* stream) {
  if (!dw || !a || a->B <= 0 || a->W <= 0 || a->T <= 0 || a->Tpad < a->T || a->Lmax < 1) return EM_ERR_BAD_ARG;
  if (a->pos < 0 || a->pos >= a->Lmax || a->pos >= dw->pe_len) return EM_ERR_BAD_ARG;
  if (!a->tok || !a->anc || !a->xlens || !a->self_k || !a->self_v || !a->mem_kv || !a->mem_vT || !a->x || !a->xn ||
      !a->qkv || !a->qs || !a->ctx || !a->hbuf || !a->logits)
    return EM_ERR_BAD_ARG;
  DecStep s{a->B, a->W, a->T, a->Tpad, a->Lmax, a->pos, nullptr, a->tok, a->anc, a->anc, a->xlens, a->self_k,
            a->self_v, a->mem_kv, a->mem_vT, a->x, a->xn, a->qkv, a->qs, a->ctx, a->hbuf, a->logits};
  return decoder_step(dtype, dw, s, stream);
}

extern "C" int em_lm_step(int dtype, const EmSearchParams* p, const EmSearchBuffers* b, int32_t i, void* stream) {
  if (!p || !b || !b->lm || !b->lm_logp || !b->tok || p->B <= 0 || p->W <= 0 || i < 0 || i >= p->Lmax)
    return EM_ERR_BAD_ARG;
  if (b->step) return EM_ERR_BAD_ARG;  // host-driven positions only
  return lm_step(dtype, p, b, i, stream);
}

extern "C" int em_ctc_log_probs_t(int dtype, const void* enc_act, int32_t B, int32_t T, int32_t d_model,
                                  const void* ctc_w, const float* ctc_b, int32_t V, float* lpT, void* stream) {
  if (B <= 0 || T <= 0 || V <= 1 || !lpT) return EM_ERR_BAD_ARG;
  EmSearchParams p = {};
  p.B = B; p.T = T; p.V = V;
  EmSearchBuffers b = {};
  b.ctc_lpT = lpT;
  return ctc_log_probs(dtype, &p, &b, enc_act, d_model, ctc_w, ctc_b, stream);
}

extern "C" int em_ctc_prefix_init(const float* lpT, const int32_t* xlens, int32_t B, int32_t T, int32_t blank,
                                  float* r0, void* stream) {
  if (!lpT || !xlens || !r0 || B <= 0 || T <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(ctc_prefix_init_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, lpT, xlens, B, T, blank,
                     (float2*)r0);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_ctc_prefix_score(const float* lpT, const int32_t* xlens, const float* r_prev, const float* s_prev,
                                   const int32_t* last_ids, const int32_t* cand_ids, int32_t B, int32_t W, int32_t S,
                                   int32_t T, int32_t V, int32_t out_len, int32_t eos, int32_t blank, float* log_psi,
                                   float* scores, void* stream) {
  if (!lpT || !xlens || !r_prev || !last_ids || !log_psi || B <= 0 || W <= 0 || T <= 0 || T > CTC_TMAX || V <= 1 ||
      out_len < 0)
    return EM_ERR_BAD_ARG;
  if (scores && !s_prev) return EM_ERR_BAD_ARG;
  if (cand_ids ? (S <= 0 || S >= V) : S != V) return EM_ERR_BAD_ARG;
  const int NC = cand_ids ? S + 1 : V;
  const long waves = (long)B * W * NC;
  hipLaunchKernelGGL(ctc_prefix_score_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     lpT, xlens, (const float2*)r_prev, s_prev, last_ids, cand_ids, B, W, S, NC, T, out_len, eos,
                     blank, log_psi, scores);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_ctc_prefix_state(const float* lpT, const int32_t* xlens, const float* r_prev,
                                   const int32_t* last_ids, const int32_t* rows, const int32_t* toks, int32_t m,
                                   int32_t B, int32_t W, int32_t T, int32_t out_len, int32_t blank, float* r_out,
                                   void* stream) {
  if (!lpT || !xlens || !r_prev || !last_ids || !rows || !toks || !r_out || m <= 0 || B <= 0 || W <= 0 || T <= 0 ||
      T > CTC_TMAX || out_len < 0)
    return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(ctc_prefix_state_kernel, dim3(m), dim3(64), 0, (hipStream_t)stream, lpT, xlens,
                     (const float2*)r_prev, last_ids, rows, toks, B, W, T, out_len, blank, (float2*)r_out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
