// phf_single_level.hip — gfx950 kernels for PyHillFit's single-level (non-hierarchical) sampler.
//
// One lane = one Markov chain.  A 64-lane wavefront advances 64 chains of the SAME (pair, temperature)
// problem in lock-step, so the dose/response points are wave-uniform: they are staged once per block into
// LDS and every per-point branch (censored / uncensored) is uniform.  Chain state lives in registers for
// the whole launch (loaded/stored once, struct-of-arrays, 512 B per wave-instruction); the only traffic
// inside the loop is the thinned sample store.  No MFMA: this is scalar-per-chain fp64 arithmetic.
//
// Arithmetic follows the reference exactly where it defines the result (python/doseresponse.py:84-88,
// 151-189,203-248,304-317; python/PyHillFit.py:830-848) and uses phf_math.h's fixed-order elementary
// functions so that the host twin in oracle/ reproduces every chain bit for bit (compile with
// -ffp-contract=off: every fma below is explicit).
#include <hip/hip_runtime.h>

#include "../../include/pyhillfit_amd.h"
#include "phf_common.h"
#include "phf_hier3_isa.h"
#include "phf_model.h"

namespace {

constexpr int kBlock = 64;
thread_local int g_last_sl_kernel = 0;        // phf_single_level_last_kernel(): 1 hipcc, 2 gfx950 assembly, 3 gfx950 assembly as a work queue
constexpr int kQueuePoison = 0x40000000;     // task counter value that makes every wavefront of a queued launch leave
constexpr int kQueueMaxRounds = 16;          // beyond this many rounds of the chip's wavefront slots the tail of a launch is negligible

template <int MODEL> struct Dim { static constexpr int d = (MODEL == 1) ? 2 : 3; };

#ifndef PHF_CHOL_FUSED_PIVOT
#define PHF_CHOL_FUSED_PIVOT 1
#endif
// packed lower-triangular Cholesky, D <= 3, fully unrolled; non-positive pivot -> zero column
template <int D>
__device__ __forceinline__ void chol_packed(const double* c, double* l) {
  double inv[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = c[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s = phf_fma(-l[i * (i + 1) / 2 + k], l[j * (j + 1) / 2 + k], s);
      if (i == j) {   // sqrt and reciprocal are computed unconditionally (no divergent branch); a non-positive pivot selects 0 —
                      // ONE predicate for both selects (sqrt(s) > 0 exactly when s > 0)
        const bool pos = s > 0.0;
        double ir = 0.0;
#if PHF_CHOL_FUSED_PIVOT
        const double rs = (i + 1 < D) ? phf_sqrt_rcp_pos(s, &ir) : phf_sqrt_pos(s);    // the last pivot's reciprocal is not used
#else
        const double rs = phf_sqrt_pos(s);                  // (A/B builds: root and reciprocal from an estimate each)
        ir = phf_rcp(rs);
#endif
        const double r = pos ? rs : 0.0;
        l[i * (i + 1) / 2 + i] = r;
        inv[i] = pos ? ir : 0.0;
      } else {
        l[i * (i + 1) / 2 + j] = s * inv[j];
      }
    }
  }
}

struct AdvanceArgs {
  phf_points pts;
  phf_problems prob;
  phf_mh_config cfg;
  int64_t t_begin, t_end;
  double* state;
  double* rows;
  double* moments;
  int64_t moments_after;
  int32_t blocks_per_problem;
  // work queue (phf_single_level_advance_queued): queue[0] = next task, queue[1 + b] = quanta of block b that are complete
  int32_t* queue;
  int32_t quantum;
};

// stage one pair's entries into LDS (lc = s_pts, y = s_pts + stride, w = s_pts + 2 stride); returns counts
__device__ __forceinline__ void stage_points(const phf_points& pts, int pair, double* s_pts, int& n_other,
                                             int& n_zero, int& n_hundred) {
  const int32_t* cnt = pts.counts + 4 * pair;
  n_other = cnt[0]; n_zero = cnt[1]; n_hundred = cnt[2];
  const int n = n_other + n_zero + n_hundred;
  for (int j = threadIdx.x; j < n; j += kBlock) {
    s_pts[j] = pts.ln_conc[(size_t)pair * pts.stride + j];
    s_pts[pts.stride + j] = pts.response[(size_t)pair * pts.stride + j];
    s_pts[2 * pts.stride + j] = pts.weight[(size_t)pair * pts.stride + j];
  }
  __syncthreads();
}

// The whole launch of one wavefront: state in, iterations t_begin+1..t_end, state out.
// KO / KC >= 0: the pair's numbers of uncensored / censored entries are these compile-time constants (the point loops of
// phf_sl_log_target fold away and an iteration is straight-line code the scheduler can interleave: measured 17 % faster
// on Amiodarone-hERG); -1: read from the data at run time.  Either way the same operations in the same order.
template <int MODEL, bool MOMENTS, int KO, int KC, bool LONE_WAVE>
__device__ __forceinline__ void advance_body(const AdvanceArgs& a, const double* s_pts, int q, int c, int pair,
                                             int n_other_rt, int n_cens_rt, const int64_t t_begin, const int64_t t_end) {
  constexpr int D = Dim<MODEL>::d;
  constexpr int NTRI = D * (D + 1) / 2;
  const int C = a.prob.chains_per_problem;
  const int n_other = (KO >= 0) ? KO : n_other_rt;
  const int n_cens = (KC >= 0) ? KC : n_cens_rt;
  const double* lc = s_pts;
  const double* yv = s_pts + a.pts.stride;
  const double* wv = s_pts + 2 * a.pts.stride;
  const double pi_bit = a.pts.pi_bit[pair];
  const double n_other_points = a.pts.extra[2 * pair], ss_within = a.pts.extra[2 * pair + 1];
  const double temperature = a.prob.temperature[q];
  const uint32_t pid = a.prob.problem_id[q];
  const uint32_t cid = a.prob.chain_id_base + (a.prob.chain_offset ? a.prob.chain_offset[q] : 0u) + (uint32_t)c;
  uint32_t seed_lo = (uint32_t)a.cfg.seed, seed_hi = (uint32_t)(a.cfg.seed >> 32);
  // Two wavefronts per SIMD (256 registers): the Philox key words as VECTOR values.  As scalars the ten round keys are loop
  // invariants hipcc hoists into 20 SGPRs and then spills to VGPR lanes (a v_readlane + s_nop per round); as vector values they
  // cost registers the scratch spills outside the loop absorb: C3 159.1 -> 158.0 ms per 8 000 iterations on one box, twice;
  // measured again after the table-driven functions (no scalar-memory fetches left to compete for SGPRs): 106.7 as scalars, 104.4
  // as vectors.  The lone-wavefront build keeps them scalar (round 2: scalar work is free there, vector issue slots are not).
  if (!LONE_WAVE) { asm volatile("" : "+v"(seed_lo)); asm volatile("" : "+v"(seed_hi)); }
  const size_t nchains = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;

  // ---- load state (SoA, coalesced) ----
  double th[D], mean[D], cov[NTRI], L[NTRI];
  double* sp = a.state + g;
#pragma unroll
  for (int i = 0; i < D; ++i) th[i] = sp[(size_t)i * nchains];
  double lt = sp[(size_t)D * nchains];
#pragma unroll
  for (int i = 0; i < D; ++i) mean[i] = sp[(size_t)(D + 1 + i) * nchains];
#pragma unroll
  for (int i = 0; i < NTRI; ++i) cov[i] = sp[(size_t)(2 * D + 1 + i) * nchains];
  double loga = sp[(size_t)(2 * D + 1 + NTRI) * nchains];
  double nacc = sp[(size_t)(2 * D + 2 + NTRI) * nchains];
  double ll1 = sp[(size_t)(2 * D + 3 + NTRI) * nchains];   // untempered log-likelihood of the current state
  // exp and log coefficients: in VGPRs for the whole launch (8 doubles)
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  chol_packed<D>(cov, L);
  double sc = phf_exp_fast_k(0.5 * loga, k_exp);

  double m1[D + 1], m2[D + 1], mll = 0.0;
  constexpr bool want_moments = MOMENTS;               // compiled out (and its 18 VGPRs freed) when no moments are requested
  if (want_moments) {
#pragma unroll
    for (int i = 0; i <= D; ++i) {
      m1[i] = a.moments[(size_t)i * nchains + g];
      m2[i] = a.moments[(size_t)(D + 1 + i) * nchains + g];
    }
    mll = a.moments[(size_t)(2 * D + 2) * nchains + g];
  }

  const int thin = a.cfg.thinning;
  int until_save = thin - (int)(t_begin % thin);   // iterations until the next t with t % thin == 0
  // rows[r][q][f][c]
  const bool save_rows = a.rows != nullptr;             // wave-uniform (scalar) test, not a per-lane pointer compare
  const size_t row_stride = (size_t)a.prob.num_problems * (D + 1) * C;
  // a.rows holds the rows of the whole launch (from a.t_begin on); this call may start later (a quantum of a queued launch)
  double* out = a.rows + ((size_t)q * (D + 1)) * C + c + (size_t)(t_begin / thin - a.t_begin / thin) * row_stride;

  // The iteration is ONE basic block up to the sample store: the adaptation is applied unconditionally with gamma = 0 before
  // it starts (cov, mean, loga are then reproduced exactly: (1-0) x + 0 y), and the random numbers of iteration t+1 — a
  // function of (chain, t+1) only — are drawn next to the factorisation of iteration t, so that the two long dependency
  // chains of an iteration (Philox -> log -> sqrt, and sqrt -> divide -> sqrt -> divide -> sqrt) overlap each other and
  // the likelihood instead of being exposed one after the other on a wavefront that has its SIMD to itself.
  double z[3];
  double log_u = phf_mh_draws(D, cid, pid, (uint32_t)(t_begin + 1), seed_lo, seed_hi, k_log, z);
  const bool reset_mean = a.cfg.reset_mean_at_adapt_start != 0;
  for (int64_t t = t_begin + 1; t <= t_end; ++t) {
    // ---- proposal: theta* = theta + e^(loga/2) L z  (PyHillFit.py:831) ----
    double star[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double v = L[i * (i + 1) / 2 + i] * z[i];
#pragma unroll
      for (int k = i - 1; k >= 0; --k) v = phf_fma(L[i * (i + 1) / 2 + k], z[k], v);
      star[i] = phf_fma(sc, v, th[i]);
    }
    // ---- target and accept test (PyHillFit.py:833-838) ----
    double lik_star, prior_star, ll1_star;
    phf_sl_log_target(MODEL, lc, yv, wv, n_other, n_cens, n_other_points, ss_within, pi_bit, temperature, star, k_exp, k_log, &lik_star, &prior_star, &ll1_star);
    const double lt_star = lik_star + prior_star;
    const bool acc = log_u < lt_star - lt;
    if (acc) {
#pragma unroll
      for (int i = 0; i < D; ++i) th[i] = star[i];
      lt = lt_star;
      ll1 = ll1_star;
    }
    nacc += acc ? 1.0 : 0.0;
    // ---- adaptation (PyHillFit.py:840-846; PyHillTemp.py:114-122), wave-uniform selects instead of branches ----
    const bool adapting = t > a.cfg.adapt_start;
    const bool reset_now = reset_mean && t == a.cfg.adapt_start;                       // PyHillTemp.py:114-115: mean <- theta
    const double gs_tab = a.cfg.gamma[adapting ? t - a.cfg.adapt_start : 0];
    const double gs = adapting ? gs_tab : 0.0;
    const double omg = 1.0 - gs;
    const double gm = reset_now ? 1.0 : gs, omm = reset_now ? 0.0 : omg;               // mean: 1 theta + 0 mean on the reset step
    {
      double v[D];
#pragma unroll
      for (int i = 0; i < D; ++i) v[i] = th[i] - mean[i];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
          cov[i * (i + 1) / 2 + j] = phf_fma(gs, v[i] * v[j], omg * cov[i * (i + 1) / 2 + j]);
#pragma unroll
      for (int i = 0; i < D; ++i) mean[i] = phf_fma(gm, th[i], omm * mean[i]);
      loga = phf_fma(gs, acc ? 0.75 : -0.25, loga);                                      // accepted - target_acceptance (1 - 0.25 is exact)
    }
    // ---- draws of the next iteration, factor and scale of the next proposal ----
    double z_next[3];
    const double log_u_next = phf_mh_draws(D, cid, pid, (uint32_t)(t + 1), seed_lo, seed_hi, k_log, z_next);
    chol_packed<D>(cov, L);
    sc = phf_exp_fast_k(0.5 * loga, k_exp);
#pragma unroll
    for (int i = 0; i < 3; ++i) z[i] = z_next[i];
    log_u = log_u_next;
    // ---- thinning + sample store (PyHillFit.py:847-848) ----
    if (--until_save == 0) {
      until_save = thin;
      if (save_rows) {
#pragma unroll
        for (int i = 0; i < D; ++i) out[(size_t)i * C] = th[i];
        out[(size_t)D * C] = lt;
        out += row_stride;
      }
      if (want_moments && t > a.moments_after) {
#pragma unroll
        for (int i = 0; i < D; ++i) { m1[i] += th[i]; m2[i] = phf_fma(th[i], th[i], m2[i]); }
        m1[D] += lt; m2[D] = phf_fma(lt, lt, m2[D]);
        mll += ll1;
      }
    }
  }

  // ---- store state ----
#pragma unroll
  for (int i = 0; i < D; ++i) sp[(size_t)i * nchains] = th[i];
  sp[(size_t)D * nchains] = lt;
#pragma unroll
  for (int i = 0; i < D; ++i) sp[(size_t)(D + 1 + i) * nchains] = mean[i];
#pragma unroll
  for (int i = 0; i < NTRI; ++i) sp[(size_t)(2 * D + 1 + i) * nchains] = cov[i];
  sp[(size_t)(2 * D + 1 + NTRI) * nchains] = loga;
  sp[(size_t)(2 * D + 2 + NTRI) * nchains] = nacc;
  sp[(size_t)(2 * D + 3 + NTRI) * nchains] = ll1;
  if (want_moments) {
#pragma unroll
    for (int i = 0; i <= D; ++i) {
      a.moments[(size_t)i * nchains + g] = m1[i];
      a.moments[(size_t)(D + 1 + i) * nchains + g] = m2[i];
    }
    a.moments[(size_t)(2 * D + 2) * nchains + g] = mll;
  }
}

// Entry-count shapes with a straight-line body: 1..4 uncensored x 0..4 censored entries = 203 of the 210 Crumb pairs
// (the set has 4 nominal doses per pair); anything else runs the run-time loops.  The moment-accumulating builds (what the
// command lines and the thermodynamic-integration path run) have the same bodies: with the run-time loops a wavefront that has
// its SIMD to itself — 64 chains per pair — was 25 % slower (C2 shape: 4.34 ms against 3.27).
#define PHF_SHAPE_CASE(ko, kc) \
  case (ko) * 8 + (kc): advance_body<MODEL, MOMENTS, ko, kc, WPS == 1>(a, s_pts, q, c, pair, n_other, n_cens, t_begin, t_end); break;
#define PHF_SHAPE_ROW(ko) PHF_SHAPE_CASE(ko, 0) PHF_SHAPE_CASE(ko, 1) PHF_SHAPE_CASE(ko, 2) PHF_SHAPE_CASE(ko, 3) PHF_SHAPE_CASE(ko, 4)

// One block (= one wavefront = 64 chains of one problem) from iteration t_begin to t_end: state in, samples out, state out.
template <int MODEL, bool MOMENTS, int WPS>
__device__ __forceinline__ void run_block(const AdvanceArgs& a, double* s_pts, int block, const int64_t t_begin, const int64_t t_end) {
  const int slot = block / a.blocks_per_problem;
  const int chunk = block - slot * a.blocks_per_problem;
  const int q = a.prob.launch_order ? a.prob.launch_order[slot] : slot;     // which problem this wavefront works on (wave-uniform)
  const int c = chunk * kBlock + threadIdx.x;
  int n_other, n_zero, n_hundred;
  const int pair = a.prob.pair_index[q];
  stage_points(a.pts, pair, s_pts, n_other, n_zero, n_hundred);
  if (c >= a.prob.chains_per_problem) return;
  const int n_cens = n_zero + n_hundred;
  if (n_other <= 4 && n_cens <= 4) {
    switch (n_other * 8 + n_cens) {                     // wave-uniform
      PHF_SHAPE_ROW(1) PHF_SHAPE_ROW(2) PHF_SHAPE_ROW(3) PHF_SHAPE_ROW(4)
      default: advance_body<MODEL, MOMENTS, -1, -1, WPS == 1>(a, s_pts, q, c, pair, n_other, n_cens, t_begin, t_end); break;
    }
    return;
  }
  advance_body<MODEL, MOMENTS, -1, -1, WPS == 1>(a, s_pts, q, c, pair, n_other, n_cens, t_begin, t_end);
}

// WPS = wavefronts per SIMD the register allocation allows for: 2 (256 registers) for launches that fill the chip more than
// once, 1 (512 registers: VGPRs + AGPRs, nothing spills to scratch) when every wavefront has a SIMD to itself anyway.
//
// Plain launch (a.queue == NULL): block b of the grid runs block b of the batch for the whole launch.
// Queued launch: the grid is only as large as the chip holds at once and its wavefronts PULL tasks from a counter in HBM.  A task
// is one QUANTUM (a.quantum iterations) of one block; tasks are numbered quantum-major, so that within every quantum the blocks
// come most expensive first (launch_order) and the last tasks of the launch are short: pairs have 2..8 entries, an iteration
// 550..1 500 instructions, and with one whole-launch job per block the chip idles ~5 % at the end of a 210-pair launch even in
// the best order (1.7 % with 4 quanta, 0.7 % with 8: list-scheduling simulation).  A block's quanta chain
// through its chain state in HBM: the wavefront that finishes quantum k of block b releases (agent scope) and publishes
// queue[1 + b] = k + 1; the one that pulled quantum k + 1 — issued a whole round of tasks later, so normally long complete —
// polls that word, acquires, and loads the state.  No cycle is possible (a task waits only for a task with a smaller number,
// which a RUNNING wavefront holds), whatever part of the grid is resident; a poll that does not end within ~2^24 sleeps poisons
// the counter so that every wavefront drains.
template <int MODEL, bool MOMENTS, int WPS>
__global__ __launch_bounds__(kBlock, WPS) void mh_advance_kernel(const AdvanceArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  PHF_LOGPHI_TABLE_TO_LDS();
  extern __shared__ double s_pts[];
  const bool queued = a.queue != nullptr;                         // wave-uniform; ONE call site of run_block serves both kinds of launch
  const int nblocks = a.blocks_per_problem * a.prob.num_problems;
  const int total = queued ? nblocks * (int)((a.t_end - a.t_begin + a.quantum - 1) / a.quantum) : 0;
  for (;;) {
    int b = blockIdx.x, k = 0;
    int64_t t0 = a.t_begin, t1 = a.t_end;
    if (queued) {
      int task = 0;
      if (threadIdx.x == 0) task = atomicAdd(a.queue, 1);
      task = __builtin_amdgcn_readfirstlane(task);
      if (task >= total) break;
      k = task / nblocks; b = task - k * nblocks;
      if (k > 0) {
        int polls = 0;
        while (__hip_atomic_load(a.queue + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
          __builtin_amdgcn_s_sleep(16);
          // cannot happen in a correct run: give up after ~2^24 sleeps — or as soon as another wavefront has — and DRAIN instead of
          // hanging: poison the task counter, raise the sticky fault word (never cleared by the library; the host reads it through
          // phf_single_level_queue_status) and leave WITHOUT advancing this block
          ++polls;
          const bool gave_up = polls > (1 << 24) ||
                               ((polls & 1023) == 0 && __hip_atomic_load(a.queue + 1 + nblocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
          if (gave_up) {
            if (threadIdx.x == 0) {
              __hip_atomic_store(a.queue + 1 + nblocks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(a.queue, kQueuePoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      t0 = a.t_begin + (int64_t)k * a.quantum;
      t1 = (t0 + a.quantum < a.t_end) ? t0 + a.quantum : a.t_end;
    }
    run_block<MODEL, MOMENTS, WPS>(a, s_pts, b, t0, t1);
    if (!queued) break;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");           // every lane: its part of this block's state (and rows) is in HBM ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // ... all lanes' parts (a barrier, so the hand-over does not lean on the
                                                                 // block being ONE lock-step wavefront), and s_pts is free to be restaged ...
    if (threadIdx.x == 0) __hip_atomic_store(a.queue + 1 + b, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the word that says so
  }
}

struct InitArgs {
  phf_points pts;
  phf_problems prob;
  int32_t cov_identity;
  double cov_scale;
  const double* theta0;
  double* state;
  double* row0;
  int32_t blocks_per_problem;
};

template <int MODEL>
__global__ __launch_bounds__(kBlock) void mh_init_kernel(const InitArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_LOGPHI_TABLE_TO_LDS();
  constexpr int D = Dim<MODEL>::d;
  constexpr int NTRI = D * (D + 1) / 2;
  extern __shared__ double s_pts[];
  const int q = blockIdx.x / a.blocks_per_problem;
  const int chunk = blockIdx.x - q * a.blocks_per_problem;
  const int C = a.prob.chains_per_problem;
  const int c = chunk * kBlock + threadIdx.x;
  int n_other, n_zero, n_hundred;
  const int pair = a.prob.pair_index[q];
  stage_points(a.pts, pair, s_pts, n_other, n_zero, n_hundred);
  if (c >= C) return;
  const size_t nchains = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;
  double th[D];
#pragma unroll
  for (int i = 0; i < D; ++i) th[i] = a.theta0[(size_t)i * nchains + g];
  double lik0, prior0, ll10;
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  phf_sl_log_target(MODEL, s_pts, s_pts + a.pts.stride, s_pts + 2 * a.pts.stride, n_other, n_zero + n_hundred,
                    a.pts.extra[2 * pair], a.pts.extra[2 * pair + 1], a.pts.pi_bit[pair], a.prob.temperature[q],
                    th, k_exp, k_log, &lik0, &prior0, &ll10);
  const double lt = lik0 + prior0;
  double* sp = a.state + g;
#pragma unroll
  for (int i = 0; i < D; ++i) { sp[(size_t)i * nchains] = th[i]; sp[(size_t)(D + 1 + i) * nchains] = th[i]; }
  sp[(size_t)D * nchains] = lt;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
      sp[(size_t)(2 * D + 1 + i * (i + 1) / 2 + j) * nchains] =
          (i != j) ? 0.0 : (a.cov_identity ? a.cov_scale : a.cov_scale * __builtin_fabs(th[i]));
  sp[(size_t)(2 * D + 1 + NTRI) * nchains] = 0.0;
  sp[(size_t)(2 * D + 2 + NTRI) * nchains] = 0.0;
  sp[(size_t)(2 * D + 3 + NTRI) * nchains] = ll10;
  if (a.row0) {
    double* o = a.row0 + ((size_t)q * (D + 1)) * C + c;
#pragma unroll
    for (int i = 0; i < D; ++i) o[(size_t)i * C] = th[i];
    o[(size_t)D * C] = lt;
  }
}

template <int MODEL>
__global__ __launch_bounds__(256) void log_target_kernel(const phf_points pts, int64_t m, const int32_t* pair_index,
                                                          const double* temperature, const double* theta,
                                                          double* out_lik, double* out_prior) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_LOGPHI_TABLE_TO_LDS();
  constexpr int D = Dim<MODEL>::d;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int pair = pair_index[i];
  const int32_t* cnt = pts.counts + 4 * pair;
  double th[D];
#pragma unroll
  for (int k = 0; k < D; ++k) th[k] = theta[(size_t)k * m + i];
  double lik, prior, ll1;
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  phf_sl_log_target(MODEL, pts.ln_conc + (size_t)pair * pts.stride, pts.response + (size_t)pair * pts.stride,
                    pts.weight + (size_t)pair * pts.stride, cnt[0], cnt[1] + cnt[2], pts.extra[2 * pair], pts.extra[2 * pair + 1],
                    pts.pi_bit[pair], temperature[i], th, k_exp, k_log, &lik, &prior, &ll1);
  if (out_lik) out_lik[i] = lik;
  if (out_prior) out_prior[i] = prior;
}

__global__ void debug_math_kernel(int fn, int64_t n, const double* in, double* out) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_LOGPHI_TABLE_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  double r, s, c;
  switch (fn) {
    case 0: r = phf_exp(x); break;
    case 1: r = phf_log(x); break;
    case 2: r = phf_erfcx_nonneg(x); break;
    case 3: r = phf_log_ndtr(x); break;
    case 4: r = phf_ndtr(x); break;
    case 5: r = phf_sqrt(x); break;
    case 6: r = 1.0 / x; break;
    case 7: phf_sincos_2pi_u32((uint32_t)x, &s, &c); r = s; break;
    case 9: r = phf_exp_fast(x); break;
    case 10: r = phf_log_fast(x); break;
    case 11: r = phf_log_ndtr_nonpos(x); break;
    case 12: r = phf_rcp(x); break;
    case 13: r = phf_sqrt_pos(x); break;
    case 14: r = phf_div(PHF_LN10, x); break;
    case 15: r = phf_sqrt_nonneg(x); break;
    case 16: r = phf_div(x, PHF_LN10); break;
    case 17: r = phf_normal_u32((uint32_t)x); break;
    case 18: r = phf_log_ndtr_tab(x, -x * PHF_INV_SQRT2); break;
    case 19: r = phf_erfc_tab(x); break;
    case 20: (void)phf_sqrt_rcp_pos(x, &r); break;      // the reciprocal half; 21: the square-root half
    case 21: r = phf_sqrt_rcp_pos(x, &s); break;
    default: phf_sincos_2pi_u32((uint32_t)x, &s, &c); r = c; break;
  }
  out[i] = r;
}

template <int ROUNDS>
__global__ void debug_philox_kernel(int64_t n, const uint32_t* ck, uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* c = ck + 6 * i;
  const phf_u32x4 r = phf_philox4x32_r(ROUNDS, c[0], c[1], c[2], c[3], c[4], c[5]);
  for (int k = 0; k < 4; ++k) out[4 * i + k] = r.w[k];
}

int check_common(const phf_points* pts, const phf_problems* prob, int model) {
  if (!pts || !prob) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null points/problems");
  if (model != 1 && model != 2) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "model must be 1 or 2");
  if (pts->num_pairs <= 0 || pts->stride <= 0 || !pts->ln_conc || !pts->response || !pts->weight || !pts->counts || !pts->pi_bit || !pts->extra)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "incomplete phf_points");
  if (prob->num_problems <= 0 || prob->chains_per_problem <= 0 || !prob->pair_index || !prob->temperature || !prob->problem_id)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "incomplete phf_problems");
  if ((size_t)pts->stride * 24 > 160 * 1024) return phf_fail(PHF_ERR_UNSUPPORTED, "pair does not fit in LDS");
  const int64_t bpp = (prob->chains_per_problem + kBlock - 1) / kBlock;
  if (bpp * prob->num_problems > 0x7fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many chains for one launch");
  return PHF_OK;
}

}  // namespace

extern "C" {

int phf_version(void) { return PHF_ABI_VERSION; }
int phf_single_level_last_kernel(void) { return g_last_sl_kernel; }

int phf_single_level_state_size(int model) {
  if (model == 1) return 2 * 2 + 3 + 4;
  if (model == 2) return 2 * 3 + 6 + 4;
  return phf_fail(PHF_ERR_INVALID_ARGUMENT, "model must be 1 or 2");
}

int phf_single_level_init(const phf_points* pts, const phf_problems* prob, int model, int cov_identity,
                          double cov_scale, const double* theta0, double* state, double* row0, void* stream) {
  if (int rc = check_common(pts, prob, model)) return rc;
  if (!theta0 || !state) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null theta0/state");
  phf_forget_device_memory_verdicts();              // a new sampler: every buffer's kind is asked of the runtime again (phf_common.h)
  InitArgs a{*pts, *prob, cov_identity, cov_scale, theta0, state, row0, 0};
  a.blocks_per_problem = (prob->chains_per_problem + kBlock - 1) / kBlock;
  const dim3 grid((unsigned)(a.blocks_per_problem * prob->num_problems)), block(kBlock);
  const size_t lds = (size_t)pts->stride * 24;
  if (model == 1) hipLaunchKernelGGL(mh_init_kernel<1>, grid, block, lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mh_init_kernel<2>, grid, block, lds, (hipStream_t)stream, a);
  return phf_check_launch("phf_single_level_init");
}

static int advance_impl(const phf_points* pts, const phf_problems* prob, const phf_mh_config* cfg, int64_t t_begin, int64_t t_end,
                        double* state, double* rows, double* moments, int64_t moments_after, int32_t quantum, int32_t* queue,
                        void* stream) {
  if (!cfg) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null config");
  if (int rc = check_common(pts, prob, cfg->model)) return rc;
  if (!state) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null state");
  if (cfg->thinning <= 0) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "thinning must be positive");
  if (t_begin < 0 || t_end < t_begin || t_end > 0xffffffffLL) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad iteration range");
  if (prob->kernel_hint & ~48u)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "kernel_hint of a single-level launch: bit 4 = the hipcc kernel, bit 5 = the gfx950 assembly build; the other bits must be 0");
  // gamma[0] is read (and multiplied by zero) on every iteration before the adaptation starts: the table is always needed
  if (!cfg->gamma) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "gamma table required (gamma[0] is read even before adaptation starts)");
  if (t_end == t_begin) return PHF_OK;
  if (int rc = phf_require_device_memory(state, "state")) return rc;
  if (int rc = phf_require_device_memory(moments, "moments")) return rc;
  if (int rc = phf_require_device_memory(queue, "queue workspace")) return rc;
  AdvanceArgs a{*pts, *prob, *cfg, t_begin, t_end, state, rows, moments, moments_after, 0, nullptr, 0};
  a.blocks_per_problem = (prob->chains_per_problem + kBlock - 1) / kBlock;
  const int64_t nblocks = (int64_t)a.blocks_per_problem * prob->num_problems;
  const size_t lds = (size_t)pts->stride * 24;
  const bool lone = nblocks <= phf_simd_count();            // one wavefront per SIMD at most: let it have the whole register file
  hipStream_t s = (hipStream_t)stream;
  int64_t grid_blocks = nblocks;
  if (queue && quantum > 0) {
    // queued: worth it only when the launch is several rounds of the chip's 2-per-SIMD slots and has at least two quanta
    const int64_t slots = 2 * phf_simd_count();
    const int64_t nquanta = (t_end - t_begin + quantum - 1) / quantum;
    if (nquanta * nblocks > 0x3fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many tasks for one queued launch");
    if (!lone && nquanta >= 2 && nblocks > slots && nblocks <= kQueueMaxRounds * slots) {
      if (hipMemsetAsync(queue, 0, (size_t)(1 + nblocks) * sizeof(int32_t), s) != hipSuccess) return phf_check_launch("phf_single_level_advance_queued (memset)");
      a.queue = queue; a.quantum = quantum;
      grid_blocks = slots;
    }
  }
  // The hand-allocated gfx950 build of the model-2 iteration (phf_hier3_isa.hip: phf_sl3_advance; tools/gen_sl_isa_main.py): OPT-IN —
  // kernel_hint bit 5 per launch, or PHF_SL_ISA=1 in the environment (read once) — for launches that do not ask for moments, give more
  // than one wavefront per SIMD and whose pairs have at most PHF_ISA_SL_MAX_STRIDE entries; plain or queued exactly as decided above.
  // Same numbers bit for bit.  Not the default because it is not faster: 487 vector instructions per iteration against hipcc's 510, 126
  // registers (four wavefronts per SIMD) against 256 — and 311 ms against 304 per 24 000 C3 iterations: both builds keep the vector pipe
  // ~100 % busy, 74 % of its cycles on the 360 fp64 instructions the twin's operation sequence fixes (4 cycles each); what the hand
  // allocation removes are 2-cycle integer instructions and copies (profiles/r05/c3_assembly_kernel.txt).
  static const bool isa_on = [] { const char* e = getenv("PHF_SL_ISA"); return e && e[0] == '1' && e[1] == 0; }();
  if (cfg->model == 2 && !moments && !lone && (isa_on || (prob->kernel_hint & 32u)) && !(prob->kernel_hint & 16u) && pts->stride <= PHF_ISA_SL_MAX_STRIDE &&
      cfg->adapt_start >= 0 && t_end < 0xffffffffLL && nblocks <= 0x7fffffffLL && phf_sl3_isa_available()) {
    phf_sl3_isa_args g{};
    g.state = state; g.rows = rows; g.gamma = cfg->gamma;
    g.ln_conc = pts->ln_conc; g.response = pts->response; g.weight = pts->weight; g.counts = pts->counts; g.pi_bit = pts->pi_bit; g.extra = pts->extra;
    g.pair_index = prob->pair_index; g.temperature = prob->temperature; g.problem_id = prob->problem_id;
    g.launch_order = prob->launch_order; g.chain_offset = prob->chain_offset;
    g.t_begin = (uint32_t)t_begin; g.t_end = (uint32_t)t_end;
    g.adapt_start = (uint32_t)(cfg->adapt_start > 0xffffffffLL ? 0xffffffffLL : cfg->adapt_start);
    g.thinning = cfg->thinning; g.reset_mean = cfg->reset_mean_at_adapt_start != 0 ? 1u : 0u;
    g.chains = prob->chains_per_problem; g.num_problems = prob->num_problems; g.bpp = a.blocks_per_problem;
    g.bpp_magic = phf_isa_magic((uint32_t)a.blocks_per_problem); g.total_waves = (int32_t)nblocks;
    g.seed_lo = (uint32_t)cfg->seed; g.seed_hi = (uint32_t)(cfg->seed >> 32);
    g.chain_id_base = prob->chain_id_base; g.pts_stride = pts->stride;
    g.until_save0 = cfg->thinning - (int32_t)(t_begin % cfg->thinning);
    bool ok = true;
    if (a.queue) {
      // a quantum must save whole rows: a multiple of the thinning from a start that is one (what the queued callers use)
      const int64_t nquanta = (t_end - t_begin + a.quantum - 1) / a.quantum;
      ok = a.quantum % cfg->thinning == 0 && t_begin % cfg->thinning == 0 && nquanta * nblocks < (1LL << 31);
      g.queue = a.queue; g.quantum = (uint32_t)a.quantum; g.num_tasks = (int32_t)(nquanta * nblocks);
      g.blocks_magic = phf_isa_magic((uint32_t)nblocks); g.rows_per_quantum = (uint32_t)(a.quantum / cfg->thinning);
    }
    if (ok) {
      g_last_sl_kernel = a.queue ? 3 : 2;
      // 126 VGPRs: FOUR wavefronts per SIMD — a queued launch's persistent grid is that large (the hipcc kernels': two per SIMD)
      const int64_t isa_slots = 4LL * phf_simd_count();
      const int64_t grid_waves = a.queue ? (nblocks < isa_slots ? nblocks : isa_slots) : nblocks;
      return phf_sl3_isa_advance(&g, (int)grid_waves, s);
    }
  }
  g_last_sl_kernel = 1;
  const dim3 grid((unsigned)grid_blocks), block(kBlock);
#define PHF_LAUNCH_ADVANCE(M, MOM, W) hipLaunchKernelGGL((mh_advance_kernel<M, MOM, W>), grid, block, lds, s, a)
  if (cfg->model == 1) {
    if (moments) { if (lone) PHF_LAUNCH_ADVANCE(1, true, 1); else PHF_LAUNCH_ADVANCE(1, true, 2); }
    else { if (lone) PHF_LAUNCH_ADVANCE(1, false, 1); else PHF_LAUNCH_ADVANCE(1, false, 2); }
  } else {
    if (moments) { if (lone) PHF_LAUNCH_ADVANCE(2, true, 1); else PHF_LAUNCH_ADVANCE(2, true, 2); }
    else { if (lone) PHF_LAUNCH_ADVANCE(2, false, 1); else PHF_LAUNCH_ADVANCE(2, false, 2); }
  }
#undef PHF_LAUNCH_ADVANCE
  return phf_check_launch("phf_single_level_advance");
}

int phf_single_level_advance(const phf_points* pts, const phf_problems* prob, const phf_mh_config* cfg,
                             int64_t t_begin, int64_t t_end, double* state, double* rows, double* moments,
                             int64_t moments_after, void* stream) {
  return advance_impl(pts, prob, cfg, t_begin, t_end, state, rows, moments, moments_after, 0, nullptr, stream);
}

int phf_single_level_advance_queued(const phf_points* pts, const phf_problems* prob, const phf_mh_config* cfg,
                                    int64_t t_begin, int64_t t_end, double* state, double* rows, double* moments,
                                    int64_t moments_after, int32_t quantum, int32_t* queue, void* stream) {
  if (quantum <= 0 || !queue) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "queued advance needs a positive quantum and a queue workspace");
  return advance_impl(pts, prob, cfg, t_begin, t_end, state, rows, moments, moments_after, quantum, queue, stream);
}

int phf_single_level_queue_status(const int32_t* queue, int64_t num_blocks, void* stream) {
  if (!queue || num_blocks <= 0) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad arguments to phf_single_level_queue_status");
  int32_t fault = 0;
  if (hipMemcpyAsync(&fault, queue + 1 + num_blocks, sizeof(fault), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return phf_check_launch("phf_single_level_queue_status");
  if (fault != 0)
    return phf_fail(PHF_ERR_DRAINED, "a queued single-level launch drained (a wavefront's wait for its block's previous quantum did not end): "
                                     "states, rows and moments written since are stale");
  return PHF_OK;
}

int phf_single_level_log_target(const phf_points* pts, int model, int64_t m, const int32_t* pair_index,
                                const double* temperature, const double* theta, double* out_lik,
                                double* out_prior, void* stream) {
  if (!pts || (model != 1 && model != 2) || m < 0 || !pair_index || !temperature || !theta)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad arguments to phf_single_level_log_target");
  if (m == 0) return PHF_OK;
  const dim3 grid((unsigned)((m + 255) / 256)), block(256);
  if (model == 1) hipLaunchKernelGGL(log_target_kernel<1>, grid, block, 0, (hipStream_t)stream, *pts, m, pair_index, temperature, theta, out_lik, out_prior);
  else hipLaunchKernelGGL(log_target_kernel<2>, grid, block, 0, (hipStream_t)stream, *pts, m, pair_index, temperature, theta, out_lik, out_prior);
  return phf_check_launch("phf_single_level_log_target");
}

int phf_debug_math(int fn, int64_t n, const double* in, double* out, void* stream) {
  if (fn < 0 || fn > 21 || n < 0 || !in || !out) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad arguments to phf_debug_math");
  if (n == 0) return PHF_OK;
  hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fn, n, in, out);
  return phf_check_launch("phf_debug_math");
}

int phf_philox_rounds(void) { return PHF_PHILOX_ROUNDS; }

int phf_debug_philox_rounds(int rounds, int64_t n, const uint32_t* counter_key, uint32_t* out, void* stream) {
  if ((rounds != 7 && rounds != 10) || n < 0 || !counter_key || !out) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad arguments to phf_debug_philox_rounds (rounds: 7 or 10)");
  if (n == 0) return PHF_OK;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (rounds == 7) hipLaunchKernelGGL(debug_philox_kernel<7>, grid, block, 0, (hipStream_t)stream, n, counter_key, out);
  else hipLaunchKernelGGL(debug_philox_kernel<10>, grid, block, 0, (hipStream_t)stream, n, counter_key, out);
  return phf_check_launch("phf_debug_philox_rounds");
}

int phf_debug_philox(int64_t n, const uint32_t* counter_key, uint32_t* out, void* stream) {
  return phf_debug_philox_rounds(PHF_PHILOX_ROUNDS, n, counter_key, out, stream);
}

}  // extern "C"
