// phf_hierarchical.hip — gfx950 kernels for PyHillFit's hierarchical sampler (python/PyHillFit.py:113-193,429-511).
//
// One lane = one Markov chain of dimension dim = 5 + 2 Ne (11..17 for the Crumb set); a wavefront advances 64
// chains of the same (drug, channel) pair in lock-step.  The kernel is compiled per Ne so that theta, mean and the
// proposal live in registers with static indices.  What does not fit in registers is the adapted covariance: its factors
// cov = L diag(d) L' (PHF_LDL_COLUMN below; dim(dim+1)/2 = 66..153 doubles per chain) live in LDS as
// L[element][lane] (stride 64 doubles: every ds_read/ds_write is conflict-free, 28..105 KB per wavefront; the one-lane kernels keep the diagonal in registers) for the
// whole launch, so that its two passes per iteration (y = L z, and the rank-one adaptation update) cost no HBM
// traffic at all.  HBM sees the state once per launch and the thinned samples.
#include <hip/hip_runtime.h>

#include <cstring>

#include <atomic>
#include <cstdlib>

#define PHF_FMA_K_AS_BUILTIN 1   // one wavefront per SIMD: every s_nop is a lost issue slot (phf_math.h)
#include "../../include/pyhillfit_amd.h"
#include "phf_common.h"
#include "phf_hier3_isa.h"
#include "phf_hier_model.h"

namespace {

constexpr int kBlock = 64;

// *p += v without reading it back: global_atomic_add_f64, no return value, nothing to wait for
__device__ __forceinline__ void phf_accumulate(double* p, double v) { (void)unsafeAtomicAdd(p, v); }

// The adapted covariance is carried as cov = L diag(d) L' — L unit lower triangular, stored as the packed lower triangle with d_k in
// the diagonal slot — and the reference's recursion cov <- (1-g) cov + g v v' (python/PyHillFit.py:498-499) is applied to the
// factors: D scales by (1-g), then the rank-one update of Gill, Golub, Murray & Saunders (1974, method C1), column by column:
//     p = w_k;  dk = (1-g) d_k;  dn = dk + alpha p^2;  beta = alpha p / dn;  alpha <- alpha dk / dn   (alpha starts at g, w at v)
//     for the rows i > k:   w_i <- w_i - p L_ik;   L_ik <- L_ik + beta w_i
// Two fused multiply-adds per element and ONE short dependent chain per column (multiply, fma, reciprocal, multiply) — against
// five operations per element and a square root AND a reciprocal in every column's chain for the Givens sweep of a Cholesky
// factor that rounds 1 and 2 carried: with that sweep cut out the one-lane Ne = 3 kernel ran 23 % faster although the sweep was
// 13.5 % of its instructions (a lone wavefront exposes every dependent latency).  No square root is needed to UPDATE; the proposal
// takes sqrt(d_k) of the eleven diagonals at the start of an iteration, where they overlap the draws.  dn = 0 (a direction that
// has no variance and gets none: d_k = 0 and p = 0 — a start point with a component exactly 0 freezes that component in the
// reference too, PyHillFit.py:431,498-499) leaves the column and alpha as they are.
#define PHF_LDL_COLUMN(omg_, alpha_, p_, d_, dn_, beta_)                  \
  do {                                                                   \
    const double dk_ = (omg_) * (d_);                                    \
    const double ap_ = (alpha_) * (p_);                                  \
    (dn_) = phf_fma(ap_, (p_), dk_);                                     \
    const double inv_ = ((dn_) > 0.0) ? phf_rcp(dn_) : 0.0;              \
    (beta_) = ap_ * inv_;                                                \
    (alpha_) = ((dn_) > 0.0) ? ((alpha_) * dk_) * inv_ : (alpha_);       \
  } while (0)

struct HierArgs {
  phf_hier_points pts;
  phf_problems prob;
  phf_hier_prior prior;
  phf_mh_config cfg;
  int64_t t_begin, t_end;
  double* state;
  double* rows;
  double* moments;
  int64_t moments_after;
  int32_t blocks_per_problem;
  int32_t* queue;          // phf_hierarchical_advance_queued: the work-queue workspace (gfx950 assembly build only) and the quantum asked for
  int32_t quantum;
  // init only
  double cov_scale;
  const double* theta0;
  double* row0;
};

// One lane per chain: what of the strictly lower triangle of L does not stay in registers lives in LDS ([slot][64] doubles: 0 KB at
// Ne = 3, 11.5 KB at Ne = 4, 28..76 KB at Ne = 5..8; at most 4 wavefronts per CU, one per SIMD), the diagonal d in registers,
// and all 512 registers for the iteration.  (LDS is what the four groups of C4 compete for — a CU holds its four wavefronts only
// if their footprints add up to 160 KB: +2.5 KB per wavefront cost 5 % of C4 on one box — so the diagonal, 11..21 doubles that the
// exp / log tables' shorter coefficient sets pay for, left LDS (C4 18.4 -> 17.5 ms), then rows 1..3 (16.9 -> 16.4), then — with
// the 3 KB table of the inverse-CDF normals coming in and the sin/cos coefficients going out — rows 4..6 (16.5 -> 16.0), then
// rows 7..10 for Ne <= 4 (15.8 -> 15.2).)
// rows 1..reg_rows of L (reg_rows (reg_rows + 1) / 2 elements) stay in registers next to the diagonal: all of them for Ne = 3 (55
// elements; hipcc parks them in the accumulation registers), ten rows of twelve for Ne = 4, six beyond (Ne = 5 with ten rows: +32 %,
// it spills) — measured on C4 and per group, profiles/r03/math_tables_ab.txt items 8-10
#ifndef PHF_HIER_REG_ROWS_SMALL
#define PHF_HIER_REG_ROWS_SMALL 10
#endif
#ifndef PHF_HIER_REG_ROWS_BIG
#define PHF_HIER_REG_ROWS_BIG 6
#endif
template <int NE>
struct Lds {
  static constexpr int dim = 5 + 2 * NE;
  static constexpr int want_rows = (NE <= 4) ? PHF_HIER_REG_ROWS_SMALL : PHF_HIER_REG_ROWS_BIG;
  static constexpr int reg_rows = want_rows < dim - 1 ? want_rows : dim - 1;
  static constexpr int reg_elems = reg_rows * (reg_rows + 1) / 2;
  static constexpr int slots = dim * (dim - 1) / 2 - reg_elems;
  static size_t point_bytes(int stride) { return (size_t)stride * 16 + (NE + 1) * 4 + 8; }
  static size_t bytes(int stride) { return (size_t)slots * kBlock * 8 + point_bytes(stride); }
};

template <int NE>
__device__ __forceinline__ void stage(const phf_hier_points& pts, int pair, double* s_lc, double* s_y, int* s_es) {
  const int n = pts.expt_start[(size_t)pair * (NE + 1) + NE];
  for (int j = threadIdx.x; j < n; j += kBlock) {
    s_lc[j] = pts.ln_conc[(size_t)pair * pts.stride + j];
    s_y[j] = pts.response[(size_t)pair * pts.stride + j];
  }
  if (threadIdx.x <= NE) s_es[threadIdx.x] = pts.expt_start[(size_t)pair * (NE + 1) + threadIdx.x];
  __syncthreads();
}

// The whole launch of one wavefront.  FIXED_N > 0: every experiment of the pair has exactly FIXED_N points, known at compile
// time (the point loops unroll: straight-line iteration); 0: run-time experiment boundaries.
template <int NE, int FIXED_N>
__device__ __forceinline__ void hier_advance_body(const HierArgs& a, double* s_mem, const double* s_lc, const double* s_y,
                                                  const int* s_es, int q, int c) {
  constexpr int D = 5 + 2 * NE;
  constexpr int TRI = D * (D + 1) / 2;
  double* sL = s_mem + threadIdx.x;                        // element (i, k < i) of L: sL[(i (i - 1) / 2 + k) * 64]; the diagonal d: dg[]
  constexpr int kRegRows = Lds<NE>::reg_rows, kRegElems = Lds<NE>::reg_elems;
  double Lreg[kRegElems > 0 ? kRegElems : 1];              // rows 1..kRegRows
#define PHF_LIDX(i, k) ((i) * ((i) - 1) / 2 + (k))
#define PHF_LGET(i, k) (((i) <= kRegRows) ? Lreg[((i) <= kRegRows) ? PHF_LIDX(i, k) : 0] : sL[(PHF_LIDX(i, k) - kRegElems) * kBlock])
#define PHF_LSET(i, k, v) do { if ((i) <= kRegRows) Lreg[((i) <= kRegRows) ? PHF_LIDX(i, k) : 0] = (v); else sL[(PHF_LIDX(i, k) - kRegElems) * kBlock] = (v); } while (0)
  const int C = a.prob.chains_per_problem;
  const uint32_t pid = a.prob.problem_id[q];
  const uint32_t cid = a.prob.chain_id_base + (a.prob.chain_offset ? a.prob.chain_offset[q] : 0u) + (uint32_t)c;
  const uint32_t seed_lo = (uint32_t)a.cfg.seed, seed_hi = (uint32_t)(a.cfg.seed >> 32);
  const size_t nchains = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;

  double th[D], mean[D], dg[D];
  // element k of this lane's state: a wave-uniform base (scalar registers, recomputed by the scalar unit) + the lane number, so that
  // the ~60 distinct addresses of an iteration cost no vector registers (as per-lane 64-bit pointers they were hipcc's first spill victims)
  double* const sbase = a.state + ((size_t)q * C + (size_t)(c - (int)threadIdx.x));
  const uint32_t lane = threadIdx.x;
#define PHF_SP(k) (sbase + (size_t)(k) * nchains)[lane]
#pragma unroll
  for (int i = 0; i < D; ++i) th[i] = PHF_SP(i);
  double lt = PHF_SP(D);
#pragma unroll
  for (int i = 0; i < D; ++i) mean[i] = PHF_SP(D + 1 + i);
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) PHF_LSET(i, k, PHF_SP(2 * D + 1 + i * (i + 1) / 2 + k));
    dg[i] = PHF_SP(2 * D + 1 + i * (i + 1) / 2 + i);
  }
  double loga = PHF_SP(2 * D + 1 + TRI);
  double nacc = PHF_SP(2 * D + 2 + TRI);
  // coefficient tables in VGPRs for the launch: a lone wavefront cannot hide the scalar-load latency
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  double sc = phf_exp_fast_k(0.5 * loga, k_exp);

  const bool want_moments = a.moments != nullptr;
  const int thin = a.cfg.thinning;
  int until_save = thin - (int)(a.t_begin % thin);
  double* out = a.rows ? a.rows + ((size_t)q * (D + 1)) * C + c : nullptr;
  const size_t row_stride = (size_t)a.prob.num_problems * (D + 1) * C;

  // y = L sqrt(D) z of the NEXT proposal, element by element:  y_i = (sum over k < i, ascending, of L_ik u_k, from +0) + u_i  with
  // u_k = sqrt(d_k) z_k.  Read from LDS here (the first iteration of a launch, and every iteration before the adaptation
  // starts); once adapting, the update sweep below produces it from the elements it has in registers anyway.
#define PHF_PLAIN_LU(zz, yy)                                                                              \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < D; ++i) (zz)[i] = phf_sqrt_nonneg(dg[i]) * (zz)[i];             \
    _Pragma("unroll") for (int i = 0; i < D; ++i) {                                                       \
      double v_ = 0.0;                                                                                    \
      _Pragma("unroll") for (int k = 0; k < i; ++k) v_ = phf_fma(PHF_LGET(i, k), (zz)[k], v_);            \
      (yy)[i] = v_ + (zz)[i];                                                                             \
    }                                                                                                     \
  } while (0)
  double y[D];
  double log_u;
  {
    double z0[D];
    log_u = phf_hier_draws_k(D, cid, pid, (uint32_t)(a.t_begin + 1), seed_lo, seed_hi, k_log, z0, 1);
    PHF_PLAIN_LU(z0, y);
  }
  for (int64_t t = a.t_begin + 1; t <= a.t_end; ++t) {
    // ---- proposal theta* = theta + e^(loga/2) L sqrt(D) z   (PyHillFit.py:485) ----
    double star[D];
#pragma unroll
    for (int i = 0; i < D; ++i) star[i] = phf_fma(sc, y[i], th[i]);
    // ---- target, accept (:486-492) ----
    const double lt_star = phf_hier_log_target_n(NE, FIXED_N, s_es, s_lc, s_y, star, 1, &a.prior, k_exp, k_log);
    const bool acc = log_u < lt_star - lt;
    if (acc) {
#pragma unroll
      for (int i = 0; i < D; ++i) th[i] = star[i];
      lt = lt_star;
    }
    nacc += acc ? 1.0 : 0.0;
    // ---- the factors for the sweep, read from LDS HERE — ahead of the draws, whose ~700 instructions cover the LDS latency a lone
    // wavefront would otherwise wait out element by element inside the sweep (same box: Ne = 3 group 13.80 -> 13.62 ms, Ne = 4 6.37 ->
    // 6.17, C4 19.11 -> 18.86) — and the draws of iteration t + 1 (a function of (chain, t + 1) alone) ----
    double Lr[TRI - D];
#pragma unroll
    for (int i = 1; i < D; ++i)
#pragma unroll
      for (int k = 0; k < i; ++k) Lr[i * (i - 1) / 2 + k] = PHF_LGET(i, k);
    double zn[D];
    log_u = phf_hier_draws_k(D, cid, pid, (uint32_t)(t + 1), seed_lo, seed_hi, k_log, zn, 1);
    // ---- adaptation (:495-501): cov <- (1-g) cov + g v v' applied to the L D L' factors as a rank-one update (PHF_LDL_COLUMN),
    // and the next proposal's y = L' sqrt(D') z' accumulated from the new elements while they are in registers.  Before the
    // adaptation starts the sweep runs with g = 0 — an exact no-op on mean, loga, d and L (x + 0 y) that still delivers y — so
    // that the iteration is one straight-line body ----
    if (t > a.cfg.adapt_start) {
      const double gs = a.cfg.gamma[t - a.cfg.adapt_start];
      const double omg = 1.0 - gs;
      double w[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        w[i] = th[i] - mean[i];
        mean[i] = phf_fma(gs, th[i], omg * mean[i]);
        y[i] = 0.0;
      }
      loga = phf_fma(gs, (acc ? 1.0 : 0.0) - 0.25, loga);
      double alpha = gs;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        double dn, beta;
        PHF_LDL_COLUMN(omg, alpha, w[k], dg[k], dn, beta);
        dg[k] = dn;
        zn[k] = phf_sqrt_nonneg(dn) * zn[k];               // u_k of the next proposal
#pragma unroll
        for (int i = k + 1; i < D; ++i) {
          const double lik = Lr[i * (i - 1) / 2 + k];
          w[i] = phf_fma(-w[k], lik, w[i]);
          const double nl = phf_fma(beta, w[i], lik);
          PHF_LSET(i, k, nl);
          y[i] = phf_fma(nl, zn[k], y[i]);
        }
        y[k] = y[k] + zn[k];
      }
      sc = phf_exp_fast_k(0.5 * loga, k_exp);
    } else {
      PHF_PLAIN_LU(zn, y);
    }
    // ---- thinning + sample store (:502-503) ----
    if (--until_save == 0) {
      until_save = thin;
      if (out) {
#pragma unroll
        for (int i = 0; i < D; ++i) out[(size_t)i * C] = th[i];
        out[(size_t)D * C] = lt;
        out += row_stride;
      }
      if (want_moments && t > a.moments_after) {
        // registers and LDS are full (512 registers + AGPR spills, 28..105 KB of factor): the running sums live in HBM and are
        // advanced by fire-and-forget fp64 atomics — one lane per address, so the sums are the sequential ones; a
        // load-add-store here made the lone wavefront wait for 24 loads per saved sample (C4 +5.6 %)
#pragma unroll
        for (int i = 0; i < D; ++i) {
          phf_accumulate(&a.moments[(size_t)i * nchains + g], th[i]);
          phf_accumulate(&a.moments[(size_t)(D + 1 + i) * nchains + g], th[i] * th[i]);
        }
        phf_accumulate(&a.moments[(size_t)D * nchains + g], lt);
        phf_accumulate(&a.moments[(size_t)(2 * D + 1) * nchains + g], lt * lt);
      }
    }
  }

#pragma unroll
  for (int i = 0; i < D; ++i) PHF_SP(i) = th[i];
  PHF_SP(D) = lt;
#pragma unroll
  for (int i = 0; i < D; ++i) PHF_SP(D + 1 + i) = mean[i];
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) PHF_SP(2 * D + 1 + i * (i + 1) / 2 + k) = PHF_LGET(i, k);
    PHF_SP(2 * D + 1 + i * (i + 1) / 2 + i) = dg[i];
  }
  PHF_SP(2 * D + 1 + TRI) = loga;
  PHF_SP(2 * D + 2 + TRI) = nacc;
#undef PHF_PLAIN_LU
#undef PHF_LGET
#undef PHF_LSET
#undef PHF_LIDX
#undef PHF_SP
}

template <int NE>
__global__ __launch_bounds__(kBlock, 1) void hier_advance_kernel(const HierArgs a) {
  extern __shared__ double s_mem[];
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  double* s_lc = s_mem + (size_t)Lds<NE>::slots * kBlock;
  double* s_y = s_lc + a.pts.stride;
  int* s_es = reinterpret_cast<int*>(s_y + a.pts.stride);
  const int slot = blockIdx.x / a.blocks_per_problem;
  const int chunk = blockIdx.x - slot * a.blocks_per_problem;
  const int q = a.prob.launch_order ? a.prob.launch_order[slot] : slot;     // include/pyhillfit_amd.h: phf_problems.launch_order
  const int c = chunk * kBlock + threadIdx.x;
  const int pair = a.prob.pair_index[q];
  stage<NE>(a.pts, pair, s_lc, s_y, s_es);
  if (c >= a.prob.chains_per_problem) return;
  // 4 points in every experiment (147 of the 210 Crumb pairs: 3 experiments x 4 doses)?  scalar loads: a uniform branch
  bool four_each = true, four_then_one = NE == 4;          // ... or 4 + 4 + 4 + 1 (32 of the 41 pairs with four experiments)
#pragma unroll
  for (int i = 1; i <= NE; ++i) {
    const int e = a.pts.expt_start[(size_t)pair * (NE + 1) + i];
    four_each = four_each && (e == 4 * i);
    four_then_one = four_then_one && (e == (i < NE ? 4 * i : 4 * (NE - 1) + 1));
  }
  if (four_each) { hier_advance_body<NE, PHF_HIER_SHAPE(4, 0)>(a, s_mem, s_lc, s_y, s_es, q, c); return; }
  if constexpr (NE == 4) {
    if (four_then_one) { hier_advance_body<NE, PHF_HIER_SHAPE(4, 1)>(a, s_mem, s_lc, s_y, s_es, q, c); return; }
  }
  hier_advance_body<NE, 0>(a, s_mem, s_lc, s_y, s_es, q, c);
}

// ---------------------------------------------------------------------------------------------------------------
// TWO LANES PER CHAIN.  One lane per chain keeps 2 dim + dim(dim+1)/2 doubles of state per lane (88 at Ne = 3) beside a target
// that wants every register: one wavefront per SIMD, which then issues one instruction per ~3.3 ns whatever its kind.  Here
// lanes 2c and 2c+1 of a wavefront share chain c (32 chains per wavefront) and split
//   the state   lane h owns rows 2a+h (a = 0..A-1, A = ceil(dim/2)) of theta, mean, the update vector and the factor: half the
//               registers, half the LDS (A^2 slots of 64 doubles: 18 KB per wavefront at Ne = 3 — eight wavefronts per CU);
//   the draws   Philox block b is evaluated by lane b mod 2 (normals 4b..4b+3, or the accept uniform), broadcast by DPP;
//   the target  lane h computes half h of the log target (phf_hier_model.h), one cross-lane addition joins the halves;
//   the factor  y = L sqrt(d) z and the rank-one update row by row: a column's coefficients (dn, beta, alpha: one reciprocal) are
//               computed by both lanes from the owner's p and d, every lane then updates its own rows.
// Every number is produced by the same operations in the same order as in the one-lane kernel and the twin (rows of a column
// are independent; the halves of the target are defined per half): chains, states and moments are bit-identical.
//
// Factor element (row 2a+h, column k <= 2a) of lane h sits in slot a^2 + k of the lane's LDS column ([slot][64 lanes] doubles):
// both lanes address "(a, k)" with the SAME immediate offset from the same base register, so one instruction stream serves both
// and no per-lane address is kept.  What exists on one lane only runs under a lane-parity mask: lane 0's last row (dim is odd, so
// lane 1 has no row in pair-row A-1), and lane 1's diagonal elements (a, 2a+1) — stored in the slots (A-1)^2 + a that lane 1's
// missing last row leaves free in its column.
template <int NE>
struct Lds2 {
  static constexpr int dim = 5 + 2 * NE;
  static constexpr int A = (dim + 1) / 2;
  static constexpr int slots = A * A;                      // pair-row a: 2a+1 slots; lane 1's diagonals in its unused last pair-row
  // constants read through LDS (wave-uniform addresses: broadcast reads on the LDS pipe) instead of occupying scalar or vector
  // registers: the 15 prior parameters
  static constexpr int consts = 15;
  static size_t point_bytes(int stride) { return (size_t)stride * 16 + (NE + 1) * 4 + 8; }
  static size_t bytes(int stride) { return (size_t)(slots * kBlock + consts) * 8 + point_bytes(stride); }
};

__device__ __forceinline__ double phf_dpp_quad(double v, const int ctrl_tag) {
  // quad_perm within groups of four lanes, applied to the two halves of a double
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (ctrl_tag == 0) { lo = __builtin_amdgcn_mov_dpp(lo, 0xA0, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xA0, 0xf, 0xf, true); }        // [0,0,2,2]: from the even lane
  else if (ctrl_tag == 1) { lo = __builtin_amdgcn_mov_dpp(lo, 0xF5, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xF5, 0xf, 0xf, true); }   // [1,1,3,3]: from the odd lane
  else { lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true); }                       // [1,0,3,2]: the partner's
  return __hiloint2double(hi, lo);
}
#define PHF_FROM_LANE(hsrc, v) phf_dpp_quad((v), (hsrc))
#define PHF_FROM_PARTNER(v) phf_dpp_quad((v), 2)

// WPS = wavefronts per SIMD the body is compiled for.  2: 256 registers — the prior
// parameters are read through LDS (s_k) where they are used; 1: 512 registers — every table resident in registers, the prior in
// scalar registers (launches whose wavefronts have a SIMD each: nothing would hide an LDS or scalar-cache latency).
template <int NE, int FIXED_N, int WPS>
__device__ __forceinline__ void hier_advance2_body(const HierArgs& a, double* s_mem, const double* s_k, const double* s_lc,
                                                   const double* s_y, const int* s_es, int q, int c0) {
  constexpr int D = 5 + 2 * NE;
  constexpr int TRI = D * (D + 1) / 2;
  constexpr int A = (D + 1) / 2;
  constexpr bool ODD = (D & 1) != 0;                       // lane 1 has no row in the last pair-row (its row index would be D)
  static_assert(ODD, "dim = 5 + 2 Ne is odd: lane 1's diagonal elements live in the slots of the last pair-row it does not have");
  const uint32_t lane = threadIdx.x;
  const int h = (int)(lane & 1u);
  const int cl = (int)(lane >> 1);                         // chain within the wavefront
  const int c = c0 + cl;
  double* const sLane = s_mem + lane;
#define PHF_L2(a_, k_) sLane[((a_) * (a_) + (k_)) * kBlock]           /* k_ <= 2 a_ */
#define PHF_L2D(a_) sLane[((A - 1) * (A - 1) + (a_)) * kBlock]        /* lane 1 only: its diagonal element (a_, 2 a_ + 1) */
  // does this lane have a row in pair-row a_ ?  (a literal a_: folds to true except for lane 1 in the last pair-row of an odd dim)
#define PHF_HAS_ROW(a_) (!(ODD && (a_) == A - 1) || h == 0)
  const int C = a.prob.chains_per_problem;
  const uint32_t pid = a.prob.problem_id[q];
  const uint32_t cid = a.prob.chain_id_base + (a.prob.chain_offset ? a.prob.chain_offset[q] : 0u) + (uint32_t)c;
  const uint32_t seed_lo = (uint32_t)a.cfg.seed, seed_hi = (uint32_t)(a.cfg.seed >> 32);
  const size_t nchains = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;
  double* const sbase = a.state + ((size_t)q * C + (size_t)c0);      // wave-uniform; element k of chain cl: (sbase + k nchains)[cl]
#define PHF_SP(k) (sbase + (size_t)(k) * nchains)[cl]

  double th[A], mean[A];                                   // rows 2a+h; the missing row of lane 1 is carried as zeros (and stays zero)
#pragma unroll
  for (int a_ = 0; a_ < A; ++a_) {
    th[a_] = 0.0; mean[a_] = 0.0;
    if (PHF_HAS_ROW(a_)) {
      const int i = 2 * a_ + h;
      th[a_] = PHF_SP(i);
      mean[a_] = PHF_SP(D + 1 + i);
#pragma unroll
      for (int k = 0; k <= 2 * a_; ++k) PHF_L2(a_, k) = PHF_SP(2 * D + 1 + i * (i + 1) / 2 + k);
      if (h) PHF_L2D(a_) = PHF_SP(2 * D + 1 + i * (i + 1) / 2 + i);
    }
  }
  double lt = PHF_SP(D);
  double loga = PHF_SP(2 * D + 1 + TRI);
  double nacc = PHF_SP(2 * D + 2 + TRI);
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);                      // used by every polynomial of the iteration: registers
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  const phf_hier_prior* const prior = (WPS == 1) ? &a.prior : reinterpret_cast<const phf_hier_prior*>(s_k);
  double sc = phf_exp_fast_k(0.5 * loga, k_exp);

  const bool want_moments = a.moments != nullptr;
  const int thin = a.cfg.thinning;
  int until_save = thin - (int)(a.t_begin % thin);
  // rows 2a+h of this lane: ONE per-lane pointer (row h of the chain) + wave-uniform offsets 2a C
  double* out = a.rows ? a.rows + ((size_t)q * (D + 1) + h) * C + c : nullptr;
  const size_t row_stride = (size_t)a.prob.num_problems * (D + 1) * C;
  double* const mom = want_moments ? a.moments + (size_t)h * nchains + g : nullptr;

  for (int64_t t = a.t_begin + 1; t <= a.t_end; ++t) {
    // ---- draws (phf_hier_draws): block b by lane b mod 2 — word j of block b is normal 4b + j, the last word of block NB - 1 the accept uniform ----
    double z[D];
    double log_u = 0.0;
    {
      constexpr int NB = (D + 3) / 4;
      constexpr int J = (NB + 1) / 2;                      // evaluations per lane: blocks 2j + h, j < J, cover 0..NB-1
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int b0 = 2 * j, b1 = 2 * j + 1;              // lane 0's block, lane 1's block
        const phf_u32x4 w = phf_philox_mh(cid, pid, (uint32_t)t, (uint32_t)(2 * j + h), seed_lo, seed_hi);
        const bool need1 = (4 * b0 + 1 < D) || (b1 < NB && 4 * b1 + 1 < D);
        const bool need2 = (4 * b0 + 2 < D) || (b1 < NB && 4 * b1 + 2 < D);
        const bool need3 = (4 * b0 + 3 < D) || (b1 < NB && 4 * b1 + 3 < D);
        const double n0 = phf_normal_u32(w.w[0]);
        const double n1 = need1 ? phf_normal_u32(w.w[1]) : 0.0, n2 = need2 ? phf_normal_u32(w.w[2]) : 0.0, n3 = need3 ? phf_normal_u32(w.w[3]) : 0.0;
        z[4 * b0] = PHF_FROM_LANE(0, n0);
        if (4 * b0 + 1 < D) z[4 * b0 + 1] = PHF_FROM_LANE(0, n1);
        if (4 * b0 + 2 < D) z[4 * b0 + 2] = PHF_FROM_LANE(0, n2);
        if (4 * b0 + 3 < D) z[4 * b0 + 3] = PHF_FROM_LANE(0, n3);
        if (b1 < NB) {
          z[4 * b1] = PHF_FROM_LANE(1, n0);
          if (4 * b1 + 1 < D) z[4 * b1 + 1] = PHF_FROM_LANE(1, n1);
          if (4 * b1 + 2 < D) z[4 * b1 + 2] = PHF_FROM_LANE(1, n2);
          if (4 * b1 + 3 < D) z[4 * b1 + 3] = PHF_FROM_LANE(1, n3);
        }
        if (b0 == NB - 1 || b1 == NB - 1) {
          const double v = phf_log_pos_k(phf_unit_open32(w.w[3]), k_log);
          log_u = PHF_FROM_LANE((b0 == NB - 1) ? 0 : 1, v);
        }
      }
    }
    // ---- proposal theta* = theta + e^(loga/2) L sqrt(D) z, own rows (PyHillFit.py:485); then every lane gets the whole vector ----
    double star[D];
    {
      // u = sqrt(d) z: each lane takes the square root of ITS rows' diagonals (one instruction stream: A square roots, not D)
      double u_o[A], u[D];
#pragma unroll
      for (int a_ = 0; a_ < A; ++a_) {
        u_o[a_] = 0.0;
        if (PHF_HAS_ROW(a_)) {
          const double d_e = PHF_L2(a_, 2 * a_);                                 // lane 0: its diagonal d; lane 1: an off-diagonal
          double d_own = d_e;
          if (2 * a_ + 1 < D) { const double d_o = PHF_L2D(a_); d_own = h ? d_o : d_e; }
          const double z_own = (2 * a_ + 1 < D) ? (h ? z[2 * a_ + 1] : z[2 * a_]) : z[2 * a_];
          u_o[a_] = phf_sqrt_nonneg(d_own) * z_own;
        }
      }
#pragma unroll
      for (int a_ = 0; a_ < A; ++a_) {
        u[2 * a_] = PHF_FROM_LANE(0, u_o[a_]);
        if (2 * a_ + 1 < D) u[2 * a_ + 1] = PHF_FROM_LANE(1, u_o[a_]);
      }
      double star_o[A];
#pragma unroll
      for (int a_ = 0; a_ < A; ++a_) {
        star_o[a_] = 0.0;
        if (PHF_HAS_ROW(a_)) {
          // row i = 2a+h: v = (sum over k < i, ascending, of L_ik u_k, from +0) + u_i.  Slot (a, 2a) is lane 1's element (2a+1, 2a)
          // but lane 0's DIAGONAL: lane 0 multiplies u_2a by zero there (fma(0, u, v) == v; a zero's sign cannot reach theta*)
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < 2 * a_; ++k) v = phf_fma(PHF_L2(a_, k), u[k], v);
          {
            const double e = PHF_L2(a_, 2 * a_);
            v = phf_fma(h ? e : 0.0, u[2 * a_], v);
          }
          v = v + u_o[a_];
          star_o[a_] = phf_fma(sc, v, th[a_]);
        }
      }
#pragma unroll
      for (int a_ = 0; a_ < A; ++a_) {
        star[2 * a_] = PHF_FROM_LANE(0, star_o[a_]);
        if (2 * a_ + 1 < D) star[2 * a_ + 1] = PHF_FROM_LANE(1, star_o[a_]);
      }
    }
    // ---- target: this lane's half + the partner's (:486), accept (:487-492) ----
    const int bad = phf_hier_out_of_support(NE, star, 1, prior);
    const double half = phf_hier_target_half(h, NE, FIXED_N, s_es, s_lc, s_y, star, 1, prior, k_exp, k_log, 0);
    const double lt_star = bad ? -PHF_INF : half + PHF_FROM_PARTNER(half);
    const bool acc = log_u < lt_star - lt;
    if (acc) {                                             // own rows out of the full vector (not kept apart across the target: registers)
#pragma unroll
      for (int a_ = 0; a_ < A; ++a_) th[a_] = (2 * a_ + 1 < D) ? (h ? star[2 * a_ + 1] : star[2 * a_]) : (h ? 0.0 : star[2 * a_]);
      lt = lt_star;
    }
    nacc += acc ? 1.0 : 0.0;
    // ---- adaptation (:495-501): rank-one update of L D L' (PHF_LDL_COLUMN); a column's (dn, beta, alpha) by both lanes from the
    // owner's p and d, every lane then updates its own rows ----
    if (t > a.cfg.adapt_start) {
      const double gs = a.cfg.gamma[t - a.cfg.adapt_start];
      const double omg = 1.0 - gs;
      double w[A];
#pragma unroll
      for (int a_ = 0; a_ < A; ++a_) {
        w[a_] = th[a_] - mean[a_];
        mean[a_] = phf_fma(gs, th[a_], omg * mean[a_]);
      }
      loga = phf_fma(gs, (acc ? 1.0 : 0.0) - 0.25, loga);
      double alpha = gs;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int m = k >> 1;
        const bool even = (k & 1) == 0;                    // owner of the diagonal: lane 0 (even k: row 2m) or lane 1 (row 2m+1)
        const bool below = even && (k + 1 < D);            // even k: lane 1's element of pair-row m is L[k+1][k], a row to update
        double e_own = 0.0;                                // slot (m, k): lane 0's d_k (even k) / lane 1's L[k+1][k]; or lane 1's d_k (odd k)
        if (even) { if (below || h == 0) e_own = PHF_L2(m, k); }
        else { if (h) e_own = PHF_L2D(m); }
        const double dcur = PHF_FROM_LANE(even ? 0 : 1, e_own);
        const double pk = PHF_FROM_LANE(even ? 0 : 1, w[m]);
        double dn, beta;
        PHF_LDL_COLUMN(omg, alpha, pk, dcur, dn, beta);
        if (below) {
          const double nw = phf_fma(-pk, e_own, w[m]);
          const double nl = phf_fma(beta, nw, e_own);
          PHF_L2(m, k) = h ? nl : dn;
          w[m] = nw;                                       // lane 0's w[m] is not read again
        } else if (even) { if (h == 0) PHF_L2(m, k) = dn; }
        else { if (h) PHF_L2D(m) = dn; }
#pragma unroll
        for (int a_ = m + 1; a_ < A; ++a_) {
          if (PHF_HAS_ROW(a_)) {
            const double lik = PHF_L2(a_, k);
            w[a_] = phf_fma(-pk, lik, w[a_]);
            PHF_L2(a_, k) = phf_fma(beta, w[a_], lik);
          }
        }
      }
      sc = phf_exp_fast_k(0.5 * loga, k_exp);
    }
    // ---- thinning + sample store (:502-503): own rows; the log target by the even lane ----
    if (--until_save == 0) {
      until_save = thin;
      if (out) {
#pragma unroll
        for (int a_ = 0; a_ < A; ++a_)
          if (PHF_HAS_ROW(a_)) out[(size_t)(2 * a_) * C] = th[a_];
        if (h == 0) out[(size_t)D * C] = lt;
        out += row_stride;
      }
      if (want_moments && t > a.moments_after) {
#pragma unroll
        for (int a_ = 0; a_ < A; ++a_)
          if (PHF_HAS_ROW(a_)) {
            phf_accumulate(&mom[(size_t)(2 * a_) * nchains], th[a_]);
            phf_accumulate(&mom[(size_t)(D + 1 + 2 * a_) * nchains], th[a_] * th[a_]);
          }
        if (h == 0) {
          phf_accumulate(&mom[(size_t)D * nchains], lt);
          phf_accumulate(&mom[(size_t)(2 * D + 1) * nchains], lt * lt);
        }
      }
    }
  }

#pragma unroll
  for (int a_ = 0; a_ < A; ++a_) {
    if (PHF_HAS_ROW(a_)) {
      const int i = 2 * a_ + h;
      PHF_SP(i) = th[a_];
      PHF_SP(D + 1 + i) = mean[a_];
#pragma unroll
      for (int k = 0; k <= 2 * a_; ++k) PHF_SP(2 * D + 1 + i * (i + 1) / 2 + k) = PHF_L2(a_, k);
      if (h) PHF_SP(2 * D + 1 + i * (i + 1) / 2 + i) = PHF_L2D(a_);
    }
  }
  if (h == 0) {
    PHF_SP(D) = lt;
    PHF_SP(2 * D + 1 + TRI) = loga;
    PHF_SP(2 * D + 2 + TRI) = nacc;
  }
#undef PHF_L2
#undef PHF_L2D
#undef PHF_HAS_ROW
#undef PHF_SP
}

constexpr int kChains2 = kBlock / 2;                       // chains per wavefront of the two-lane kernel

template <int NE, int WPS>
__global__ __launch_bounds__(kBlock, WPS) void hier_advance2_kernel(const HierArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  extern __shared__ double s_mem[];
  double* s_k = s_mem + (size_t)Lds2<NE>::slots * kBlock;
  double* s_lc = s_k + Lds2<NE>::consts;
  double* s_y = s_lc + a.pts.stride;
  int* s_es = reinterpret_cast<int*>(s_y + a.pts.stride);
  const int slot = blockIdx.x / a.blocks_per_problem;
  const int chunk = blockIdx.x - slot * a.blocks_per_problem;
  const int q = a.prob.launch_order ? a.prob.launch_order[slot] : slot;
  const int c0 = chunk * kChains2;
  const int pair = a.prob.pair_index[q];
#pragma unroll
  for (int i = 0; i < 5; ++i) {                            // static indices: a run-time index would copy the argument block to scratch
    if (threadIdx.x == i) s_k[i] = a.prior.shape_m1[i];
    if (threadIdx.x == 5 + i) s_k[5 + i] = a.prior.inv_scale[i];
    if (threadIdx.x == 10 + i) s_k[10 + i] = a.prior.loc[i];
  }
  stage<NE>(a.pts, pair, s_lc, s_y, s_es);
  if (c0 + (int)(threadIdx.x >> 1) >= a.prob.chains_per_problem) return;    // both lanes of a chain leave together
  bool four_each = true;
#pragma unroll
  for (int i = 1; i <= NE; ++i) four_each = four_each && (a.pts.expt_start[(size_t)pair * (NE + 1) + i] == 4 * i);
  if (four_each) hier_advance2_body<NE, 4, WPS>(a, s_mem, s_k, s_lc, s_y, s_es, q, c0);
  else hier_advance2_body<NE, 0, WPS>(a, s_mem, s_k, s_lc, s_y, s_es, q, c0);
}
template <int NE>
__global__ __launch_bounds__(kBlock) void hier_init_kernel(const HierArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  constexpr int D = 5 + 2 * NE;
  constexpr int TRI = D * (D + 1) / 2;
  extern __shared__ double s_mem[];
  double* s_lc = s_mem;
  double* s_y = s_lc + a.pts.stride;
  int* s_es = reinterpret_cast<int*>(s_y + a.pts.stride);
  const int q = blockIdx.x / a.blocks_per_problem;
  const int chunk = blockIdx.x - q * a.blocks_per_problem;
  const int C = a.prob.chains_per_problem;
  const int c = chunk * kBlock + threadIdx.x;
  const int pair = a.prob.pair_index[q];
  stage<NE>(a.pts, pair, s_lc, s_y, s_es);
  if (c >= C) return;
  const size_t nchains = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;
  double th[D];
#pragma unroll
  for (int i = 0; i < D; ++i) th[i] = a.theta0[(size_t)i * nchains + g];
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  const double lt = phf_hier_log_target(NE, s_es, s_lc, s_y, th, 1, &a.prior, k_exp, k_log);
  double* sp = a.state + g;
#pragma unroll
  for (int i = 0; i < D; ++i) { sp[(size_t)i * nchains] = th[i]; sp[(size_t)(D + 1 + i) * nchains] = th[i]; }
  sp[(size_t)D * nchains] = lt;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)   // L D L' factors of diag(cov_scale |theta0|)  (PyHillFit.py:431): L = I, d on the diagonal slots
      sp[(size_t)(2 * D + 1 + i * (i + 1) / 2 + j) * nchains] = (i != j) ? 0.0 : a.cov_scale * __builtin_fabs(th[i]);
  sp[(size_t)(2 * D + 1 + TRI) * nchains] = 0.0;
  sp[(size_t)(2 * D + 2 + TRI) * nchains] = 0.0;
  if (a.row0) {
    double* o = a.row0 + ((size_t)q * (D + 1)) * C + c;
#pragma unroll
    for (int i = 0; i < D; ++i) o[(size_t)i * C] = th[i];
    o[(size_t)D * C] = lt;
  }
}

template <int NE>
__global__ __launch_bounds__(kBlock) void hier_log_target_kernel(const phf_hier_points pts, const phf_hier_prior prior, int64_t m,
                                                                 const int32_t* pair_index, const double* theta, double* out) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  constexpr int D = 5 + 2 * NE;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const int pair = pair_index[i];
  double th[D];
#pragma unroll
  for (int k = 0; k < D; ++k) th[k] = theta[(size_t)k * m + i];
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  out[i] = phf_hier_log_target(NE, pts.expt_start + (size_t)pair * (NE + 1), pts.ln_conc + (size_t)pair * pts.stride,
                               pts.response + (size_t)pair * pts.stride, th, 1, &prior, k_exp, k_log);
}


// ---------------------------------------------------------------------------------------------------------------
// Pairs with more experiments than the per-Ne kernels are compiled for (Ne > PHF_HIER_FAST_EXPTS, up to PHF_HIER_MAX_EXPTS = 64:
// dim = 5 + 2 Ne known only at run time).

__device__ __forceinline__ double gen_target(const HierArgs& a, int ne, int pair, const double* th, int ts, phf_ktab k_exp, phf_ktab k_log) {
  return phf_hier_log_target_any(ne, a.pts.expt_start + (size_t)pair * (ne + 1), a.pts.ln_conc + (size_t)pair * a.pts.stride,
                                 a.pts.response + (size_t)pair * a.pts.stride, th, ts, &a.prior, k_exp, k_log);
}

// ---------------------------------------------------------------------------------------------------------------
// One WAVEFRONT per chain, for pairs with many experiments (dim = 5 + 2 Ne up to 133): the chain's factor (dim(dim+1)/2
// doubles: 44.5 KB at dim 105), theta, mean, proposal and normals live in LDS; the 64 lanes take one experiment each in
// the target (phf_hier_experiment_terms) and the rows lane, lane+64, ... of the factor in the proposal and in the update
// sweep.  Every element is computed by the same operations in the same order as in the twin (the per-experiment sums are
// added in experiment order by every lane), so results are bit-identical.  (A first version kept the state in HBM, 16 chains
// per block: 1.2 ms per iteration at dim 105 against tens of us here.  It was unreachable once this kernel covered every
// supported dimension — 71 KB of LDS at Ne = 64 — and has been removed.)
struct WaveLds {
  int D, tri, stride, ne;
  __host__ __device__ size_t doubles() const { return (size_t)tri + 4 * (size_t)D + 3 * 64 + 2 * (size_t)stride; }
  __host__ __device__ size_t bytes() const { return doubles() * 8 + (size_t)(ne + 1) * 4 + 8; }
};

__global__ __launch_bounds__(kBlock) void hier_wave_advance_kernel(const HierArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  extern __shared__ double s_mem[];
  const int ne = a.pts.n_expts;
  const int D = 5 + 2 * ne;
  const int TRI = D * (D + 1) / 2;
  const int lane = threadIdx.x;
  double* sLm = s_mem;                           // packed lower triangle, row-major
  double* s_th = sLm + TRI;
  double* s_mean = s_th + D;
  double* s_star = s_mean + D;
  double* s_z = s_star + D;                      // normals; reused as the update vector w
  double* s_part = s_z + D;                      // [3][64]: per-experiment sse, trunc, hyper
  double* s_lc = s_part + 3 * 64;
  double* s_y = s_lc + a.pts.stride;
  int* s_es = reinterpret_cast<int*>(s_y + a.pts.stride);
  const int C = a.prob.chains_per_problem;
  const int slot = blockIdx.x / C;
  const int c = blockIdx.x - slot * C;
  const int q = a.prob.launch_order ? a.prob.launch_order[slot] : slot;
  const int pair = a.prob.pair_index[q];
  const uint32_t pid = a.prob.problem_id[q];
  const uint32_t cid = a.prob.chain_id_base + (a.prob.chain_offset ? a.prob.chain_offset[q] : 0u) + (uint32_t)c;
  const uint32_t seed_lo = (uint32_t)a.cfg.seed, seed_hi = (uint32_t)(a.cfg.seed >> 32);
  const size_t nch = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;
  double* sp = a.state + g;                      // element e of this chain's state: sp[e * nch]
  // ---- stage the pair's points and the chain's state ----
  const int n_pts = a.pts.expt_start[(size_t)pair * (ne + 1) + ne];
  for (int j = lane; j < n_pts; j += kBlock) {
    s_lc[j] = a.pts.ln_conc[(size_t)pair * a.pts.stride + j];
    s_y[j] = a.pts.response[(size_t)pair * a.pts.stride + j];
  }
  for (int i = lane; i <= ne; i += kBlock) s_es[i] = a.pts.expt_start[(size_t)pair * (ne + 1) + i];
  for (int i = lane; i < D; i += kBlock) { s_th[i] = sp[(size_t)i * nch]; s_mean[i] = sp[(size_t)(D + 1 + i) * nch]; }
  for (int e = lane; e < TRI; e += kBlock) sLm[e] = sp[(size_t)(2 * D + 1 + e) * nch];
  __syncthreads();
  double lt = sp[(size_t)D * nch], loga = sp[(size_t)(2 * D + 1 + TRI) * nch], nacc = sp[(size_t)(2 * D + 2 + TRI) * nch];
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  double sc = phf_exp_fast_k(0.5 * loga, k_exp);
  const int thin = a.cfg.thinning;
  int until_save = thin - (int)(a.t_begin % thin);
  double* out = a.rows ? a.rows + ((size_t)q * (D + 1)) * C + c : nullptr;
  const size_t row_stride = (size_t)a.prob.num_problems * (D + 1) * C;
  const int nb = (D + 3) / 4;

  for (int64_t t = a.t_begin + 1; t <= a.t_end; ++t) {
    // ---- draws (phf_hier_draws): Philox block b -> normals 4b..4b+3, lane b; the accept uniform (last word of the last block) in every lane ----
    for (int b = lane; b < nb; b += kBlock) {
      const phf_u32x4 w = phf_philox_mh(cid, pid, (uint32_t)t, (uint32_t)b, seed_lo, seed_hi);
      const int i = 4 * b;
      s_z[i] = phf_normal_u32(w.w[0]);
      if (i + 1 < D) s_z[i + 1] = phf_normal_u32(w.w[1]);
      if (i + 2 < D) s_z[i + 2] = phf_normal_u32(w.w[2]);
      if (i + 3 < D) s_z[i + 3] = phf_normal_u32(w.w[3]);
    }
    const phf_u32x4 wu = phf_philox_mh(cid, pid, (uint32_t)t, (uint32_t)(nb - 1), seed_lo, seed_hi);
    const double log_u = phf_log_pos_k(phf_unit_open32(wu.w[3]), k_log);
    __syncthreads();
    // ---- proposal theta* = theta + e^(loga/2) L sqrt(D) z: u = sqrt(d) z in place, then rows lane, lane+64, ... ----
    for (int i = lane; i < D; i += kBlock) s_z[i] = phf_sqrt_nonneg(sLm[i * (i + 1) / 2 + i]) * s_z[i];
    __syncthreads();
    for (int i = lane; i < D; i += kBlock) {
      const double* row = sLm + i * (i + 1) / 2;
      double v = 0.0;
      for (int k = 0; k < i; ++k) v = phf_fma(row[k], s_z[k], v);
      s_star[i] = phf_fma(sc, v + s_z[i], s_th[i]);
    }
    __syncthreads();
    // ---- target: one experiment per lane, sums in experiment order ----
    const phf_hier_common cm = phf_hier_common_terms(s_star[0], s_star[1], s_star[2], s_star[3], s_star[D - 1], &a.prior, k_log);
    int bad_mine = 0;
    for (int i = lane; i < ne; i += kBlock) {                     // PHF_HIER_MAX_EXPTS = 64: at most one per lane
      double e_sse, e_trunc, e_hyper;
      phf_hier_experiment_terms(&cm, s_star[4 + 2 * i], s_star[5 + 2 * i], s_lc + s_es[i], s_y + s_es[i], s_es[i + 1] - s_es[i],
                                k_exp, k_log, &e_sse, &e_trunc, &e_hyper, &bad_mine);
      s_part[i] = e_sse; s_part[64 + i] = e_trunc; s_part[128 + i] = e_hyper;
    }
    const int bad = __any(bad_mine) ? 1 : 0;
    __syncthreads();
    double sse = 0.0, trunc = 0.0, hyper = 0.0;
    for (int i = 0; i < ne; ++i) { sse += s_part[i]; trunc += s_part[64 + i]; hyper += s_part[128 + i]; }
    const double lt_star = phf_hier_combine(&cm, n_pts, sse, trunc, hyper, bad);
    // ---- accept (the same decision in every lane) ----
    const bool acc = log_u < lt_star - lt;
    if (acc) {
      for (int i = lane; i < D; i += kBlock) s_th[i] = s_star[i];
      lt = lt_star;
    }
    nacc += acc ? 1.0 : 0.0;
    __syncthreads();
    // ---- adaptation: rank-one update of L D L' (PHF_LDL_COLUMN), column by column; rows of a column in parallel ----
    if (t > a.cfg.adapt_start) {
      const double gs = a.cfg.gamma[t - a.cfg.adapt_start];
      const double omg = 1.0 - gs;
      double* s_w = s_z;
      for (int i = lane; i < D; i += kBlock) {
        const double thi = s_th[i], mi = s_mean[i];
        s_w[i] = thi - mi;
        s_mean[i] = phf_fma(gs, thi, omg * mi);
      }
      loga = phf_fma(gs, (acc ? 1.0 : 0.0) - 0.25, loga);
      __syncthreads();
      double alpha = gs;                                          // the same chain in every lane
      for (int k = 0; k < D; ++k) {
        const double pk = s_w[k];
        double dn, beta;
        PHF_LDL_COLUMN(omg, alpha, pk, sLm[k * (k + 1) / 2 + k], dn, beta);
        __syncthreads();                                          // everybody has read d_k and w_k
        if (lane == 0) sLm[k * (k + 1) / 2 + k] = dn;
        for (int i = k + 1 + lane; i < D; i += kBlock) {
          const double lik = sLm[i * (i + 1) / 2 + k];
          const double wi = phf_fma(-pk, lik, s_w[i]);
          s_w[i] = wi;
          sLm[i * (i + 1) / 2 + k] = phf_fma(beta, wi, lik);
        }
        __syncthreads();                                          // w_{k+1} is final before the next column reads it
      }
      sc = phf_exp_fast_k(0.5 * loga, k_exp);
    }
    // ---- thinning + sample store ----
    if (--until_save == 0) {
      until_save = thin;
      if (out) {
        for (int i = lane; i < D; i += kBlock) out[(size_t)i * C] = s_th[i];
        if (lane == 0) out[(size_t)D * C] = lt;
        out += row_stride;
      }
      if (a.moments && t > a.moments_after) {
        for (int i = lane; i < D; i += kBlock) {
          const double x = s_th[i];
          phf_accumulate(&a.moments[(size_t)i * nch + g], x);
          phf_accumulate(&a.moments[(size_t)(D + 1 + i) * nch + g], x * x);
        }
        if (lane == 0) {
          phf_accumulate(&a.moments[(size_t)D * nch + g], lt);
          phf_accumulate(&a.moments[(size_t)(2 * D + 1) * nch + g], lt * lt);
        }
      }
    }
  }
  __syncthreads();
  for (int i = lane; i < D; i += kBlock) { sp[(size_t)i * nch] = s_th[i]; sp[(size_t)(D + 1 + i) * nch] = s_mean[i]; }
  for (int e = lane; e < TRI; e += kBlock) sp[(size_t)(2 * D + 1 + e) * nch] = sLm[e];
  if (lane == 0) {
    sp[(size_t)D * nch] = lt;
    sp[(size_t)(2 * D + 1 + TRI) * nch] = loga;
    sp[(size_t)(2 * D + 2 + TRI) * nch] = nacc;
  }
}

__global__ __launch_bounds__(64) void hier_generic_init_kernel(const HierArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  const int ne = a.pts.n_expts;
  const int D = 5 + 2 * ne;
  const int q = blockIdx.x / a.blocks_per_problem;
  const int chunk = blockIdx.x - q * a.blocks_per_problem;
  const int C = a.prob.chains_per_problem;
  const int c = chunk * 64 + threadIdx.x;
  if (c >= C) return;
  const int pair = a.prob.pair_index[q];
  const size_t nch = (size_t)a.prob.num_problems * C;
  const size_t g = (size_t)q * C + c;
  double* sp = a.state + g;
  for (int i = 0; i < D; ++i) {
    const double v = a.theta0[(size_t)i * nch + g];
    sp[(size_t)i * nch] = v; sp[(size_t)(D + 1 + i) * nch] = v;
    for (int j = 0; j <= i; ++j)
      sp[(size_t)(2 * D + 1 + i * (i + 1) / 2 + j) * nch] = (i != j) ? 0.0 : a.cov_scale * __builtin_fabs(v);
  }
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  const double lt = gen_target(a, ne, pair, sp, (int)nch, k_exp, k_log);
  sp[(size_t)D * nch] = lt;
  const int tri = D * (D + 1) / 2;
  sp[(size_t)(2 * D + 1 + tri) * nch] = 0.0;
  sp[(size_t)(2 * D + 2 + tri) * nch] = 0.0;
  if (a.row0) {
    double* o = a.row0 + ((size_t)q * (D + 1)) * C + c;
    for (int i = 0; i < D; ++i) o[(size_t)i * C] = sp[(size_t)i * nch];
    o[(size_t)D * C] = lt;
  }
}

__global__ __launch_bounds__(64) void hier_generic_log_target_kernel(const phf_hier_points pts, const phf_hier_prior prior, int64_t m,
                                                                     const int32_t* pair_index, const double* theta, double* out) {
  PHF_MATH_TABLES_TO_LDS();
  PHF_ERFC_TABLE_TO_LDS();
  PHF_NORMAL_TABLE_TO_LDS();
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= m) return;
  const int ne = pts.n_expts;
  const int pair = pair_index[i];
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(k_log, phf_k_log, PHF_K_LOG_N);
  out[i] = phf_hier_log_target_any(ne, pts.expt_start + (size_t)pair * (ne + 1), pts.ln_conc + (size_t)pair * pts.stride,
                               pts.response + (size_t)pair * pts.stride, theta + i, (int)m, &prior, k_exp, k_log);
}

int check(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior) {
  if (!pts || !prior) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null hierarchical points/prior");
  if (pts->n_expts < 1 || pts->n_expts > PHF_HIER_MAX_EXPTS)
    return phf_fail(PHF_ERR_UNSUPPORTED, "hierarchical sampler supports 1..64 experiments per pair");
  if (pts->num_pairs <= 0 || pts->stride <= 0 || !pts->ln_conc || !pts->response || !pts->expt_start)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "incomplete phf_hier_points");
  if (prob) {
    if (prob->num_problems <= 0 || prob->chains_per_problem <= 0 || !prob->pair_index || !prob->problem_id)
      return phf_fail(PHF_ERR_INVALID_ARGUMENT, "incomplete phf_problems");
    const int64_t bpp = (pts->n_expts > PHF_HIER_FAST_EXPTS) ? prob->chains_per_problem : (prob->chains_per_problem + kBlock - 1) / kBlock;
    if (bpp * prob->num_problems > 0x7fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many chains for one launch");
    if ((int64_t)prob->num_problems * prob->chains_per_problem > 0x7fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many chains");
    if ((prob->kernel_hint & 3u) == 3u || ((prob->kernel_hint >> 2) & 3u) == 3u || (prob->kernel_hint & ~(31u | 64u)))
      return phf_fail(PHF_ERR_INVALID_ARGUMENT, "kernel_hint: bits 0-1 and 2-3 hold 0, 1 or 2; bit 4 = not the gfx950 assembly build; bit 6 = the queue workspace "
                                                "holds phf_hierarchical_queue_words() words; the other bits must be 0");
  }
  if (pts->points_per_expt != 0) {                          // a shape code: per | last << 4, or (bit 30) a nibble per experiment
    const int code = pts->points_per_expt;
    bool ok = code > 0;
    int n_pts = 0;
    if (ok && (code & (1 << 30))) {
      ok = pts->n_expts <= 7 && (pts->n_expts == 7 || ((code & ~(1 << 30)) >> (4 * pts->n_expts)) == 0);
      for (int i = 0; ok && i < pts->n_expts; ++i) { const int n = (code >> (4 * i)) & 15; ok = n > 0; n_pts += n; }
    } else if (ok) {
      const int per = code & 15, last = code >> 4;
      ok = per > 0 && last <= 15;
      n_pts = (pts->n_expts - 1) * per + (last ? last : per);
    }
    if (!ok || n_pts > pts->stride)
      return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_hier_points.points_per_expt: 0 (unknown / no such code), per | last << 4 (the points of EVERY experiment, and of the "
                                                "last one if it differs), or bit 30 + a nibble per experiment — of EVERY pair");
  }
  return PHF_OK;
}

constexpr int kMaxDevices = 64;
thread_local int g_last_kernel = 0;               // phf_hierarchical_last_kernel(): what this thread's last advance launched

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return (dev >= 0 && dev < kMaxDevices) ? dev : 0;
}

// Which kernel runs a group of pairs with Ne <= PHF_HIER_FAST_EXPTS experiments (measured on C4's groups and on the command
// lines' 64 chains per pair, tools/diag_hier_lanes.py, profiles/r02/hier_lanes_ab.txt):
//   - a launch whose two-lane wavefronts each get a SIMD of their own (blocks <= SIMDs: the command-line regime, where the run
//     time is the LATENCY of one wavefront-iteration) runs the two-lane kernel built for one wavefront per SIMD: about half the
//     instructions per iteration;
//   - bigger launches run the one-lane kernel: a lone wavefront already issues an instruction every ~2.2 ns, the rate at which
//     the fp64 pipe retires them, so two wavefronts per SIMD gain nothing and the two-lane split costs ~18 % more vector
//     instructions per chain (rotations and draws computed twice, cross-lane moves).
// phf_hierarchical_set_kernel_policy(lanes, wps) forces one / the register build of the two-lane kernel (tests, A/B timing); the
// environment variables PHF_HIER_LANES / PHF_HIER_WPS = 1|2 give the initial values and are read once per process.
constexpr int kMinNe2 = 3, kMaxNe2 = 6;                           // the two-lane kernel is compiled for the Crumb set's Ne = 3..6

int env_1_or_2(const char* name) {
  const char* e = getenv(name);
  return (e && (e[0] == '1' || e[0] == '2') && e[1] == 0) ? e[0] - '0' : 0;
}
struct HierPolicy { std::atomic<int> lanes, wps; };
HierPolicy& hier_policy() {                                       // first use reads the environment; later changes through the ABI only
  static HierPolicy p{{env_1_or_2("PHF_HIER_LANES")}, {env_1_or_2("PHF_HIER_WPS")}};
  return p;
}
// process-wide policy first, then the launch's own hint (phf_problems.kernel_hint), then 0 = decide from the launch size
int hier_lanes_override(const HierArgs& a) {
  const int forced = hier_policy().lanes.load(std::memory_order_relaxed);
  const int hint = (int)(a.prob.kernel_hint & 3u);
  return forced ? forced : (hint <= 2 ? hint : 0);
}
// The hand-allocated gfx950 build of the Ne = 3 iteration (phf_hier3_isa.hip, tools/gen_hier_isa.py): 256 registers, two wavefronts
// per SIMD, 256-thread workgroups sharing one copy of the function tables.  Used for a launch that (a) holds pairs with exactly four
// points in each of three experiments (phf_hier_points.points_per_expt == 4: the caller's statement about its data, like n_expts),
// (b) would get the one-lane kernel, (c) is not told otherwise: kernel_hint bit 4 or PHF_HIER_ISA=0 in the environment (read once)
// select the hipcc kernels — same numbers bit for bit, kept for A/B timing and for the bit-identity tests.
bool hier_isa_enabled() {
  static const bool on = [] { const char* e = getenv("PHF_HIER_ISA"); return !(e && e[0] == '0' && e[1] == 0); }();
  return on;
}

// layout of the queue workspace (phf_hierarchical_queue_words): [task counter][progress word per block][sticky fault word], then — for
// the kernels with a scratch tier — up to 512 bytes of alignment and 512 bytes x slots for each of min(blocks, the chip's wavefront slots)
// (+ 3: a grid is whole workgroups of four) resident wavefronts
double* hier_queue_scratch(int32_t* queue, int64_t total) {
  const uintptr_t p = reinterpret_cast<uintptr_t>(queue + 2 + total);
  return reinterpret_cast<double*>((p + 511) & ~(uintptr_t)511);
}
int64_t hier_queue_words(int n_expts, int shape_code, int64_t total) {
  int64_t words = 2 + total;
  const int scratch_slots = phf_hier_isa_scratch_slots(n_expts, shape_code);
  if (scratch_slots > 0) {
    const int64_t slots = 2LL * phf_simd_count();
    words += 128 + ((total < slots ? total : slots) + 3) * scratch_slots * 128;
  }
  return words;
}

// the argument block of one launch group for the gfx950 kernels (generated/phf_hier3_isa_layout.h): everything but the queue-level fields
void fill_isa_args(const HierArgs& a, phf_hier3_isa_args* gp) {
  phf_hier3_isa_args& g = *gp;
  const int64_t bpp = a.blocks_per_problem;
  const int64_t total = bpp * a.prob.num_problems;
  g.state = a.state; g.rows = a.rows; g.moments = a.moments; g.gamma = a.cfg.gamma;
  g.ln_conc = a.pts.ln_conc; g.response = a.pts.response; g.pair_index = a.prob.pair_index; g.problem_id = a.prob.problem_id;
  g.launch_order = a.prob.launch_order; g.chain_offset = a.prob.chain_offset;
  g.t_begin = (uint32_t)a.t_begin; g.t_end = (uint32_t)a.t_end;
  g.adapt_start = (uint32_t)(a.cfg.adapt_start > 0xffffffffLL ? 0xffffffffLL : a.cfg.adapt_start);
  g.thinning = a.cfg.thinning;
  g.moments_after = (uint32_t)(a.moments_after < 0 ? 0 : (a.moments_after > 0xffffffffLL ? 0xffffffffLL : a.moments_after));
  g.chains = a.prob.chains_per_problem; g.num_problems = a.prob.num_problems; g.bpp = (int32_t)bpp;
  g.bpp_magic = phf_isa_magic((uint32_t)bpp);
  g.total_waves = (int32_t)total;
  g.seed_lo = (uint32_t)a.cfg.seed; g.seed_hi = (uint32_t)(a.cfg.seed >> 32);
  g.chain_id_base = a.prob.chain_id_base; g.pts_stride = a.pts.stride;
  g.until_save0 = a.cfg.thinning - (int32_t)(a.t_begin % a.cfg.thinning);
  for (int i = 0; i < 5; ++i) { g.prior_loc[i] = a.prior.loc[i]; g.prior_inv_scale[i] = a.prior.inv_scale[i]; g.prior_shape_m1[i] = a.prior.shape_m1[i]; }
  g.three_twelve[0] = 3.0; g.three_twelve[1] = 12.0;
}

int launch_isa(const HierArgs& a, hipStream_t stream, bool* launched) {
  *launched = false;
  const int64_t bpp = a.blocks_per_problem;
  const int64_t total = bpp * a.prob.num_problems;
  if (a.cfg.adapt_start < 0 || a.t_end >= 0xffffffffLL || total > 0x7fffffffLL) return PHF_OK;   // hipcc kernels
  // a kernel that keeps part of the state in device-memory scratch needs the caller's word that the workspace holds it (kernel_hint bit 6)
  const int scratch_slots = phf_hier_isa_scratch_slots(a.pts.n_expts, a.pts.points_per_expt);
  if (scratch_slots > 0 && (!a.queue || !(a.prob.kernel_hint & 64u))) return PHF_OK;
  const int which = phf_hier_isa_find(a.pts.n_expts, a.pts.points_per_expt);
  if (which < 0) return PHF_OK;
  phf_hier3_isa_args g{};
  fill_isa_args(a, &g);
  // Work queue (phf_hierarchical_advance_queued): the launch is cut into quanta, the grid is only as large as the chip holds (two
  // wavefronts per SIMD) and its wavefronts pull (quantum, block) tasks, so that the last round of a launch is a round of short tasks:
  // 147 pairs x 1 024 chains are 2 352 wavefronts on 2 048 slots — 1.15 rounds that cost two without the queue.  Worth it only when
  // the blocks do not fit the chip at once and the launch has at least two quanta; the quantum is a multiple of the thinning, so that
  // every quantum saves the same number of rows.
  const int64_t slots = 2LL * phf_simd_count();
  // the library's quantum: as long as leaves ~16 rounds of tasks on the chip's slots (the ragged last round is then ~3 % of the launch;
  // every quantum costs a block one trip of its 91-double state through HBM), at least 100 iterations: 140 for C4's 2 000-iteration
  // steps (2 352 blocks), 1 435 for the command line's 20 000-iteration segments at that width
  int64_t quantum = a.quantum;
  if (quantum <= 0) {
    quantum = (a.t_end - a.t_begin) * total / (16 * slots);
    if (quantum < 100) quantum = 100;
  }
  quantum -= quantum % a.cfg.thinning;
  if (quantum < a.cfg.thinning) quantum = a.cfg.thinning;
  const int64_t nquanta = (a.t_end - a.t_begin + quantum - 1) / quantum;
  int64_t grid_waves = total;
  if (a.queue && total > slots && nquanta >= 2 && nquanta * total < (1LL << 31) && a.t_begin % a.cfg.thinning == 0) {
    if (hipMemsetAsync(a.queue, 0, (size_t)(1 + total) * sizeof(int32_t), stream) != hipSuccess)
      return phf_check_launch("phf_hierarchical_advance_queued (memset)");
    g.queue = a.queue; g.quantum = (uint32_t)quantum; g.num_tasks = (int32_t)(nquanta * total);
    g.blocks_magic = phf_isa_magic((uint32_t)total);
    g.rows_per_quantum = (uint32_t)(quantum / a.cfg.thinning);
    grid_waves = slots;
  }
  if (scratch_slots > 0) {
    if (grid_waves > slots) return PHF_OK;                  // more blocks than the chip holds and no queue to pull them through: hipcc kernels
    g.scratch = hier_queue_scratch(a.queue, total);
  }
  *launched = true;
  g_last_kernel = g.queue ? PHF_HIER_KERNEL_GFX950_ISA_QUEUED : PHF_HIER_KERNEL_GFX950_ISA;
  return phf_hier_isa_advance(which, &g, (int)grid_waves, stream);
}

int hier_wps_override(const HierArgs& a) {
  const int forced = hier_policy().wps.load(std::memory_order_relaxed);
  const int hint = (int)((a.prob.kernel_hint >> 2) & 3u);
  return forced ? forced : (hint <= 2 ? hint : 0);
}

// a workgroup's 160 KB of LDS less the static part every kernel here has: the exp / log, erfc and normal tables of phf_math.h
constexpr size_t kMaxDynamicLds = 160 * 1024 - PHF_MATH_LDS_BYTES - PHF_ERFC_TAB_N * sizeof(phf_erfctab) - PHF_NORMAL_TAB_N * sizeof(phf_normtab);

template <typename K>
int allow_big_lds(K kernel, bool* configured) {                   // the attribute is per function AND per device
  const int dev = current_device();
  if (!configured[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds) != hipSuccess)
      return phf_check_launch("hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB less the math tables) for a hierarchical kernel");
    configured[dev] = true;
  }
  return PHF_OK;
}

template <int NE>
int launch_advance1(const HierArgs& a, hipStream_t stream) {
  const size_t lds = Lds<NE>::bytes(a.pts.stride);
  if (lds > kMaxDynamicLds) return phf_fail(PHF_ERR_UNSUPPORTED, "proposal factor does not fit in LDS");
  const dim3 grid((unsigned)(a.blocks_per_problem * a.prob.num_problems)), block(kBlock);
  static bool configured[kMaxDevices] = {};
  if (int rc = allow_big_lds(&hier_advance_kernel<NE>, configured)) return rc;
  hipLaunchKernelGGL((hier_advance_kernel<NE>), grid, block, lds, stream, a);
  g_last_kernel = PHF_HIER_KERNEL_ONE_LANE;
  return phf_check_launch("phf_hierarchical_advance");
}

template <int NE, int WPS>
int launch_advance2_wps(const HierArgs& a, size_t lds, hipStream_t stream) {
  static bool configured[kMaxDevices] = {};
  if (int rc = allow_big_lds(&hier_advance2_kernel<NE, WPS>, configured)) return rc;
  const dim3 grid((unsigned)(a.blocks_per_problem * a.prob.num_problems)), block(kBlock);
  hipLaunchKernelGGL((hier_advance2_kernel<NE, WPS>), grid, block, lds, stream, a);
  g_last_kernel = PHF_HIER_KERNEL_TWO_LANES;
  return phf_check_launch("phf_hierarchical_advance (two lanes per chain)");
}

template <int NE>
int launch_advance2(HierArgs a, hipStream_t stream) {
  const size_t lds = Lds2<NE>::bytes(a.pts.stride);
  if (lds > kMaxDynamicLds) return phf_fail(PHF_ERR_UNSUPPORTED, "proposal factor does not fit in LDS");
  a.blocks_per_problem = (a.prob.chains_per_problem + kChains2 - 1) / kChains2;
  const int64_t blocks = (int64_t)a.blocks_per_problem * a.prob.num_problems;
  if (blocks > 0x7fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many chains for one launch");
  const int wps = hier_wps_override(a);
  if (wps ? wps == 1 : blocks <= phf_simd_count()) return launch_advance2_wps<NE, 1>(a, lds, stream);
  return launch_advance2_wps<NE, 2>(a, lds, stream);
}

template <int NE>
int launch_advance(const HierArgs& a, hipStream_t stream) {
#ifndef PHF_EXP_NO_TWO_LANE
  if constexpr (NE >= kMinNe2 && NE <= kMaxNe2) {
    const int force = hier_lanes_override(a);
    const int64_t blocks2 = (int64_t)((a.prob.chains_per_problem + kChains2 - 1) / kChains2) * a.prob.num_problems;
    const bool two = force ? force == 2 : blocks2 <= phf_simd_count();
    if (two) return launch_advance2<NE>(a, stream);
  }
#endif
  if (a.pts.points_per_expt != 0 && !(a.prob.kernel_hint & 16u) && hier_isa_enabled()) {
    bool launched = false;
    const int rc = launch_isa(a, stream, &launched);
    if (rc != PHF_OK || launched) return rc;
  }
  return launch_advance1<NE>(a, stream);
}

template <int NE>
int launch_init(const HierArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.blocks_per_problem * a.prob.num_problems)), block(kBlock);
  hipLaunchKernelGGL(hier_init_kernel<NE>, grid, block, (size_t)a.pts.stride * 16 + (NE + 1) * 4 + 8, stream, a);
  return phf_check_launch("phf_hierarchical_init");
}

// PHF_EXP_ONLY_NE (timing experiments, tools/build_exp.sh): compile the kernels of ONE Ne only — a minute less per build
#ifdef PHF_EXP_ONLY_NE
#define PHF_DISPATCH_NE(ne, FN, ...)                                                                            \
  if ((ne) != PHF_EXP_ONLY_NE) return phf_fail(PHF_ERR_UNSUPPORTED, "experiment build: kernels of one Ne only");  \
  return FN<PHF_EXP_ONLY_NE>(__VA_ARGS__);
#else
#define PHF_DISPATCH_NE(ne, FN, ...)                                \
  switch (ne) {                                                     \
    case 1: return FN<1>(__VA_ARGS__);                              \
    case 2: return FN<2>(__VA_ARGS__);                              \
    case 3: return FN<3>(__VA_ARGS__);                              \
    case 4: return FN<4>(__VA_ARGS__);                              \
    case 5: return FN<5>(__VA_ARGS__);                              \
    case 6: return FN<6>(__VA_ARGS__);                              \
    case 7: return FN<7>(__VA_ARGS__);                              \
    default: return FN<8>(__VA_ARGS__);                             \
  }
#endif

int launch_wave_advance(const HierArgs& a, hipStream_t stream, bool* launched) {
  WaveLds w;
  w.ne = a.pts.n_expts; w.D = 5 + 2 * w.ne; w.tri = w.D * (w.D + 1) / 2; w.stride = a.pts.stride;
  *launched = false;
  if (w.bytes() > kMaxDynamicLds) return PHF_OK;                     // dimension too large for LDS (cannot happen for Ne <= 64): caller reports it
  const int64_t blocks = (int64_t)a.prob.num_problems * a.prob.chains_per_problem;
  if (blocks > 0x7fffffffLL) return PHF_OK;
  static bool configured[kMaxDevices] = {};
  if (int rc = allow_big_lds(&hier_wave_advance_kernel, configured)) return rc;
  hipLaunchKernelGGL(hier_wave_advance_kernel, dim3((unsigned)blocks), dim3(kBlock), w.bytes(), stream, a);
  *launched = true;
  g_last_kernel = PHF_HIER_KERNEL_WAVE;
  return phf_check_launch("phf_hierarchical_advance (wave per chain)");
}

int launch_generic_advance(HierArgs a, hipStream_t stream) {
  bool launched = false;
  const int rc = launch_wave_advance(a, stream, &launched);
  if (rc != PHF_OK || launched) return rc;
  return phf_fail(PHF_ERR_UNSUPPORTED, "hierarchical state does not fit in LDS (dimension too large, or too many chains for one launch)");
}

int launch_generic_init(HierArgs a, hipStream_t stream) {
  a.blocks_per_problem = (a.prob.chains_per_problem + 63) / 64;
  hipLaunchKernelGGL(hier_generic_init_kernel, dim3((unsigned)(a.blocks_per_problem * a.prob.num_problems)), dim3(64), 0, stream, a);
  return phf_check_launch("phf_hierarchical_init (generic Ne)");
}

int dispatch_advance(const HierArgs& a, hipStream_t s) {
  if (a.pts.n_expts > PHF_HIER_FAST_EXPTS) return launch_generic_advance(a, s);
  PHF_DISPATCH_NE(a.pts.n_expts, launch_advance, a, s)
}
int dispatch_init(const HierArgs& a, hipStream_t s) {
  if (a.pts.n_expts > PHF_HIER_FAST_EXPTS) return launch_generic_init(a, s);
  PHF_DISPATCH_NE(a.pts.n_expts, launch_init, a, s)
}

template <int NE>
int launch_log_target(const phf_hier_points& pts, const phf_hier_prior& prior, int64_t m, const int32_t* pair_index,
                      const double* theta, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(hier_log_target_kernel<NE>, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, pts, prior, m,
                     pair_index, theta, out);
  return phf_check_launch("phf_hierarchical_log_target");
}

}  // namespace

extern "C" {

int phf_hierarchical_state_size(int n_expts) {
  if (n_expts < 1 || n_expts > PHF_HIER_MAX_EXPTS) return phf_fail(PHF_ERR_UNSUPPORTED, "hierarchical sampler supports 1..64 experiments per pair");
  const int d = 5 + 2 * n_expts;
  return 2 * d + d * (d + 1) / 2 + 3;
}

int phf_hierarchical_last_kernel(void) { return g_last_kernel; }

int phf_hierarchical_set_kernel_policy(int lanes, int wps) {
  if (lanes < 0 || lanes > 2 || wps < 0 || wps > 2) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "kernel policy: lanes and wps must be 0 (automatic), 1 or 2");
  hier_policy().lanes.store(lanes, std::memory_order_relaxed);
  hier_policy().wps.store(wps, std::memory_order_relaxed);
  return PHF_OK;
}

int phf_hierarchical_init(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                          double cov_scale, const double* theta0, double* state, double* row0, void* stream) {
  if (!prob) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null problems");
  if (int rc = check(pts, prob, prior)) return rc;
  if (!theta0 || !state) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null theta0/state");
  phf_forget_device_memory_verdicts();              // a new sampler: every buffer's kind is asked of the runtime again (phf_common.h)
  HierArgs a{};
  a.pts = *pts; a.prob = *prob; a.prior = *prior; a.state = state; a.cov_scale = cov_scale; a.theta0 = theta0; a.row0 = row0;
  a.blocks_per_problem = (prob->chains_per_problem + kBlock - 1) / kBlock;
  return dispatch_init(a, (hipStream_t)stream);
}

static int hier_advance_impl(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                             const phf_mh_config* cfg, int64_t t_begin, int64_t t_end, double* state, double* rows,
                             double* moments, int64_t moments_after, int32_t quantum, int32_t* queue, void* stream) {
  if (!prob || !cfg) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null problems/config");
  if (int rc = check(pts, prob, prior)) return rc;
  if (!state) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "null state");
  if (cfg->thinning <= 0) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "thinning must be positive");
  if (t_begin < 0 || t_end < t_begin || t_end > 0xffffffffLL) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad iteration range");
  if (t_end > cfg->adapt_start && !cfg->gamma) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "gamma table required once adapting");
  if (t_end == t_begin) return PHF_OK;
  if (int rc = phf_require_device_memory(state, "state")) return rc;
  if (int rc = phf_require_device_memory(moments, "moments")) return rc;
  HierArgs a{};
  a.pts = *pts; a.prob = *prob; a.prior = *prior; a.cfg = *cfg; a.t_begin = t_begin; a.t_end = t_end; a.state = state; a.rows = rows;
  a.moments = moments; a.moments_after = moments_after;
  a.blocks_per_problem = (prob->chains_per_problem + kBlock - 1) / kBlock;
  a.queue = queue; a.quantum = quantum;
  if (int rc = phf_require_device_memory(queue, "queue workspace")) return rc;
  return dispatch_advance(a, (hipStream_t)stream);
}

int64_t phf_hierarchical_queue_words(const phf_hier_points* pts, const phf_problems* prob) {
  if (!pts || !prob || prob->num_problems <= 0 || prob->chains_per_problem <= 0 || pts->n_expts < 1)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_hierarchical_queue_words: incomplete phf_hier_points / phf_problems");
  const int64_t total = (int64_t)((prob->chains_per_problem + kBlock - 1) / kBlock) * prob->num_problems;
  return hier_queue_words(pts->n_expts, pts->points_per_expt, total);
}

// ---- every launch group of a run through ONE persistent grid (phf_hier_fused_advance: a body per (experiments, point shape)) ----
static int fused_plan(int32_t n_groups, const phf_hier_group* groups, int* body_of, int64_t* blocks_of, int64_t* total, bool* scratch) {
  if (n_groups < 1 || n_groups > PHF_ISA_FUSED_BODIES || !groups)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "fused hierarchical launch: 1..PHF_ISA_FUSED_BODIES groups");
  *total = 0; *scratch = false;
  bool taken[PHF_ISA_FUSED_BODIES] = {};
  for (int i = 0; i < n_groups; ++i) {
    const phf_hier_group& g = groups[i];
    if (!g.pts || !g.prob || g.prob->num_problems <= 0 || g.prob->chains_per_problem <= 0)
      return phf_fail(PHF_ERR_INVALID_ARGUMENT, "fused hierarchical launch: incomplete group");
    int b = -1;
    for (int j = 0; j < PHF_ISA_FUSED_BODIES; ++j)
      if (phf_isa_hier_kernels[j].n_expts == g.pts->n_expts && phf_isa_hier_kernels[j].shape_code == g.pts->points_per_expt) b = j;
    if (b < 0) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: no gfx950 kernel for a group's (n_expts, points_per_expt)");
    if (taken[b]) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: two groups of one (n_expts, points_per_expt)");
    taken[b] = true;
    body_of[i] = b;
    blocks_of[i] = (int64_t)((g.prob->chains_per_problem + kBlock - 1) / kBlock) * g.prob->num_problems;
    *total += blocks_of[i];
    if (phf_isa_hier_kernels[b].scratch_slots > 0) *scratch = true;
  }
  if (*total > 0x3fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many chains for one launch");
  return PHF_OK;
}

int64_t phf_hierarchical_fused_queue_words(int32_t n_groups, const phf_hier_group* groups) {
  int body_of[PHF_ISA_FUSED_BODIES]; int64_t blocks_of[PHF_ISA_FUSED_BODIES]; int64_t total; bool scratch;
  if (int rc = fused_plan(n_groups, groups, body_of, blocks_of, &total, &scratch)) return rc;
  int64_t words = 2 + total;
  if (scratch) {
    const int64_t slots = 2LL * phf_simd_count();
    int max_slots = 0;
    for (int j = 0; j < PHF_ISA_FUSED_BODIES; ++j) max_slots = phf_isa_hier_kernels[j].scratch_slots > max_slots ? phf_isa_hier_kernels[j].scratch_slots : max_slots;
    words += 128 + ((total < slots ? total : slots) + 3) * max_slots * 128;
  }
  return words;
}

int phf_hierarchical_advance_fused(int32_t n_groups, const phf_hier_group* groups, const phf_hier_prior* prior, int64_t t_begin, int64_t t_end,
                                   int64_t moments_after, int32_t quantum_in, int32_t* queue, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int body_of[PHF_ISA_FUSED_BODIES]; int64_t blocks_of[PHF_ISA_FUSED_BODIES]; int64_t total; bool scratch;
  if (int rc = fused_plan(n_groups, groups, body_of, blocks_of, &total, &scratch)) return rc;
  if (!prior || !queue || quantum_in < 0) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "fused hierarchical launch: prior, queue workspace, quantum >= 0");
  if (t_begin < 0 || t_end < t_begin || t_end >= 0xffffffffLL) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad iteration range");
  if (t_end == t_begin) return PHF_OK;
  if (!hier_isa_enabled()) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: PHF_HIER_ISA=0");
  const int thinning = groups[0].cfg ? groups[0].cfg->thinning : 0;
  if (thinning <= 0) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "thinning must be positive");
  if (t_begin % thinning) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: t_begin must be a multiple of the thinning");
  if (int rc = phf_require_device_memory(queue, "queue workspace")) return rc;
  static thread_local phf_hier_fused_args fa;
  fa = phf_hier_fused_args{};
  for (int i = 0; i < n_groups; ++i) {
    const phf_hier_group& g = groups[i];
    if (!g.cfg || !g.state) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "fused hierarchical launch: a group without config / state");
    if (int rc = check(g.pts, g.prob, prior)) return rc;
    if (g.cfg->thinning != thinning) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: the groups' thinning differs");
    if (g.cfg->adapt_start < 0) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: negative adapt_start");
    if (t_end > g.cfg->adapt_start && !g.cfg->gamma) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "gamma table required once adapting");
    if (int rc = phf_require_device_memory(g.state, "state")) return rc;
    if (int rc = phf_require_device_memory(g.moments, "moments")) return rc;
    HierArgs a{};
    a.pts = *g.pts; a.prob = *g.prob; a.prior = *prior; a.cfg = *g.cfg; a.t_begin = t_begin; a.t_end = t_end; a.state = g.state; a.rows = g.rows;
    a.moments = g.moments; a.moments_after = moments_after;
    a.blocks_per_problem = (g.prob->chains_per_problem + kBlock - 1) / kBlock;
    phf_hier3_isa_args full{};
    fill_isa_args(a, &full);
    static_assert(sizeof(phf_hier_body_args) <= sizeof(phf_hier3_isa_args), "a body's block is the head of phf_hier3_isa_args");
    std::memcpy(&fa.body[body_of[i]], &full, sizeof(phf_hier_body_args));
  }
  for (int i = 0; i < 5; ++i) { fa.prior_loc[i] = prior->loc[i]; fa.prior_inv_scale[i] = prior->inv_scale[i]; fa.prior_shape_m1[i] = prior->shape_m1[i]; }
  // bounds[b]: first block of body b's group, bodies in table order; a body without a group gets an empty range
  int64_t run = 0;
  for (int b = 0; b < 16; ++b) {
    fa.bounds[b] = (uint32_t)run;
    for (int i = 0; i < n_groups; ++i) if (body_of[i] == b) run += blocks_of[i];
  }
  const int64_t slots = 2LL * phf_simd_count();
  int64_t quantum = quantum_in;
  if (quantum <= 0) {
    quantum = (t_end - t_begin) * total / (16 * slots);
    if (quantum < 100) quantum = 100;
  }
  quantum -= quantum % thinning;
  if (quantum < thinning) quantum = thinning;
  const int64_t nquanta = (t_end - t_begin + quantum - 1) / quantum;
  if (nquanta * total >= (1LL << 31)) return phf_fail(PHF_ERR_UNSUPPORTED, "fused hierarchical launch: too many tasks");
  if (hipMemsetAsync(queue, 0, (size_t)(1 + total) * sizeof(int32_t), stream) != hipSuccess)
    return phf_check_launch("phf_hierarchical_advance_fused (memset)");
  fa.queue = queue; fa.scratch = scratch ? hier_queue_scratch(queue, total) : nullptr;
  fa.t_begin = (uint32_t)t_begin; fa.t_end = (uint32_t)t_end;
  fa.quantum = (uint32_t)quantum; fa.num_tasks = (uint32_t)(nquanta * total); fa.blocks_magic = phf_isa_magic((uint32_t)total);
  fa.rows_per_quantum = (uint32_t)(quantum / thinning); fa.total_blocks = (int32_t)total;
  const int64_t grid_waves = total < slots ? total : slots;
  g_last_kernel = PHF_HIER_KERNEL_GFX950_ISA_FUSED;
  return phf_hier_isa_fused_advance(&fa, (int)grid_waves, stream);
}

int phf_hierarchical_advance(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                             const phf_mh_config* cfg, int64_t t_begin, int64_t t_end, double* state, double* rows,
                             double* moments, int64_t moments_after, void* stream) {
  return hier_advance_impl(pts, prob, prior, cfg, t_begin, t_end, state, rows, moments, moments_after, 0, nullptr, stream);
}

int phf_hierarchical_advance_queued(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                                    const phf_mh_config* cfg, int64_t t_begin, int64_t t_end, double* state, double* rows,
                                    double* moments, int64_t moments_after, int32_t quantum, int32_t* queue, void* stream) {
  if (quantum < 0 || !queue) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "queued advance needs a queue workspace and a quantum >= 0 (0 = the library's)");
  return hier_advance_impl(pts, prob, prior, cfg, t_begin, t_end, state, rows, moments, moments_after, quantum, queue, stream);
}

int phf_hierarchical_log_target(const phf_hier_points* pts, const phf_hier_prior* prior, int64_t m,
                                const int32_t* pair_index, const double* theta, double* out, void* stream) {
  if (int rc = check(pts, nullptr, prior)) return rc;
  if (m < 0 || !pair_index || !theta || !out) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "bad arguments to phf_hierarchical_log_target");
  if (m == 0) return PHF_OK;
  if (m > 0x7fffffffLL) return phf_fail(PHF_ERR_UNSUPPORTED, "too many evaluations for one call");
  if (pts->n_expts > PHF_HIER_FAST_EXPTS) {
    hipLaunchKernelGGL(hier_generic_log_target_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, (hipStream_t)stream, *pts, *prior, m,
                       pair_index, theta, out);
    return phf_check_launch("phf_hierarchical_log_target (generic Ne)");
  }
  PHF_DISPATCH_NE(pts->n_expts, launch_log_target, *pts, *prior, m, pair_index, theta, out, (hipStream_t)stream)
}

}  // extern "C"
