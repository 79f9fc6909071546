/* pyhillfit_amd.h — C ABI of the MI355X-native Metropolis-Hastings engine for PyHillFit's sampling step.
 *
 * The reference (mirams/PyHillFit) has no FFI: its hot path is a Python loop.  These entry points are
 * what a ctypes binding inside the reference would call instead of that loop; each cites the reference
 * code it replaces (file:line into the reference repository).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer marked "device" is a HIP device pointer owned by the caller
 *     (the Python host passes torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default)
 *   - every function returns PHF_OK (0) or a negative error code and never throws or exits; the message is
 *     available from phf_last_error() (thread-local).  The reference's sys.exit() on NaN
 *     (python/PyHillFit.py:126-131,188-191) becomes: NaN proposals are rejected, nothing aborts.
 *   - launches are asynchronous on `stream`; the library allocates nothing and keeps no global state, so one
 *     host thread/process per GPU may call it concurrently.
 *   - all arithmetic is IEEE fp64.  Chain-major arrays are struct-of-arrays with the chain index fastest
 *     ("[f][chain]"), so that the 64 lanes of a wavefront touch 512 contiguous bytes.
 *   - MEMORY KIND: state, moments and the queue workspace must be ordinary coarse-grained device memory (hipMalloc /
 *     torch.empty(device="cuda")): the hierarchical kernels accumulate moments with hardware fp64 atomics (global_atomic_add_f64
 *     without return) and the queued launch hands a block's state over with agent-scope release/acquire — on fine-grained, managed
 *     or host-pinned memory neither is guaranteed to take effect.  The advance calls CHECK these three (hipPointerGetAttributes:
 *     host, managed and fine-grained allocations are refused with PHF_ERR_INVALID_ARGUMENT; an address the runtime cannot classify
 *     passes; the verdicts of the calling thread's last 8 (address, device) pairs are remembered and forgotten at every *_init, so a
 *     buffer re-allocated as another kind at the same address must not be swapped in between two advances of ONE sampler).  rows
 *     and sums are written with plain stores and read by nobody inside a launch: any device-accessible memory works, and they are
 *     not checked.
 */
#ifndef PYHILLFIT_AMD_H
#define PYHILLFIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHF_ABI_VERSION 7

enum {
  PHF_OK = 0,
  PHF_ERR_INVALID_ARGUMENT = -1,
  PHF_ERR_HIP = -2,
  PHF_ERR_UNSUPPORTED = -3,
  PHF_ERR_DRAINED = -4        /* a queued launch gave up waiting and drained (phf_single_level_queue_status): results are stale */
};

/* Dose-response data of P (drug, channel) pairs for the single-level (non-hierarchical) models.
 * Replaces the per-pair arrays built at python/PyHillFit.py:661-683 (concs, responses, the three boolean
 * masks and pi_bit).  A pair is a list of ENTRIES (ln concentration, response, weight), stored masked-group by
 * masked-group: first the k_other entries with 0 < y < 100 (where_r_other), then the k_zero entries with y == 0
 * (where_r_0), then the k_hundred entries with y == 100 (where_r_100); points outside [0,100] belong to no mask
 * and are dropped, exactly as the reference ignores them.
 * An entry stands for `weight` data points measured at the same concentration (the Crumb set repeats 4 doses
 * over 3-6 experiments).  The likelihood sums of doseresponse.py:244-247 regroup exactly —
 *     sum_j (y_j - pred)^2 = sum_j (y_j - ybar)^2 + w (ybar - pred)^2,      sum_j log Phi(z(pred)) = w log Phi(z(pred))
 * — so an uncensored entry carries the MEAN response of its points, and extra[1] the theta-independent within-group
 * sum of squares.  One entry per point (weight 1, extra[1] = 0) is equally valid; merging only saves arithmetic. */
typedef struct phf_points {
  int32_t num_pairs;        /* P */
  int32_t stride;           /* doubles per pair row in ln_conc/response/weight (>= k_other+k_zero+k_hundred) */
  const double* ln_conc;    /* device [P][stride]  natural log of the dose in uM (-inf for dose 0) */
  const double* response;   /* device [P][stride]  percent inhibition (mean over the entry's points) */
  const double* weight;     /* device [P][stride]  number of data points the entry stands for */
  const int32_t* counts;    /* device [P][4]       k_other, k_zero, k_hundred entries, n_total data rows (incl. dropped points) */
  const double* pi_bit;     /* device [P]          0.5*n_total*ln(2 pi), python/doseresponse.py:299-301 */
  const double* extra;      /* device [P][2]       number of uncensored POINTS (sum of their weights; the n ln sigma term,
                                                   doseresponse.py:246), within-group sum of squares of the merged points */
} phf_points;

/* The batch of independent Markov chains one call advances: Q problems x C chains.
 * A problem = (pair, temperature): python/PyHillFit.py runs one per pair at temperature 1
 * (:57,978-981); python/PyHillTemp.py runs one per rung of the ladder (:151-161).
 * problem_id / chain_id_base feed the Philox counter so that a chain's random stream is the same whichever
 * GPU or launch runs it.                                                                                      */
typedef struct phf_problems {
  int32_t num_problems;          /* Q */
  int32_t chains_per_problem;    /* C */
  const int32_t* pair_index;     /* device [Q]  row of phf_points */
  const double* temperature;     /* device [Q]  power the likelihood is raised to */
  const uint32_t* problem_id;    /* device [Q]  global problem number (Philox counter word 1) */
  uint32_t chain_id_base;        /* global number of local chain 0 (Philox counter word 0) */
  uint32_t kernel_hint;          /* hierarchical launches (ABI 5; was `reserved`; single-level: 0, bit 4 or bit 5 (phf_single_level_last_kernel)): which kernel THIS launch
                                    should get — bits 0-1 lanes per chain (1 | 2), bits 2-3 register build of the two-lane kernel (1 | 2
                                    wavefronts per SIMD), bit 4 (ABI 6) = 1: not the gfx950 assembly build of the iteration (A/B timing,
                                    bit-identity tests), bit 6 (ABI 7) = 1: the queue workspace of phf_hierarchical_advance_queued holds
                                    phf_hierarchical_queue_words() words; 0 = the library decides from the launch size.  A host that runs several
                                    groups side by side sets it per launch (the groups together fill the chip although each alone
                                    would not); a process-wide policy (phf_hierarchical_set_kernel_policy, PHF_HIER_LANES / _WPS) overrides it.
                                    Every choice gives the same numbers bit for bit. */
  const int32_t* launch_order;   /* device [Q] or NULL: a permutation of 0..Q-1 — the order in which the problems' wavefronts are
                                    handed to the GPU (ABI 3).  Results do not depend on it (every problem writes its own rows and
                                    state); what does is the tail of a launch whose problems differ in cost: pairs have 2..8 entries,
                                    an iteration 550..1 500 instructions, and most expensive first (what the reference's pool gets by
                                    luck or not) took 6.5 % off the full-Crumb-set launch.  NULL = 0, 1, 2, ... */
  const uint32_t* chain_offset;  /* device [Q] or NULL (ABI 4): global number of problem q's local chain 0 is chain_id_base +
                                    chain_offset[q].  Lets a launch hold ANY subset of a batch's (problem, 64-chain block) units — a
                                    rank's share when a batch is split over GPUs by blocks rather than by whole pairs
                                    (pyhillfit_amd/distributed.py:shard_blocks) — with every chain keeping its Philox stream. */
} phf_problems;

/* Adaptive-Metropolis schedule: python/PyHillFit.py:787-848 / python/PyHillTemp.py:76-123. */
typedef struct phf_mh_config {
  int32_t model;                 /* 1: (pIC50, sigma), Hill fixed to 1;  2: (pIC50, Hill, sigma); doseresponse.py:250-279 */
  int32_t thinning;              /* save every thinning-th iteration (-t) */
  int64_t adapt_start;           /* when_to_adapt: 1000*d (PyHillFit.py:787, PyHillTemp.py:83) */
  int32_t reset_mean_at_adapt_start; /* PyHillTemp.py:114-115 */
  int32_t reserved;
  uint64_t seed;                 /* Philox key (the reference seeds numpy with 25: PyHillFit.py:824-825) */
  const double* gamma;           /* device [>= max(1, t_end-adapt_start+1)]  gamma[s] = 1/(s+1)**0.6 (PyHillFit.py:841-842);
                                    never NULL: gamma[0] (any finite value) is read before the adaptation starts */
} phf_mh_config;

int phf_version(void);
const char* phf_last_error(void);

/* SIMDs of the current HIP device (4 per compute unit on CDNA: 1 024 on an MI355X in SPX mode, fewer on a partitioned one), looked up
 * once per device (ABI 4).  The launchers below decide with it which register build a launch gets (one wavefront per SIMD: the whole
 * register file) and whether a queued launch can pay; exported so that a host never has to hard-code the chip. */
int phf_simd_count(void);

/* doubles of per-chain state for the single-level sampler: theta[d], log-target, mean[d], cov[d(d+1)/2]
 * (packed lower triangle, row-major), loga, accepted-count, untempered log-likelihood of the current state
 * ->  2d + d(d+1)/2 + 4.                                                                                     */
int phf_single_level_state_size(int model);

/* Start Q*C chains.  Replaces python/PyHillFit.py:748-751,789,796-798,814 (PyHillTemp.py:63-80):
 *   theta = mean = theta0;  cov = cov_scale*diag(|theta0|) (cov_identity == 0; PyHillFit 0.05) or
 *   cov_scale*I (cov_identity != 0; PyHillTemp);  log-target of theta0;  loga = 0;  acceptance = 0.
 *   theta0    device [d][Q*C]
 *   state     device [S][Q*C]   (S = phf_single_level_state_size)
 *   row0      device [Q][d+1][C] or NULL: receives chain row 0 = (theta0, log-target)                      */
int phf_single_level_init(const phf_points* pts, const phf_problems* prob, int model, int cov_identity,
                          double cov_scale, const double* theta0, double* state, double* row0, void* stream);

/* Run MH iterations t_begin+1 .. t_end for every chain, in lock-step.  Replaces the loop
 * python/PyHillFit.py:830-856 (python/PyHillTemp.py:87-123).
 *   state     device [S][Q*C]   read, advanced, written back (so calls can be chained; also the checkpoint)
 *   rows      device [R][Q][d+1][C] or NULL, R = t_end/thinning - t_begin/thinning: the saved samples
 *             (theta, log-target) of iterations t with t % thinning == 0, in order (PyHillFit.py:847-848)
 *   moments   device [2(d+1)+1][Q*C] or NULL: running sums of x and x*x (x = theta, log-target) over the saved samples
 *             with t > moments_after (on-device replacement for reading the chain file back to get posterior
 *             means/variances); last row: sum of log_data_likelihood(theta, t = 1) over the same samples — the
 *             expectation python/compute_bayes_factors.py:11-27 needs per temperature rung (thermodynamic
 *             integration), at no extra cost.  Accumulated into, never zeroed.                               */
int phf_single_level_advance(const phf_points* pts, const phf_problems* prob, const phf_mh_config* cfg,
                             int64_t t_begin, int64_t t_end, double* state, double* rows,
                             double* moments, int64_t moments_after, void* stream);

/* Which kernel the calling thread's last phf_single_level_advance[_queued] launched (ABI 6; 0 = none yet): 1 = mh_advance_kernel (hipcc),
 * 2 = phf_sl3_advance, the hand-allocated gfx950 build of the model-2 iteration — OPT-IN (phf_problems.kernel_hint bit 5, or PHF_SL_ISA=1
 * in the environment; bit 4 vetoes it): launches without moments, more than one wavefront per SIMD, at most 32 entries per pair; same
 * numbers bit for bit, not faster than the hipcc kernel (why: DESIGN.md section 3) —, 3 = the same as a work queue. */
int phf_single_level_last_kernel(void);

/* The same advance as a WORK QUEUE inside one launch (ABI 3).  The launch is cut into quanta of `quantum` iterations; the grid is
 * only as large as the chip holds at once and its wavefronts pull (quantum, block) tasks — quantum-major, blocks in
 * prob->launch_order — from a counter, a block's quanta chaining through its state in HBM with agent-scope release/acquire.
 * Results are identical to phf_single_level_advance (same chains, rows, state, moments); what changes is the tail of a launch
 * whose problems differ in cost (the reference's unit of work, one pair, varies 2.5x in cost over the Crumb set; its process pool
 * balances that dynamically too, python/PyHillFit.py:997-1003).  Falls back to the plain launch when queueing cannot pay
 * (fewer blocks than the chip's 2 x phf_simd_count() wavefront slots, more than 16 rounds of them — the tail is then negligible —,
 * or fewer than two quanta).
 *   queue   device int32 [2 + Q * ceil(C/64)] workspace owned by the caller, zeroed by the caller when it is allocated.  Words
 *           0 .. Q*ceil(C/64) (task counter, per-block progress) are zeroed here, on the stream, before every launch; the LAST word
 *           is a sticky fault flag (ABI 4) the library only ever sets: a wavefront whose wait for a predecessor quantum does not
 *           end (cannot happen in a correct run) sets it, poisons the counter so that the launch drains, and returns WITHOUT
 *           advancing its block — the call has long returned PHF_OK by then, so check phf_single_level_queue_status() wherever the
 *           host synchronises anyway. */
int phf_single_level_advance_queued(const phf_points* pts, const phf_problems* prob, const phf_mh_config* cfg,
                                    int64_t t_begin, int64_t t_end, double* state, double* rows, double* moments,
                                    int64_t moments_after, int32_t quantum, int32_t* queue, void* stream);

/* Synchronise `stream` and read the sticky fault flag of a queue workspace (ABI 4): PHF_OK, or PHF_ERR_DRAINED if any queued launch
 * on it drained — states, rows and moments written since the flag was last zero are then stale and must be discarded.
 *   num_blocks = Q * ceil(C/64), as for the launches that used the workspace. */
int phf_single_level_queue_status(const int32_t* queue, int64_t num_blocks, void* stream);

/* Batch evaluation of the single-level log-likelihood and log-prior at M parameter vectors.
 * Replaces calls of dr.log_data_likelihood / dr.log_priors / dr.log_target
 * (python/doseresponse.py:187-189,203-248,166-184), e.g. the Bayes-factor sweep
 * python/compute_bayes_factors.py:18-21.
 *   pair_index  device [M], temperature device [M], theta device [d][M]
 *   out_lik, out_prior  device [M] (either may be NULL);  log_target = out_lik + out_prior                   */
int phf_single_level_log_target(const phf_points* pts, int model, int64_t m, const int32_t* pair_index,
                                const double* temperature, const double* theta, double* out_lik,
                                double* out_prior, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Hierarchical model (python/PyHillFit.py --hierarchical: log_target_distribution :113-193, loop :429-511).
 * theta = [alpha, beta, mu, s, pIC50_1, Hill_1, ..., pIC50_Ne, Hill_Ne, sigma], dim = 5 + 2 Ne (:178-181).
 * One call handles problems whose pairs all have the same number of experiments Ne (1 <= Ne <= PHF_HIER_MAX_EXPTS);
 * the host groups the pairs by Ne (Crumb: Ne = 3..6).  Ne <= PHF_HIER_FAST_EXPTS runs kernels compiled per Ne (state in
 * registers, proposal factor in LDS) — one lane per chain, or, for Ne = 3..6 when the launch is small enough to give every
 * wavefront a SIMD of its own, TWO lanes per chain (half the instructions per iteration; same results bit for bit;
 * phf_hierarchical_set_kernel_policy forces one or the other) —; larger Ne (the reference's synthetic set has Ne = 50,
 * dim 105) runs one WAVEFRONT per chain with the whole state in LDS: lanes take an experiment each in the target and a row
 * each in the factor.                                                                                            */
#define PHF_HIER_MAX_EXPTS 64
#define PHF_HIER_FAST_EXPTS 8

/* Points of P pairs, stored experiment by experiment (python/doseresponse.py:60-67 keeps one array per experiment). */
typedef struct phf_hier_points {
  int32_t num_pairs;          /* P */
  int32_t stride;             /* doubles per pair row in ln_conc/response */
  int32_t n_expts;            /* Ne, the same for every pair of this set */
  int32_t points_per_expt;    /* ABI 6 (was `reserved`): the point SHAPE of every pair of this set, the caller's statement about its data (like
                                 n_expts; expt_start must say the same): per | last << 4 — EVERY experiment of EVERY pair has `per` points
                                 (1..15), except that the last one has `last` (1..15) if those bits are not 0; so n > 0 alone = n points in
                                 every experiment; or (ABI 7) bit 30 + a nibble per experiment, experiment 0 in the lowest (up to 7 experiments of
                                 1..15 points: 4 + 4 + 4 + 1 + 1 = 0x40011444); 0 = the pairs differ, the shape has no such code, or unknown.  (PHF_HIER_SHAPE of
                                 pyhillfit_amd/csrc/phf_hier_model.h.)  Launches that get one lane per chain run the hand-allocated gfx950
                                 build of the iteration (two wavefronts per SIMD: same numbers) where the library has one for (n_expts,
                                 shape): n_expts == 3 with 4 + 4 + 4 points (147 of the Crumb set's 210 pairs), 2 + 2 + 2 (6), 5 + 5 + 4 (1);
                                 ABI 7: every n_expts == 4 shape of the Crumb set (4 + 4 + 4 + 1 (32 pairs), + 2 (5), + 3 (2), 2 + 2 + 2 + 1, 5 + 5 + 5 + 1) and
                                 every n_expts == 5 and 6 one (4 + 4 + 4 + 1 + 1 (5), 4 + 4 + 4 + 2 + 1 (5), 4 + 4 + 4 + 4 + 4, 5 + 5 + 4 + 2 + 2; 4 + 4 + 4 + 1 + 1 + 1 (2),
                                 4 + 4 + 4 + 4 + 2 + 1: the last two, like 2 + 2 + 2 + 1, 5 + 5 + 5 + 1 and the n_expts == 5 ones, in the fused kernel only) — through
                                 phf_hierarchical_advance_queued / _fused with a workspace of phf_hierarchical_queue_words() words and kernel_hint bit 6. */
  const double* ln_conc;      /* device [P][stride] */
  const double* response;     /* device [P][stride] */
  const int32_t* expt_start;  /* device [P][Ne+1]  first point of each experiment; [Ne] = number of points */
} phf_hier_points;

#ifndef PHF_HIER_PRIOR_DEFINED
#define PHF_HIER_PRIOR_DEFINED
/* Shifted-Gamma hyper-priors of (alpha, beta, mu, s, sigma): python/PyHillFit.py:301,340-364 */
typedef struct phf_hier_prior {
  double shape_m1[5];         /* shapes - 1 */
  double inv_scale[5];        /* 1/scales */
  double loc[5];              /* lower bounds (locs) */
} phf_hier_prior;
#endif

/* doubles of per-chain state: theta[dim], log-target, mean[dim], F[dim(dim+1)/2], loga, accepted-count.
 * F holds the adapted covariance as cov = L diag(d) L' — L unit lower triangular — in ONE packed lower triangle (row-major):
 * slot (i, j < i) = L_ij, slot (i, i) = d_i.  The factors are carried instead of the covariance: the reference's update
 * cov <- (1-g) cov + g v v' (PyHillFit.py:498-499) is applied to them as d <- (1-g) d followed by a rank-one update (Gill, Golub,
 * Murray & Saunders 1974, method C1), which is the same matrix in exact arithmetic and needs one pass over dim(dim+1)/2 numbers
 * per iteration — two fused multiply-adds per element, no square root — instead of an O(dim^3) refactorisation; the proposal
 * (PyHillFit.py:485) is theta + e^(loga/2) L sqrt(d) z.  (ABI <= 3 carried a Cholesky factor in the same slots.)          */
int phf_hierarchical_state_size(int n_expts);

/* Start chains: theta = mean = theta0, cov = diag(cov_scale*|theta0|) (PyHillFit.py:431, cov_scale 0.01), loga = 0.
 *   theta0 device [dim][Q*C];  state device [S][Q*C];  row0 device [Q][dim+1][C] or NULL.
 *   prob->temperature is ignored (the hierarchical sampler is not tempered).                                  */
int phf_hierarchical_init(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                          double cov_scale, const double* theta0, double* state, double* row0, void* stream);

/* MH iterations t_begin+1 .. t_end (python/PyHillFit.py:484-511); arguments as phf_single_level_advance with
 * d = dim; cfg->model is ignored, cfg->adapt_start is 100*dim in the reference (:440).                        */
int phf_hierarchical_advance(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                             const phf_mh_config* cfg, int64_t t_begin, int64_t t_end, double* state, double* rows,
                             double* moments, int64_t moments_after, void* stream);

/* The same advance as a WORK QUEUE inside one launch (ABI 6), as phf_single_level_advance_queued: where the launch runs the gfx950
 * assembly build (n_expts == 3, points_per_expt == 4, one lane per chain) AND has more 64-chain blocks than the chip holds wavefronts
 * (2 x phf_simd_count()) AND at least two quanta, it is cut into quanta of `quantum` iterations (0 = the library's choice: about 16 rounds
 * of tasks on the chip's wavefront slots, at least 100 iterations; rounded down to a multiple of the thinning) and a grid as large as the chip pulls (quantum, block) tasks from a counter, a block's quanta
 * chaining through its state in HBM with agent-scope release / acquire — the ragged last round of a launch becomes a round of short
 * tasks (147 pairs x 1 024 chains: 2 352 wavefronts on 2 048 slots).  Every other launch runs exactly as phf_hierarchical_advance.
 * Results are identical either way (same chains, rows, state, moments).
 *   queue   device int32 [2 + Q * ceil(C / 64)], owned by the caller, zeroed by the caller when allocated; words 0 .. Q ceil(C/64) are
 *           zeroed here, on the stream, before a queued launch; word 1 + Q ceil(C/64) is the sticky fault flag of
 *           phf_single_level_queue_status (same layout, same meaning: check it there wherever the host synchronises anyway).
 *           ABI 7: a workspace of phf_hierarchical_queue_words(pts, prob) words — which the caller states with kernel_hint bit 6 —
 *           also holds the device-memory scratch in which the gfx950 build of the Ne = 4 iteration keeps the last rows of the proposal
 *           factor (21 KB per resident wavefront, at most 44 MB); without that statement such launches run the hipcc kernels. */
/* CAUTION (profiles/r05/queue_progress_word_hazard.txt): do not run a queued launch of the gfx950 build BESIDE other launches of that build on the same
 * GPU (several groups of one run, a stream each) — a wavefront has been seen to wait for a block's progress word until it gave up (PHF_ERR_DRAINED
 * through the fault flag); several groups go through phf_hierarchical_advance_fused (one grid), or one after the other, or kernel_hint bit 4. */
int phf_hierarchical_advance_queued(const phf_hier_points* pts, const phf_problems* prob, const phf_hier_prior* prior,
                                    const phf_mh_config* cfg, int64_t t_begin, int64_t t_end, double* state, double* rows,
                                    double* moments, int64_t moments_after, int32_t quantum, int32_t* queue, void* stream);

/* int32 words the queue workspace of phf_hierarchical_advance_queued has to hold for launches of this (pts, prob) shape (ABI 7):
 * 2 + Q ceil(C / 64), plus the scratch of a gfx950 kernel that keeps part of the chain state in device memory (n_expts == 4 with
 * 4 + 4 + 4 + 1 / 2 / 3 points).  Reads only n_expts, points_per_expt, num_problems and chains_per_problem; no device call.  < 0: error. */
int64_t phf_hierarchical_queue_words(const phf_hier_points* pts, const phf_problems* prob);

/* EVERY launch group of a run through ONE persistent grid (ABI 7; phf_hier_fused_advance of the gfx950 code object: a body per (n_expts, point
 * shape), a wavefront that finishes a task of one group pulls the next task whatever group it belongs to).  Separate launches side by side —
 * one stream per group, what python/PyHillFit.py's pool amounts to (:997-1003) — leave a chip's workgroup slots to whichever persistent grid
 * got them first; one queue does not.  groups: 1..14 of them, each with a (n_expts, points_per_expt) the code object has a kernel for
 * (PHF_ERR_UNSUPPORTED otherwise: launch such groups one by one), no two alike; the same thinning in every cfg; t_begin a multiple of it.
 * Every chain's numbers are those of phf_hierarchical_advance, bit for bit.  queue: device int32 [phf_hierarchical_fused_queue_words()],
 * zeroed by the caller when allocated; word 1 + (the groups' blocks) is the sticky fault flag. */
typedef struct phf_hier_group {
  const phf_hier_points* pts;
  const phf_problems* prob;
  const phf_mh_config* cfg;      /* seed, thinning, adapt_start, gamma of THIS group (adapt_start = 100 dim differs with n_expts) */
  double* state;                 /* device [phf_hierarchical_state_size(n_expts)][Q C] */
  double* rows;                  /* device [rows][Q][dim + 1][C] or NULL */
  double* moments;               /* device [2 (dim + 1)][Q C] or NULL */
} phf_hier_group;
int64_t phf_hierarchical_fused_queue_words(int32_t n_groups, const phf_hier_group* groups);
int phf_hierarchical_advance_fused(int32_t n_groups, const phf_hier_group* groups, const phf_hier_prior* prior, int64_t t_begin, int64_t t_end,
                                   int64_t moments_after, int32_t quantum, int32_t* queue, void* stream);

/* Which kernel runs groups with Ne = 3..6, PROCESS-WIDE (ABI 4; A/B timing and the bit-identity tests): lanes 1 | 2 = one | two lanes
 * per chain, wps 1 | 2 = the register build of the two-lane kernel (512 | 256 registers); 0 = not forced: the launch's own
 * phf_problems.kernel_hint, else the launch size, decides.  The environment variables PHF_HIER_LANES / PHF_HIER_WPS give the initial
 * values and are read ONCE, here in the library, at the first use; the two words are plain atomics (any thread may set them). */
int phf_hierarchical_set_kernel_policy(int lanes, int wps);

/* Which kernel the calling thread's last phf_hierarchical_advance launched (ABI 6; 0 = none yet): the tests that compare kernels
 * against each other assert through it that the kernel they mean is the one that ran. */
#define PHF_HIER_KERNEL_ONE_LANE 1     /* hier_advance_kernel<Ne>: one lane per chain, hipcc */
#define PHF_HIER_KERNEL_TWO_LANES 2    /* hier_advance2_kernel<Ne, wps> */
#define PHF_HIER_KERNEL_WAVE 3         /* hier_wave_advance_kernel: one wavefront per chain (Ne > 8) */
#define PHF_HIER_KERNEL_GFX950_ISA 4   /* phf_hier3_advance: the hand-allocated gfx950 build (Ne = 3, four points per experiment) */
#define PHF_HIER_KERNEL_GFX950_ISA_QUEUED 5   /* ... as a work queue (phf_hierarchical_advance_queued) */
#define PHF_HIER_KERNEL_GFX950_ISA_FUSED 6    /* ... every group of a run in one persistent grid (phf_hierarchical_advance_fused, ABI 7) */
int phf_hierarchical_last_kernel(void);

/* log_target_distribution (python/PyHillFit.py:173-193) at M parameter vectors: theta device [dim][M]. */
int phf_hierarchical_log_target(const phf_hier_points* pts, const phf_hier_prior* prior, int64_t m,
                                const int32_t* pair_index, const double* theta, double* out, void* stream);

/* Evaluate one of the device elementary functions on an array (parity tests: the device must reproduce the
 * host build of pyhillfit_amd/csrc/phf_math.h bit for bit).
 * fn: 0 exp, 1 log, 2 erfcx(y>=0), 3 log_ndtr, 4 ndtr, 5 sqrt, 6 reciprocal, 7 sin(2 pi w/2^32), 8 cos(...)
 * 9 exp_fast, 10 log_fast, 11 log_ndtr_nonpos — the branch-free forms the kernels use
 * (for 7/8 the input doubles hold integer values w in [0, 2^32));
 * the MH loops' own division / square root without exponent-range handling (phf_math.h; correctly rounded for operands within
 * 2^-600..2^600): 12 phf_rcp(x), 13 phf_sqrt_pos(x), 14 phf_div(ln 10, x), 15 phf_sqrt_nonneg(x) (0 -> 0), 16 phf_div(x, ln 10);
 * 17 phf_normal_u32(w): the single-level sampler's standard normal of a 32-bit word (input doubles hold integer values w in [0, 2^32));
 * 18 phf_log_ndtr_tab(x): log Phi(x) from the censored likelihood's table (valid for -185 000 < x <= 0; any x is safe to pass);
 * 19 phf_erfc_tab(y): erfc(y), y >= 0, to 3.6e-17 absolute from the hierarchical target's table, 0 from y = 6 on (any y is safe to pass);
 * 20 / 21 phf_sqrt_rcp_pos(x): the reciprocal 1 / sqrt(x) (of the ROUNDED root) / the root itself — a Cholesky pivot and its reciprocal
 * from one hardware estimate (ABI 5; python/PyHillFit.py:831 draws through numpy's factorisation of the same covariance). */
int phf_debug_math(int fn, int64_t n, const double* in, double* out, void* stream);

/* The same for the hand-allocated gfx950 code object (ABI 6; tools/gen_hier_isa.py, tools/isa/phf_isa_math.py): every elementary
 * function of the assembly build of the hierarchical Ne = 3 iteration, evaluated on an array by a unit kernel of the same code object —
 * each must reproduce its C namesake in pyhillfit_amd/csrc/phf_math.h bit for bit (tests/test_gpu_isa.py).
 * fn: 0 phf_exp_fast_k, 1 phf_exp_capped_k, 2 phf_log_pos_k, 3 phf_log_fast_k, 4 phf_erfc_tab, 5 phf_rcp, 6 phf_sqrt_nonneg (in, out:
 * device double [n]); 7 phf_normal_u32, 8 phf_log_pos_k(phf_unit_open32(w)) (in: device uint32 [n], out: device double [n]);
 * 9 Philox4x32-7 (in: device uint32 [n][6] = counter words 0..3, key words 0..1 — the KEY OF ELEMENT 0 is used for the whole call,
 * as the kernels advance the key schedule on the scalar unit; out: device uint32 [n][4]). */
int phf_debug_isa(int fn, int64_t n, const void* in, void* out, void* stream);

/* The four Philox4x32-R words of n (counter, key) tuples: in device uint32 [n][6], out device uint32 [n][4].
 * phf_debug_philox: R = the rounds the samplers draw with, phf_philox_rounds() (7 since ABI 5; rounds 1-3 of this build: 10);
 * phf_debug_philox_rounds: R = 7 or 10 (both are held to the Random123 known-answer vectors).  Plays the role of the reference's
 * numpy RandomState (python/PyHillFit.py:825,831,834; python/PyHillTemp.py:88,100). */
int phf_philox_rounds(void);
int phf_debug_philox(int64_t n, const uint32_t* counter_key, uint32_t* out, void* stream);
int phf_debug_philox_rounds(int rounds, int64_t n, const uint32_t* counter_key, uint32_t* out, void* stream);

/* ---- posterior-predictive curves -------------------------------------------------------------------
 * Replaces construct_posterior_predictive_cdfs (python/construct_hierarchical_cdfs.py:32-58): for every problem q
 * and every hierarchical sample (alpha, beta, mu, s) add fisk.cdf/pdf(hill_x; c=beta, scale=alpha) and
 * logistic.cdf/pdf(pic50_x; mu, s) into running sums; the caller divides by the number of samples (:54-57).
 *   rows     device [num_rows][num_problems][row_stride][num_chains] — the row buffer phf_hierarchical_advance wrote
 *            (row_stride = dim+1; columns 0..3 = alpha, beta, mu, s), or any tensor of that shape (a chain read back
 *            from a reference-format file: row_stride 4, num_chains 1).  Chains 0..chains_used-1 of every row are used.
 *   hill_x, pic50_x  device [grid_points]  (the reference: 501 points on [0,4] and [-2,12], :33-39)
 *   chunk    samples per partial sum (fixes the order of the additions; 4096 is a good value)
 *   sums     device [num_problems][4][grid_points], curves in the order hill cdf, pic50 cdf, hill pdf, pic50 pdf;
 *            ADDED to (zero it before the first call; call once per segment of rows)
 *   scratch  device, at least phf_predictive_scratch_bytes(num_problems, num_rows*chains_used, grid_points, chunk) */
size_t phf_predictive_scratch_bytes(int num_problems, int64_t samples_per_problem, int grid_points, int chunk);
int phf_predictive_accumulate(int num_problems, const double* rows, int64_t num_rows, int row_stride, int num_chains,
                              int chains_used, int grid_points, const double* hill_x, const double* pic50_x, int chunk,
                              double* sums, double* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYHILLFIT_AMD_H */
