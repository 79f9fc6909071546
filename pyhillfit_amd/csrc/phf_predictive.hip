// phf_predictive.hip — posterior-predictive CDFs/PDFs of Hill and pIC50 from hierarchical samples
// (python/construct_hierarchical_cdfs.py:32-58: a Python loop over ~75 000 samples with four scipy.stats calls on a
// 501-point grid each, ~30 s per pair).
//
// Mapping: one lane = TWO grid points of each axis (a 256-thread block covers 512 >= 501 points), samples are the
// loop.  A block takes one chunk of samples of one pair: it turns them into (ln alpha, beta, mu, 1/s) once,
// 512 at a time, in LDS (every lane then reads the same address: a broadcast, no bank conflict) and keeps its
// 8 running sums in registers.  Per (sample, grid point) the cost is two exponentials and a quarter of a division;
// the kernel is fp64-VALU bound like the samplers.  Chunk sums go to a scratch buffer and are added in chunk order by
// a second kernel, so the result does not depend on the launch shape and the CPU twin reproduces it bit for bit.
// The samples are read in place from the hierarchical sampler's row buffer [rows][pairs][dim+1][chains]
// (columns 0..3 = alpha, beta, mu, s), chain fastest: coalesced, and no copy of the chain is ever made.
#include <hip/hip_runtime.h>

#include "../../include/pyhillfit_amd.h"
#include "phf_common.h"
#include "phf_predictive_model.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPointsPerBlock = 2 * kThreads;

struct PredArgs {
  const double* rows;
  int64_t num_rows;
  int32_t num_problems, row_stride, num_chains, chains_used;
  int64_t samples;                 // num_rows * chains_used, per problem
  int32_t grid_points, chunk, num_chunks;
  const double* hill_x;
  const double* pic50_x;
  double* partial;                 // [problem][chunk][curve][grid point]
  double* sums;                    // [problem][curve][grid point]
};

__global__ __launch_bounds__(kThreads) void pred_partial_kernel(const PredArgs a) {
  PHF_MATH_TABLES_TO_LDS();
  __shared__ double s_par[4][PHF_PRED_TILE];
  const int q = blockIdx.z, tile = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  PHF_KFETCH_V(k_exp, phf_k_exp, PHF_K_EXP_N);

  double lnx[2], inv_x[2], px[2];
  int g[2];
  PHF_UNROLL
  for (int j = 0; j < 2; ++j) {
    g[j] = tile * kPointsPerBlock + j * kThreads + tid;
    const int gg = g[j] < a.grid_points ? g[j] : a.grid_points - 1;       // lanes past the grid repeat its last point
    phf_pred_hill_axis(a.hill_x[gg], &lnx[j], &inv_x[j]);
    px[j] = a.pic50_x[gg];
  }
  double acc[2][PHF_PRED_CURVES] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};

  const int64_t m_begin = (int64_t)chunk * a.chunk;
  const int64_t m_end = m_begin + a.chunk < a.samples ? m_begin + a.chunk : a.samples;
  const size_t col = (size_t)a.num_chains;
  for (int64_t m0 = m_begin; m0 < m_end; m0 += PHF_PRED_TILE) {
    const int n = (int)(m_end - m0 < PHF_PRED_TILE ? m_end - m0 : PHF_PRED_TILE);
    __syncthreads();                                                       // previous tile fully consumed
    for (int i = tid; i < n; i += kThreads) {
      const int64_t m = m0 + i;
      const int64_t r = m / a.chains_used;
      const int c = (int)(m - r * a.chains_used);
      const double* p = a.rows + (((size_t)r * a.num_problems + q) * a.row_stride) * col + c;
      phf_pred_prepare(p[0], p[col], p[2 * col], p[3 * col], &s_par[0][i], &s_par[1][i], &s_par[2][i], &s_par[3][i]);
    }
    __syncthreads();
    for (int i = 0; i < n; ++i)
      phf_pred_accumulate2(s_par[0][i], s_par[1][i], s_par[2][i], s_par[3][i], lnx, inv_x, px, k_exp, acc);
  }
  PHF_UNROLL
  for (int j = 0; j < 2; ++j) {
    if (g[j] >= a.grid_points) continue;
    double* out = a.partial + (((size_t)q * a.num_chunks + chunk) * PHF_PRED_CURVES) * a.grid_points + g[j];
    PHF_UNROLL
    for (int f = 0; f < PHF_PRED_CURVES; ++f) out[(size_t)f * a.grid_points] = acc[j][f];
  }
}

// sums[q][f][g] += partial[q][0][f][g] + partial[q][1][f][g] + ... (left to right); Hill curves are 0 at x <= 0
__global__ __launch_bounds__(kThreads) void pred_reduce_kernel(const PredArgs a) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t per_q = (int64_t)PHF_PRED_CURVES * a.grid_points;
  if (i >= per_q * a.num_problems) return;
  const int q = (int)(i / per_q);
  const int f = (int)((i - q * per_q) / a.grid_points);
  const int g = (int)(i - q * per_q - (int64_t)f * a.grid_points);
  double s = a.sums[i];
  const double* p = a.partial + ((size_t)q * a.num_chunks * PHF_PRED_CURVES + f) * a.grid_points + g;
  for (int c = 0; c < a.num_chunks; ++c) s += p[(size_t)c * per_q];
  if ((f == 0 || f == 2) && !(a.hill_x[g] > 0.0)) s = 0.0;
  a.sums[i] = s;
}

int64_t chunks_of(int64_t samples, int chunk) { return (samples + chunk - 1) / chunk; }

}  // namespace

extern "C" size_t phf_predictive_scratch_bytes(int num_problems, int64_t samples_per_problem, int grid_points, int chunk) {
  if (num_problems <= 0 || samples_per_problem <= 0 || grid_points <= 0 || chunk <= 0) return 0;
  return (size_t)num_problems * (size_t)chunks_of(samples_per_problem, chunk) * PHF_PRED_CURVES * (size_t)grid_points * sizeof(double);
}

extern "C" int phf_predictive_accumulate(int num_problems, const double* rows, int64_t num_rows, int row_stride,
                                         int num_chains, int chains_used, int grid_points, const double* hill_x,
                                         const double* pic50_x, int chunk, double* sums, double* scratch,
                                         size_t scratch_bytes, void* stream) {
  if (num_problems < 0 || num_rows < 0) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: negative size");
  if (num_problems == 0 || num_rows == 0) return PHF_OK;
  if (!rows || !hill_x || !pic50_x || !sums || !scratch) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: null pointer");
  if (row_stride < 4) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: a row holds alpha, beta, mu, s in its first 4 columns");
  if (num_chains < 1 || chains_used < 1 || chains_used > num_chains)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: need 1 <= chains_used <= num_chains");
  if (grid_points < 1 || chunk < 1) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: grid_points and chunk must be positive");
  PredArgs a;
  a.rows = rows; a.num_rows = num_rows; a.num_problems = num_problems; a.row_stride = row_stride;
  a.num_chains = num_chains; a.chains_used = chains_used; a.samples = num_rows * chains_used;
  a.grid_points = grid_points; a.chunk = chunk;
  const int64_t nchunks = chunks_of(a.samples, chunk);
  const int tiles = (grid_points + kPointsPerBlock - 1) / kPointsPerBlock;
  if (nchunks > 0x7fffffff || tiles > 65535 || num_problems > 65535)
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: launch grid too large (use a larger chunk / fewer problems per call)");
  a.num_chunks = (int32_t)nchunks;
  if (scratch_bytes < phf_predictive_scratch_bytes(num_problems, a.samples, grid_points, chunk))
    return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_predictive_accumulate: scratch smaller than phf_predictive_scratch_bytes()");
  a.hill_x = hill_x; a.pic50_x = pic50_x; a.partial = scratch; a.sums = sums;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pred_partial_kernel, dim3((unsigned)nchunks, tiles, num_problems), dim3(kThreads), 0, s, a);
  int rc = phf_check_launch("pred_partial_kernel");
  if (rc != PHF_OK) return rc;
  const int64_t outs = (int64_t)num_problems * PHF_PRED_CURVES * grid_points;
  hipLaunchKernelGGL(pred_reduce_kernel, dim3((unsigned)((outs + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
  return phf_check_launch("pred_reduce_kernel");
}
