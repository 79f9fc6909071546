// microbench.hip — instruction issue/latency probes for the fp64 VALU roof of the MH kernels (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench.hip -o /tmp/microbench ; run on the GPU box.
// Each probe runs REPS x UNROLL copies of an instruction pattern per wave and reports shader cycles per
// instruction per wave (s_memtime), at 1, 2 and 4 waves per SIMD (blocks of 64 threads, 256 CUs x 4 SIMDs).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int REPS = 2000;

template <int KIND>
__global__ __launch_bounds__(64) void probe(double* out, unsigned long long* cyc, double seed) {
  double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 0.999999, c = 1e-7;
  unsigned int u0 = threadIdx.x + 12345u, u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < REPS; ++r) {
    if (KIND == 0) {  // 8 dependent fma (one chain)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
    } else if (KIND == 1) {  // 2 chains x 4
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
      }
    } else if (KIND == 2) {  // 4 chains x 2
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
      }
    } else if (KIND == 3) {  // 8 independent chains
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a5) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a6) : "v"(m), "v"(c));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a7) : "v"(m), "v"(c));
    } else if (KIND == 4) {  // 4 fma + 4 s_mov interleaved (does SALU steal VALU issue slots of the same wave?)
      unsigned int s;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        asm volatile("s_mov_b32 %0, 0x12345678" : "=s"(s));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
        asm volatile("s_mov_b32 %0, 0x12345678" : "=s"(s));
      }
    } else if (KIND == 5) {  // 8 v_mul_f64 independent
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(m));
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(m)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(m));
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(m)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(m));
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(m)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(m));
    } else if (KIND == 6) {  // 8 v_add_f64 independent
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(c));
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(c));
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "v"(c));
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a7) : "v"(c));
    } else if (KIND == 7) {  // 8 v_rcp_f64 independent
      asm volatile("v_rcp_f64 %0, %0" : "+v"(a0)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a1));
      asm volatile("v_rcp_f64 %0, %0" : "+v"(a2)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a3));
      asm volatile("v_rcp_f64 %0, %0" : "+v"(a4)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a5));
      asm volatile("v_rcp_f64 %0, %0" : "+v"(a6)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a7));
    } else if (KIND == 8) {  // 8 v_sqrt_f64
      asm volatile("v_sqrt_f64 %0, %0" : "+v"(a0)); asm volatile("v_sqrt_f64 %0, %0" : "+v"(a1));
      asm volatile("v_sqrt_f64 %0, %0" : "+v"(a2)); asm volatile("v_sqrt_f64 %0, %0" : "+v"(a3));
      asm volatile("v_sqrt_f64 %0, %0" : "+v"(a4)); asm volatile("v_sqrt_f64 %0, %0" : "+v"(a5));
      asm volatile("v_sqrt_f64 %0, %0" : "+v"(a6)); asm volatile("v_sqrt_f64 %0, %0" : "+v"(a7));
    } else if (KIND == 9) {  // 8 v_mul_hi_u32 (4 chains)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u0) : "v"(0xD2511F53u)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u1) : "v"(0xD2511F53u));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u2) : "v"(0xD2511F53u)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u3) : "v"(0xD2511F53u));
      }
    } else if (KIND == 10) {  // 8 v_mul_lo_u32
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(0xD2511F53u)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u1) : "v"(0xD2511F53u));
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u2) : "v"(0xD2511F53u)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u3) : "v"(0xD2511F53u));
      }
    } else if (KIND == 11) {  // 8 v_mad_u64_u32 (full 32x32->64 multiply-add)
      unsigned long long w0 = u0, w1 = u1, w2 = u2, w3 = u3;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w0) : "v"(u0), "v"(0xD2511F53u) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w1) : "v"(u1), "v"(0xD2511F53u) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w2) : "v"(u2), "v"(0xD2511F53u) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w3) : "v"(u3), "v"(0xD2511F53u) : "vcc");
      }
      u0 ^= (unsigned)w0; u1 ^= (unsigned)(w1 >> 32); u2 ^= (unsigned)w2; u3 ^= (unsigned)(w3 >> 32);
    } else if (KIND == 12) {  // 8 v_xor_b32 (fp32-rate int op)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u1) : "v"(u2));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u2) : "v"(u3)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u3) : "v"(u0));
      }
    } else if (KIND == 13) {  // 8 v_mov_b64
      double t;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mov_b64 %0, %1" : "=v"(t) : "v"(a0));
      a1 += t;
    } else if (KIND == 14) {  // 8 v_cndmask_b32
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u0) : "v"(u1)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u1) : "v"(u2));
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u2) : "v"(u3)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u3) : "v"(u0));
      }
    } else if (KIND == 15) {  // 8 v_ldexp_f64
      asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a0)); asm volatile("v_ldexp_f64 %0, %0, -1" : "+v"(a0));
      asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a1)); asm volatile("v_ldexp_f64 %0, %0, -1" : "+v"(a1));
      asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a2)); asm volatile("v_ldexp_f64 %0, %0, -1" : "+v"(a2));
      asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a3)); asm volatile("v_ldexp_f64 %0, %0, -1" : "+v"(a3));
    } else if (KIND == 16) {  // 8 fma with SGPR-held constant operand (constant bus)
      double sc;
      asm volatile("s_mov_b64 %0, 0x3ff0000000000000" : "=s"(sc));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "s"(sc)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "s"(sc));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "s"(sc)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "s"(sc));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(m), "s"(sc)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a5) : "v"(m), "s"(sc));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a6) : "v"(m), "s"(sc)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a7) : "v"(m), "s"(sc));
    } else if (KIND == 17) {  // 8 v_fmac_f64 (VOP2) independent
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a4) : "v"(m), "v"(c)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a5) : "v"(m), "v"(c));
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a6) : "v"(m), "v"(c)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a7) : "v"(m), "v"(c));
    } else if (KIND == 18) {  // 8 v_cvt_f64_u32 / v_rndne mix
      asm volatile("v_rndne_f64 %0, %0" : "+v"(a0)); asm volatile("v_rndne_f64 %0, %0" : "+v"(a1));
      asm volatile("v_rndne_f64 %0, %0" : "+v"(a2)); asm volatile("v_rndne_f64 %0, %0" : "+v"(a3));
      asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a4) : "v"(u0)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a5) : "v"(u1));
      asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a6) : "v"(u2)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a7) : "v"(u3));

    } else if (KIND == 19) {  // 8 independent v_cndmask_b32 (VOP2, vcc)
      unsigned int r0, r1, r2, r3;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r0) : "v"(u0), "v"(u1)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r1) : "v"(u1), "v"(u2));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r2) : "v"(u2), "v"(u3)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r3) : "v"(u3), "v"(u0));
      }
      u0 ^= r0 ^ r1 ^ r2 ^ r3;
    } else if (KIND == 20) {  // 8 v_cmp_lt_f64 (writes vcc)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc");
    } else if (KIND == 21) {  // 8 v_max_f64 independent
      asm volatile("v_max_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(a1) : "v"(c));
      asm volatile("v_max_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(a3) : "v"(c));
      asm volatile("v_max_f64 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(a5) : "v"(c));
      asm volatile("v_max_f64 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(a7) : "v"(c));
    } else if (KIND == 22) {  // cmp + 2 cndmask (a 64-bit select as the compiler emits it) x 2, + 2 more cmp
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc");
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u0) : "v"(u1)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u2) : "v"(u3));
        asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a2), "v"(a3) : "vcc");
      }
    } else if (KIND == 23) {  // 8 v_and_b32 dependent ring
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(u1) : "v"(u2));
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(u2) : "v"(u3)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(u3) : "v"(u0));
      }
    } else if (KIND == 24) {  // full IEEE division a/b as the compiler expands it, 2 independent per rep (counted as 8 "instr")
      a0 = m / a0; a1 = m / a1; a2 = c / a2; a3 = c / a3; a4 = m / a4; a5 = m / a5; a6 = c / a6; a7 = c / a7;
    } else if (KIND == 25) {  // 8 v_mov_b32 literal
      unsigned int r0;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mov_b32 %0, 0x12345678" : "=v"(r0));
      u0 ^= r0;
    } else if (KIND == 26) {  // 8 s_mov_b32 only
      unsigned int s;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("s_mov_b32 %0, 0x12345678" : "=s"(s));
    } else if (KIND == 27) {  // 8 v_cndmask_b32 e64 with SGPR-pair mask
      unsigned long long msk = 0x5555555555555555ull;
      unsigned int r0, r1, r2, r3;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r0) : "v"(u0), "v"(u1), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r1) : "v"(u1), "v"(u2), "s"(msk));
        asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r2) : "v"(u2), "v"(u3), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r3) : "v"(u3), "v"(u0), "s"(msk));
      }
      u0 ^= r0 ^ r1 ^ r2 ^ r3;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 ^ u1 ^ u2 ^ u3);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, double* out, unsigned long long* cyc) {
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    // cycles per instruction per wave (s_memtime ticks at 100 MHz on some parts: report wall-based too)
    const double ninstr = 8.0 * REPS;
    printf("  w/simd=%d: %6.2f tick/instr, wall %6.2f ns/instr/wave-slot", wps, mean / ninstr, ms * 1e6 / ninstr);
  }
  printf("\n");
}

int main() {
  double* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(double)));
  CHECK(hipMalloc(&cyc, 256 * 4 * 8 * sizeof(unsigned long long)));
  run<0>("fma_f64 1 dependent chain", out, cyc);
  run<1>("fma_f64 2 chains", out, cyc);
  run<2>("fma_f64 4 chains", out, cyc);
  run<3>("fma_f64 8 chains", out, cyc);
  run<4>("fma_f64 2 chains + s_mov each", out, cyc);
  run<5>("mul_f64 8 indep", out, cyc);
  run<6>("add_f64 8 indep", out, cyc);
  run<7>("rcp_f64 8 indep", out, cyc);
  run<8>("sqrt_f64 8 indep", out, cyc);
  run<9>("mul_hi_u32 4 chains", out, cyc);
  run<10>("mul_lo_u32 4 chains", out, cyc);
  run<11>("mad_u64_u32 4 indep", out, cyc);
  run<12>("xor_b32 dependent ring", out, cyc);
  run<13>("mov_b64 8", out, cyc);
  run<14>("cndmask_b32 ring", out, cyc);
  run<15>("ldexp_f64 4 chains", out, cyc);
  run<16>("fma_f64 8 chains, SGPR operand", out, cyc);
  run<17>("fmac_f64 8 chains", out, cyc);
  run<18>("rndne_f64 x4 + cvt_f64_u32 x4", out, cyc);
  run<19>("cndmask_b32 vcc, 8 indep", out, cyc);
  run<20>("cmp_lt_f64 x8", out, cyc);
  run<21>("max_f64 8 indep", out, cyc);
  run<22>("cmp,cnd,cnd,cmp pattern", out, cyc);
  run<23>("and_b32 ring", out, cyc);
  run<24>("IEEE f64 division (8 per rep)", out, cyc);
  run<25>("v_mov_b32 literal x8", out, cyc);
  run<26>("s_mov_b32 x8", out, cyc);
  run<27>("cndmask_b32 e64 sgpr mask, 8 indep", out, cyc);
  return 0;
}
