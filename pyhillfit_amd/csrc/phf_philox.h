/* phf_philox.h — Philox4x32-R counter-based generator (Salmon et al., SC'11), one block of
 * four 32-bit words per (counter, key); the number of rounds R is a literal at the call site.  The samplers draw with
 * PHF_PHILOX_ROUNDS = 7 (round 4): Philox4x32-7 is the fewest rounds that pass BigCrush ("Crush-resistant", Salmon et al. table 2);
 * the 10 of the default are a safety margin a Metropolis proposal does not need, and three rounds are 6 % of the generator-bound
 * part of an iteration.  Ten rounds stay available (phf_philox4x32_10, -DPHF_PHILOX_ROUNDS=10).
 * Stateless: the MH kernels address the stream as
 *   counter = (chain index within problem, global problem id, MH iteration, draw block)
 *   key     = (seed low word, seed high word)
 * so a chain's random numbers do not depend on which GPU, block or launch segment runs it
 * (needed for the 1/2/4/8-GPU parity check and for resume).  Plays the role of the
 * reference's global numpy RandomState (python/PyHillFit.py:825,831,834).
 * Checked against the Random123 known-answer vectors (kat_vectors: philox4x32 7 and philox4x32 10) in tests/test_math_philox.py
 * on the host and tests/test_gpu_parity.py on the device.                                                                     */
#ifndef PHF_PHILOX_H
#define PHF_PHILOX_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PHF_PHILOX_HD static __host__ __device__ __forceinline__
#else
#define PHF_PHILOX_HD static inline
#endif

typedef struct { uint32_t w[4]; } phf_u32x4;

/* a ^ b ^ c: ONE instruction on gfx950 (v_bitop3_b32 with the truth table of a three-input xor; hipcc does not fuse two
 * v_xor_b32 into it by itself): 20 instead of 40 logic instructions per Philox block */
#if defined(__HIP_DEVICE_COMPILE__)
#define PHF_XOR3(a, b, c) ((uint32_t)__builtin_amdgcn_bitop3_b32((int)(a), (int)(b), (int)(c), 0x96))
#else
#define PHF_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif

#ifndef PHF_PHILOX_ROUNDS
#define PHF_PHILOX_ROUNDS 7
#endif

/* rounds: a literal (the loop unrolls) */
PHF_PHILOX_HD phf_u32x4 phf_philox4x32_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint32_t k0, uint32_t k1) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int round = 0; round < rounds; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = PHF_XOR3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = PHF_XOR3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  phf_u32x4 out;
  out.w[0] = c0; out.w[1] = c1; out.w[2] = c2; out.w[3] = c3;
  return out;
}

PHF_PHILOX_HD phf_u32x4 phf_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return phf_philox4x32_r(10, c0, c1, c2, c3, k0, k1);
}
PHF_PHILOX_HD phf_u32x4 phf_philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return phf_philox4x32_r(7, c0, c1, c2, c3, k0, k1);
}
/* the block the samplers draw from */
PHF_PHILOX_HD phf_u32x4 phf_philox_mh(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return phf_philox4x32_r(PHF_PHILOX_ROUNDS, c0, c1, c2, c3, k0, k1);
}

#endif /* PHF_PHILOX_H */
