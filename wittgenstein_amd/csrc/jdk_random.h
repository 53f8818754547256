// java.util.Random for host and device, with O(log k) jump-ahead.
//
// The reference shares ONE generator `rd = new Random(0)` between node construction, every send()
// and protocol logic (C/Network.java:32,377,430; P/Handel.java:789), consumed in event order. The
// engine reproduces that stream by giving every draw its index in the global order (an exclusive
// scan) and jumping the 48-bit LCG straight to it:  s_k = A_k * s_0 + C_k  (mod 2^48).
// Algorithm per the java.util.Random Javadoc (JDK 9, build.gradle:8-9).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WG_HD __host__ __device__
#else
#define WG_HD
#endif

namespace wg {

constexpr uint64_t LCG_A = 0x5DEECE66DULL;
constexpr uint64_t LCG_C = 0xBULL;
constexpr uint64_t LCG_MASK = (1ULL << 48) - 1;

WG_HD inline uint64_t lcg_scramble(int64_t seed) { return ((uint64_t)seed ^ LCG_A) & LCG_MASK; }
WG_HD inline uint64_t lcg_step(uint64_t s) { return (s * LCG_A + LCG_C) & LCG_MASK; }

// the transform of 2^i steps, s -> LCG_POW.a[i] * s + LCG_POW.c[i] (mod 2^48), for every i: compile-time constants
struct LcgPow {
  uint64_t a[48], c[48];
};
constexpr LcgPow lcg_pow_table() {
  LcgPow t{};
  uint64_t a = LCG_A, c = LCG_C;
  for (int i = 0; i < 48; i++) {
    t.a[i] = a;
    t.c[i] = c;
    c = (c * (a + 1)) & LCG_MASK;
    a = (a * a) & LCG_MASK;
  }
  return t;
}
#if defined(__HIPCC__)
__device__ __constant__ const LcgPow LCG_POW = lcg_pow_table();
#endif

// state after k steps from s
WG_HD inline uint64_t lcg_skip(uint64_t s, uint64_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  // one multiply-add per SET bit of k on the state itself (the 2^i-step transforms commute): composing the transform first
  // — four 64-bit multiplies per bit of k, set or not — made k_resolve ALU-bound in the ms in which every node sends
  // (8 M records of draw indices up to 2^19: 230 us at 24 copies of 32 768 nodes)
  for (int i = 0; i < 48 && (k >> i); i++)
    if ((k >> i) & 1ULL) s = (LCG_POW.a[i] * s + LCG_POW.c[i]) & LCG_MASK;
  return s;
#else
  uint64_t a = LCG_A, c = LCG_C;  // transform for 2^i steps
  uint64_t A = 1, Cc = 0;         // accumulated transform
  while (k) {
    if (k & 1) {
      A = (A * a) & LCG_MASK;
      Cc = (Cc * a + c) & LCG_MASK;
    }
    c = (c * (a + 1)) & LCG_MASK;
    a = (a * a) & LCG_MASK;
    k >>= 1;
  }
  return (A * s + Cc) & LCG_MASK;
#endif
}

// next(bits) on an explicit state
WG_HD inline int32_t lcg_next(uint64_t& s, int bits) {
  s = lcg_step(s);
  return (int32_t)(int64_t)(s >> (48 - bits));
}

// Random.nextInt(bound) on an explicit state; *consumed = number of next() calls (1 + rejections).
WG_HD inline int32_t lcg_next_int_bounded(uint64_t& s, int32_t bound, int* consumed) {
  int n = 1;
  int32_t r = lcg_next(s, 31);
  int32_t m = bound - 1;
  if ((bound & m) == 0) {
    r = (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
  } else {
    int32_t u = r;
    for (;;) {
      r = u % bound;
      if ((int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m) >= 0) break;
      u = lcg_next(s, 31);
      n++;
    }
  }
  if (consumed) *consumed = n;
  return r;
}

// Host-side generator object (init() code paths).
struct JavaRandom {
  uint64_t s;
  explicit JavaRandom(int64_t seed = 0) : s(lcg_scramble(seed)) {}
  void setSeed(int64_t seed) { s = lcg_scramble(seed); }
  int32_t nextInt() { return lcg_next(s, 32); }
  int32_t nextInt(int32_t bound) { return lcg_next_int_bounded(s, bound, nullptr); }
  bool nextBoolean() { return lcg_next(s, 1) != 0; }
  double nextDouble() {
    int64_t hi = lcg_next(s, 26);
    int64_t lo = lcg_next(s, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
  }
};

// Network.getPseudoRandom (C/Network.java:493-503): xorshift of the node id, xor seed, |x % 100|.
WG_HD inline int32_t pseudo_delta(int32_t nodeId, int32_t seed) {
  uint32_t a = (uint32_t)nodeId;
  a ^= a << 13;
  a ^= a >> 17;
  a ^= a << 5;
  int32_t x = (int32_t)(a ^ (uint32_t)seed);
  int32_t r = x % 100;
  return r < 0 ? -r : r;
}

}  // namespace wg
