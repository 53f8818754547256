// Micro-benchmark: does the width of the per-lane access decide the memory system's ceiling for block transfers?
// One wavefront per random block of BW 64-bit words (a Handel level block / payload / snapshot): read it from one
// random place of a 32 GB buffer and write it to another, with 8-byte, 16-byte (dwordx4) per-lane accesses.
// Reports GB/s (read + write) and blocks/us. (profiles/r07h: the node-visit kernels sit at ~ 20-27 G 64-byte
// requests/s whatever their access pattern — if a 16-byte-per-lane access makes 128-byte requests, bulk rows move faster.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
struct alignas(16) V2 { uint64_t a, b; };
template <int W16>
__global__ void __launch_bounds__(256) copy_blocks(uint64_t* base, uint64_t blocks, int bw, int iters) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; it++) {
    const uint64_t q = (uint64_t)it * nWaves + wave;
    const uint64_t s = (mix(q * 2 + 1) & (blocks - 1)) * (uint64_t)bw, t = (mix(q * 2 + 2) & (blocks - 1)) * (uint64_t)bw;
    if (W16) {
      const V2* src = (const V2*)(base + s);
      V2* dst = (V2*)(base + t);
      V2 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int j = u * 64 + lane; if (j < bw / 2) v[u] = src[j]; }
#pragma unroll
      for (int u = 0; u < 4; u++) { const int j = u * 64 + lane; if (j < bw / 2) dst[j] = v[u]; }
    } else {
      uint64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int j = u * 64 + lane; if (j < bw) v[u] = base[s + j]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int j = u * 64 + lane; if (j < bw) base[t + j] = v[u]; }
    }
  }
}
int main() {
  const uint64_t words = 1ull << 32;  // 32 GB
  uint64_t* buf;
  CK(hipMalloc((void**)&buf, words * 8));
  CK(hipMemset(buf, 1, words * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("%8s %6s %8s | %10s %12s\n", "waves", "words", "access", "GB/s r+w", "blocks/us");
  for (int blocksPerGrid : {2048, 8192}) for (int bw : {8, 32, 64, 256, 512}) for (int w16 = 0; w16 < 2; w16++) {
    const uint64_t nblk = words / bw;
    const int iters = 64;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(a));
      if (w16) hipLaunchKernelGGL(copy_blocks<1>, dim3(blocksPerGrid), dim3(256), 0, 0, buf, nblk, bw, iters);
      else hipLaunchKernelGGL(copy_blocks<0>, dim3(blocksPerGrid), dim3(256), 0, 0, buf, nblk, bw, iters);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double nb = (double)blocksPerGrid * 4 * iters;
    printf("%8d %6d %8s | %10.1f %12.2f\n", blocksPerGrid * 4, bw, w16 ? "16 B" : "8 B", nb * bw * 16 / (ms * 1e6), nb / (ms * 1e3));
  }
  return 0;
}
