// Micro-benchmark behind DESIGN.md's layout decision: latency of DEPENDENT 64-byte reads spread over a
// footprint of F GB (what a wave-per-node visit does when every field lives in its own N-strided array),
// versus the same number of reads confined to one random 512 KB region per visit (one arena per node).
// hipcc --offload-arch=gfx950 -O3 tlb_probe.hip -o tlb_probe && ./tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// every wave: `visits` visits, each = `steps` dependent rounds of `par` independent 64-B line reads
__global__ void probe(const uint64_t* base, uint64_t lines, uint64_t regionLines, int visits, int steps, int par, uint64_t* out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t h = mix(wave * 0x9E3779B97F4A7C15ULL + 1);
  uint64_t acc = 0;
  for (int v = 0; v < visits; v++) {
    const uint64_t region = regionLines ? (mix(h + v) % (lines / regionLines)) * regionLines : 0;
    for (int s = 0; s < steps; s++) {
      uint64_t got = 0;
      for (int p = 0; p < par; p++) {
        uint64_t r = mix(h ^ (acc + (uint64_t)(v * 131 + s * 17 + p)));
        uint64_t line = regionLines ? region + r % regionLines : r % lines;
        if (lane < 8) got += base[line * 8 + lane];  // one 64-byte line per read
      }
      acc += __shfl(got, 0, 64) + 1;  // the next round depends on this one
    }
  }
  if (lane == 0) out[wave] = acc;
}

int main() {
  const int blocks = 1024, threads = 256;  // 4096 waves = 4 per SIMD, as the delivery kernel runs
  uint64_t* out;
  CK(hipMalloc((void**)&out, 8 * blocks * 4));
  size_t freeB, totalB;
  CK(hipMemGetInfo(&freeB, &totalB));
  printf("free %.1f GB of %.1f GB\n", freeB / 1e9, totalB / 1e9);
  const double sizesGB[] = {0.25, 2, 16, 64, 160};
  for (double gb : sizesGB) {
    size_t bytes = (size_t)(gb * (1ull << 30));
    if (bytes + (4ull << 30) > freeB) continue;
    uint64_t* buf;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess) { printf("%.2f GB: alloc failed\n", gb); continue; }
    CK(hipMemset(buf, 0, bytes));
    const uint64_t lines = bytes / 64;
    for (int mode = 0; mode < 2; mode++) {
      const uint64_t regionLines = mode ? (512 * 1024) / 64 : 0;
      const int visits = 64, steps = 4, par = 6;
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, buf, lines, regionLines, 4, steps, par, out);  // warm
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, buf, lines, regionLines, visits, steps, par, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("footprint %6.2f GB  %-28s  %.2f us per visit (4 dependent rounds x 6 lines), %.2f us per round\n", gb,
             mode ? "one 512 KB region per visit" : "lines anywhere in footprint", ms * 1e3 / visits, ms * 1e3 / visits / steps);
    }
    CK(hipFree(buf));
  }
  return 0;
}
