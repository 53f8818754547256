// Micro-benchmark: what bounds a wave-per-node visit — latency or the rate of scattered requests?
// Every wave runs visits of `steps` DEPENDENT rounds; a round issues `par` independent reads of `bytes`
// contiguous bytes (bytes/8 lanes x 8 B) at random places of a 16 GB footprint. Prints us per round and the
// aggregate request rate for several (waves in flight, par, bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void probe(const uint64_t* base, uint64_t chunks, int lanesPerRead, int visits, int steps, int par, int write, uint64_t* out, uint64_t* wbuf) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t h = mix(wave * 0x9E3779B97F4A7C15ULL + 1), acc = 0;
  for (int v = 0; v < visits; v++)
    for (int s = 0; s < steps; s++) {
      uint64_t got = 0;
      for (int p = 0; p < par; p++) {
        uint64_t r = mix(h ^ (acc + (uint64_t)(v * 131 + s * 17 + p))) % chunks;
        if (lane < lanesPerRead) {
          if (write) wbuf[r * 64 + lane] = acc; else got += base[r * 64 + lane];
        }
      }
      acc += __shfl(got, 0, 64) + 1;
    }
  if (lane == 0) out[wave] = acc;
}
int main() {
  uint64_t *out, *buf;
  const size_t bytes = 16ull << 30;
  CK(hipMalloc((void**)&out, 8 * 8192 * 4));
  CK(hipMalloc((void**)&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  const uint64_t chunks = bytes / 512;
  printf("%8s %4s %6s %5s | %10s %12s %10s\n", "waves", "par", "bytes", "write", "us/round", "Greq/s", "GB/s");
  const int grids[] = {1, 256, 1024, 2048, 4096};
  for (int g : grids)
    for (int par : {1, 6, 16})
      for (int lanesPerRead : {8, 16, 64})
        for (int write = 0; write < 2; write++) {
          if (g == 1 && (par != 1 || write)) continue;
          if (write && lanesPerRead == 16) continue;
          const int visits = g == 1 ? 256 : 32, steps = 4;
          hipEvent_t a, b;
          CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
          hipLaunchKernelGGL(probe, dim3(g), dim3(256), 0, 0, buf, chunks, lanesPerRead, 2, steps, par, write, out, buf);
          CK(hipEventRecord(a));
          hipLaunchKernelGGL(probe, dim3(g), dim3(256), 0, 0, buf, chunks, lanesPerRead, visits, steps, par, write, out, buf);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          const double rounds = (double)visits * steps, us = ms * 1e3 / rounds;
          const double reqs = (double)g * 4 * par * rounds;
          printf("%8d %4d %6d %5d | %10.2f %12.2f %10.1f\n", g * 4, par, lanesPerRead * 8, write, us, reqs / (ms * 1e6), reqs * lanesPerRead * 8 / (ms * 1e6));
        }
  return 0;
}
