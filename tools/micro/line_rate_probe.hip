// Micro-benchmark behind bench.py's LINE_RATE_CEILING: the rate at which the chip's memory side
// serves SCATTERED lines when reads and writes are MIXED the way the delivery pass mixes them — mlp_probe.hip measured
// scattered 64-byte READS only (21 G lines/s), and the pass sends 27 G requests/s of which 47 % are writes, so that figure
// was no ceiling for it.
// One LANE = one access stream (as the lane-per-node kernels): every round a lane issues RD independent 16-byte loads,
// each from its own random 64-byte line of a FOOT-GiB footprint, then WR stores of WB bytes (16 / 32 / 64: one, two or
// four 16-byte stores to one random line); the next round's addresses depend on what was loaded (a visit's chain).
// Full-chip occupancy: 2048 workgroups x 256 threads, 8 wavefronts a SIMD. Printed: logical lines/s (reads, writes, sum);
// the same kernels under `rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum` give the rate in the
// PMC's own unit (tools/line_rate.sh), which is what bench.py's roofline.line_rate compares the pass with.
//   hipcc --offload-arch=gfx950 -O2 -o line_rate_probe line_rate_probe.hip && ./line_rate_probe [footprint GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct alignas(16) U4 { uint32_t x, y, z, w; };
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int RD, int WR, int WB>
__global__ void __launch_bounds__(256) probe(U4* base, uint64_t lines, int rounds, uint64_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = mix(tid * 0x9E3779B97F4A7C15ULL + 1);
  for (int s = 0; s < rounds; s++) {
    U4 v[RD > 0 ? RD : 1];
#pragma unroll
    for (int p = 0; p < RD; p++) {
      const uint64_t r = mix(acc + (uint64_t)p * 0x51ED27ULL) & (lines - 1);  // (lines is a power of two)
      v[p] = base[r * 4 + (tid & 3)];  // 16 bytes of the lane's own random line
    }
    uint32_t got = 0;
#pragma unroll
    for (int p = 0; p < RD; p++) got += v[p].x;
#pragma unroll
    for (int q = 0; q < WR; q++) {
      const uint64_t w = mix(acc + 77 + (uint64_t)q * 0x7F4A7C15ULL) & (lines - 1);
      U4 o;
      o.x = got | 1u, o.y = (uint32_t)s, o.z = o.w = 0;
#pragma unroll
      for (int k = 0; k < WB / 16; k++) base[w * 4 + ((tid + k) & 3)] = o;
    }
    acc += (got & 1u) + 1;
  }
  if ((threadIdx.x & 63) == 0) out[tid >> 6] = acc;
}
template <int RD, int WR, int WB>
int run(U4* buf, uint64_t lines, uint64_t* out, int blocks) {
  const int rounds = 128;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((probe<RD, WR, WB>), dim3(blocks), dim3(256), 0, 0, buf, lines, 4, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((probe<RD, WR, WB>), dim3(blocks), dim3(256), 0, 0, buf, lines, rounds, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double n = (double)blocks * 256 * rounds, s = ms * 1e-3;
  printf("RD %2d WR %2d x %2d B | %8.3f ms | reads %7.2f G lines/s  writes %7.2f G lines/s  sum %7.2f G/s  (writes %.0f %%)\n", RD, WR, WB, ms,
         n * RD / s / 1e9, n * WR / s / 1e9, n * (RD + WR) / s / 1e9, 100.0 * WR / (RD + WR > 0 ? RD + WR : 1));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return 0;
}
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 128;
  uint64_t* out;
  U4* buf;
  size_t bytes = gib << 30;
  uint64_t lines = 1;
  while (lines * 2 * 64 <= bytes) lines *= 2;
  bytes = lines * 64;
  CK(hipMalloc((void**)&out, 8 * 2048 * 4));
  CK(hipMalloc((void**)&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  printf("footprint %zu GiB, 2048 workgroups x 256 lanes, one random 64-byte line per access\n", bytes >> 30);
  const int B = 2048;
  // reads alone, writes alone, then the pass's 53 : 47 (8 : 7) with each write size
  if (run<8, 0, 16>(buf, lines, out, B)) return 1;
  if (run<0, 8, 16>(buf, lines, out, B)) return 1;
  if (run<0, 8, 32>(buf, lines, out, B)) return 1;
  if (run<0, 8, 64>(buf, lines, out, B)) return 1;
  if (run<8, 7, 16>(buf, lines, out, B)) return 1;
  if (run<8, 7, 32>(buf, lines, out, B)) return 1;
  if (run<8, 7, 64>(buf, lines, out, B)) return 1;
  if (run<4, 4, 16>(buf, lines, out, B)) return 1;
  if (run<2, 2, 16>(buf, lines, out, B)) return 1;
  if (run<1, 1, 16>(buf, lines, out, B)) return 1;
  return 0;
}
