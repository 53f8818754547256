// Micro-benchmark: do independent scattered loads of ONE wavefront overlap (memory-level parallelism), and what is a
// dependent round trip worth under load? Every wave runs `rounds` dependent rounds; a round issues PAR independent
// reads (compile-time PAR: the loads are all in flight before the first use) of one random 64-byte line each
// (8 lanes x 8 B) from a 16 GB footprint. us/round ~ constant in PAR  =>  a visit's cost is its number of DEPENDENT
// rounds, not its number of loads. (tools/micro/req_probe.hip's `par` loop is not unrolled, and both it and this file's first version took the line
// index with a 64-bit `%`, ~1000 cycles of ALU per load: their rates were the divider's, not the memory system's.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int PAR>
__global__ void __launch_bounds__(256) probe(const uint64_t* base, uint64_t lines, int rounds, uint64_t* out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t acc = mix(wave * 0x9E3779B97F4A7C15ULL + 1);
  for (int s = 0; s < rounds; s++) {
    uint64_t v[PAR];
#pragma unroll
    for (int p = 0; p < PAR; p++) {
      const uint64_t r = mix(acc + (uint64_t)p * 0x51ED27ULL) & (lines - 1);  // (lines is a power of two: a 64-bit % is ~1000 cycles)
      v[p] = lane < 8 ? base[r * 8 + lane] : 0;
    }
    uint64_t got = 0;
#pragma unroll
    for (int p = 0; p < PAR; p++) got += v[p];
    acc += __shfl(got, 0, 64) + 1;
  }
  if (lane == 0) out[wave] = acc;
}
// A round = one dependent load, then (ST = 1) one 64-byte store to another random line. On gfx9-family ISAs loads and
// stores share the vmcnt counter, which retires in order: waiting for the NEXT round's load also waits for this round's
// store to be acknowledged. ST = 2: the store goes to the line that was just loaded (already in L2).
template <int ST>
__global__ void __launch_bounds__(256) probe_st(uint64_t* base, uint64_t lines, int rounds, uint64_t* out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t acc = mix(wave * 0x9E3779B97F4A7C15ULL + 1);
  for (int s = 0; s < rounds; s++) {
    const uint64_t r = mix(acc) & (lines - 1);
    const uint64_t got = lane < 8 ? base[r * 8 + lane] : 0;
    acc += __shfl(got & 1, 0, 64) + 1;
    if (ST) {
      const uint64_t w = ST == 2 ? r : (mix(acc + 77) & (lines - 1));
      if (lane < 8) base[w * 8 + lane] = got | 1;
    }
  }
  if (lane == 0) out[wave] = acc;
}
template <int ST>
int run_st(uint64_t* buf, uint64_t lines, uint64_t* out, int blocks) {
  const int rounds = 256;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe_st<ST>, dim3(blocks), dim3(256), 0, 0, buf, lines, 8, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(probe_st<ST>, dim3(blocks), dim3(256), 0, 0, buf, lines, rounds, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%8d  store=%d | %9.3f us/round\n", blocks * 4, ST, ms * 1e3 / rounds);
  return 0;
}
template <int PAR>
int run(const uint64_t* buf, uint64_t lines, uint64_t* out, int blocks) {
  const int rounds = 256;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<PAR>, dim3(blocks), dim3(256), 0, 0, buf, lines, 8, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(probe<PAR>, dim3(blocks), dim3(256), 0, 0, buf, lines, rounds, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const int resident = blocks < 1024 ? blocks : 1024;  // 4 blocks of 256 threads per CU at <= 128 VGPRs; probe uses few
  printf("%8d %4d | %9.3f us/round (if all waves resident) | %8.2f G lines/s\n", blocks * 4, PAR, ms * 1e3 / rounds,
         (double)blocks * 4 * PAR * rounds / (ms * 1e6));
  (void)resident;
  return 0;
}
int main() {
  uint64_t *out, *buf;
  const size_t bytes = 16ull << 30;
  CK(hipMalloc((void**)&out, 8 * 65536));
  CK(hipMalloc((void**)&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  const uint64_t lines = bytes / 64;
  printf("%8s %4s\n", "waves", "PAR");
  for (int blocks : {64, 1024, 2048, 4096}) {
    if (run<1>(buf, lines, out, blocks)) return 1;
    if (run<2>(buf, lines, out, blocks)) return 1;
    if (run<4>(buf, lines, out, blocks)) return 1;
    if (run<8>(buf, lines, out, blocks)) return 1;
    if (run<16>(buf, lines, out, blocks)) return 1;
  }
  printf("dependent load (+ store) per round\n");
  for (int blocks : {64, 1024, 2048}) {
    if (run_st<0>(buf, lines, out, blocks)) return 1;
    if (run_st<1>(buf, lines, out, blocks)) return 1;
    if (run_st<2>(buf, lines, out, blocks)) return 1;
  }
  return 0;
}
