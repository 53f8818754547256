// Micro-probe: does the per-node STRIDE of an array matter to the memory side? Every lane reads the 64-byte line at
// node * STRIDE + OFF of a random node (16-byte load), as the lane-per-node kernels read "the same field of many nodes" — if the
// address -> HBM channel mapping used few address bits, a stride that is a multiple of a large power of two (Handel's rows:
// 12 KB a node; a level-15 signature slot: 32 KB a node) would put every node's line on a handful of channels.
//   hipcc --offload-arch=gfx950 -O2 -o stride_probe stride_probe.hip && ./stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct alignas(16) U4 { uint32_t x, y, z, w; };
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) probe(const char* base, uint64_t nodes, uint64_t stride, uint64_t off, int rounds, int wr, uint64_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = mix(tid * 0x9E3779B97F4A7C15ULL + 1);
  for (int s = 0; s < rounds; s++) {
    U4 v[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const uint64_t node = mix(acc + (uint64_t)p * 0x51ED27ULL) % nodes;
      v[p] = *(const U4*)(base + node * stride + off + 16 * (tid & 3));
    }
    uint32_t got = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) got += v[p].x;
    if (wr) {
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const uint64_t node = mix(acc + 77 + (uint64_t)p * 0x7F4A7C15ULL) % nodes;
        U4 o; o.x = got | 1u; o.y = (uint32_t)s; o.z = o.w = 0;
        *(U4*)(base + node * stride + off + 16 * (tid & 3)) = o;
      }
    }
    acc += (got & 1u) + 1;
  }
  if ((threadIdx.x & 63) == 0) out[tid >> 6] = acc;
}
int main() {
  const size_t bytes = 96ull << 30;
  char* buf; uint64_t* out;
  CK(hipMalloc((void**)&out, 8 * 2048 * 4));
  CK(hipMalloc((void**)&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  // (the same NUMBER of nodes for every stride — 1 Mi, a 32-copy batch of 32 768-node networks — so that the lines touched are
  // the same 64 MB whatever the stride: what differs is where they lie)
  const uint64_t strides[] = {64, 640, 640 + 64, 2048, 2048 + 64, 4096, 4096 + 64, 5120, 5120 + 64, 8192, 8192 + 64, 12288, 12288 + 64, 16384, 16384 + 64,
                              32768, 32768 + 64, 32768 + 128, 32768 + 256, 65536, 65536 + 64};
  printf("1 Mi nodes; 8 scattered 16-byte reads (+ 4 writes) per lane and round, 2048 x 256 lanes\n");
  for (int wr = 0; wr < 2; wr++)
    for (uint64_t st : strides) {
      const uint64_t nodes = 1ull << 20;
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, buf, nodes, st, 128, 4, wr, out);
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, buf, nodes, st, 128, 64, wr, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double n = 2048.0 * 256 * 64 * (8 + (wr ? 4 : 0));
      printf("stride %8llu B, %s: %8.3f ms  %7.2f G lines/s\n", (unsigned long long)st, wr ? "reads + writes" : "reads only    ", ms, n / (ms * 1e-3) / 1e9);
    }
  return 0;
}
