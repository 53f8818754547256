// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the delivery kernels
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an
// absolute"). Every kernel moves a KNOWN number of bytes over a 8 GiB footprint (beyond the 256 MiB Infinity Cache);
// the program prints that number per kernel, tools/pmc_calib.sh prints the counters next to it.
//   stream16   every lane 16 B, coalesced (1 KiB per wave instruction)
//   line64     a wave instruction touches ONE random 64-byte line with 8 lanes x 8 B           (header / row words)
//   line64x4   a wave instruction touches 4 random 64-byte lines, 16 lanes... 4 x (4 lanes x 16 B)   (record moves)
//   lane8      every lane 8 B in its OWN random 64-byte line (64 lines per instruction)        (lane-per-node gathers)
//   lane16     every lane 16 B in its own random line                                           (envelope records)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

__global__ void r_stream16(const uint4* p, size_t n, uint64_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  uint64_t a = 0;
  for (; i < n; i += st) { uint4 v = p[i]; a += v.x + v.w; }
  if (a == 0x1234567) out[0] = a;
}
__global__ void w_stream16(uint4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// mode 0: line64 (8 lanes x 8 B, one line); 1: line64x4 (16 lanes x 16 B over 4 lines); 2: lane8; 3: lane16
template <int MODE, bool WRITE>
__global__ void scat(uint64_t* base, uint64_t lines, int iters, uint64_t* out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t acc = 0;
  for (int it = 0; it < iters; it++) {
    const uint64_t h = mix(((uint64_t)wave << 20) + it + 1);
    if (MODE == 0) {
      uint64_t* q = base + (h % lines) * 8 + lane;
      if (lane < 8) { if (WRITE) *q = h; else acc += *q; }
    } else if (MODE == 1) {
      uint64_t* q = base + (mix(h + (lane >> 2)) % lines) * 8 + (lane & 3) * 2;
      if (lane < 16) { if (WRITE) *(ulonglong2*)q = make_ulonglong2(h, h); else { ulonglong2 v = *(const ulonglong2*)q; acc += v.x + v.y; } }
    } else if (MODE == 2) {
      uint64_t* q = base + (mix(h + lane) % lines) * 8 + (lane & 7);
      if (WRITE) *q = h; else acc += *q;
    } else {
      uint64_t* q = base + (mix(h + lane) % lines) * 8 + (lane & 3) * 2;
      if (WRITE) *(ulonglong2*)q = make_ulonglong2(h, h); else { ulonglong2 v = *(const ulonglong2*)q; acc += v.x + v.y; }
    }
  }
  if (acc == 0x1234567) out[0] = acc;
}
template <int MODE, bool WRITE>
static int run(const char* name, uint64_t* buf, uint64_t lines, uint64_t* out, double bytesPerInstr) {
  const int grid = 4096, block = 256, iters = 256;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((scat<MODE, WRITE>), dim3(grid), dim3(block), 0, 0, buf, lines, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double instr = (double)grid * (block / 64) * iters, bytes = instr * bytesPerInstr;
  printf("CALIB %-28s known_bytes %14.0f  wave_instr %10.0f  %8.3f ms  %8.1f GB/s\n", name, bytes, instr, ms, bytes / (ms * 1e6));
  return 0;
}
int main() {
  uint64_t *out, *buf;
  const size_t bytes = 8ull << 30;
  CK(hipMalloc((void**)&out, 64));
  CK(hipMalloc((void**)&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  CK(hipDeviceSynchronize());
  const uint64_t lines = bytes / 64;
  {
    hipEvent_t a, b; float ms;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(r_stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("CALIB %-28s known_bytes %14.0f  wave_instr %10.0f  %8.3f ms  %8.1f GB/s\n", "r_stream16", (double)bytes, bytes / 1024.0, ms, bytes / (ms * 1e6));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(w_stream16, dim3(8192), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("CALIB %-28s known_bytes %14.0f  wave_instr %10.0f  %8.3f ms  %8.1f GB/s\n", "w_stream16", (double)bytes, bytes / 1024.0, ms, bytes / (ms * 1e6));
  }
  if (run<0, false>("scat<0, false> r_line64", buf, lines, out, 64)) return 1;
  if (run<1, false>("scat<1, false> r_line64x4", buf, lines, out, 256)) return 1;
  if (run<2, false>("scat<2, false> r_lane8", buf, lines, out, 512)) return 1;
  if (run<3, false>("scat<3, false> r_lane16", buf, lines, out, 1024)) return 1;
  if (run<0, true>("scat<0, true> w_line64", buf, lines, out, 64)) return 1;
  if (run<1, true>("scat<1, true> w_line64x4", buf, lines, out, 256)) return 1;
  if (run<2, true>("scat<2, true> w_lane8", buf, lines, out, 512)) return 1;
  if (run<3, true>("scat<3, true> w_lane16", buf, lines, out, 1024)) return 1;
  return 0;
}
