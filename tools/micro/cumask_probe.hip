// Probe: a stream created with hipExtStreamCreateWithCUMask for ONE XCD — where do its workgroups run (HW_REG_XCC_ID), for
#include <chrono>
// the two plausible bit layouts of the mask (CU i of the agent -> XCD i % 8, or XCD i / 32), and do kernels of eight such
// streams run side by side?    hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && timeout 60 ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) where(uint32_t* cnt) {
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) atomicAdd(&cnt[id & 7u], 1u);
}
__global__ void __launch_bounds__(256) spin(uint64_t* out, int iters) {
  uint64_t a = threadIdx.x;
  for (int i = 0; i < iters; i++) a = a * 6364136223846793005ULL + 1442695040888963407ULL;
  if (a == 42) out[0] = a;
}
int main() {
  uint32_t* cnt;
  uint64_t* out;
  CK(hipMalloc(&cnt, 32));
  CK(hipMalloc(&out, 8));
  for (int layout = 0; layout < 2; layout++) {
    for (int k = 0; k < 8; k += 3) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 256; i++) {
        const bool on = layout == 0 ? (i % 8) == k : (i / 32) == k;
        if (on) mask[i / 32] |= 1u << (i % 32);
      }
      hipStream_t s;
      CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
      CK(hipMemsetAsync(cnt, 0, 32, s));
      where<<<dim3(2048), 256, 0, s>>>(cnt);
      CK(hipStreamSynchronize(s));
      uint32_t h[8];
      CK(hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost));
      printf("layout %s, XCD %d wanted: blocks per XCC_ID =", layout == 0 ? "i %% 8" : "i / 32", k);
      for (int q = 0; q < 8; q++) printf(" %u", h[q]);
      printf("\n");
      CK(hipStreamDestroy(s));
    }
  }
  // eight masked streams (layout i % 8), one long kernel each: side by side or one after the other?
  hipStream_t st[8];
  for (int k = 0; k < 8; k++) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 256; i++)
      if ((i % 8) == k) mask[i / 32] |= 1u << (i % 32);
    CK(hipExtStreamCreateWithCUMask(&st[k], 8, mask));
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int n = 1; n <= 8; n *= 2) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    CK(hipStreamSynchronize(0));
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; k++) spin<<<dim3(32 * 4), 256, 0, st[k]>>>(out, 2000000);
    CK(hipDeviceSynchronize());
    auto t1 = std::chrono::steady_clock::now();
    printf("%d masked streams, one 128-block kernel each: %.2f ms\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count());
  }
  return 0;
}
