// Micro-probe behind engine_kernels.hip.h wg_place: on which XCD does workgroup (bx, by) of a 2-D grid (gx, R) run?
// Every block reads HW_REG_XCC_ID; the report is the share of blocks whose XCD equals (bx + gx * by) % 8 — the dispatcher's
// observed (not contracted) round-robin by LINEAR workgroup id — and, with the same re-deal wg_place does, the share of blocks
// of engine e that run on XCD e % 8.          hipcc --offload-arch=gfx950 -O2 -o xcd_probe xcd_probe.hip && ./xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) probe(uint32_t* xcc) {
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[blockIdx.x + gridDim.x * blockIdx.y] = id & 0xFu;
}
int main() {
  const int shapes[][2] = {{40, 31}, {88, 12}, {8, 31}, {136, 31}, {1, 31}, {16, 256}, {520, 31}};
  for (auto& sh : shapes) {
    const int gx = sh[0], R = sh[1], T = gx * R;
    uint32_t* d;
    CK(hipMalloc(&d, 4 * T));
    for (int rep = 0; rep < 3; rep++) probe<<<dim3(gx, R), 256>>>(d);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(T);
    CK(hipMemcpy(h.data(), d, 4 * T, hipMemcpyDeviceToHost));
    int lin = 0, placed = 0, counted = 0;
    for (int L = 0; L < T; L++) {
      lin += (int)h[L] == (L & 7);
      if (R >= 8 && gx >= 2) {  // wg_place's re-deal
        const int xcd = L & 7, j = L >> 3, blocksHere = (T - xcd + 7) >> 3, enginesHere = (R - xcd + 7) >> 3;
        int per = blocksHere / enginesHere;
        if (per > gx) per = gx;
        const int ei = j / per;
        if (ei >= enginesHere) continue;
        const int e = xcd + 8 * ei;
        counted++;
        placed += (int)h[L] == (e & 7);
      }
    }
    printf("grid (%d, %d): XCD == linear id %% 8 for %d of %d blocks; engine e on XCD e %% 8 for %d of %d re-dealt blocks\n", gx, R, lin, T,
           placed, counted);
    CK(hipFree(d));
  }
  return 0;
}
