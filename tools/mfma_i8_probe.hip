// Layout probe for v_mfma_i32_16x16x64_i8 (tool): A = one-hot at (i0, k0) with the ASSUMED operand map (lane l: row l&15,
// bytes = k (l>>4)*16 .. +15), B[j][k] = (7 j + 3 k) % 100 in the same map -> D[i0][j] should be B[j][k0] at lane (j, i0>>2),
// register i0 & 3.  Prints where the non-zero results actually land.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(int i0, int k0, int* out) {
  const int lane = threadIdx.x, l15 = lane & 15, q4 = lane >> 4;
  union { v4i v; int8_t b[16]; } a, b;
  for (int j = 0; j < 16; ++j) {
    const int k = q4 * 16 + j;
    a.b[j] = (l15 == i0 && k == k0) ? 1 : 0;
    b.b[j] = (int8_t)((7 * l15 + 3 * k) % 100);
  }
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  int h[256];
  int bad = 0;
  for (int i0 = 0; i0 < 16; ++i0)
    for (int k0 = 0; k0 < 64; k0 += 5) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, i0, k0, d);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int j = lane & 15, i = (lane >> 4) * 4 + r;
          const int want = (i == i0) ? (7 * j + 3 * k0) % 100 : 0;
          if (h[lane * 4 + r] != want) {
            if (bad < 12) printf("i0=%d k0=%d: lane %d (j=%d q=%d) reg %d holds %d, expected %d\n", i0, k0, lane, j, lane >> 4, r, h[lane * 4 + r], want);
            ++bad;
          }
        }
    }
  printf("mismatches: %d\n", bad);
  return 0;
}
