// Issue-rate probe for the attention kernel's instruction mix on gfx950 (round 6): cycles per wave-instruction of v_exp_f32,
// v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_cvt_pk_bf16_f32, v_maximum3_f32, v_permlane16_swap and v_mfma_f32_16x16x32_bf16 alone,
// and of MFMA + k VALU fillers interleaved in ONE wave - with one and with two waves per SIMD (blockDim 256 / 512 on one
// workgroup per CU).  Register-only loops, independent operands; time from s_memtime of wave 0.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define REP8(X) X X X X X X X X

// MODE 0 exp, 1 fma, 2 pk_fma, 3 cvt_pk_bf16, 4 maximum3, 5 permlane16_swap, 6 mfma only, 7 mfma + 1 exp, 8 mfma + 2 exp,
// 9 mfma + 3 fma, 10 mfma + 1 exp + 2 fma, 11 mfma + 4 fma, 12 mfma + 2 pk_fma, 13 pk_mul, 14 mfma + 2 exp + 2 fma, 15 mfma + 6 fma
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = (f32x2){x[i], x[i + 8]};
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  union { bf16x8 v; unsigned u[4]; } a, b;
  a.u[0] = a.u[1] = a.u[2] = a.u[3] = 0x3f803f80u + threadIdx.x;
  b.u[0] = b.u[1] = b.u[2] = b.u[3] = 0x3f003f00u + threadIdx.x;
  unsigned w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = threadIdx.x * 7 + i;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); asm volatile("v_exp_f32 %0, %0" : "+v"(x[i + 8])); }
      if (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i + 8])); }
      if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i])); asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[(i + 4) & 7])); }
      if (MODE == 13) { asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i])); asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[(i + 4) & 7])); }
      if (MODE == 3) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(x[i]), "v"(x[i + 8])); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[(i + 4) & 7]) : "v"(x[i + 8]), "v"(x[i])); }
      if (MODE == 4) { asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15])); asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(x[i + 8]) : "v"(x[(i + 9) & 15]), "v"(x[(i + 10) & 15])); }
      if (MODE == 5) { asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[i]), "+v"(w[(i + 4) & 7])); asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[(i + 1) & 7]), "+v"(w[(i + 5) & 7])); }
      if (MODE >= 6 && MODE != 13) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[i], 0, 0, 0);
        if (MODE == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (MODE == 8) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); asm volatile("v_exp_f32 %0, %0" : "+v"(x[i + 8])); }
        if (MODE == 9) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i + 8])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 3) & 15])); }
        if (MODE == 10) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i + 8])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 3) & 15])); }
        if (MODE == 11) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i + 8])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 3) & 15])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 5) & 15])); }
        if (MODE == 12) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i])); asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[(i + 4) & 7])); }
        if (MODE == 14) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); asm volatile("v_exp_f32 %0, %0" : "+v"(x[i + 8])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 3) & 15])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 5) & 15])); }
        if (MODE == 15) {
#pragma unroll
          for (int q = 0; q < 6; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 2 * q + 1) & 15]));
        }
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + p[i][0] + p[i][1] + (float)w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int n_mfma, int n_valu, float* out, long long* cyc) {
  for (int threads : {256, 512}) {
    const int iters = 20000;
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_iter = (double)c / iters / 8.0;   // cycles (s_memtime units, 100 MHz? -> use wall time too) per inner group
    // wall-time based: cycles at the measured clock are unknown; report ns per group per wave and the counter
    printf("%-28s waves/SIMD %d: %7.2f ns per group of (%d mfma + %d valu) per wave  [counter %.2f/group]\n", name, threads / 256,
           ms * 1e6 / iters / 8.0, n_mfma, n_valu, per_iter);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0>("2 x v_exp_f32", 0, 2, out, cyc);
  run<1>("2 x v_fma_f32", 0, 2, out, cyc);
  run<2>("2 x v_pk_fma_f32", 0, 2, out, cyc);
  run<13>("2 x v_pk_mul_f32", 0, 2, out, cyc);
  run<3>("2 x v_cvt_pk_bf16_f32", 0, 2, out, cyc);
  run<4>("2 x v_maximum3_f32", 0, 2, out, cyc);
  run<5>("2 x v_permlane16_swap", 0, 2, out, cyc);
  run<6>("mfma16x16x32 alone", 1, 0, out, cyc);
  run<7>("mfma + 1 exp", 1, 1, out, cyc);
  run<8>("mfma + 2 exp", 1, 2, out, cyc);
  run<9>("mfma + 3 fma", 1, 3, out, cyc);
  run<10>("mfma + 1 exp + 2 fma", 1, 3, out, cyc);
  run<11>("mfma + 4 fma", 1, 4, out, cyc);
  run<12>("mfma + 2 pk_fma", 1, 2, out, cyc);
  run<14>("mfma + 2 exp + 2 fma", 1, 4, out, cyc);
  run<15>("mfma + 6 fma", 1, 6, out, cyc);
  return 0;
}
