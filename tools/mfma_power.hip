// Power-capped MFMA ceiling on MI355X: register-only MFMA loops (no LDS, no memory traffic) with zero vs random bf16
// operands, for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16, one or two waves per SIMD.  Tells how much of
// the 2.5 PFLOP/s nominal peak the package power budget allows on random data - the regime the encoder GEMMs run in.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power && tools/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int SHAPE>   // 0: 16x16x32 (16 independent accumulators), 1: 32x32x16 (8 independent accumulators)
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ ops, float* out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  union { u32x4 u; bf16x8 v; } a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i].u = ops[(size_t)tid * 8 + i];
    b[i].u = ops[(size_t)tid * 8 + 4 + i];
  }
  float s = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3].v, b[i >> 2].v, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3].v, b[i >> 2].v, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  out[tid] = s;
}

static unsigned short rnd_bf16(unsigned long long& st, int mode) {
  st = st * 6364136223846793005ULL + 1442695040888963407ULL;
  if (mode == 0) return 0;
  const unsigned r = (unsigned)(st >> 33);
  // uniform in about [-1, 1): random sign, exponent 2^-8..2^-1, random 7-bit mantissa
  const unsigned sign = (r >> 30) & 1, e = 119 + ((r >> 20) & 7), m = r & 0x7f;
  return (unsigned short)((sign << 15) | (e << 7) | m);
}

int main() {
  const int blocks = 256;
  for (int threads : {256, 512}) {
    const size_t n = (size_t)blocks * threads;
    u32x4* d_ops;
    float* d_out;
    hipMalloc(&d_ops, n * 8 * sizeof(u32x4));
    hipMalloc(&d_out, n * sizeof(float));
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<unsigned short> h(n * 8 * 8);
      unsigned long long st = 12345;
      for (auto& x : h) x = rnd_bf16(st, mode);
      hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
      for (int shape = 0; shape < 2; ++shape) {
        const int iters = 6000000;
        auto launch = [&](int it) {
          if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, 0, d_ops, d_out, it);
          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, 0, d_ops, d_out, it);
        };
        launch(iters / 4);   // warm-up / let the clocks settle
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_iter = shape == 0 ? 16.0 * 2 * 16 * 16 * 32 : 8.0 * 2 * 32 * 32 * 16;
        const double flops = per_iter * iters * (double)(n / 64);
        printf("%s  %d waves/SIMD  %-6s operands: %8.1f TFLOP/s  (%.1f ms)\n", shape == 0 ? "16x16x32" : "32x32x16",
               threads / 256, mode == 0 ? "zero" : "random", flops / (ms * 1e-3) / 1e12, ms);
        fflush(stdout);
      }
    }
    hipFree(d_ops); hipFree(d_out);
  }
  return 0;
}
