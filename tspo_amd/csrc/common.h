// Shared host/device helpers for libtspo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/tspo_hip.h"

namespace tspo {

// ---- thread-local error string (no exceptions cross the ABI) --------------
char* err_buf();
int set_err(int code, const char* fmt, ...);
int check_launch(const char* what);

#define TSPO_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return ::tspo::set_err(TSPO_EINVAL, __VA_ARGS__); \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Carves sub-buffers out of the caller's workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* r = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return r;
  }
  size_t bytes() const { return align_up(off, 256); }
};

}  // namespace tspo

// ---- device helpers -------------------------------------------------------
#define WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide reductions through a small LDS scratch (>= 32 floats). All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

// bf16 <-> f32 (round-to-nearest-even), bit-level so no header type juggling.
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even)
typedef __attribute__((ext_vector_type(2))) float tspo_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 tspo_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const tspo_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, tspo_bf16x2));
}
