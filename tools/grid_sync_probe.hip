// What does a grid-wide dependency cost INSIDE a launch, against a kernel boundary?  (round 6; VERDICT r5 #6 asked for the
// cooperative single-launch tail of the policy step to be measured before launch count is called the floor.)
//   chain   : N dependent launches of a tiny 256-workgroup kernel (each workgroup touches its own 1 KB) on one stream
//   fused   : ONE launch of the same N phases separated by a grid barrier (monotonic device-scope counter, release fence before
//             the arrive, relaxed polling + s_sleep, acquire fence after - the "barrier-counter" form; one workgroup per CU)
//   fused-x : the same with an XCD-hierarchical barrier (per-XCD counter -> leader -> top counter -> per-XCD generation word)
// Prints us per phase for each.  hipcc --offload-arch=gfx950 -O3 tools/grid_sync_probe.hip -o tools/grid_sync_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void phase_body(float* buf, int it) {
  float* p = buf + (size_t)blockIdx.x * 256 + threadIdx.x;
  *p = *p * 1.0001f + (float)it;
}

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int it) { phase_body(buf, it); }

__global__ __launch_bounds__(256) void fused_counter(float* buf, unsigned* ctr, int n) {
  const unsigned nwg = gridDim.x;
  for (int it = 0; it < n; ++it) {
    phase_body(buf, it);
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope (system by default: conservative)
      const unsigned target = (unsigned)(it + 1) * nwg;
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
  }
}

// XCD-hierarchical: workgroup b sits on XCD b & 7 (observed dispatch order; used for speed only - correctness needs only the counts)
__global__ __launch_bounds__(256) void fused_xcd(float* buf, unsigned* ctr, int n) {
  unsigned* xc = ctr + 64;            // 8 per-XCD arrival counters, 64 B apart
  unsigned* gen = ctr + 64 + 8 * 16;  // 8 per-XCD generation words
  unsigned* top = ctr;                // top-level counter
  const unsigned x = blockIdx.x & 7, per = gridDim.x >> 3;
  for (int it = 0; it < n; ++it) {
    phase_body(buf, it);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned prev = __hip_atomic_fetch_add(xc + x * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == (unsigned)(it + 1) * per - 1) {   // last arriver of this XCD: the leader of this round
        __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1) * 8u) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(gen + x * 16, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(gen + x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

int main() {
  const int nwg = 256, n = 16, reps = 200;
  float* buf; unsigned* ctr;
  hipMalloc(&buf, nwg * 256 * 4); hipMalloc(&ctr, 4096);
  hipMemset(buf, 0, nwg * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int warm = 0; warm < 2; ++warm) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
      for (int it = 0; it < n; ++it) hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), 0, 0, buf, it);
    hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("chain   : %6.2f us per phase (%d dependent launches of a 256-workgroup kernel, one stream)\n", ms * 1e3 / (reps * n), n);
  for (int form = 0; form < 2; ++form) {
    for (int warm = 0; warm < 2; ++warm) {
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(ctr, 0, 4096, 0);
        if (form == 0) hipLaunchKernelGGL(fused_counter, dim3(nwg), dim3(256), 0, 0, buf, ctr, n);
        else hipLaunchKernelGGL(fused_xcd, dim3(nwg), dim3(256), 0, 0, buf, ctr, n);
      }
      hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s: %6.2f us per phase (ONE launch, %d phases separated by a %s grid barrier; incl. the launch and a 4 KB memset per %d phases)\n",
           form == 0 ? "fused   " : "fused-x ", ms * 1e3 / (reps * n), n, form == 0 ? "single-counter" : "XCD-hierarchical", n);
  }
  return 0;
}
