// Store-path probe for the GEMM epilogue (tool, not part of the library): 256 workgroups x 4 waves write 256x256 bf16
// tiles of a [M][N] matrix with 16-byte stores in different lane->address mappings; reports bytes/clk/CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip && tools/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int PAT, bool NT>
__global__ __launch_bounds__(256, 1) void store_kernel(uint16_t* C, int N, int tilesN, int ntiles, int valu, int hot) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q4 = lane >> 4, wm = wid >> 1, wn = wid & 1;
  uint4 v = make_uint4(tid, tid * 3, tid * 5, tid * 7);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // hot: every workgroup rewrites ONE 64 KB region of its own (2 MB per XCD: stays in that XCD's 4 MB L2)
    const int m0 = hot ? blockIdx.x * 256 : (t / tilesN) * 256, n0 = hot ? 0 : (t % tilesN) * 256;
    for (int nhs = 0; nhs < (hot ? 1 : 2); ++nhs)
      for (int mi = 0; mi < 8; ++mi) {
        for (int i = 0; i < valu; ++i) {   // stand-in for the epilogue arithmetic between the stores
          v.x = v.x * 1664525u + 1013904223u; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          int row, col;
          if (PAT == 0) {          // current epilogue: 16 rows x 64 B per instruction
            row = mi * 16 + l15; col = (2 * pr + (q4 & 1)) * 16 + (q4 >> 1) * 8;
          } else if (PAT == 1) {   // full lines: 8 rows x 128 B per instruction
            row = mi * 16 + pr * 8 + (l15 & 7); col = ((l15 >> 3) * 4 + q4) * 8;
          } else {                 // 4 rows x 256 B (both slices of a row; not reachable from the MFMA layout cheaply)
            row = mi * 16 + pr * 8 + (l15 & 3) + 4 * nhs; col = -nhs * 64 + ((l15 >> 2) * 4 + q4) * 8;
          }
          uint4* p = reinterpret_cast<uint4*>(C + (size_t)(m0 + wm * 128 + row) * N + n0 + (wn * 2 + nhs) * 64 + col);
          typedef unsigned int u4 __attribute__((ext_vector_type(4)));
          const u4 vv = {v.x, v.y, v.z, v.w};
          if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<u4*>(p)); else *reinterpret_cast<u4*>(p) = vv;
        }
      }
  }
}

// loads (L2-resident source, 8 rows x 128 B per instruction like the GEMM's staging loads) and stores (L2-resident target)
// issued by the same waves: do the two directions share the CU's ~16 B/clk store path?
template <int LD, int ST>
__global__ __launch_bounds__(256, 1) void mix_kernel(const uint16_t* A, uint16_t* C, int iters, uint32_t* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4* src = reinterpret_cast<const u4*>(A + (size_t)blockIdx.x * 32768) + wid * 1024 + lane;   // 64 KB per workgroup
  u4* dst = reinterpret_cast<u4*>(C + (size_t)blockIdx.x * 32768) + wid * 1024 + lane;
  u4 acc = {0, 0, 0, 0}, v = {(unsigned)tid, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < LD) { const u4 x = __builtin_nontemporal_load(src + ((it + j) & 15) * 64); acc ^= x; }
      if (j < ST) dst[((it + j) & 15) * 64] = v;
    }
  }
  if (acc.x == 0x12345678u) sink[0] = acc.y;
}

// load shapes: one wave instruction = 1 KB contiguous, or 8 rows x 128 B at a 2 KB / 8 KB row pitch (the GEMM's staging
// loads of a [M][K] row-major operand, K = 1024 / 4096), source L2-resident (256 KB per workgroup) or streamed from HBM
template <int PITCH>
__global__ __launch_bounds__(256, 1) void load_kernel(const uint16_t* A, int iters, size_t wg_stride, int wrap, uint32_t* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const char* base = reinterpret_cast<const char*>(A) + (size_t)blockIdx.x * wg_stride;
  const int loff = PITCH == 0 ? lane * 16 : (lane >> 3) * PITCH + (lane & 7) * 16;
  u4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const int blk = it % wrap;
#pragma unroll
    for (int j = 0; j < 16; ++j) {   // 16 pieces of 1 KB per wave and iteration = 64 KB per workgroup, like one K-step
      const int piece = wid * 16 + j;
      const size_t off = PITCH == 0 ? ((size_t)blk * 64 + piece) * 1024 : ((size_t)(piece * 8) * PITCH + (size_t)blk * 128);
      acc ^= __builtin_nontemporal_load(reinterpret_cast<const u4*>(base + off + loff));
    }
  }
  if (acc.x == 0x12345678u) sink[0] = acc.y;
}

// cost of publishing a workgroup's results to the rest of the chip inside one launch (agent-scope release + atomic):
// every workgroup writes 64 KB of its own, then FENCE: __threadfence() + atomicAdd on a per-group counter
template <bool FENCE>
__global__ __launch_bounds__(256, 1) void publish_kernel(uint16_t* C, unsigned* counters, int iters) {
  const int tid = threadIdx.x;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4* dst = reinterpret_cast<u4*>(C + (size_t)blockIdx.x * 32768);
  const u4 v = {(unsigned)tid, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * 256 + tid] = v;
    if (FENCE) {
      __threadfence();
      __syncthreads();
      if (tid == 0) atomicAdd(counters + (blockIdx.x >> 3), 1u);
    }
  }
}

int main() {
  const int M = 257 * 1024, N = 3072, tilesN = N / 256, ntiles = (M / 256) * tilesN;
  uint16_t* C;
  if (hipMalloc(&C, (size_t)M * N * 2) != hipSuccess) return 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int valu, int nt) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, N, tilesN, nt, valu, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, N, tilesN, nt, valu, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)nt * 256 * 256 * 2;
    printf("%-34s valu %3d tiles %5d: %7.3f ms  %6.2f TB/s  %6.2f us/tile-round  (%5.1f B/clk/CU at 2.0 GHz)\n", name, valu, nt, ms,
           bytes / ms / 1e9, ms * 1e3 / (nt / 256.0), bytes / 256 / (ms * 1e-3 * 2.0e9));
  };
  for (int valu : {0, 8, 24}) {
    run("16 rows x 64 B (current)", store_kernel<0, false>, valu, ntiles);
    run("8 rows x 128 B (full lines)", store_kernel<1, false>, valu, ntiles);
    run("4 rows x 256 B", store_kernel<2, false>, valu, ntiles);
    run("16 rows x 64 B, nontemporal", store_kernel<0, true>, valu, ntiles);
    run("8 rows x 128 B, nontemporal", store_kernel<1, true>, valu, ntiles);
  }
  // L2-resident target: the CU -> L2 store path alone (bytes = half a tile per iteration)
  auto runhot = [&](const char* name, auto kern, int valu) {
    const int nt = 256 * 200;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, 3072, 12, nt, valu, 1);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, 3072, 12, nt, valu, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)nt * 256 * 128 * 2;
    printf("HOT %-30s valu %3d: %7.3f ms  %6.2f TB/s  (%5.1f B/clk/CU at 2.0 GHz)\n", name, valu, ms, bytes / ms / 1e9,
           bytes / 256 / (ms * 1e-3 * 2.0e9));
  };
  for (int valu : {0, 8, 24}) {
    runhot("16 rows x 64 B", store_kernel<0, false>, valu);
    runhot("8 rows x 128 B", store_kernel<1, false>, valu);
    runhot("16 rows x 64 B, nontemporal", store_kernel<0, true>, valu);
    runhot("8 rows x 128 B, nontemporal", store_kernel<1, true>, valu);
  }
  {
    uint32_t* sink; hipMalloc(&sink, 64);
    auto runmix = [&](const char* name, auto kern, int ld, int st) {
      const int iters = 2000;
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, C + 16 * 1024 * 1024, iters, sink);
      hipEventRecord(e0);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, C + 16 * 1024 * 1024, iters, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      const double lb = (double)iters * ld * 4096 * 256, sb = (double)iters * st * 4096 * 256;
      printf("MIX %-26s: %7.3f ms  loads %5.1f B/clk/CU  stores %5.1f B/clk/CU (at 2.0 GHz)\n", name, ms, lb / 256 / (ms * 1e-3 * 2.0e9),
             sb / 256 / (ms * 1e-3 * 2.0e9));
    };
    runmix("16 loads", mix_kernel<16, 0>, 16, 0);
    runmix("16 stores", mix_kernel<0, 16>, 0, 16);
    runmix("16 loads + 16 stores", mix_kernel<16, 16>, 16, 16);
    runmix("16 loads + 4 stores", mix_kernel<16, 4>, 16, 4);
    runmix("16 loads + 2 stores", mix_kernel<16, 2>, 16, 2);
  }
  {
    uint32_t* sink; hipMalloc(&sink, 64);
    auto runld = [&](const char* name, auto kern, size_t wg_stride, int wrap) {
      const int iters = 1024;
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, iters, wg_stride, wrap, sink);
      hipEventRecord(e0);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, iters, wg_stride, wrap, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      const double b = (double)iters * 65536 * 256;
      printf("LOAD %-44s: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (at 2.0 GHz)\n", name, ms, b / ms / 1e9, b / 256 / (ms * 1e-3 * 2.0e9));
    };
    // L2-resident: each workgroup cycles over 64 KB (wrap 1)
    runld("1 KB contiguous, L2-resident", load_kernel<0>, 65536, 1);
    runld("8 rows x 128 B pitch 2 KB, L2-resident", load_kernel<2048>, 512 * 2048, 1);
    runld("8 rows x 128 B pitch 8 KB, L2-resident", load_kernel<8192>, 512 * 8192, 1);
    // streamed: every iteration touches new memory (wrap = many blocks); 1.6 GB buffer = 256 WGs x 6 MB
    runld("1 KB contiguous, streamed", load_kernel<0>, 6u << 20, 96);
    runld("8 rows x 128 B pitch 2 KB, streamed (16 K-steps)", load_kernel<2048>, 512 * 2048 * 4, 16);
    runld("8 rows x 128 B pitch 8 KB, streamed (64 K-steps)", load_kernel<8192>, 512 * 8192, 64);
  }
  {
    unsigned* ctr; hipMalloc(&ctr, 4096); hipMemset(ctr, 0, 4096);
    auto runpub = [&](const char* name, auto kern) {
      const int iters = 200;
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, ctr, iters);
      hipEventRecord(e0);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, C, ctr, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      printf("PUBLISH %-34s: %7.3f ms for %d x 64 KB per workgroup = %6.2f us per publication\n", name, ms, iters, ms * 1e3 / iters);
    };
    runpub("64 KB stores only", publish_kernel<false>);
    runpub("64 KB stores + fence + atomic", publish_kernel<true>);
  }
  // fewer workgroups: is ~16 B/clk a per-CU limit or the XCD's L2 write bandwidth shared by its 32 CUs?
  for (int grid : {8, 32, 64, 128, 256}) {
    const int per = 200;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((store_kernel<1, false>), dim3(grid), dim3(256), 0, 0, C, 3072, 12, grid * per, 0, 1);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_kernel<1, false>), dim3(grid), dim3(256), 0, 0, C, 3072, 12, grid * per, 0, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)grid * per * 256 * 128 * 2;
    printf("HOT grid %3d workgroups: %7.3f ms  %6.2f TB/s  (%5.1f B/clk/CU at 2.0 GHz)\n", grid, ms, bytes / ms / 1e9, bytes / grid / (ms * 1e-3 * 2.0e9));
  }
  return 0;
}
