// bf16 MFMA GEMMs of the CLIP encoder for gfx950 (MI355X): C = A * W^T with fused epilogues.
// Production kernel: gemm_bf16_p256_kernel (persistent 256x256x64, LDS-DMA ring).  The other kernels are the
// small-problem path and A/B variants reachable through tspo_gemm_bf16's test hook.
#include "gemm_bf16.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
__host__ __device__ constexpr bool epi_has_bias(int epi) {
  return epi == GE_BIAS || epi == GE_GELU || epi == GE_RESID || epi == GE_BIAS_LN || epi == GE_GELU_LN || epi == GE_RESID_ST;
}


// ===========================================================================
// GEMM  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous, bf16)
// 128x128x64 workgroup tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32.
// LDS image per operand tile: [128 rows][128 B], 16-B chunk c of row r stored
// at chunk (c ^ (r & 7)) - the XOR is applied on the global SOURCE address of
// the LDS-DMA (destination must stay lane-linear) and again on the ds_read.
// The MFMA is issued "swapped" (A-operand = W fragment, B-operand = A fragment)
// so that each lane ends up with 4 consecutive N for one M -> 8-byte stores.
// ===========================================================================
#define GT_BM 128
#define GT_BN 128
#define GT_BK 64
#define GT_STAGE_BYTES (2 * 128 * 128)  // A tile + W tile, 16 KB each

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int rows_total, int row0, int K, int kt,
                                           char* lds_tile, int wid, int lane) {
  // 16 pieces of 1 KB (8 rows x 128 B); wave `wid` issues pieces wid*4 .. wid*4+3
  const int rin = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int piece = wid * 4 + p;
    const int r = piece * 8 + rin;
    int gr = row0 + r;
    gr = gr < rows_total ? gr : rows_total - 1;
    const int c = slot ^ (r & 7);
    const bf16_t* src = G + (size_t)gr * K + (size_t)kt * GT_BK + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

// x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)): one v_exp_f32 + one v_rcp_f32 (1 ulp; the result is rounded
// to bf16 anyway) instead of a full-precision division sequence
__device__ __forceinline__ float quick_gelu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * GT_STAGE_BYTES];  // 2 stages x (A 16K | W 16K); the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  // XCD-aware, bijective remap: consecutive workgroups of one XCD walk the N tiles of one M row-panel
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = g.nwg >> 3, rr = g.nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / g.tilesN, tn = wg - tm * g.tilesN;
  const int m0 = tm * GT_BM, n0 = tn * GT_BN;
  const int wm = wid >> 1, wn = wid & 1;
  const int nk = g.K / GT_BK;

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage_tile(g.A, g.M, m0, g.K, 0, lds, wid, lane);
  stage_tile(g.W, g.N, n0, g.K, 0, lds + 16384, wid, lane);

  // per-lane read offsets (row-dependent swizzle is loop invariant)
  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + l15;
    const int rw = wn * 64 + i * 16 + l15;
    offA[i] = ra * 128;
    offW[i] = 16384 + rw * 128;
  }
  const int sw = l15 & 7;  // (row & 7) == (l15 & 7) because every row base is a multiple of 16

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // stage kt landed (vmcnt(0) folded in by the compiler) and compute(kt-1) is done everywhere
    char* cur = lds + (kt & 1) * GT_STAGE_BYTES;
    if (kt + 1 < nk) {
      char* nxt = lds + ((kt + 1) & 1) * GT_STAGE_BYTES;
      stage_tile(g.A, g.M, m0, g.K, kt + 1, nxt, wid, lane);
      stage_tile(g.W, g.N, n0, g.K, kt + 1, nxt + 16384, wid, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q4) ^ sw) << 4);
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i] + co);
        fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  // epilogue: lane holds n = nb + q4*4 + r (r = 0..3), m = mb + l15
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + l15;
    if (m >= g.M) continue;
    size_t orow = (size_t)m;
    int prow = 0;
    if (EPI == GE_PATCH) {
      const int f = m / g.P;
      prow = 1 + (m - f * g.P);
      orow = (size_t)f * (g.P + 1) + prow;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
      if (n >= g.N) continue;
      f32x4 v = acc[ni][mi];
      if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        v += bv;
      }
      if (EPI == GE_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
      }
      if (EPI == GE_PATCH) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(g.pos + (size_t)prow * g.N + n);
        v += pv;
      }
      const size_t o = orow * g.N + n;
      if (EPI == GE_RESID) {
        const uint2 rv = *reinterpret_cast<const uint2*>(g.R + o);
        v[0] += bf16_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
        v[2] += bf16_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
      }
      if (EPI == GE_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
      } else {
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + o) = pk;
      }
    }
  }
}


// ===========================================================================
// GEMM v2: persistent 256x128x64, 8 waves (4x2, 64x64 each), 3-stage LDS ring
// (3 x 48 KB) filled by LDS-DMA that stays in flight across barriers (counted
// s_waitcnt vmcnt + raw s_barrier), and a ring that runs CONTINUOUSLY across the
// tiles a workgroup owns, so the next tile's first stages stream in under the
// current tile's last MFMAs and its epilogue stores.  Measured motivation
// (profiles/r1_a_*): with the 2-stage kernel above one K-step took ~3.3k cycles
// for ~1.1k cycles of MFMA because each step waited for its own loads.
// Tiles are dealt per XCD (blockIdx % 8 observed = XCD): the 32 workgroups of an
// XCD walk the N tiles of consecutive M panels together, so an A panel is read
// from HBM once per XCD and served from that XCD's L2 to the others.
// ===========================================================================
#define G2_BM 256
#define G2_BN 128
#define G2_STAGE (G2_BM * 128 + G2_BN * 128)  // 49152 B
#define G2_NSTAGE 3

__device__ __forceinline__ void g2_stage(const GemmArgs& g, int m0, int n0, int kt, char* buf, int wid, int lane) {
  const int rin = lane >> 3, slot = lane & 7;
  // A: 32 pieces of 8 rows; wave takes pieces wid*4..+3
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int piece = wid * 4 + p;
    const int r = piece * 8 + rin;
    int gr = m0 + r;
    gr = gr < g.M ? gr : g.M - 1;
    const bf16_t* src = g.A + (size_t)gr * g.K + (size_t)kt * GT_BK + ((slot ^ rin) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
  }
  // W: 16 pieces; wave takes pieces wid*2, wid*2+1
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int piece = wid * 2 + p;
    const int r = piece * 8 + rin;
    int gr = n0 + r;
    gr = gr < g.N ? gr : g.N - 1;
    const bf16_t* src = g.W + (size_t)gr * g.K + (size_t)kt * GT_BK + ((slot ^ rin) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(buf + G2_BM * 128 + piece * 1024), 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_p3_kernel(GemmArgs g, int tilesM, int ngrp) {
  // the ONLY LDS object: 3 stages + the whole bias vector (<= 4096 floats).  Keeping the bias in LDS matters:
  // an ordinary global load in the epilogue makes hipcc drain vmcnt(0), i.e. the LDS-DMA ring, at its first use.
  __shared__ __attribute__((aligned(16))) char lds[G2_NSTAGE * G2_STAGE + 16384];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  float* lbias = reinterpret_cast<float*>(lds + G2_NSTAGE * G2_STAGE);
  if (epi_has_bias(EPI))
    for (int i = tid; i < g.N; i += 512) lbias[i] = g.bias[i];
  const int nk = g.K / GT_BK;
  // Tile ownership per XCD (blockIdx % 8, observed placement - speed only).  The N tiles are split into `ngrp`
  // groups so that one XCD only ever touches W rows worth <= ~2.5 MB (its 4 MB L2 keeps them resident instead of
  // cycling the whole W through LRU), and the M panels are dealt round-robin over the 8/ngrp XCDs of a group.
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  const int total_it = my_tiles * nk;
  if (total_it == 0) return;

  // issue-side cursor (runs 2 stages ahead of the compute cursor)
  int i_it = 0, i_kt = 0, i_s = wl;
  int i_m0 = ((i_s / n_per) * npset + pset) * G2_BM, i_n0 = (grp * n_per + i_s % n_per) * G2_BN;
  int i_rot = g.P < 0 ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
  auto issue_next = [&]() {
    // K-rotation: the workgroups that share an A panel (different N tiles, same XCD) start their K loops at
    // different offsets, so each 64-wide K slice of the panel is pulled from HBM by ONE of them while the others
    // find it in the XCD's L2 instead of all of them stalling on the same miss together.
    int kt_eff = i_kt + i_rot;
    kt_eff = kt_eff >= nk ? kt_eff - nk : kt_eff;
    g2_stage(g, i_m0, i_n0, kt_eff, lds + (i_it % G2_NSTAGE) * G2_STAGE, wid, lane);
    ++i_it;
    if (++i_kt == nk) {
      i_kt = 0;
      i_s += nwl;
      i_m0 = ((i_s / n_per) * npset + pset) * G2_BM;
      i_n0 = (grp * n_per + i_s % n_per) * G2_BN;
      i_rot = g.P < 0 ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
    }
  };
  issue_next();
  if (total_it > 1) issue_next();

  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    offA[i] = (wm * 64 + i * 16 + l15) * 128;
    offW[i] = G2_BM * 128 + (wn * 64 + i * 16 + l15) * 128;
  }
  const int sw = l15 & 7;

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int c_kt = 0, c_s = wl, stage = 0;
  bool drained = false;  // true right after an epilogue: its stores share the VM counter with the loads
  for (int it = 0; it < total_it; ++it) {
    // stage `it` must have landed for every wave; stage it+1 (6 LDS-DMA ops per wave) may stay in flight
    if (it + 1 < total_it && !drained) {
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    drained = false;
    if (i_it < total_it && g.P != -2) issue_next();  // refills the buffer whose reads finished before the barrier above
    const char* cur = lds + stage * G2_STAGE;
    if (g.P != -3)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q4) ^ sw) << 4);
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i] + co);
        fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
    stage = stage == G2_NSTAGE - 1 ? 0 : stage + 1;
    if (++c_kt == nk) {
      // ---- epilogue of tile c_s (the ring keeps streaming the next tile meanwhile) ----
      const int m0 = ((c_s / n_per) * npset + pset) * G2_BM, n0 = (grp * n_per + c_s % n_per) * G2_BN;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + l15;
        size_t orow = (size_t)m;
        int prow = 0;
        if (EPI == GE_PATCH) {
          const int f = m / g.P;
          prow = 1 + (m - f * g.P);
          orow = (size_t)f * (g.P + 1) + prow;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
          f32x4 v = acc[ni][mi];
          acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (m >= g.M || n >= g.N) continue;
          if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID) v += *reinterpret_cast<const f32x4*>(lbias + n);
          if (EPI == GE_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
          }
          if (EPI == GE_PATCH) v += *reinterpret_cast<const f32x4*>(g.pos + (size_t)prow * g.N + n);
          const size_t o = orow * g.N + n;
          if (EPI == GE_RESID) {
            const uint2 rv = *reinterpret_cast<const uint2*>(g.R + o);
            v[0] += bf16_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
            v[2] += bf16_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
          }
          if (EPI == GE_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
          } else {
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + o) = pk;
          }
        }
      }
      c_kt = 0;
      c_s += nwl;
      drained = true;
    }
  }
}

template <int EPI>
int launch_gemm_p3(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G2_BM - 1) / G2_BM;
  g.tilesN = (g.N + G2_BN - 1) / G2_BN;
  g.nwg = tilesM * g.tilesN;
  int grid = 256;  // one persistent workgroup per CU (160 KB of LDS each)
  int ngrp = 1;
  const double wbytes = (double)g.N * g.K * 2.0;
  while (ngrp < 8 && wbytes / ngrp > 2.5e6 && g.tilesN % (ngrp * 2) == 0) ngrp *= 2;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_p3_kernel<EPI>), dim3(grid), dim3(512), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_p3");
}

template <int EPI>
int launch_gemm_v1(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + GT_BM - 1) / GT_BM;
  g.tilesN = (g.N + GT_BN - 1) / GT_BN;
  g.nwg = tilesM * g.tilesN;
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(g.nwg), dim3(256), 0, st, g);
  return tspo::check_launch("gemm_bf16");
}


// ===========================================================================
// GEMM v3: persistent 256x256x64, 8 waves (2x4, 128x64 each = 8x4 MFMA tiles, 128 fp32 accumulators per lane),
// 2-stage LDS ring (2 x 64 KB) + bias (16 KB) = 144 KB.  Why: measured on the 256x128 kernel, the LDS-DMA stream
// alone tops out at ~12.6 TB/s chip-wide (~26 B/clk/CU) whatever the L2 hit rate, i.e. ~1.07 PFLOP/s at the
// 85 FLOP/B of a 256x128 tile; a 256x256 tile needs 128 FLOP/B (ceiling ~1.6 PFLOP/s) and reads 25 % fewer LDS
// bytes per MFMA.  Same continuous ring across the tiles a workgroup owns, same XCD/N-group ownership, same K-rotation.
// ===========================================================================
#define G3_BM 256
#define G3_BN 256
#define G3_STAGE (G3_BM * 128 + G3_BN * 128)  // 65536 B

// One stage = 64 LDS-DMA pieces of 1 KB.  Only the 4 waves of ONE wave-row (one per SIMD) issue them, the row
// alternating every K-step: an LDS-DMA instruction costs its issuing wave ~60-180 cycles, so while a loader wave
// is busy issuing, its SIMD partner (the other wave-row) has the matrix pipe to itself instead of both waves
// queueing DMA issues and then both queueing MFMAs.
__device__ __forceinline__ void g3_stage(const GemmArgs& g, int m0, int n0, int kt, char* buf, int j, int lane) {
  const int rin = lane >> 3, slot = lane & 7;
  const size_t koff = (size_t)kt * GT_BK + ((slot ^ rin) << 3);
#pragma unroll
  for (int p = 0; p < 8; ++p) {  // A: 32 pieces of 8 rows, 8 per loader wave
    const int piece = j * 8 + p;
    int gr = m0 + piece * 8 + rin;
    gr = gr < g.M ? gr : g.M - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.A + (size_t)gr * g.K + koff),
                                     (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
  }
  if (g.P == -4) return;  // test hook: A half only
#pragma unroll
  for (int p = 0; p < 8; ++p) {  // W: 32 pieces of 8 rows
    const int piece = j * 8 + p;
    int gr = n0 + piece * 8 + rin;
    gr = gr < g.N ? gr : g.N - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W + (size_t)gr * g.K + koff),
                                     (__attribute__((address_space(3))) void*)(buf + G3_BM * 128 + piece * 1024), 16, 0, 0);
  }
}

// Epilogue of one 256x256 tile, shared by the ring kernels.  A wave's columns come in 64-column slices (4 MFMA tiles
// of 16): g3_epi_cols holds what a slice needs once per tile, g3_epi_row finishes ONE 16-row block of a slice (this
// lane: row m, 4 x 4 consecutive columns) and g3_epilogue_t walks MI row blocks of accumulators held in VGPRs.
struct EpiCols { f32x4 lc[4]; f32x4 bias[4]; };   // bias[] only for the PRE (register-prefetched) form of g3_epi_row
template <int EPI>
__device__ __forceinline__ void g3_epi_cols(const GemmArgs& g, int n0, int wn, int q4, EpiCols& ec) {
  if (EPI == GE_BIAS_LN || EPI == GE_GELU_LN) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
      ec.lc[ni] = *reinterpret_cast<const f32x4*>(g.lnc + (n < g.N ? n : 0));
    }
  }
}
__device__ __forceinline__ float2 g3_epi_rowstat(const GemmArgs& g, int m) {   // (rstd, -mean*rstd) of row m
  return *reinterpret_cast<const float2*>(g.rstats + 2 * (size_t)(m < g.M ? m : g.M - 1));
}

// v[ni] = the 4 accumulators of column tile ni for row m (lane l15 of the 16-row block), rst = g3_epi_rowstat(m)
// PRE: bias / folded bias comes from ec.bias and the residual from rpre[ni] (this lane's 4 bf16 of column tile ni, MFMA
// layout) - both fetched by the caller ahead of time - instead of being loaded here.
template <int EPI, bool PRE = false>
__device__ __forceinline__ void g3_epi_row(const GemmArgs& g, f32x4 (&vv)[4], const EpiCols& ec, float2 rst, int m, int n0,
                                           int wn, int q4, const float* lbias, const uint2* rpre = nullptr) {
  constexpr bool LN = EPI == GE_BIAS_LN || EPI == GE_GELU_LN;
  constexpr bool RES = EPI == GE_RESID || EPI == GE_RESID_ST;
  size_t orow = (size_t)m;
  int prow = 0;
  if (EPI == GE_PATCH) {
    const int f = m / g.P;
    prow = 1 + (m - f * g.P);
    orow = (size_t)f * (g.P + 1) + prow;
  }
  const float rs = rst.x, mu = rst.y;   // rstd, -mean * rstd
  float ssum = 0.f, ssq = 0.f;
  uint2 pk[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
    f32x4 v = vv[ni];
    const bool ok = m < g.M && n < g.N;
    if (LN) {   // rstats holds (rstd, -mean*rstd): y = acc*rstd + (-mean*rstd)*c[n] + d[n], two FMAs per value
      const f32x4 dv = PRE ? ec.bias[ni] : *reinterpret_cast<const f32x4*>(lbias + (n < g.N ? n : 0));
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], rs, fmaf(mu, ec.lc[ni][r], dv[r]));
    } else if (epi_has_bias(EPI)) {
      v += PRE ? ec.bias[ni] : *reinterpret_cast<const f32x4*>(lbias + (n < g.N ? n : 0));
    }
    if (EPI == GE_GELU || EPI == GE_GELU_LN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
    }
    const size_t o = orow * g.N + n;
    if (EPI == GE_PATCH && ok) v += *reinterpret_cast<const f32x4*>(g.pos + (size_t)prow * g.N + n);
    if (RES && (PRE || ok)) {
      const uint2 rv = PRE ? rpre[ni] : *reinterpret_cast<const uint2*>(g.R + o);
      v[0] += bf16_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
      v[2] += bf16_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
    }
    if (EPI == GE_F32) {
      if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
    } else {
      pk[ni].x = pack_bf16x2(v[0], v[1]);
      pk[ni].y = pack_bf16x2(v[2], v[3]);
      if (EPI == GE_RESID_ST && ok) {   // statistics of the values as stored (bf16-rounded): what the next GEMM reads
        const float r0 = __uint_as_float(pk[ni].x << 16), r1 = __uint_as_float(pk[ni].x & 0xffff0000u);
        const float r2 = __uint_as_float(pk[ni].y << 16), r3 = __uint_as_float(pk[ni].y & 0xffff0000u);
        ssum += (r0 + r1) + (r2 + r3);
      }
    }
  }
  if (EPI == GE_RESID_ST) {   // the row's 64 columns of this slice live in the 4 lanes that share l15
    // per-slice (mean, centred sum of squares): two passes over the 16 stored values of this lane, so the later
    // combination of the N/64 slices (Chan et al.) is as robust as a two-pass LayerNorm
    ssum += __shfl_xor(ssum, 16, 64);
    ssum += __shfl_xor(ssum, 32, 64);
    const float smean = ssum * (1.0f / 64.0f);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const float d0 = __uint_as_float(pk[ni].x << 16) - smean, d1 = __uint_as_float(pk[ni].x & 0xffff0000u) - smean;
      const float d2 = __uint_as_float(pk[ni].y << 16) - smean, d3 = __uint_as_float(pk[ni].y & 0xffff0000u) - smean;
      ssq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    ssq += __shfl_xor(ssq, 16, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    const int cslice = (n0 >> 6) + wn;
    if (q4 == 0 && m < g.M && cslice * 64 < g.N)
      *reinterpret_cast<float2*>(g.spart + ((size_t)m * (g.N >> 6) + cslice) * 2) = make_float2(smean, ssq);
  }
  if (EPI != GE_F32) {
    // widen the stores: v_permlane16_swap exchanges the odd 16-lane rows of tile a with the even rows of tile
    // b, after which row q4 holds 16 contiguous bytes of tile (q4 & 1 ? b : a) at column (q4 >> 1) * 8
    // -> 16 instead of 32 store instructions per wave and tile (the epilogue is store-issue bound)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const auto w0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr].x, pk[2 * pr + 1].x, false, false);
      const auto w1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr].y, pk[2 * pr + 1].y, false, false);
      const int n = n0 + wn * 64 + (2 * pr + (q4 & 1)) * 16 + (q4 >> 1) * 8;
      if (m < g.M && n < g.N) {
        uint4 st;
        st.x = w0[0]; st.y = w1[0]; st.z = w0[1]; st.w = w1[1];
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + orow * g.N + n) = st;
      }
    }
  }
}

template <int EPI, int MI>
__device__ __forceinline__ void g3_epilogue_t(const GemmArgs& g, f32x4 (&acc)[4][MI], int m0, int n0, int rbase, int wn,
                                              int l15, int q4, const float* lbias) {
  constexpr bool LN = EPI == GE_BIAS_LN || EPI == GE_GELU_LN;
  EpiCols ec;
  g3_epi_cols<EPI>(g, n0, wn, q4, ec);
  float2 rst[MI];   // (rstd, -mean*rstd) of this lane's MI rows: all loads in flight together (one latency, not MI)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) rst[mi] = LN ? g3_epi_rowstat(g, m0 + rbase + mi * 16 + l15) : make_float2(1.f, 0.f);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    f32x4 vv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      vv[ni] = acc[ni][mi];
      acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    g3_epi_row<EPI>(g, vv, ec, rst[mi], m0 + rbase + mi * 16 + l15, n0, wn, q4, lbias);
  }
}

template <int EPI>
__device__ __forceinline__ void g3_epilogue(const GemmArgs& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn,
                                            int l15, int q4, const float* lbias) {
  g3_epilogue_t<EPI, 8>(g, acc, m0, n0, wm * 128, wn, l15, q4, lbias);
}

template <int EPI, int MODE, int PF, int EARLY>
__global__ __launch_bounds__(512) void gemm_bf16_p256_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE + 16384 + 256];  // the ONLY LDS object (+256 B sink of the L2 prefetch)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 2, wn = wid & 3;
  float* lbias = reinterpret_cast<float*>(lds + 2 * G3_STAGE);
  if (epi_has_bias(EPI))
    for (int i = tid; i < g.N; i += 512) lbias[i] = g.bias[i];
  const int nk = g.K / GT_BK;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((size_t)g.M * g.K * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  const int total_it = my_tiles * nk;
  if (total_it == 0) return;

  int i_it = 0, i_kt = 0, i_s = wl;
  int i_m0 = ((i_s / n_per) * npset + pset) * G3_BM, i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
  int i_rot = g.P != -6 ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
  auto issue_next = [&]() {
    int kt_eff = i_kt + i_rot;
    kt_eff = kt_eff >= nk ? kt_eff - nk : kt_eff;
    if ((wid >> 2) == (i_it & 1)) g3_stage(g, i_m0, i_n0, kt_eff, lds + (i_it & 1) * G3_STAGE, wid & 3, lane);
    ++i_it;
    if (++i_kt == nk) {
      i_kt = 0;
      i_s += nwl;
      i_m0 = ((i_s / n_per) * npset + pset) * G3_BM;
      i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
      i_rot = g.P != -6 ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
    }
  };
  issue_next();

  int offA[8], offW[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) offA[i] = (wm * 128 + i * 16 + l15) * 128;
#pragma unroll
  for (int i = 0; i < 4; ++i) offW[i] = G3_BM * 128 + (wn * 64 + i * 16 + l15) * 128;
  const int sw = l15 & 7;

  f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int c_kt = 0, c_s = wl;
  unsigned long long dbg_vm = 0, dbg_bar = 0;
  const unsigned long long dbg_t0 = EARLY == 4 ? __builtin_readcyclecounter() : 0;
  bool after_epi = true;   // first wait: nothing but the first stage is outstanding
  for (int it = 0; it < total_it; ++it) {
    // stage `it` landed everywhere; buffer (it+1)&1 is free.  Wave 0 may leave its 4 (younger) L2-prefetch ops in flight,
    // except right after an epilogue whose stores are younger still.
    unsigned long long tw0 = 0;
    if (EARLY == 4) tw0 = __builtin_readcyclecounter();
    if (PF > 0 && (wid == 0 || (PF > 100 && wid == 1)) && !after_epi) {
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      if (EARLY == 4) { const unsigned long long t1 = __builtin_readcyclecounter(); dbg_vm += t1 - tw0; tw0 = t1; }
      asm volatile("s_barrier" ::: "memory");
    } else if (g.P != -7) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (EARLY == 4) { const unsigned long long t1 = __builtin_readcyclecounter(); dbg_vm += t1 - tw0; tw0 = t1; }
      asm volatile("s_barrier" ::: "memory");
    }
    if (EARLY == 4) dbg_bar += __builtin_readcyclecounter() - tw0;
    after_epi = false;
    const char* cur = lds + (it & 1) * G3_STAGE;
    if (MODE == 1) {
      // default: every wave issues its own 8 pieces, two behind each group of 16 MFMAs of the first half K-step
      const bool more = i_it < total_it && g.P > -2;
      int kt_eff = i_kt + i_rot;
      kt_eff = kt_eff >= nk ? kt_eff - nk : kt_eff;
      char* nbuf = lds + (i_it & 1) * G3_STAGE;
      const int rin = lane >> 3, slot = lane & 7;
      const size_t koff = (size_t)kt_eff * GT_BK + ((slot ^ rin) << 3);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int co = (((kk * 4 + q4) ^ sw) << 4);
        bf16x8 fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
          bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(cur + offA[2 * mp] + co);
          bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(cur + offA[2 * mp + 1] + co);
          if (more && kk == 0 && (EARLY != 1 || mp < 2))
#pragma unroll
          for (int e = 0; e < (EARLY == 1 ? 2 : 1); ++e) {
            // the 8 pieces of this wave go out during the FIRST half of the K-step (2 per group of 16 MFMAs) so the
            // last one still has half a K-step of MFMAs to land behind  (EARLY: 4 per group, first quarter)
            const int piece = wid * 4 + (EARLY == 1 ? mp * 2 + e : mp);
            if (EARLY == 3) {
              // A/B: buffer_load ... lds through a resource descriptor (32-bit offsets, hardware range check instead
              // of the row clamp)
              const unsigned oa = ((unsigned)(i_m0 + piece * 8 + rin) * (unsigned)g.K + (unsigned)koff) * 2u;
              const unsigned ow = ((unsigned)(i_n0 + piece * 8 + rin) * (unsigned)g.K + (unsigned)koff) * 2u;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(nbuf + piece * 1024), 16, oa, 0, 0, 0);
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(nbuf + G3_BM * 128 + piece * 1024), 16, ow, 0, 0, 0);
            } else {
            int gr = i_m0 + piece * 8 + rin;
            gr = gr < g.M ? gr : g.M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.A + (size_t)gr * g.K + koff),
                                             (__attribute__((address_space(3))) void*)(nbuf + piece * 1024), 16, 0, 0);
            gr = i_n0 + piece * 8 + rin;
            gr = gr < g.N ? gr : g.N - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W + (size_t)gr * g.K + koff),
                                             (__attribute__((address_space(3))) void*)(nbuf + G3_BM * 128 + piece * 1024), 16, 0, 0);
            }
          }
          if (EARLY == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            acc[ni][2 * mp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa0, acc[ni][2 * mp], 0, 0, 0);
            acc[ni][2 * mp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa1, acc[ni][2 * mp + 1], 0, 0, 0);
          }
          if (EARLY == 2) __builtin_amdgcn_s_setprio(0);
          if (PF > 100 && more && kk == 0 && mp == 3 && wid == 1) {   // A/B probe: also prefetch the W slice (wave 1)
            int p_kt = i_kt + (PF - 100), p_n0 = i_n0;
            const bool pv = i_it + (PF - 100) < total_it;
            if (p_kt >= nk) {
              p_kt -= nk;
              const int s2 = i_s + nwl;
              p_n0 = (grp * n_per + s2 % n_per) * G3_BN;
            }
            if (pv) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                int gr = p_n0 + j * 64 + lane;
                gr = gr < g.N ? gr : g.N - 1;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(g.W + (size_t)gr * g.K + (size_t)p_kt * GT_BK),
                    (__attribute__((address_space(3))) void*)(lds + 2 * G3_STAGE + 16384), 4, 0, 0);
              }
            }
          }
          if (PF > 0 && more && kk == 0 && mp == 3 && wid == 0) {
            // L2 prefetch of the A slice this workgroup will stage PF K-steps from now: one 4-byte LDS-DMA per
            // 128-B line (64 lines per instruction, destination = a 256-B sink), so the real 16-B pieces issued PF
            // steps later find the first-touch lines of the panel in L2 instead of waiting on HBM
            const int PFD = PF > 100 ? PF - 100 : PF;
            int p_kt = i_kt + PFD, p_m0 = i_m0;
            bool pv = i_it + PFD < total_it;
            if (p_kt >= nk) {
              p_kt -= nk;
              const int s2 = i_s + nwl;
              p_m0 = ((s2 / n_per) * npset + pset) * G3_BM;
            }
            if (pv) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                int gr = p_m0 + j * 64 + lane;
                gr = gr < g.M ? gr : g.M - 1;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(g.A + (size_t)gr * g.K + (size_t)p_kt * GT_BK),
                    (__attribute__((address_space(3))) void*)(lds + 2 * G3_STAGE + 16384), 4, 0, 0);
              }
            }
          }
          asm volatile("" ::: "memory");
        }
      }
      if (more) {
        ++i_it;
        if (++i_kt == nk) {
          i_kt = 0;
          i_s += nwl;
          i_m0 = ((i_s / n_per) * npset + pset) * G3_BM;
          i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
          i_rot = g.P != -6 ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
        }
      }
    } else {
    if (i_it < total_it && g.P != -2) issue_next();
    if (g.P > -3)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q4) ^ sw) << 4);
      bf16x8 fa[8], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i] + co);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
    }
    if (++c_kt == nk) {
      const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
      g3_epilogue<EPI>(g, acc, m0, n0, wm, wn, l15, q4, lbias);
      c_kt = 0;
      c_s += nwl;
      after_epi = true;
    }
  }
  if (EARLY == 4 && lane == 0 && g.pos) {   // timing probe: per wave {cycles in vmcnt wait, cycles in barrier, total, K-steps}
    float* d = const_cast<float*>(g.pos) + (blockIdx.x * 8 + wid) * 4;
    d[0] = (float)dbg_vm; d[1] = (float)dbg_bar; d[2] = (float)(__builtin_readcyclecounter() - dbg_t0); d[3] = (float)total_it;
  }
}

template <int EPI, int MODE = 0, int PF = 0, int EARLY = 0>
int launch_gemm_p256(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  // N groups: 2 when W is too big for one XCD's 4 MB L2 and there are enough N tiles to split (measured: QKV 928 vs
  // 912 TFLOP/s, fc1 875 vs 872; fc2 / out-proj are best un-split).  K-rotation between the workgroups of a panel is
  // OFF: with it the FETCH_SIZE counter showed ~7 GB of L2 misses for a GEMM whose operands are 0.55 GB (the
  // workgroups sharing an A panel no longer touched the same lines at the same time), and it ran 6-10 % slower.
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_p256_kernel<EPI, MODE, PF, EARLY>), dim3(256), dim3(512), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_p256");
}


// ===========================================================================
// GEMM v4 ("role-split"): same 256x256x64 tile, ring, ownership and epilogue as v3, but the two wave-rows of the
// workgroup run HALF A K-STEP OUT OF PHASE.  Waves w and w+4 share a SIMD; while one of them issues its 32 MFMAs
// of a half K-step back to back (all fragments already in registers, s_setprio 1), the other one is in its LOAD
// segment: 12 ds_read_b128 for its next half K-step plus its LDS-DMA pieces for the next stage.  So the matrix pipe
// of a SIMD is fed by exactly one wave at a time and never waits behind LDS reads or DMA issue of that same wave.
// Segments are separated by workgroup barriers (4 per K-step); group B (waves 4-7) takes one extra barrier up
// front, group A one at the end.  Stage it+1 is issued by each wave in its LOAD segment of the first half of
// K-step it - the first point at which every read of the buffer's previous contents is known to be complete - and
// every wave drains its own DMA (vmcnt(0)) before the barrier that closes global segment 4*it+3.
// ===========================================================================
#define G4_BAR() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G4_BAR_VM() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_s256_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE + 16384 + 256];  // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 2, wn = wid & 3;   // wm = wave-row = phase group (0: A leads, 1: B trails by one segment)
  float* lbias = reinterpret_cast<float*>(lds + 2 * G3_STAGE);
  if (epi_has_bias(EPI))
    for (int i = tid; i < g.N; i += 512) lbias[i] = g.bias[i];
  const int nk = g.K / GT_BK;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  const int total_it = my_tiles * nk;
  if (total_it == 0) return;

  // issue-side cursor: stage index i_it of tile i_s, K-step i_kt
  int i_it = 0, i_kt = 0, i_s = wl;
  int i_m0 = ((i_s / n_per) * npset + pset) * G3_BM, i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
  const int rin = lane >> 3, slot = lane & 7;
  auto issue_stage = [&]() {   // this wave's 8 pieces (4 A + 4 W) of stage i_it, then advance the cursor
    char* nbuf = lds + (i_it & 1) * G3_STAGE;
    const size_t koff = (size_t)i_kt * GT_BK + ((slot ^ rin) << 3);
#pragma unroll
    for (int pce = 0; pce < 4; ++pce) {
      const int piece = wid * 4 + pce;
      int gr = i_m0 + piece * 8 + rin;
      gr = gr < g.M ? gr : g.M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.A + (size_t)gr * g.K + koff),
                                       (__attribute__((address_space(3))) void*)(nbuf + piece * 1024), 16, 0, 0);
      gr = i_n0 + piece * 8 + rin;
      gr = gr < g.N ? gr : g.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W + (size_t)gr * g.K + koff),
                                       (__attribute__((address_space(3))) void*)(nbuf + G3_BM * 128 + piece * 1024), 16, 0, 0);
    }
    if (wid == 0) {  // L2 prefetch of the A slice 6 K-steps ahead (4-byte LDS-DMA per 128-B line into a sink)
      int p_kt = i_kt + 6, p_m0 = i_m0;
      if (p_kt >= nk) {
        p_kt -= nk;
        const int s2 = i_s + nwl;
        p_m0 = ((s2 / n_per) * npset + pset) * G3_BM;
      }
      if (i_it + 6 < total_it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int gr = p_m0 + j * 64 + lane;
          gr = gr < g.M ? gr : g.M - 1;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(g.A + (size_t)gr * g.K + (size_t)p_kt * GT_BK),
              (__attribute__((address_space(3))) void*)(lds + 2 * G3_STAGE + 16384), 4, 0, 0);
        }
      }
    }
    ++i_it;
    if (++i_kt == nk) {
      i_kt = 0;
      i_s += nwl;
      i_m0 = ((i_s / n_per) * npset + pset) * G3_BM;
      i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
    }
  };

  int offA[8], offW[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) offA[i] = (wm * 128 + i * 16 + l15) * 128;
#pragma unroll
  for (int i = 0; i < 4; ++i) offW[i] = G3_BM * 128 + (wn * 64 + i * 16 + l15) * 128;
  const int sw = l15 & 7;

  f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: stage 0 everywhere, visible to all
  issue_stage();
  G4_BAR_VM();
  if (wm == 1) G4_BAR();   // group B trails by one segment

  int c_kt = 0, c_s = wl;
  const int P = 2 * total_it;
  for (int p = 0; p < P; ++p) {
    const int it = p >> 1, kk = p & 1;
    // ---------------- LOAD segment ----------------
    const char* cur = lds + (it & 1) * G3_STAGE;
    const int co = (((kk * 4 + q4) ^ sw) << 4);
    bf16x8 fa[8], fw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i] + co);
    if (kk == 0 && i_it < total_it) issue_stage();   // stage it+1 -> the buffer whose last readers finished a barrier ago
    if (kk == 1 && wm == 1) G4_BAR_VM(); else G4_BAR();   // B closes global segment 4*it+3 here: its DMA must have landed
    // ---------------- COMPUTE segment ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (kk == 1 && ++c_kt == nk) {
      const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
      g3_epilogue<EPI>(g, acc, m0, n0, wm, wn, l15, q4, lbias);
      c_kt = 0;
      c_s += nwl;
    }
    if (kk == 1 && wm == 0) G4_BAR_VM(); else G4_BAR();   // A closes global segment 4*it+3 here
  }
  if (wm == 0) G4_BAR();   // balance the barrier count of the two groups
}

template <int EPI>
int launch_gemm_s256(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_s256_kernel<EPI>), dim3(256), dim3(512), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_s256");
}


// ===========================================================================
// GEMM v5: the same persistent 256x256x64 tile / 2-stage ring, but 16 waves per workgroup (4x4, 64x64 each = 4x4
// MFMA tiles, 64 accumulators) = FOUR waves per SIMD.  Motivation (s_memtime probe on v3, tools/probe_gemm_wait.py):
// per K-step of ~3650 cycles only ~80 are spent waiting for the LDS-DMA (the data has landed), but ~760 at the
// barrier because the older of the two waves of a SIMD races ahead and then idles while the younger one cannot keep
// the matrix pipe busy on its own.  With four lighter waves per SIMD some wave is always ready to issue MFMAs.
// ===========================================================================
template <int EPI>
__global__ __launch_bounds__(1024) void gemm_bf16_w16_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE + 16384 + 256];  // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 2, wn = wid & 3;
  float* lbias = reinterpret_cast<float*>(lds + 2 * G3_STAGE);
  if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID)
    for (int i = tid; i < g.N; i += 1024) lbias[i] = g.bias[i];
  const int nk = g.K / GT_BK;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  const int total_it = my_tiles * nk;
  if (total_it == 0) return;

  int i_it = 0, i_kt = 0, i_s = wl;
  int i_m0 = ((i_s / n_per) * npset + pset) * G3_BM, i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
  const int rin = lane >> 3, slot = lane & 7;
  auto advance = [&]() {
    ++i_it;
    if (++i_kt == nk) {
      i_kt = 0;
      i_s += nwl;
      i_m0 = ((i_s / n_per) * npset + pset) * G3_BM;
      i_n0 = (grp * n_per + i_s % n_per) * G3_BN;
    }
  };
  auto issue_piece = [&](int q) {   // q = 0..3: pieces 2*wid, 2*wid+1 of A then of W, for stage i_it
    char* nbuf = lds + (i_it & 1) * G3_STAGE;
    const size_t koff = (size_t)i_kt * GT_BK + ((slot ^ rin) << 3);
    const int piece = wid * 2 + (q & 1);
    if (q < 2) {
      int gr = i_m0 + piece * 8 + rin;
      gr = gr < g.M ? gr : g.M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.A + (size_t)gr * g.K + koff),
                                       (__attribute__((address_space(3))) void*)(nbuf + piece * 1024), 16, 0, 0);
    } else {
      int gr = i_n0 + piece * 8 + rin;
      gr = gr < g.N ? gr : g.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W + (size_t)gr * g.K + koff),
                                       (__attribute__((address_space(3))) void*)(nbuf + G3_BM * 128 + piece * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_piece(q);
  advance();

  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    offA[i] = (wm * 64 + i * 16 + l15) * 128;
    offW[i] = G3_BM * 128 + (wn * 64 + i * 16 + l15) * 128;
  }
  const int sw = l15 & 7;
  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int c_kt = 0, c_s = wl;
  for (int it = 0; it < total_it; ++it) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const char* cur = lds + (it & 1) * G3_STAGE;
    const bool more = i_it < total_it;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q4) ^ sw) << 4);
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
        fa[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i] + co);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (more && kk == 0) issue_piece(mi);   // one piece behind each group of 4 MFMAs of the first half K-step
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
      }
    }
    if (more) advance();
    if (++c_kt == nk) {
      const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
      g3_epilogue_t<EPI, 4>(g, acc, m0, n0, wm * 64, wn, l15, q4, lbias);
      c_kt = 0;
      c_s += nwl;
    }
  }
}

template <int EPI>
int launch_gemm_w16(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_w16_kernel<EPI>), dim3(256), dim3(1024), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_w16");
}

// ===========================================================================
// GEMM v6 ("a4"): persistent 256x256x64 tile, FOUR waves per workgroup (2x2) = ONE wave per SIMD, each wave owning
// 128x128 outputs = 8x8 MFMA 16x16x32 tiles = 256 fp32 accumulators per lane.  hipcc cannot allocate that (it spills,
// DESIGN 4.1; with "+a" inline-asm operands its allocator still spills accumulators as soon as anything else moves),
// so the accumulators live in the accumulator half of the unified register file under LITERAL names a[0:255]: every
// MFMA, every accumulator read and every zeroing is an inline-asm statement that names its AGPRs itself and hipcc
// allocates only the ~150 architectural VGPRs around them.  (tests/test_abi.py audits the generated code: no spill and
// no compiler-emitted AGPR access may exist in this kernel.)  Per K-step the four waves read 128 KB of fragments from LDS
// instead of the 192 KB of the 8-wave kernel (-33 % LDS bytes per MFMA) and no SIMD is shared: the matrix pipe of a
// SIMD is fed by ONE in-order instruction stream, software-pipelined so that it never waits on its own memory ops:
//   [A] 64 MFMAs on K-half 0 of K-step `it`, the 16 fragment reads of K-half 1 going out behind the first 16 of them
//       s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier   (stage it+1 landed - issued a K-step ago; buffer it&1 fully read)
//   [B] 16 fragment reads of K-half 0 of K-step it+1, then 64 MFMAs on K-half 1 with one LDS-DMA piece of stage it+2
//       (into the buffer just freed) behind every 4th MFMA
// so a fragment read has ~half a K-step and a DMA piece a whole K-step (~2000 cycles) to complete, and the only
// synchronisation is one barrier per K-step between four waves running the same stream on separate SIMDs.
// ===========================================================================
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(static_cast<F&&>(f));
  }
}
#define A4_ALL_AGPRS                                                                                                     \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",  \
      "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",   \
      "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52",   \
      "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",   \
      "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86",   \
      "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102",       \
      "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117",  \
      "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132",  \
      "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147",  \
      "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162",  \
      "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177",  \
      "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192",  \
      "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207",  \
      "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222",  \
      "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237",  \
      "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252",  \
      "a253", "a254", "a255"

// accumulator tile (nn, mi) = a[(nn*8 + mi)*4 .. +3]; nn = column tile 0..7 (16 columns each), mi = row tile 0..7.
// EVERY statement that touches the accumulators lists ALL of a[0:255] as clobbered: hipcc then cannot keep any value of
// its own in an AGPR across such a statement (with one wave per SIMD it is otherwise free to allocate loads / spills into
// "unused" AGPRs - which are this kernel's accumulators).
#define A4_MFMA(NN, MI, WF, AF)                                                                              \
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(WF), "v"(AF), "i"(((NN)*8 + (MI)) * 4), \
               "i"(((NN)*8 + (MI)) * 4 + 3) : A4_ALL_AGPRS)
template <int IDX>
__device__ __forceinline__ float a4_acc_take() {   // read a[IDX] and zero it for the next tile
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1\n\tv_accvgpr_write_b32 a%c1, 0" : "=v"(x) : "i"(IDX) : A4_ALL_AGPRS);
  return x;
}
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_a4_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE + 16384];  // the ONLY LDS object: ring + bias
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  float* lbias = reinterpret_cast<float*>(lds + 2 * G3_STAGE);
  if (epi_has_bias(EPI))
    for (int i = tid; i < g.N; i += 256) lbias[i] = g.bias[i];
  const int nk = g.K / GT_BK;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  const int total_it = my_tiles * nk;
  if (total_it == 0) return;
  // claims a[0:255] for this kernel (allocation granule + "clobbered here") and zeroes them
  asm volatile("" ::: A4_ALL_AGPRS);
  sfor<0, 256>([&](auto i_) { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"i"(decltype(i_)::value) : A4_ALL_AGPRS); });

  // ---- issue side: this wave's 16 LDS-DMA pieces (8 of A, 8 of W; 8 rows x 128 B each) of stage i_it ----
  // buffer_load ... lds through a per-tile resource descriptor: ONE per-lane byte offset (row-in-piece * K + swizzled
  // chunk, the same for A and W), the piece's row block in the scalar offset, rows past the matrix end fall outside
  // num_records and read as zeros - a piece costs no VALU work at all.
  const int rin = lane >> 3, slot = lane & 7;
  const unsigned lane_off = ((unsigned)rin * (unsigned)g.K + (unsigned)((slot ^ rin) << 3)) * 2u;
  const unsigned piece_stride = 8u * (unsigned)g.K * 2u;      // bytes between consecutive 8-row pieces
  int i_it = 0, i_kt = 0, i_s = wl;
  __amdgpu_buffer_rsrc_t rsA, rsW;
  auto tile_rsrc = [&]() {
    const int m0 = ((i_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + i_s % n_per) * G3_BN;
    const long ra = m0 < g.M ? ((long)(g.M - m0) * g.K * 2) : 0, rw = n0 < g.N ? ((long)(g.N - n0) * g.K * 2) : 0;
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)(m0 < g.M ? m0 : 0) * g.K), 0,
                                            (int)(ra < 0x40000000L ? ra : 0x40000000L), 0x00020000);
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)(n0 < g.N ? n0 : 0) * g.K), 0,
                                            (int)(rw < 0x40000000L ? rw : 0x40000000L), 0x00020000);
  };
  auto piece = [&](int q) {            // q = 0..15: A pieces first, then W pieces, of stage i_it
    char* nbuf = lds + (i_it & 1) * G3_STAGE;
    const int p = wid * 8 + (q & 7);
    const unsigned voff = lane_off + (unsigned)i_kt * (GT_BK * 2u);
    if (q < 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(nbuf + p * 1024), 16, voff,
                                               (unsigned)p * piece_stride, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(nbuf + G3_BM * 128 + p * 1024),
                                               16, voff, (unsigned)p * piece_stride, 0, 0);
  };
  auto advance = [&]() {
    ++i_it;
    if (++i_kt == nk) {
      i_kt = 0;
      i_s += nwl;
      tile_rsrc();
    }
  };
  tile_rsrc();
#pragma unroll
  for (int q = 0; q < 16; ++q) piece(q);
  advance();
#pragma unroll
  for (int q = 0; q < 16; ++q) piece(q);   // (past the last stage the cursor addresses rows outside num_records: zeros, unused)
  advance();

  // ---- fragment addressing: rows wm*128 + mi*16 + l15 of A, wn*128 + ni*16 + l15 of W; chunk (kk*4 + q4) ^ (row & 7)
  const int sw = l15 & 7;
  const int fbaseA = (wm * 128 + l15) * 128, fbaseW = G3_BM * 128 + (wn * 128 + l15) * 128;
  i32x4 fa0[8], fw0[8], fa1[8], fw1[8];   // K-half 0 / K-half 1 fragments (double buffered)
  auto read_frags = [&](const char* buf, int kk, i32x4 (&fa)[8], i32x4 (&fw)[8]) {
    const int co = (((kk * 4 + q4) ^ sw) << 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const i32x4*>(buf + fbaseW + i * 2048 + co);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const i32x4*>(buf + fbaseA + i * 2048 + co);
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  read_frags(lds, 0, fa0, fw0);

  int c_kt = 0, c_s = wl;
  for (int it = 0; it < total_it; ++it) {
    const char* cur = lds + (it & 1) * G3_STAGE;
    const char* nxt = lds + ((it + 1) & 1) * G3_STAGE;
    // ---- [A] ----  (the K-half-1 reads go out behind the first 16 MFMAs: the lgkmcnt wait hipcc places in front of
    // the first MFMA then covers only the K-half-0 reads issued half a K-step ago, not 16 reads issued just now)
    sfor<0, 8>([&](auto nn_) {
      constexpr int nn = decltype(nn_)::value;
      if (nn == 2) read_frags(cur, 1, fa1, fw1);
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        const i32x4 wf = fw0[nn], af = fa0[mi];
        A4_MFMA(nn, mi, wf, af);
      });
    });
    // stage it+1 has landed (issued a K-step ago) and this wave's reads of buffer it&1 are complete
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- [B] ----
    sfor<0, 8>([&](auto nn_) {
      constexpr int nn = decltype(nn_)::value;
      // K-half 0 of K-step it+1, again behind the first 16 MFMAs (hipcc's lgkmcnt ladder cannot count past 15 ops, so
      // 16 fresh reads in front of the first MFMA would make it wait for some of them).  After the last K-step this
      // reads a stale buffer; the values are never used.
      if (nn == 2) read_frags(nxt, 0, fa0, fw0);
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        // one piece of stage it+2 behind every 4th MFMA.  Unconditional (no branches in the stream): past the last
        // stage the pieces read zeros into a buffer nobody reads again.
        if ((mi & 3) == 0) piece(nn * 2 + (mi >> 2));
        const i32x4 wf = fw1[nn], af = fa1[mi];
        A4_MFMA(nn, mi, wf, af);
      });
    });
    advance();
    if (++c_kt == nk) {
      const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last MFMA's result -> first v_accvgpr_read
      sfor<0, 2>([&](auto nh_) {                          // the wave's two 64-column slices
        constexpr int nh = decltype(nh_)::value;
        EpiCols ec;
        g3_epi_cols<EPI>(g, n0, wn * 2 + nh, q4, ec);
        float2 rst[8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
          rst[mi] = (EPI == GE_BIAS_LN || EPI == GE_GELU_LN) ? g3_epi_rowstat(g, m0 + wm * 128 + mi * 16 + l15) : make_float2(1.f, 0.f);
        sfor<0, 8>([&](auto mi_) {
          constexpr int mi = decltype(mi_)::value;
          f32x4 vv[4];
          sfor<0, 4>([&](auto ni_) {
            constexpr int ni = decltype(ni_)::value;
            constexpr int base = ((nh * 4 + ni) * 8 + mi) * 4;
            vv[ni][0] = a4_acc_take<base>(); vv[ni][1] = a4_acc_take<base + 1>();
            vv[ni][2] = a4_acc_take<base + 2>(); vv[ni][3] = a4_acc_take<base + 3>();
          });
          g3_epi_row<EPI>(g, vv, ec, rst[mi], m0 + wm * 128 + mi * 16 + l15, n0, wn * 2 + nh, q4, lbias);
        });
      });
      asm volatile("s_nop 1");   // (zeroed accumulators are VALU writes; keep them clear of the next MFMA's read)
      c_kt = 0;
      c_s += nwl;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
}

template <int EPI>
int launch_gemm_a4(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_a4_kernel<EPI>), dim3(256), dim3(256), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_a4");
}

// Variant choice: the persistent 256x256 kernel whenever there is at least one tile per CU; the small
// 128x128 kernel otherwise.  Variants 2..29 are reachable only through tspo_gemm_bf16's test hook (act >> 8):
// 2 = 256x128 ring, 3 = 2 without K-rotation, 4/5 = 2 compute-only / loads-only, 6 = 256x256, 7/8/9 = 6 compute-only /
// loads-only / A-loads-only, 1x = 2 with x N-groups, 2x = 2 loads-only with x N-groups.
// ===========================================================================
// GEMM v7 ("a5"): the a4 wave layout (4 waves, one per SIMD, 128x128 outputs each, accumulators in literal AGPRs) with a
// DEEPER, FINER ring.  Measured on a4 and on the 8-wave kernel alike: a K-step takes ~3800 cycles for ~2100 cycles of
// MFMA because with two 64 KB stages only ONE stage (64 KB per CU) can be in flight and it has exactly one K-step to
// land - the loop runs at one stage per fill latency (~9 TB/s of L2->LDS traffic chip-wide), whatever the waves do.
// Here the ring holds FIVE half-stages of 32 KB (K = 32: A [256 rows][64 B] | W [256 rows][64 B]) = all 160 KB of LDS:
// while half-step h computes, half-stages h+2, h+3 and h+4 (96 KB) are in flight and each has three half-steps (1.5
// K-steps) to land.  64-byte rows: 16-byte chunk c of row r is stored at chunk c ^ f(r), f = t ^ ((t & 1) << 1) with
// t = (r >> 2) & 3, which makes every ds_read_b128 lane group hit 16 distinct 16-byte slots (no bank conflicts); the
// XOR sits on the DMA source address (the LDS destination of an LDS-DMA is lane-linear) and on the fragment read.
//   half-step h:  16 MFMAs on fragments F(h) | 16 ds_read_b128 of F(h+1) | 48 MFMAs with the wave's 8 DMA pieces of
//                 half-stage h+4 (into the buffer F(h-1) was read from) behind every 6th | vmcnt(16) lgkmcnt(0) s_barrier
// The first half-step of a tile issues its MFMAs with C = 0 (no zeroing pass); the bias vector is not kept in LDS any
// more: bias / LayerNorm vectors / row statistics / the residual tile (16-byte loads in the store mapping, un-swapped
// with v_permlane16_swap) of a 64-column slice are fetched together into registers before the slice is finished.
// ===========================================================================
#define H5_BYTES 32768
#define H5_N 5
#define A5_MFMA_Z(NN, MI, WF, AF)                                                                  \
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(WF), "v"(AF), "i"(((NN)*8 + (MI)) * 4), \
               "i"(((NN)*8 + (MI)) * 4 + 3) : A4_ALL_AGPRS)
template <int IDX>
__device__ __forceinline__ float a5_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(IDX) : A4_ALL_AGPRS);
  return x;
}

// Epilogue of the kernels that keep the wave's 128x128 accumulators in a[0:255] (tile (nn, mi) = a[(nn*8 + mi)*4 .. +3]):
// per 64-column slice, everything the slice needs from memory - bias / folded bias, LayerNorm column sums, row statistics,
// the residual tile (16-byte loads in the store mapping, un-swapped with v_permlane16_swap) - is fetched together into
// registers (one memory latency per slice, not one per row block), then the 8 row blocks are finished by g3_epi_row.
template <int EPI>
__device__ __forceinline__ void agpr_epilogue(const GemmArgs& g, int m0, int n0, int wm, int wn, int l15, int q4) {
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last MFMA's result -> first v_accvgpr_read
  constexpr bool LN = EPI == GE_BIAS_LN || EPI == GE_GELU_LN;
  constexpr bool RES = EPI == GE_RESID || EPI == GE_RESID_ST;
  sfor<0, 2>([&](auto nh_) {                          // the wave's two 64-column slices
    constexpr int nhs = decltype(nh_)::value;
    const int ws = wn * 2 + nhs;
    EpiCols ec;
    g3_epi_cols<EPI>(g, n0, ws, q4, ec);
    if (epi_has_bias(EPI)) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + ws * 64 + ni * 16 + q4 * 4;
        ec.bias[ni] = *reinterpret_cast<const f32x4*>(g.bias + (n < g.N ? n : 0));
      }
    }
    float2 rst[8];
    uint4 rres[8][2];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int m = m0 + wm * 128 + mi * 16 + l15;
      rst[mi] = LN ? g3_epi_rowstat(g, m) : make_float2(1.f, 0.f);
      if (RES) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {   // 16 B per lane in the store mapping: 4 lanes cover 64 contiguous bytes of a row
          const int n = n0 + ws * 64 + (2 * pr + (q4 & 1)) * 16 + (q4 >> 1) * 8;
          const bool ok = m < g.M && n < g.N;
          rres[mi][pr] = ok ? *reinterpret_cast<const uint4*>(g.R + (size_t)m * g.N + n) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
    sfor<0, 8>([&](auto mi_) {
      constexpr int mi = decltype(mi_)::value;
      f32x4 vv[4];
      sfor<0, 4>([&](auto ni_) {
        constexpr int ni = decltype(ni_)::value;
        constexpr int base = ((nhs * 4 + ni) * 8 + mi) * 4;
        vv[ni][0] = a5_acc_read<base>(); vv[ni][1] = a5_acc_read<base + 1>();
        vv[ni][2] = a5_acc_read<base + 2>(); vv[ni][3] = a5_acc_read<base + 3>();
      });
      uint2 rp[4] = {};
      if (RES) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {   // back to the MFMA layout: the inverse of the store-side permlane16 swap
          const auto x = __builtin_amdgcn_permlane16_swap(rres[mi][pr].x, rres[mi][pr].z, false, false);
          const auto y = __builtin_amdgcn_permlane16_swap(rres[mi][pr].y, rres[mi][pr].w, false, false);
          rp[2 * pr] = make_uint2(x[0], y[0]);
          rp[2 * pr + 1] = make_uint2(x[1], y[1]);
        }
      }
      g3_epi_row<EPI, true>(g, vv, ec, rst[mi], m0 + wm * 128 + mi * 16 + l15, n0, ws, q4, nullptr, rp);
    });
  });
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_a5_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[H5_N * H5_BYTES];  // the ONLY LDS object: all 160 KB of the CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int nh = g.K / 32;   // half-steps per tile (K % 64 == 0: even)
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  if (my_tiles == 0) return;
  asm volatile("" ::: A4_ALL_AGPRS);   // claims a[0:255] for this kernel

  // ---- issue side: this wave's 8 LDS-DMA pieces (4 of A, 4 of W; 16 rows x 64 B each) of half-stage i_h ----
  const int prow = lane >> 2, pc = lane & 3, pt = (prow >> 2) & 3;
  const unsigned lane_off = (unsigned)prow * (unsigned)g.K * 2u + (unsigned)((pc ^ pt ^ ((pt & 1) << 1)) << 4);
  const unsigned piece_stride = 16u * (unsigned)g.K * 2u;     // bytes between consecutive 16-row pieces
  int i_kh = 0, i_s = wl, i_b = 0;                            // half-step inside the tile, tile, ring buffer
  __amdgpu_buffer_rsrc_t rsA, rsW;
  auto tile_rsrc = [&]() {
    const int m0 = ((i_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + i_s % n_per) * G3_BN;
    const long ra = m0 < g.M ? ((long)(g.M - m0) * g.K * 2) : 0, rw = n0 < g.N ? ((long)(g.N - n0) * g.K * 2) : 0;
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)(m0 < g.M ? m0 : 0) * g.K), 0,
                                            (int)(ra < 0x40000000L ? ra : 0x40000000L), 0x00020000);
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)(n0 < g.N ? n0 : 0) * g.K), 0,
                                            (int)(rw < 0x40000000L ? rw : 0x40000000L), 0x00020000);
  };
  auto piece = [&](int q) {            // q = 0..7: 4 A pieces, then 4 W pieces.  Past the last tile the cursor addresses
    char* nbuf = lds + i_b * H5_BYTES; // rows outside num_records: zeros into a buffer nobody reads again.
    const int p = wid * 4 + (q & 3);
    const unsigned voff = lane_off + (unsigned)i_kh * 64u;
    if (q < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(nbuf + p * 1024), 16, voff,
                                               (unsigned)p * piece_stride, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(nbuf + 16384 + p * 1024), 16,
                                               voff, (unsigned)p * piece_stride, 0, 0);
  };
  auto advance = [&]() {
    i_b = i_b == H5_N - 1 ? 0 : i_b + 1;
    if (++i_kh == nh) {
      i_kh = 0;
      i_s += nwl;
      tile_rsrc();
    }
  };
  tile_rsrc();
#pragma unroll
  for (int st = 0; st < 4; ++st) {     // half-stages 0..3 (always exist: nh >= 2 and the cursor may run past the end)
#pragma unroll
    for (int q = 0; q < 8; ++q) piece(q);
    advance();
  }

  // ---- fragment addressing ----
  const int ft = (l15 >> 2) & 3;
  const int flane = l15 * 64 + ((q4 ^ ft ^ ((ft & 1) << 1)) << 4);
  const int fbaseA = wm * 128 * 64 + flane, fbaseW = 16384 + wn * 128 * 64 + flane;
  i32x4 fa0[8], fw0[8], fa1[8], fw1[8];   // fragments of even / odd half-steps (double buffered)
  auto read_frags = [&](int buf, i32x4 (&fa)[8], i32x4 (&fw)[8]) {
    const char* b = lds + buf * H5_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const i32x4*>(b + fbaseW + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const i32x4*>(b + fbaseA + i * 1024);
  };

  asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // half-stages 0 and 1 have landed
  read_frags(0, fa0, fw0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int c_b = 0;   // ring buffer the CURRENT half-step's fragments came from

  // one half-step: MFMAs on (fa, fw); reads the next half-step's fragments into (na, nw); issues half-stage +4
  auto half = [&](auto zero_, i32x4 (&fa)[8], i32x4 (&fw)[8], i32x4 (&na)[8], i32x4 (&nw)[8]) {
    constexpr bool ZERO = decltype(zero_)::value;
    const int nb = c_b == H5_N - 1 ? 0 : c_b + 1;
    sfor<0, 8>([&](auto nn_) {
      constexpr int nn = decltype(nn_)::value;
      if (nn == 2) read_frags(nb, na, nw);
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        if (nn >= 2 && (nn * 8 + mi - 16) % 6 == 0) piece((nn * 8 + mi - 16) / 6);   // 8 pieces behind MFMAs 16, 22, .., 58
        const i32x4 wf = fw[nn], af = fa[mi];
        if (ZERO) A5_MFMA_Z(nn, mi, wf, af); else A4_MFMA(nn, mi, wf, af);
      });
    });
    advance();
    c_b = nb;
    // half-stage h+2 has landed (the 16 youngest pieces - h+3, h+4 - may stay in flight); this wave's reads of F(h+1) are done
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // the same lgkmcnt(0) again as a builtin: free at run time, but it tells hipcc's wait-count model that every fragment
    // read has returned - otherwise it keeps "pending" reads across the asm and, unable to count past 15, makes the
    // MFMAs behind the next 16 reads wait for the first of them
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };

  int c_s = wl;
  for (int t = 0; t < my_tiles; ++t) {
    half(std::true_type{}, fa0, fw0, fa1, fw1);
    half(std::false_type{}, fa1, fw1, fa0, fw0);
    for (int kp = 1; kp < (nh >> 1); ++kp) {
      half(std::false_type{}, fa0, fw0, fa1, fw1);
      half(std::false_type{}, fa1, fw1, fa0, fw0);
    }
    // ---- epilogue of tile c_s (the ring keeps streaming: three half-stages of the next tile are in flight / landed) ----
    const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
    agpr_epilogue<EPI>(g, m0, n0, wm, wn, l15, q4);
    c_s += nwl;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
}

template <int EPI>
int launch_gemm_a5(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_a5_kernel<EPI>), dim3(256), dim3(256), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_a5");
}

// ===========================================================================
// GEMM v8 ("a7"): the a4 wave layout and 2 x 64 KB ring, but the operands come in through REGISTERS (buffer_load_dwordx4
// -> 64 staging VGPRs per lane -> ds_write_b128) instead of LDS-DMA.  Why: a4, a5 and the 8-wave kernel - three different
// structures - all run a K-step in ~3600 cycles for ~2200 cycles of MFMA, and neither more bytes in flight (a5) nor
// fewer LDS reads (a4) moved it.  What they share is 64 LDS-DMA wave-instructions per CU and K-step: an LDS-DMA piece
// holds up the issuing wave ~60-180 cycles and the whole stream tops out at ~25 B/clk/CU (12.6 TB/s chip-wide, measured
// loads-only), i.e. ~2600 cycles of load-path time per K-step that a one-wave-per-SIMD kernel cannot hide.  Ordinary
// vector loads do not have that issue cost, and with 256 accumulators in AGPRs this kernel has the VGPRs to hold a
// whole K-step share (16 x 16 B per lane) in flight.
//   piece j (8 rows x 128 B; 0-7 = A, 8-15 = W):  S_j(x) = ds_write_b128 of the registers into stage x's buffer,
//                                                 G_j(x) = buffer_load_dwordx4 of stage x's rows into the same registers
//   K-step it:  [A] 64 MFMAs on K-half 0 | F1(it) reads | S_j(it+1), G_j(it+2) for the W pieces   | lgkmcnt(0) s_barrier
//               [B] 64 MFMAs on K-half 1 | F0(it+1) reads | S_j(it+2), G_j(it+3) for the A pieces
// Every load has a full K-step to land before its ds_write; the XOR swizzle sits on the ds_write address (conflict-free:
// the 8 lanes of a row cover all 32 banks) and the global reads are plain 128-byte rows.
// ===========================================================================
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_a7_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE];  // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int nk = g.K / GT_BK;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  if (my_tiles == 0) return;
  asm volatile("" ::: A4_ALL_AGPRS);   // claims a[0:255] for this kernel

  // ---- staging: this wave's 8 A pieces (registers 0-7) and 8 W pieces (registers 8-15) of a stage ----
  const int rin = lane >> 3, slot = lane & 7;
  const unsigned lane_goff = ((unsigned)rin * (unsigned)g.K + (unsigned)(slot << 3)) * 2u;   // plain 128-byte rows
  const int lane_woff = rin * 128 + ((slot ^ rin) << 4);                                      // swizzled LDS image
  const unsigned piece_stride = 8u * (unsigned)g.K * 2u;
  u32x4 stg[16];
  // the stage the A registers / the W registers are loaded for NEXT: K-step inside the tile, tile, resource descriptor
  int a_kt = 0, a_s = wl, w_kt = 0, w_s = wl;
  auto rsrc_a = [&](int s_) {
    const int m0 = ((s_ / n_per) * npset + pset) * G3_BM;
    const long r = m0 < g.M ? ((long)(g.M - m0) * g.K * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)(m0 < g.M ? m0 : 0) * g.K), 0,
                                             (int)(r < 0x40000000L ? r : 0x40000000L), 0x00020000);
  };
  auto rsrc_w = [&](int s_) {
    const int n0 = (grp * n_per + s_ % n_per) * G3_BN;
    const long r = n0 < g.N ? ((long)(g.N - n0) * g.K * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)(n0 < g.N ? n0 : 0) * g.K), 0,
                                             (int)(r < 0x40000000L ? r : 0x40000000L), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t a_rs = rsrc_a(a_s), w_rs = rsrc_w(w_s);
  auto adv_a = [&]() { if (++a_kt == nk) { a_kt = 0; a_s += nwl; a_rs = rsrc_a(a_s); } };
  auto adv_w = [&]() { if (++w_kt == nk) { w_kt = 0; w_s += nwl; w_rs = rsrc_w(w_s); } };
  // (past the last tile the cursors address rows outside num_records: the loads return zeros, nobody reads them)
  auto gload = [&](auto q_) {   // G_q of the cursor's stage
    constexpr int q = decltype(q_)::value;
    if constexpr (q < 8)
      stg[q] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, lane_goff + (unsigned)a_kt * (GT_BK * 2u),
                                                     (unsigned)(wid * 8 + q) * piece_stride, 0);
    else
      stg[q] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane_goff + (unsigned)w_kt * (GT_BK * 2u),
                                                     (unsigned)(wid * 8 + q - 8) * piece_stride, 0);
  };
  auto swrite = [&](auto q_, int buf) {   // S_q into ring buffer `buf`
    constexpr int q = decltype(q_)::value;
    char* d = lds + buf * G3_STAGE + (q < 8 ? 0 : G3_BM * 128) + (wid * 8 + (q & 7)) * 1024 + lane_woff;
    *reinterpret_cast<u32x4*>(d) = stg[q];
  };

  // ---- prologue: stage 0 complete in buffer 0; A of stage 1 in buffer 1, W of stage 1 in registers 8-15; A of stage 2
  //      in registers 0-7 ----
  sfor<0, 16>([&](auto q_) { gload(q_); });
  sfor<0, 16>([&](auto q_) { swrite(q_, 0); });
  adv_a(); adv_w();
  sfor<0, 16>([&](auto q_) { gload(q_); });
  sfor<0, 8>([&](auto q_) { swrite(q_, 1); });
  adv_a();
  sfor<0, 8>([&](auto q_) { gload(q_); });
  adv_a();
  adv_w();   // cw now points at stage 2 (loaded in [A] of K-step 0, after stage 1's W went to LDS)

  // ---- fragments: A double-buffered per K-half (2 x 32 VGPRs); W in ONE set of 8 x 4 VGPRs that is refilled in place:
  //      the fragment of column tile nn is dead after its 8 MFMAs, so the next K-half's fragment nn is read right behind
  //      them (a full K-half ahead of its use) - 32 VGPRs less than double-buffering both operands, which is what keeps
  //      this kernel (64 staging VGPRs) below the point where hipcc starts parking values in AGPRs ----
  const int sw = l15 & 7;
  const int fbaseA = (wm * 128 + l15) * 128, fbaseW = G3_BM * 128 + (wn * 128 + l15) * 128;
  i32x4 fa0[8], fa1[8], fw[8];
  auto co_of = [&](int kk) { return ((kk * 4 + q4) ^ sw) << 4; };
  auto read_a = [&](const char* buf, int kk, i32x4 (&fa)[8]) {
    const int co = co_of(kk);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const i32x4*>(buf + fbaseA + i * 2048 + co);
  };
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  read_a(lds, 0, fa0);
#pragma unroll
  for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const i32x4*>(lds + fbaseW + i * 2048 + co_of(0));
  __builtin_amdgcn_s_waitcnt(0xC07F);

  int it = 0;
  // one K-half: 64 MFMAs on (fa, fw); behind column tile nn's MFMAs its W fragment is refilled from (wbuf, wkk) = the next
  // K-half; na <- A fragments of the next K-half (abuf, akk) behind the first 16 MFMAs (skipped when !RDA);
  // staging pieces J0 .. J0+7: ds_write into ring buffer sbuf, then reload for the cursor's stage
  auto khalf = [&](auto zero_, auto rda_, auto j0_, i32x4 (&fa)[8], i32x4 (&na)[8], const char* abuf, int akk,
                   const char* wbuf, int wkk, int sbuf) {
    constexpr bool ZERO = decltype(zero_)::value, RDA = decltype(rda_)::value;
    constexpr int J0 = decltype(j0_)::value;
    const int wco = co_of(wkk);
    sfor<0, 8>([&](auto nn_) {
      constexpr int nn = decltype(nn_)::value;
      if (nn == 2 && RDA) read_a(abuf, akk, na);
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        if (nn >= 2 && (nn * 8 + mi - 16) % 6 == 0) {
          constexpr int j = J0 + (nn * 8 + mi - 16) / 6;
          swrite(std::integral_constant<int, j>{}, sbuf);
          gload(std::integral_constant<int, j>{});
        }
        const i32x4 wf = fw[nn], af = fa[mi];
        if (ZERO) A5_MFMA_Z(nn, mi, wf, af); else A4_MFMA(nn, mi, wf, af);
      });
      if (RDA) fw[nn] = *reinterpret_cast<const i32x4*>(wbuf + fbaseW + nn * 2048 + wco);
    });
  };
  // LAST (the tile's last K-step): the next tile's first fragments are not read in [B] - they would stay live across the
  // epilogue on top of the 64 staging registers and the epilogue's own prefetch - but after it.
  auto kstep = [&](auto zero_, auto last_) {
    constexpr bool LAST = decltype(last_)::value;
    const int cb = it & 1, nb = cb ^ 1;
    const char* cur = lds + cb * G3_STAGE;
    const char* nxt = lds + nb * G3_STAGE;
    // [A]: K-half 0; W pieces: S_j(it+1) -> buffer nb, then G_j(it+2); fragments of K-half 1 of this stage
    khalf(zero_, std::true_type{}, std::integral_constant<int, 8>{}, fa0, fa1, cur, 1, cur, 1, nb);
    adv_w();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage it+1 complete in LDS; buffer cb fully read
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // [B]: K-half 1; A pieces: S_j(it+2) -> buffer cb, then G_j(it+3); fragments of K-half 0 of stage it+1
    khalf(std::false_type{}, std::integral_constant<bool, !LAST>{}, std::integral_constant<int, 0>{}, fa1, fa0, nxt, 0, nxt, 0, cb);
    adv_a();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // (free at run time when the reads have already returned)
    ++it;
  };

  int c_s = wl;
  for (int t = 0; t < my_tiles; ++t) {
    if (nk == 1) {
      kstep(std::true_type{}, std::true_type{});
    } else {
      kstep(std::true_type{}, std::false_type{});
      for (int kt = 1; kt < nk - 1; ++kt) kstep(std::false_type{}, std::false_type{});
      kstep(std::false_type{}, std::true_type{});
    }
    const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
    agpr_epilogue<EPI>(g, m0, n0, wm, wn, l15, q4);
    c_s += nwl;
    const char* nbuf = lds + (it & 1) * G3_STAGE;   // K-half 0 of the next tile's first K-step (stale after the last tile)
    read_a(nbuf, 0, fa0);
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const i32x4*>(nbuf + fbaseW + i * 2048 + co_of(0));
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
}

template <int EPI>
int launch_gemm_a7(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_a7_kernel<EPI>), dim3(256), dim3(256), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_a7");
}

// kernel used for "big" problems: TSPO_GEMM_VARIANT (dev hook for whole-encoder A/B runs) or the default
static int default_big_variant() {
  static const int v = [] {
    const char* e = getenv("TSPO_GEMM_VARIANT");
    return e && atoi(e) > 0 ? atoi(e) : 6;
  }();
  return v;
}

template <int EPI>
int launch_gemm(GemmArgs g, hipStream_t st) {
  const bool big = tspo::gemm_bf16_is_big(g.M, g.N, g.K);
  const int v = g.variant ? g.variant : (big ? default_big_variant() : 1);
  if (v == 1) return launch_gemm_v1<EPI>(g, st);
  if (v == 6) return launch_gemm_p256<EPI, 1, 6>(g, st);     // default: interleaved DMA issue + L2 prefetch of A 6 K-steps ahead
  if (v == 65) return launch_gemm_p256<EPI, 1, 0>(g, st);    // A/B: no L2 prefetch
  if (v == 66) return launch_gemm_p256<EPI, 1, 6, 1>(g, st); // A/B: DMA pieces issued in the first quarter of the K-step
  if (v == 67) return launch_gemm_p256<EPI, 1, 6, 2>(g, st); // A/B: s_setprio(1) around each group of 8 MFMAs
  if (v == 68) return launch_gemm_p256<EPI, 1, 6, 3>(g, st); // A/B: buffer_load ... lds instead of global_load_lds
  if (v == 69) return launch_gemm_p256<EPI, 1, 6, 4>(g, st); // timing probe (s_memtime around the per-K-step wait); g.pos = debug buffer
  if (v == 70) return launch_gemm_s256<EPI>(g, st);          // role-split (staggered wave rows)
  if (v == 71) return launch_gemm_w16<EPI>(g, st);           // 16 waves per workgroup (4 per SIMD)
  if (v == 80) return launch_gemm_a4<EPI>(g, st);            // 4 waves, 128x128 per wave, AGPR accumulators
  if (v == 81) return launch_gemm_a5<EPI>(g, st);            // a4 + five 32 KB half-stages (96 KB in flight)
  if (v == 82) return launch_gemm_a7<EPI>(g, st);            // a4 with register-staged operand loads
  if (v == 7) { g.P = -2; return launch_gemm_p256<EPI, 1>(g, st); }
  if (v == 60) { g.P = -7; return launch_gemm_p256<EPI, 1>(g, st); }
  if (v == 61) return launch_gemm_p256<EPI, 1, 3>(g, st);   // + L2 prefetch of A, 3 K-steps ahead
  if (v == 63) return launch_gemm_p256<EPI, 1, 12>(g, st);
  if (v == 64) return launch_gemm_p256<EPI, 1, 106>(g, st);  // A and W prefetch, 6 ahead   // compute only, no barrier (timing probe; wrong results)
  if (v == 8) { g.P = -3; return launch_gemm_p256<EPI>(g, st); }
  if (v == 9) { g.P = -4; return launch_gemm_p256<EPI>(g, st); }
  if (v == 30) return launch_gemm_p256<EPI, 0>(g, st);
  if (v >= 40 && v < 50) { g.ngrp = v - 40; return launch_gemm_p256<EPI, 1>(g, st); }               // forced N groups (0 = auto)
  if (v >= 50 && v < 60) { g.P = -6; g.ngrp = v - 50; return launch_gemm_p256<EPI, 1>(g, st); }   // K-rotation on, forced N groups   // A/B: all DMA up front, issued by alternating wave rows   // DMA pieces interleaved between MFMA groups
  if (v >= 10 && v < 20) { g.ngrp = v - 10; return launch_gemm_p3<EPI>(g, st); }
  if (v >= 20 && v < 30) { g.ngrp = v - 20; g.P = -3; return launch_gemm_p3<EPI>(g, st); }
  if (v == 3) { g.P = -1; return launch_gemm_p3<EPI>(g, st); }
  if (v == 4) { g.P = -2; return launch_gemm_p3<EPI>(g, st); }
  if (v == 5) { g.P = -3; return launch_gemm_p3<EPI>(g, st); }
  return launch_gemm_p3<EPI>(g, st);
}

}  // namespace

bool tspo::gemm_bf16_is_big(long M, int N, int K) {
  return M * N >= (long)256 * 256 * 256 && K >= 128 && N <= 4096;
}

namespace {
// LayerNorm-folded epilogues exist only in the persistent 256x256 kernel
template <int EPI>
int launch_gemm_ln(GemmArgs g, hipStream_t st) {
  if (!tspo::gemm_bf16_is_big(g.M, g.N, g.K) || g.N % 64)
    return tspo::set_err(TSPO_EINVAL, "gemm: LayerNorm-folded epilogue %d needs the 256x256 kernel (M=%d N=%d K=%d)", EPI, g.M, g.N, g.K);
  if (EPI == GE_RESID_ST ? !g.spart : !(g.lnc && g.rstats))
    return tspo::set_err(TSPO_EINVAL, "gemm: epilogue %d without its statistics pointers", EPI);
  if (g.variant == 80 || (g.variant == 0 && default_big_variant() == 80)) return launch_gemm_a4<EPI>(g, st);
  if (g.variant == 81 || (g.variant == 0 && default_big_variant() == 81)) return launch_gemm_a5<EPI>(g, st);
  if (g.variant == 82 || (g.variant == 0 && default_big_variant() == 82)) return launch_gemm_a7<EPI>(g, st);
  return launch_gemm_p256<EPI, 1, 6>(g, st);
}
}  // namespace

int tspo::gemm_bf16(int epi, const GemmArgs& g, hipStream_t st) {
  switch (epi) {
    case GE_BIAS_LN: return launch_gemm_ln<GE_BIAS_LN>(g, st);
    case GE_GELU_LN: return launch_gemm_ln<GE_GELU_LN>(g, st);
    case GE_RESID_ST: return launch_gemm_ln<GE_RESID_ST>(g, st);
    case GE_BIAS: return launch_gemm<GE_BIAS>(g, st);
    case GE_GELU: return launch_gemm<GE_GELU>(g, st);
    case GE_RESID: return launch_gemm<GE_RESID>(g, st);
    case GE_F32: return launch_gemm<GE_F32>(g, st);
    case GE_PATCH: return launch_gemm<GE_PATCH>(g, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm: bad epilogue %d", epi);
}
