// bf16 MFMA GEMMs of the CLIP encoder for gfx950 (MI355X): C = A * W^T with fused epilogues.
// Production kernel: gemm_bf16_p256_kernel (persistent 256x256x64, LDS-DMA ring).  The other kernels are the
// small-problem path and A/B variants reachable through tspo_gemm_bf16's test hook.
#include "gemm_epilogue.h"
#include <stdlib.h>

// Development hooks (A/B kernel variants, ablation branches, s_memtime probes) exist only in a -DTSPO_DEV_HOOKS build
// (python -m tspo_amd.build --dev); the shipped library carries the small-problem kernel, the 8-wave ring kernel in its
// production configuration and the 4-wave AGPR kernel of gemm_agpr.hip - nothing that computes deliberately wrong results.
#ifdef TSPO_DEV_HOOKS
#define ABL(g, n) ((g).P == -(n))        /* ablation n requested through GemmArgs.P (tools only) */
#define ABL_LE(g, n) ((g).P <= -(n))
#else
#define ABL(g, n) false
#define ABL_LE(g, n) false
#endif

namespace {


// ===========================================================================
// GEMM  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous, bf16)
// 128x128x64 workgroup tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32.
// LDS image per operand tile: [128 rows][128 B], 16-B chunk c of row r stored
// at chunk (c ^ (r & 7)) - the XOR is applied on the global SOURCE address of
// the LDS-DMA (destination must stay lane-linear) and again on the ds_read.
// The MFMA is issued "swapped" (A-operand = W fragment, B-operand = A fragment)
// so that each lane ends up with 4 consecutive N for one M -> 8-byte stores.
// ===========================================================================
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


#ifdef TSPO_DEV_HOOKS
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

#endif  // TSPO_DEV_HOOKS

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
  if (ABL(g, 4)) return;  // dev hook: A half only
#pragma unroll
  for (int p = 0; p < 8; ++p) {  // W: 32 pieces of 8 rows
    const int piece = j * 8 + p;
    int gr = n0 + piece * 8 + rin;
    gr = gr < g.N ? gr : g.N - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W + (size_t)gr * g.K + koff),
                                     (__attribute__((address_space(3))) void*)(buf + G3_BM * 128 + piece * 1024), 16, 0, 0);
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
  int i_rot = !ABL(g, 6) ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
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
      i_rot = !ABL(g, 6) ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
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
    } else if (!ABL(g, 7)) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (EARLY == 4) { const unsigned long long t1 = __builtin_readcyclecounter(); dbg_vm += t1 - tw0; tw0 = t1; }
      asm volatile("s_barrier" ::: "memory");
    }
    if (EARLY == 4) dbg_bar += __builtin_readcyclecounter() - tw0;
    after_epi = false;
    const char* cur = lds + (it & 1) * G3_STAGE;
    if (MODE == 1) {
      // default: every wave issues its own 8 pieces, two behind each group of 16 MFMAs of the first half K-step
      const bool more = i_it < total_it && !ABL_LE(g, 2);
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
          i_rot = !ABL(g, 6) ? 0 : (int)(((long)(i_s % n_per) * nk) / n_per);
        }
      }
    } else {
    if (i_it < total_it && !ABL(g, 2)) issue_next();
    if (!ABL_LE(g, 3))
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


#ifdef TSPO_DEV_HOOKS
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

#endif  // TSPO_DEV_HOOKS

// kernel used for "big" problems: the 4-wave AGPR kernel with LDS-DMA operands of gemm_dma.hip (77; round 4) for every K that
// is a multiple of 64 and at least 128; the register-staged 4-wave kernel (82, needs K % 128 == 0) and the 8-wave LDS-DMA ring
// kernel (6) stay selectable.  In --dev builds TSPO_GEMM_VARIANT overrides it (whole-encoder A/B runs); the shipped
// library reads no environment variables.
static int default_big_variant(int K) {
#if defined(TSPO_DEV_HOOKS) || defined(TSPO_A9_LAB)
  static const int forced = [] {
    const char* e = getenv("TSPO_GEMM_VARIANT");
    return e && atoi(e) > 0 ? atoi(e) : 0;
  }();
  if (forced) return forced;
#endif
  return (K % 64 == 0 && K >= 128) ? 77 : 6;
}

template <int EPI>
int launch_gemm(GemmArgs g, hipStream_t st) {
  const bool big = tspo::gemm_bf16_is_big(g.M, g.N, g.K);
  const int v = g.variant ? g.variant : (big ? default_big_variant(g.K) : 1);
  g.variant = v;
  if (v == 1) return launch_gemm_v1<EPI>(g, st);
  if (v == 6) return launch_gemm_p256<EPI, 1, 6>(g, st);     // 8-wave LDS-DMA ring: interleaved DMA issue + L2 prefetch of A 6 K-steps ahead
  if (v >= 72 && v < 78) return tspo::gemm_bf16_dma(EPI, g, st);      // 4-wave AGPR kernel, LDS-DMA operands (gemm_dma.hip)
  if (v >= 78 && v < 100) return tspo::gemm_bf16_agpr(EPI, g, st);   // 4-wave kernel with AGPR accumulators (gemm_agpr.hip)
#ifdef TSPO_DEV_HOOKS
  // A/B variants and ablations (tools/bench_gemm.py, tools/probe_gemm_wait.py); several compute wrong results on purpose
  if (v == 65) return launch_gemm_p256<EPI, 1, 0>(g, st);    // no L2 prefetch
  if (v == 66) return launch_gemm_p256<EPI, 1, 6, 1>(g, st); // DMA pieces issued in the first quarter of the K-step
  if (v == 67) return launch_gemm_p256<EPI, 1, 6, 2>(g, st); // s_setprio(1) around each group of 8 MFMAs
  if (v == 68) return launch_gemm_p256<EPI, 1, 6, 3>(g, st); // buffer_load ... lds instead of global_load_lds
  if (v == 69) return launch_gemm_p256<EPI, 1, 6, 4>(g, st); // timing probe (s_memtime around the per-K-step wait); g.pos = debug buffer
  if (v == 70) return launch_gemm_s256<EPI>(g, st);          // role-split (staggered wave rows)
  if (v == 71) return launch_gemm_w16<EPI>(g, st);           // 16 waves per workgroup (4 per SIMD)
  if (v == 7) { g.P = -2; return launch_gemm_p256<EPI, 1>(g, st); }
  if (v == 60) { g.P = -7; return launch_gemm_p256<EPI, 1>(g, st); }
  if (v == 61) return launch_gemm_p256<EPI, 1, 3>(g, st);
  if (v == 63) return launch_gemm_p256<EPI, 1, 12>(g, st);
  if (v == 64) return launch_gemm_p256<EPI, 1, 106>(g, st);
  if (v == 8) { g.P = -3; return launch_gemm_p256<EPI>(g, st); }
  if (v == 9) { g.P = -4; return launch_gemm_p256<EPI>(g, st); }
  if (v == 30) return launch_gemm_p256<EPI, 0>(g, st);
  if (v >= 40 && v < 50) { g.ngrp = v - 40; return launch_gemm_p256<EPI, 1>(g, st); }
  if (v >= 50 && v < 60) { g.P = -6; g.ngrp = v - 50; return launch_gemm_p256<EPI, 1>(g, st); }
  if (v >= 10 && v < 20) { g.ngrp = v - 10; return launch_gemm_p3<EPI>(g, st); }
  if (v >= 20 && v < 30) { g.ngrp = v - 20; g.P = -3; return launch_gemm_p3<EPI>(g, st); }
  if (v == 3) { g.P = -1; return launch_gemm_p3<EPI>(g, st); }
  if (v == 4) { g.P = -2; return launch_gemm_p3<EPI>(g, st); }
  if (v == 5) { g.P = -3; return launch_gemm_p3<EPI>(g, st); }
  if (v == 2) return launch_gemm_p3<EPI>(g, st);
#endif
  return tspo::set_err(TSPO_EINVAL, "gemm: kernel variant %d is not part of this build", v);
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
  { const int v = g.variant ? g.variant : default_big_variant(g.K);
    if (v >= 72 && v < 78) { GemmArgs h = g; h.variant = v; return tspo::gemm_bf16_dma(EPI, h, st); }
    if (v >= 78 && v < 100) { GemmArgs h = g; h.variant = v; return tspo::gemm_bf16_agpr(EPI, h, st); } }
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
