// CLIP ViT frame encoder for gfx950 (MI355X): bf16 MFMA GEMMs (fp32 accumulate)
// with LDS-staged, XOR-swizzled tiles fed by global_load_lds (LDS-DMA), a
// 257-token flash-style attention kernel that keeps the whole score row in
// registers, wave-per-row LayerNorm and a fused patch gather (+u8 normalise).
//
// Replaces transformers' CLIPVisionTransformer + visual_projection as called
// by the reference at model/temporal_agent.py:166, tspo_trainer.py:401.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;

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
enum { GE_BIAS = 0, GE_GELU = 1, GE_RESID = 2, GE_F32 = 3, GE_PATCH = 4 };

#define GT_BM 128
#define GT_BN 128
#define GT_BK 64
#define GT_STAGE_BYTES (2 * 128 * 128)  // A tile + W tile, 16 KB each

struct GemmArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; const bf16_t* R; void* C;
  const float* pos;  // GE_PATCH: pos_emb [S, N]
  int M, N, K, tilesN, nwg, P;  // P: patches per frame (GE_PATCH row remap); P < 0 = ablation hooks (tests only)
  int variant;                  // 0 = auto; 1 = 128x128 2-stage; 2 = persistent 256x128 ring; 6 = persistent 256x256
  int ngrp;                     // 0 = auto N-group count per XCD
};

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
  if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID)
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

// Epilogue of one 256x256 tile for a wave owning rows wm*128.. and columns wn*64.. (shared by the ring kernels).
template <int EPI>
__device__ __forceinline__ void g3_epilogue(const GemmArgs& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn,
                                            int l15, int q4, const float* lbias) {
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wm * 128 + mi * 16 + l15;
    size_t orow = (size_t)m;
    int prow = 0;
    if (EPI == GE_PATCH) {
      const int f = m / g.P;
      prow = 1 + (m - f * g.P);
      orow = (size_t)f * (g.P + 1) + prow;
    }
    uint2 pk[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
      f32x4 v = acc[ni][mi];
      acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const bool ok = m < g.M && n < g.N;
      if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID) v += *reinterpret_cast<const f32x4*>(lbias + (n < g.N ? n : 0));
      if (EPI == GE_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
      }
      const size_t o = orow * g.N + n;
      if (EPI == GE_PATCH && ok) v += *reinterpret_cast<const f32x4*>(g.pos + (size_t)prow * g.N + n);
      if (EPI == GE_RESID && ok) {
        const uint2 rv = *reinterpret_cast<const uint2*>(g.R + o);
        v[0] += bf16_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
        v[2] += bf16_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
      }
      if (EPI == GE_F32) {
        if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
      } else {
        pk[ni].x = pack_bf16x2(v[0], v[1]);
        pk[ni].y = pack_bf16x2(v[2], v[3]);
      }
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
}

template <int EPI, int MODE, int PF, int EARLY>
__global__ __launch_bounds__(512) void gemm_bf16_p256_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE + 16384 + 256];  // the ONLY LDS object (+256 B sink of the L2 prefetch)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 2, wn = wid & 3;
  float* lbias = reinterpret_cast<float*>(lds + 2 * G3_STAGE);
  if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID)
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
  bool after_epi = true;   // first wait: nothing but the first stage is outstanding
  for (int it = 0; it < total_it; ++it) {
    // stage `it` landed everywhere; buffer (it+1)&1 is free.  Wave 0 may leave its 4 (younger) L2-prefetch ops in flight,
    // except right after an epilogue whose stores are younger still.
    if (PF > 0 && (wid == 0 || (PF > 100 && wid == 1)) && !after_epi) {
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (g.P != -7) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
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
  if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID)
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

// Variant choice: the persistent 256x256 kernel whenever there is at least one tile per CU; the small
// 128x128 kernel otherwise.  Variants 2..29 are reachable only through tspo_gemm_bf16's test hook (act >> 8):
// 2 = 256x128 ring, 3 = 2 without K-rotation, 4/5 = 2 compute-only / loads-only, 6 = 256x256, 7/8/9 = 6 compute-only /
// loads-only / A-loads-only, 1x = 2 with x N-groups, 2x = 2 loads-only with x N-groups.
template <int EPI>
int launch_gemm(GemmArgs g, hipStream_t st) {
  const bool big = (long)g.M * g.N >= (long)256 * 256 * 256 && g.K >= 128 && g.N <= 4096;
  const int v = g.variant ? g.variant : (big ? 6 : 1);
  if (v == 1) return launch_gemm_v1<EPI>(g, st);
  if (v == 6) return launch_gemm_p256<EPI, 1, 6>(g, st);     // default: interleaved DMA issue + L2 prefetch of A 6 K-steps ahead
  if (v == 65) return launch_gemm_p256<EPI, 1, 0>(g, st);    // A/B: no L2 prefetch
  if (v == 66) return launch_gemm_p256<EPI, 1, 6, 1>(g, st); // A/B: DMA pieces issued in the first quarter of the K-step
  if (v == 67) return launch_gemm_p256<EPI, 1, 6, 2>(g, st); // A/B: s_setprio(1) around each group of 8 MFMAs
  if (v == 68) return launch_gemm_p256<EPI, 1, 6, 3>(g, st); // A/B: buffer_load ... lds instead of global_load_lds
  if (v == 70) return launch_gemm_s256<EPI>(g, st);          // role-split (staggered wave rows)
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

int gemm_dispatch(int epi, const GemmArgs& g, hipStream_t st) {
  switch (epi) {
    case GE_BIAS: return launch_gemm<GE_BIAS>(g, st);
    case GE_GELU: return launch_gemm<GE_GELU>(g, st);
    case GE_RESID: return launch_gemm<GE_RESID>(g, st);
    case GE_F32: return launch_gemm<GE_F32>(g, st);
    case GE_PATCH: return launch_gemm<GE_PATCH>(g, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm: bad epilogue %d", epi);
}

// ===========================================================================
// LayerNorm over the last dim (C <= 4096, C % 8 == 0): one wave per row, the
// row lives in registers (16-byte loads), two-pass mean / variance in fp32.
// in_stride / out_stride in elements (lets the post-LN read only CLS rows).
// ===========================================================================
// LN_MAXCH = ceil(C / 512) 16-byte chunks per lane: a template parameter so that C = 1024 needs 16 value registers
// (8 waves/SIMD resident = twice the bytes in flight of the generic C <= 4096 instance).
template <int LN_MAXCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* in, bf16_t* out,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        long rows, int C, long in_stride, long out_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* x = in + row * in_stride;
  float v[LN_MAXCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < C) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + e);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[c][2 * i] = bf16_to_f32((uint16_t)(w[i] & 0xffff));
        v[c][2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16));
        s += v[c][2 * i] + v[c][2 * i + 1];
      }
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float qv = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < C) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; qv += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(qv) / (float)C + eps);
  bf16_t* y = out + row * out_stride;
#pragma unroll
  for (int c = 0; c < LN_MAXCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < C) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + e), g1 = *reinterpret_cast<const f32x4*>(gamma + e + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + e), b1 = *reinterpret_cast<const f32x4*>(beta + e + 4);
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = (v[c][i] - mean) * rstd * g0[i] + b0[i];
        o[4 + i] = (v[c][4 + i] - mean) * rstd * g1[i] + b1[i];
      }
      uint4 u;
      u.x = pack_bf16x2(o[0], o[1]); u.y = pack_bf16x2(o[2], o[3]);
      u.z = pack_bf16x2(o[4], o[5]); u.w = pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<uint4*>(y + e) = u;
    }
  }
}

// ===========================================================================
// Patch gather (im2col for the stride-p conv) + CLS rows.
// patches[(n*P + py*gw + px)][k = c*p*p + ky*p + kx] = pixel[n][c][py*p+ky][px*p+kx]
// Kp = roundup(3*p*p, 64), padding zero.  TSPO_U8 input gets (x/255-mean)/std fused.
// ===========================================================================
template <typename TP>
__device__ __forceinline__ float load_pixel(const TP* p, size_t i, int c);
template <> __device__ __forceinline__ float load_pixel<float>(const float* p, size_t i, int) { return p[i]; }
template <> __device__ __forceinline__ float load_pixel<bf16_t>(const bf16_t* p, size_t i, int) { return bf16_to_f32(p[i]); }
template <> __device__ __forceinline__ float load_pixel<_Float16>(const _Float16* p, size_t i, int) { return (float)p[i]; }
template <> __device__ __forceinline__ float load_pixel<uint8_t>(const uint8_t* p, size_t i, int c) {
  const float mean = c == 0 ? 0.48145466f : (c == 1 ? 0.4578275f : 0.40821073f);
  const float sd = c == 0 ? 0.26862954f : (c == 1 ? 0.26130258f : 0.27577711f);
  return ((float)p[i] / 255.0f - mean) / sd;
}

template <typename TP>
__global__ __launch_bounds__(256) void patch_gather_kernel(const TP* __restrict__ px, bf16_t* __restrict__ out,
                                                           int n_frames, int image, int patch, int Kp) {
  const int gw = image / patch, P = gw * gw, pp = patch * patch, Kreal = 3 * pp;
  const int oct = Kp / 8;
  const size_t total = (size_t)n_frames * P * oct;
  for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
    const int o8 = (int)(id % oct);
    const size_t m = id / oct;
    const int pidx = (int)(m % P);
    const size_t n = m / P;
    const int py = pidx / gw, pxx = pidx - py * gw;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = o8 * 8 + i;
      if (k < Kreal) {
        const int c = k / pp, rem = k - c * pp;
        const int ky = rem / patch, kx = rem - ky * patch;
        const size_t src = ((n * 3 + c) * image + (size_t)(py * patch + ky)) * image + (pxx * patch + kx);
        v[i] = load_pixel<TP>(px, src, c);
      } else {
        v[i] = 0.f;
      }
    }
    uint4 u;
    u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
    u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(out + m * Kp + (size_t)o8 * 8) = u;
  }
}

// x[n*S + 0][:] = pos_emb[0][:]  (class_embedding already folded into row 0)
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ pos, bf16_t* __restrict__ x, int n_frames,
                                                       int S, int C) {
  const size_t total = (size_t)n_frames * C;
  for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
    const size_t n = id / C;
    const int c = (int)(id - n * C);
    x[n * S * C + c] = f32_to_bf16(pos[c]);
  }
}

// ===========================================================================
// Attention for one (frame, head): S <= 288 tokens, head_dim 64, non-causal.
// K rows in LDS (swizzled like the GEMM tiles), V row-major in [32 keys][16 d] sub-tiles that the
// hardware transpose read (ds_read_b64_tr_b16) turns into V^T fragments.  Each wave owns 16-query tiles:
//   S^T = K Q^T  (18 key tiles -> 72 fp32 regs hold the full score row)
//   softmax over keys in registers (lane-local + 2 xor-shuffles)
//   O^T = V^T P^T with P taken straight from the S^T accumulators (the key
//   permutation inside each 32-key chunk is shared by both MFMA operands).
// ===========================================================================
#define AT_KEYS 288
#define AT_V_BYTES (9 * 4 * 1024)  // V: 9 key chunks x 4 d-tiles of row-major [32 keys][16 d] sub-tiles (1 KB each)
#define AT_LDS_BYTES (AT_KEYS * 128 + AT_V_BYTES)
typedef __attribute__((ext_vector_type(4))) short s16x4;

// NQ query tiles (16 queries each, tiles qt and qt+qstride) against all keys of one (frame, head).
template <int NQ, int S_CT>
__device__ __forceinline__ void attn_tiles(const bf16_t* __restrict__ base, bf16_t* __restrict__ out, const char* Ks,
                                           const char* Vt, int S_rt, int C, size_t ld, size_t f, int h, float scale, int qt0,
                                           int qstride, int l15, int q4) {
  const int S = S_CT ? S_CT : S_rt;   // S_CT = 257 (ViT-L/14 @ 224): every tile-skip / mask decision folds at compile time
  // V^T fragments come from ds_read_b64_tr_b16: the 16 lanes of a row each point at 4 contiguous bf16 of a
  // row-major [4 keys][16 d] block (lane i -> key i>>2, d-chunk i&3) and receive COLUMN i of it, i.e. 4 keys of
  // their own d - so V stays row-major in LDS (16-byte staging writes, no scattered 2-byte transposition).
  int voff[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) voff[dt][hh] = dt * 1024 + (hh * 16 + q4 * 4 + (l15 >> 2)) * 32 + (l15 & 3) * 8;
  const int koff = l15 * 128;
  const int ksw = l15 & 7;
  int qrow[NQ];
  bool qvalid[NQ];
  bf16x8 qf[NQ][2];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int r = (qt0 + n * qstride) * 16 + l15;
    qvalid[n] = r < S;
    qrow[n] = qvalid[n] ? r : S - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[n][kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow[n] * ld + kk * 32 + q4 * 8);
  }
  f32x4 sc[NQ][18];
  // K fragments are software-pipelined one key tile ahead (the MFMA of tile kt never waits on an LDS read issued
  // just before it; before this the loop paid the LDS latency 36 times per pass)
  const int nkt = (S + 15) >> 4;   // key tiles that hold at least one valid key (17 for S = 257)
  bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(Ks + koff + ((q4 ^ ksw) << 4));
  bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(Ks + koff + (((4 + q4) ^ ksw) << 4));
#pragma unroll
  for (int kt = 0; kt < 18; ++kt) {
#pragma unroll
    for (int n = 0; n < NQ; ++n) sc[n][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 nf0 = kf0, nf1 = kf1;
    if (kt + 1 < 18) {   // rows past S are zero-filled in LDS, so the read itself is always safe
      nf0 = *reinterpret_cast<const bf16x8*>(Ks + (kt + 1) * 2048 + koff + ((q4 ^ ksw) << 4));
      nf1 = *reinterpret_cast<const bf16x8*>(Ks + (kt + 1) * 2048 + koff + (((4 + q4) ^ ksw) << 4));
    }
    if (kt < nkt) {  // key tiles entirely past the sequence are never multiplied (uniform branch)
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        sc[n][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[n][0], sc[n][kt], 0, 0, 0);
        sc[n][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[n][1], sc[n][kt], 0, 0, 0);
      }
    }
    if ((kt + 1) * 16 > S) {  // only the tile(s) straddling / past S need the key mask
#pragma unroll
      for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[n][kt][r] = (kt * 16 + q4 * 4 + r) < S ? sc[n][kt][r] : -INFINITY;
    }
    kf0 = nf0;
    kf1 = nf1;
  }
  // softmax over keys: lane holds keys kt*16 + q4*4 + r for query l15
  const float c2 = scale * 1.4426950408889634f;  // exp(x*scale) = 2^(x*scale*log2 e): one v_fma + one v_exp per score
  float inv[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 18; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[n][kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 18; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sc[n][kt][r], c2, -mc));
        sc[n][kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    inv[n] = 1.f / sum;
  }
  f32x4 o[NQ][4];
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[n][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  union VF { bf16x8 v; s16x4 h2[2]; };
  auto ldv = [&](int step) {   // step = c * 4 + dt
    VF f;
    const int c = step >> 2, dt = step & 3;
    f.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vt + voff[dt][0] + c * 4096));
    f.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vt + voff[dt][1] + c * 4096));
    return f;
  };
  constexpr bool VPF = true;   // (A/B: without the V prefetch the paired path measured 18.0 ms vs 15.6 ms per video)
  VF vcur = ldv(0);
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    union { bf16x8 v; uint32_t u[4]; } pf[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      pf[n].u[0] = pack_bf16x2(sc[n][2 * c][0], sc[n][2 * c][1]);
      pf[n].u[1] = pack_bf16x2(sc[n][2 * c][2], sc[n][2 * c][3]);
      pf[n].u[2] = pack_bf16x2(sc[n][2 * c + 1][0], sc[n][2 * c + 1][1]);
      pf[n].u[3] = pack_bf16x2(sc[n][2 * c + 1][2], sc[n][2 * c + 1][3]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int step = c * 4 + dt;
      VF vnext = vcur;
      if (VPF && step + 1 < 36) vnext = ldv(step + 1);
      if (!VPF) vcur = ldv(step);
#pragma unroll
      for (int n = 0; n < NQ; ++n) o[n][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcur.v, pf[n].v, o[n][dt], 0, 0, 0);
      vcur = vnext;
    }
  }
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    if (qvalid[n]) {
      bf16_t* orow = out + (f * S + qrow[n]) * (size_t)C + (size_t)h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 pk;
        pk.x = pack_bf16x2(o[n][dt][0] * inv[n], o[n][dt][1] * inv[n]);
        pk.y = pack_bf16x2(o[n][dt][2] * inv[n], o[n][dt][3] * inv[n]);
        *reinterpret_cast<uint2*>(orow + dt * 16 + q4 * 4) = pk;
      }
    }
  }
}


template <int S_CT>
__global__ __launch_bounds__(256, 2) void clip_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int S_rt,
                                                        int C, float scale) {
  const int S = S_CT ? S_CT : S_rt;
  __shared__ __attribute__((aligned(16))) char lds[AT_LDS_BYTES];
  char* Ks = lds;
  char* Vt = lds + AT_KEYS * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x;
  const size_t f = blockIdx.y;
  const size_t ld = (size_t)3 * C;
  const bf16_t* base = qkv + f * S * ld + (size_t)h * 64;

  // ---- stage K (swizzled rows) and V (row-major sub-tiles) -------------------
  // All 18 16-byte loads of a thread are issued before the first LDS write (rows past S are clamped and zeroed
  // afterwards, so there is no branch around a load): the rolled, branchy version of this loop exposed one full
  // memory latency per iteration - 9 round trips per workgroup.
  {
    uint4 kv[9], vv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int id = tid + i * 256;
      const int row = id >> 3, c = id & 7;
      const int rc = row < S ? row : S - 1;
      kv[i] = *reinterpret_cast<const uint4*>(base + (size_t)rc * ld + C + c * 8);
      vv[i] = *reinterpret_cast<const uint4*>(base + (size_t)rc * ld + 2 * C + c * 8);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int id = tid + i * 256;
      const int row = id >> 3, c = id & 7;
      const uint4 z = {0u, 0u, 0u, 0u};
      const uint4 k4 = row < S ? kv[i] : z, v4 = row < S ? vv[i] : z;
      *reinterpret_cast<uint4*>(Ks + row * 128 + ((c ^ (row & 7)) << 4)) = k4;
      *reinterpret_cast<uint4*>(Vt + ((row >> 5) * 4 + (c >> 1)) * 1024 + (row & 31) * 32 + (c & 1) * 16) = v4;
    }
  }
  __syncthreads();

  const int nqt = (S + 15) >> 4;
  // wave w owns query tiles w, w+4, w+8, ...; they are processed two at a time so every K / Vt fragment read from
  // LDS feeds two MFMAs (LDS bytes per MFMA halve: at one tile per pass the kernel read 1 KB of LDS per MFMA and
  // was co-limited by LDS bandwidth).
  int qt = wid;
  while (qt < nqt) {
    asm volatile("" ::: "memory");  // keep the loop-invariant K / Vt fragment reads inside the loop (register budget)
    if (qt + 4 < nqt) {
      attn_tiles<2, S_CT>(base, out, Ks, Vt, S, C, ld, f, h, scale, qt, 4, l15, q4);
      qt += 8;
    } else {
      attn_tiles<1, S_CT>(base, out, Ks, Vt, S, C, ld, f, h, scale, qt, 4, l15, q4);
      qt += 4;
    }
  }
}

// ===========================================================================
struct ClipWs {
  bf16_t *x, *h, *qkv, *a, *u, *patches, *pooled;
  size_t bytes;
};

ClipWs clip_carve(void* ws, const tspo_clip_config& c, int n) {
  tspo::Carver cv(ws);
  ClipWs w;
  const int gw = c.image / c.patch, P = gw * gw, S = P + 1;
  const size_t M = (size_t)n * S;
  const int Kp = (int)tspo::align_up((size_t)3 * c.patch * c.patch, 64);
  w.x = cv.take<bf16_t>(M * c.hidden);
  w.h = cv.take<bf16_t>(M * c.hidden);
  w.qkv = cv.take<bf16_t>(M * 3 * c.hidden);
  w.a = cv.take<bf16_t>(M * c.hidden);
  w.u = cv.take<bf16_t>(M * c.mlp);
  w.patches = cv.take<bf16_t>((size_t)n * P * Kp);
  w.pooled = cv.take<bf16_t>((size_t)n * c.hidden);
  w.bytes = cv.bytes();
  return w;
}

int clip_check_cfg(const tspo_clip_config& c) {
  TSPO_REQUIRE(c.hidden >= 64 && c.hidden % 64 == 0 && c.hidden <= 4096, "clip: hidden=%d must be a multiple of 64 <= 4096", c.hidden);
  TSPO_REQUIRE(c.heads >= 1 && c.hidden == c.heads * 64, "clip: head_dim must be 64 (hidden=%d heads=%d)", c.hidden, c.heads);
  TSPO_REQUIRE(c.mlp >= 64 && c.mlp % 64 == 0, "clip: mlp=%d must be a multiple of 64", c.mlp);
  TSPO_REQUIRE(c.proj >= 8 && c.proj % 8 == 0, "clip: proj=%d must be a multiple of 8", c.proj);
  TSPO_REQUIRE(c.patch >= 1 && c.image >= c.patch && c.image % c.patch == 0, "clip: image=%d patch=%d", c.image, c.patch);
  const int gw = c.image / c.patch;
  TSPO_REQUIRE(gw * gw + 1 <= AT_KEYS, "clip: %d tokens exceed the %d-token attention kernel", gw * gw + 1, AT_KEYS);
  TSPO_REQUIRE(c.layers >= 0, "clip: layers=%d", c.layers);
  return TSPO_OK;
}

// Optional per-kernel-class timing with HIP events on the launch stream (bench / profiling only).
enum { PK_GEMM = 0, PK_ATTN = 1, PK_LN = 2, PK_GATHER = 3, PK_NKIND = 4 };
struct Prof {
  hipStream_t st;
  hipEvent_t ev[512];
  int kind[512];
  int n = 0;
  bool on = false;
  void start(hipStream_t s) { st = s; on = true; n = 0; tick(-1); }
  void tick(int k) {
    if (!on || n >= 512) return;
    (void)hipEventCreate(&ev[n]);
    (void)hipEventRecord(ev[n], st);
    kind[n] = k;
    ++n;
  }
  void finish(float* ms /* PK_NKIND + 2 */) {
    for (int i = 0; i < PK_NKIND + 2; ++i) ms[i] = 0.f;
    if (!on || n == 0) return;
    (void)hipEventSynchronize(ev[n - 1]);
    int ngemm = 0;
    for (int i = 1; i < n; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ev[i - 1], ev[i]);
      ms[kind[i]] += t;
      ms[PK_NKIND] += t;
      ngemm += kind[i] == PK_GEMM;
    }
    ms[PK_NKIND + 1] = (float)ngemm;
    for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]);
  }
};

int run_ln(const bf16_t* in, bf16_t* out, const float* g, const float* b, long rows, int C, long is, long os, float eps,
           hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (C <= 1024) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, in, out, g, b, rows, C, is, os, eps);
  else if (C <= 2048) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, in, out, g, b, rows, C, is, os, eps);
  else hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, st, in, out, g, b, rows, C, is, os, eps);
  return tspo::check_launch("layernorm");
}

}  // namespace

extern "C" size_t tspo_clip_workspace_bytes(const tspo_clip_config* cfg, int n_frames) {
  if (!cfg || n_frames < 1 || cfg->patch < 1 || cfg->image < cfg->patch) return 0;
  return clip_carve(nullptr, *cfg, n_frames).bytes;
}

extern "C" int tspo_gemm_bf16(const void* A, const void* W, const float* bias, const void* residual, void* C,
                              int out_dtype, int M, int N, int K, int act, tspo_stream_t stream) {
  TSPO_REQUIRE(A && W && C, "gemm_bf16: null pointer");
  TSPO_REQUIRE(M >= 1 && N >= 8 && N % 8 == 0 && K >= 64 && K % 64 == 0, "gemm_bf16: bad dims M=%d N=%d K=%d", M, N, K);
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.R = (const bf16_t*)residual; g.C = C;
  g.M = M; g.N = N; g.K = K; g.P = 1;
  const int variant = act >> 8;
  act &= 0xff;
  int epi;
  if (out_dtype == TSPO_F32) {
    TSPO_REQUIRE(!bias && !residual && act == 0, "gemm_bf16: f32 output supports no epilogue");
    epi = GE_F32;
  } else {
    TSPO_REQUIRE(out_dtype == TSPO_BF16, "gemm_bf16: out_dtype must be TSPO_BF16 or TSPO_F32");
    TSPO_REQUIRE(bias, "gemm_bf16: bf16 output needs a bias vector");
    TSPO_REQUIRE(!(residual && act), "gemm_bf16: residual and activation are exclusive");
    epi = residual ? GE_RESID : (act == 1 ? GE_GELU : GE_BIAS);
  }
  g.variant = variant;
  return gemm_dispatch(epi, g, (hipStream_t)stream);
}

static int clip_forward_impl(const tspo_clip_weights* w, const void* pixels, int pixel_dtype, int n_frames, float* feat,
                             void* workspace, size_t workspace_bytes, tspo_stream_t stream, Prof& prof) {
  TSPO_REQUIRE(w && pixels && feat && workspace, "clip_vit_forward: null pointer");
  TSPO_REQUIRE(n_frames >= 1, "clip_vit_forward: n_frames=%d", n_frames);
  const tspo_clip_config& c = w->cfg;
  if (int e = clip_check_cfg(c)) return e;
  TSPO_REQUIRE(w->patch_w && w->pos_emb && w->pre_g && w->pre_b && w->post_g && w->post_b && w->proj_w &&
                   (c.layers == 0 || w->layers),
               "clip_vit_forward: null weight pointer");
  ClipWs b = clip_carve(workspace, c, n_frames);
  if (workspace_bytes < b.bytes)
    return tspo::set_err(TSPO_EWORKSPACE, "clip_vit_forward: workspace %zu < %zu", workspace_bytes, b.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int gw = c.image / c.patch, P = gw * gw, S = P + 1, C = c.hidden;
  const int Kp = (int)tspo::align_up((size_t)3 * c.patch * c.patch, 64);
  const long M = (long)n_frames * S;
  TSPO_REQUIRE(M * (long)c.mlp < (1L << 40) && M < (1L << 31), "clip_vit_forward: n_frames too large");

  // 1. patch gather (+ normalise) and CLS rows
  {
    const size_t total = (size_t)n_frames * P * (Kp / 8);
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 65536u) nb = 65536u;
    switch (pixel_dtype) {
      case TSPO_F32: hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)pixels, b.patches, n_frames, c.image, c.patch, Kp); break;
      case TSPO_BF16: hipLaunchKernelGGL(patch_gather_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)pixels, b.patches, n_frames, c.image, c.patch, Kp); break;
      case TSPO_F16: hipLaunchKernelGGL(patch_gather_kernel<_Float16>, dim3(nb), dim3(256), 0, st, (const _Float16*)pixels, b.patches, n_frames, c.image, c.patch, Kp); break;
      case TSPO_U8: hipLaunchKernelGGL(patch_gather_kernel<uint8_t>, dim3(nb), dim3(256), 0, st, (const uint8_t*)pixels, b.patches, n_frames, c.image, c.patch, Kp); break;
      default: return tspo::set_err(TSPO_EINVAL, "clip_vit_forward: bad pixel_dtype %d", pixel_dtype);
    }
    unsigned cb = (unsigned)(((size_t)n_frames * C + 255) / 256);
    if (cb > 4096u) cb = 4096u;
    hipLaunchKernelGGL(cls_rows_kernel, dim3(cb), dim3(256), 0, st, w->pos_emb, b.x, n_frames, S, C);
    prof.tick(PK_GATHER);
  }
  // 2. patch embedding GEMM (+ position embedding, rows remapped past each frame's CLS row)
  {
    GemmArgs g{};
    g.A = b.patches; g.W = (const bf16_t*)w->patch_w; g.C = b.x; g.pos = w->pos_emb;
    g.M = n_frames * P; g.N = C; g.K = Kp; g.P = P;
    if (int e = launch_gemm<GE_PATCH>(g, st)) return e;
    prof.tick(PK_GEMM);
  }
  // 3. pre-LN (in place)
  if (int e = run_ln(b.x, b.x, w->pre_g, w->pre_b, M, C, C, C, c.ln_eps, st)) return e;
  prof.tick(PK_LN);
  // 4. transformer blocks
  for (int l = 0; l < c.layers; ++l) {
    const tspo_clip_layer& L = w->layers[l];
    TSPO_REQUIRE(L.ln1_g && L.ln1_b && L.wqkv && L.bqkv && L.wo && L.bo && L.ln2_g && L.ln2_b && L.w1 && L.b1 && L.w2 && L.b2,
                 "clip_vit_forward: null pointer in layer %d", l);
    if (int e = run_ln(b.x, b.h, L.ln1_g, L.ln1_b, M, C, C, C, c.ln_eps, st)) return e;
    prof.tick(PK_LN);
    GemmArgs g{};
    g.A = b.h; g.W = (const bf16_t*)L.wqkv; g.bias = L.bqkv; g.C = b.qkv; g.M = (int)M; g.N = 3 * C; g.K = C; g.P = 1;
    if (int e = launch_gemm<GE_BIAS>(g, st)) return e;
    prof.tick(PK_GEMM);
    if (S == 257) hipLaunchKernelGGL(clip_attn_kernel<257>, dim3(c.heads, n_frames), dim3(256), 0, st, b.qkv, b.a, S, C, 0.125f);
    else hipLaunchKernelGGL(clip_attn_kernel<0>, dim3(c.heads, n_frames), dim3(256), 0, st, b.qkv, b.a, S, C, 0.125f);
    if (int e = tspo::check_launch("clip_attn")) return e;
    prof.tick(PK_ATTN);
    g = GemmArgs{};
    g.A = b.a; g.W = (const bf16_t*)L.wo; g.bias = L.bo; g.R = b.x; g.C = b.x; g.M = (int)M; g.N = C; g.K = C; g.P = 1;
    if (int e = launch_gemm<GE_RESID>(g, st)) return e;
    prof.tick(PK_GEMM);
    if (int e = run_ln(b.x, b.h, L.ln2_g, L.ln2_b, M, C, C, C, c.ln_eps, st)) return e;
    prof.tick(PK_LN);
    g = GemmArgs{};
    g.A = b.h; g.W = (const bf16_t*)L.w1; g.bias = L.b1; g.C = b.u; g.M = (int)M; g.N = c.mlp; g.K = C; g.P = 1;
    if (int e = launch_gemm<GE_GELU>(g, st)) return e;
    prof.tick(PK_GEMM);
    g = GemmArgs{};
    g.A = b.u; g.W = (const bf16_t*)L.w2; g.bias = L.b2; g.R = b.x; g.C = b.x; g.M = (int)M; g.N = C; g.K = c.mlp; g.P = 1;
    if (int e = launch_gemm<GE_RESID>(g, st)) return e;
    prof.tick(PK_GEMM);
  }
  // 5. CLS pool + post-LN + projection
  if (int e = run_ln(b.x, b.pooled, w->post_g, w->post_b, n_frames, C, (long)S * C, C, c.ln_eps, st)) return e;
  prof.tick(PK_LN);
  {
    GemmArgs g{};
    g.A = b.pooled; g.W = (const bf16_t*)w->proj_w; g.C = feat; g.M = n_frames; g.N = c.proj; g.K = C; g.P = 1;
    if (int e = launch_gemm<GE_F32>(g, st)) return e;
    prof.tick(PK_GEMM);
  }
  return TSPO_OK;
}

extern "C" int tspo_clip_vit_forward(const tspo_clip_weights* w, const void* pixels, int pixel_dtype, int n_frames,
                                     float* feat, void* workspace, size_t workspace_bytes, tspo_stream_t stream) {
  Prof prof;
  return clip_forward_impl(w, pixels, pixel_dtype, n_frames, feat, workspace, workspace_bytes, stream, prof);
}

extern "C" int tspo_clip_vit_profile(const tspo_clip_weights* w, const void* pixels, int pixel_dtype, int n_frames,
                                     float* feat, void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                                     float* host_ms6) {
  TSPO_REQUIRE(host_ms6, "clip_vit_profile: null host_ms6");
  static thread_local Prof prof;  // 512 events: keep it off the stack
  prof.start((hipStream_t)stream);
  const int rc = clip_forward_impl(w, pixels, pixel_dtype, n_frames, feat, workspace, workspace_bytes, stream, prof);
  prof.finish(host_ms6);
  prof.on = false;
  return rc;
}
