// Temporal scoring head (MultiModal_Align) forward + backward for gfx950, fp32.
//
// fp32 end-to-end so that greedy top-k indices match the fp32 oracle: the
// dense projections run on the exact-f32 matrix cores
// (v_mfma_f32_16x16x4_f32, an fmaf chain per output), operands streamed from
// L2 straight into MFMA fragments (the whole problem - <= 50 MB - is
// L2/Infinity-Cache resident, an LDS round trip would be pure overhead).  The
// windowed attention is a banded kernel: O(T*w) work, the T x T mask and
// score tensors of the reference (temporal_agent.py:40-51, 97-104) never exist.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// Split-precision operands ("bf16x3"): x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi); a product is
// taken as hi*hi + hi*lo + lo*hi on the bf16 MFMA (fp32 accumulate), dropping only lo*lo ~ 2^-16 relative.  Optional
// mode of the selector GEMMs for training (TSPO_SEL_BF16X3): ~1e-5 relative error instead of fp32's ~1e-6, far inside
// the bf16 autocast the reference trains with, at several times the fp32-MFMA rate.
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  union { bf16x8 v; uint32_t u[4]; } h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h.u[j] = pack_bf16x2(x[2 * j], x[2 * j + 1]);
    const float r0 = x[2 * j] - __uint_as_float(h.u[j] << 16);
    const float r1 = x[2 * j + 1] - __uint_as_float(h.u[j] & 0xffff0000u);
    l.u[j] = pack_bf16x2(r0, r1);
  }
  hi = h.v;
  lo = l.v;
}

// bf16 operands, fp32 accumulation (TSPO_SEL_BF16, inference): the operand rounded to nearest-even bf16 - what a bf16 model holds
// (gen_id_tspo.py:55) - one MFMA per 32-deep slab; products of bf16 values are exact in the fp32 accumulator.
__device__ __forceinline__ bf16x8 round_bf16x8(const float (&x)[8]) {
  union { bf16x8 v; uint32_t u[4]; } h;
#pragma unroll
  for (int j = 0; j < 4; ++j) h.u[j] = pack_bf16x2(x[2 * j], x[2 * j + 1]);
  return h.v;
}

namespace {

// ---------------------------------------------------------------------------
// x_pe = x + pe, pe[t,2i] = sin((t/T) * exp(2i * -ln(1e4)/C)), pe[t,2i+1] = cos(..)
// (model/temporal_agent.py:10-19, 128-129); all fp32 like torch.
__global__ __launch_bounds__(256) void posenc_add_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int T, int D, size_t total) {
  // one thread per (row, channel pair): the pair (2i, 2i+1) shares its angle - one expf and one sincosf for two outputs
  // (same values as separate sinf / cosf calls), 8-byte loads and stores.  D is even (D % 64 == 0).
  const float c = -9.21034049987793f / (float)D;
  const size_t pairs = total >> 1;
  const int hd = D >> 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (size_t)gridDim.x * 256) {
    const int pi = (int)(i % hd);
    const int t = (int)((i / hd) % T);
    const float div = expf((float)(2 * pi) * c);
    const float pos = (float)t / (float)T;
    const float a = pos * div;
    float sn, cs;
    sincosf(a, &sn, &cs);
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    reinterpret_cast<float2*>(out)[i] = make_float2(v.x + sn, v.y + cs);
  }
}

// ---------------------------------------------------------------------------
// C[M,N] = A[M,K] * W[N,K]^T (+ bias[N]) with optional ReLU / residual add /
// relu-mask multiply.  One wave = 32x32 outputs (2x2 MFMA tiles), 4 waves =
// 64x64 per workgroup.  K % 16 == 0, N % 64 == 0, rows clamped (any M).
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_MASK = 3 };

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                          const float* __restrict__ bias, const float* __restrict__ R,
                                                          float* __restrict__ C, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 64 + (wid >> 1) * 32;
  const int n0 = blockIdx.x * 64 + (wid & 1) * 32;
  const float* ap[2];
  const float* wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = m0 + 16 * i + l15;
    r = r < M ? r : M - 1;
    ap[i] = A + (size_t)r * K + 4 * q;
    wp[i] = W + (size_t)(n0 + 16 * i + l15) * K + 4 * q;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // software pipeline, prefetch distance 2 K-steps (the problem is L2 resident but only ~2 waves share a SIMD, so
  // an un-prefetched loop exposes the whole L2 latency every 16 MFMAs: measured 34 % of the fp32 MFMA peak)
  // Loads past the end re-read the last K-step (clamped address, result unused): keeping the loads unconditional lets
  // the compiler emit counted vmcnt waits - with branches around them it falls back to vmcnt(0) and the prefetch is lost.
  f32x4 a0[2], b0[2], a1[2], b1[2], a2[2], b2[2];
  const int klast = K - 16;
  auto ld = [&](f32x4 (&a)[2], f32x4 (&b)[2], int k0) {
    k0 = k0 < klast ? k0 : klast;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const f32x4*>(ap[i] + k0);
      b[i] = *reinterpret_cast<const f32x4*>(wp[i] + k0);
    }
  };
  auto mm = [&](const f32x4 (&a)[2], const f32x4 (&b)[2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
  };
  ld(a0, b0, 0);
  ld(a1, b1, 16);
  const int nsteps = K >> 4;
  int st = 0;
  for (; st + 3 <= nsteps; st += 3) {   // three register sets rotate; prefetch distance 2 K-steps
    ld(a2, b2, (st + 2) << 4);
    mm(a0, b0);
    ld(a0, b0, (st + 3) << 4);
    mm(a1, b1);
    ld(a1, b1, (st + 4) << 4);
    mm(a2, b2);
  }
  if (st < nsteps) mm(a0, b0);
  if (st + 1 < nsteps) mm(a1, b1);
  // D layout: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 16 * j + l15;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + q * 4 + r;
        if (row < M) {
          float v = acc[i][j][r] + bv;
          const size_t o = (size_t)row * N + col;
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI == EPI_RESID) v += R[o];
          if (EPI == EPI_MASK) v = R[o] > 0.f ? v : 0.f;
          C[o] = v;
        }
      }
    }
}

// LDS-staged version of the same GEMM (used when K % 32 == 0): the fragment-shaped global loads of the kernel above
// (16 rows x 64 B per wave instruction) are address-processing bound; here each 32-float-deep operand slab is brought
// in as full 128-byte rows by LDS-DMA (XOR-swizzled like the bf16 GEMM tiles) and fragments come from ds_read_b128.
// One workgroup = 64 x (32*WN) outputs, NWM x 2 waves of (64/NWM) x (16*WN); 3-deep ring with counted vmcnt waits.
// WN = 3 (64x96 tiles) is used when N % 96 == 0: for D = 768 that makes BT=2048 x 768 exactly 256 workgroups (one
// per CU) and x 2304 exactly 3 per CU, instead of 1.5 / 4.5 with 64x64 tiles, and gives 48 MFMAs per barrier.
// Small-M form (BM = 32: NWM = 2, MI = 1, four waves): with BT = 512 rows - one prompt per micro-step, the reference's training
// configuration - 64-row tiles give 64 (DxD) or 192 (Dx3D) workgroups for 256 CUs and the launch takes one workgroup's 24 K-steps
// (26-27 us whatever N is); 32-row tiles spread the same MFMAs over 2-4x as many CUs.
template <int WN, int NT_NST, int BM = 64, int SPB = 1>
constexpr int nt_lds_bytes() { return NT_NST * SPB * (BM / 8 + 4 * WN) * 1024; }

// (body of the kernel: also instantiated inside dgrad_wgrad_kernel, which runs it next to a weight-gradient tile set)
// HALVES = 2: the workgroup has 2 x NWM*2 waves and runs TWO tiles side by side (waves 0..NW-1 one, NW..2NW-1 the other: the caller
// passes each half its own lds / bx / by; the K-loops are the same length, so the workgroup-wide barriers line up).
// SPB = 2: TWO 32-deep slabs per ring stage and barrier (K / 32 even).  A K-step of a small tile is ~650 cycles of wait + barrier + DMA
// issue + fragment-read latency around 256 cycles of dependent MFMAs (cycle-counter probe, round 4): pairing the slabs halves the
// number of such steps, and the ring (NT_NST stages of two slabs) holds twice as many slabs in flight.  Same contraction order.
template <int EPI, int WN, int NWM, int NT_NST, int SPLIT, int MI = 4 / NWM, int HALVES = 1, int SPB = 1>
__device__ __forceinline__ void gemm_f32_nt_lds_body(char* lds, const float* __restrict__ A, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ R,
                                                     float* __restrict__ C, int M, int N, int K, int bx, int by) {
  constexpr int NW = NWM * 2;                  // waves: NWM along M x 2 along N
  constexpr int BM = 16 * NWM * MI;            // A rows per slab (MI = 16-row tiles per wave): 64, or 32 in the small-M form
  constexpr int AP = BM / 8;                   // 1 KB DMA pieces of the A part
  constexpr int BN = 32 * WN;                  // W rows per slab
  constexpr int PIECES = AP + BN / 8;          // 1 KB DMA pieces per slab (A: BM/8, W: BN/8)
  constexpr int PW_HI = (PIECES + NW - 1) / NW, PW_LO = PIECES / NW;   // first NHI waves move PW_HI pieces, the rest PW_LO
  constexpr int NHI = PIECES - PW_LO * NW;     // (0 when it divides evenly)
  constexpr int SLAB = PIECES * 1024;          // ring of (A 64x128 B | W BNx128 B)
  static_assert(NT_NST * SPB * SLAB == nt_lds_bytes<WN, NT_NST, BM, SPB>(), "LDS size");
  const int lane = threadIdx.x & 63, wid = HALVES == 1 ? threadIdx.x >> 6 : (threadIdx.x >> 6) % NW;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int nk = (K >> 5) / SPB;            // ring stages (SPB slabs each; the launcher checks divisibility)
  const int rin = lane >> 3, slot = lane & 7;
  const bool hi = NHI == 0 || wid < NHI;
  const int piece0 = hi ? wid * PW_HI : NHI * PW_HI + (wid - NHI) * PW_LO;
  auto stage = [&](int kt, int ring) {
#pragma unroll
    for (int sub = 0; sub < SPB; ++sub) {
    char* buf = lds + (ring * SPB + sub) * SLAB;
    const int kslab = kt * SPB + sub;
#pragma unroll
    for (int p = 0; p < PW_HI; ++p) {
      if (p < PW_LO || hi) {
        const int piece = piece0 + p;            // wave-uniform
        const float* src;
        if (piece < AP) {
          int gr = m0 + piece * 8 + rin;
          gr = gr < M ? gr : M - 1;
          src = A + (size_t)gr * K;
        } else {
          src = W + (size_t)(n0 + (piece - AP) * 8 + rin) * K;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kslab * 32 + ((slot ^ rin) << 2)),
                                         (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
      }
    }
    }
  };
  // wait until slab kt has landed: the (NT_NST - 2) newer slabs of this wave (PW pieces each) may stay in flight
  auto wait_ahead = [&](int ahead) {
    const int cnt = ahead * SPB * (hi ? PW_HI : PW_LO);
    switch (cnt) {   // s_waitcnt needs an immediate
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (conservative)
    }
  };
  static_assert(PW_HI >= 2 && PW_HI <= 5 && PW_LO >= 2, "unexpected DMA piece split");
  f32x4 acc[MI][WN];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offA[MI], offW[WN];
#pragma unroll
  for (int i = 0; i < MI; ++i) offA[i] = (wm * 16 * MI + i * 16 + l15) * 128;
#pragma unroll
  for (int j = 0; j < WN; ++j) offW[j] = AP * 1024 + (wn * 16 * WN + j * 16 + l15) * 128;
  const int sw = l15 & 7;
#pragma unroll
  for (int t = 0; t < NT_NST - 1; ++t)
    if (t < nk) stage(t, t);
  // epilogue operands (bias, residual / mask source) are fetched NOW, behind the first slabs' DMAs: loaded in the epilogue they were
  // first-touch misses on its critical path (probe: 8-10 k of a 64 k-cycle DxD launch at BT = 2048)
  float bv[WN], rv[MI][WN][4];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = n0 + wn * 16 * WN + 16 * j + l15;
    bv[j] = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 16 * MI + 16 * i + q * 4 + r;
        rv[i][j][r] = (EPI == EPI_RESID || EPI == EPI_MASK) ? R[(size_t)(row < M ? row : M - 1) * N + col] : 0.f;
      }
  }
  int ring = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // slabs kt+1 .. kt+NT_NST-2 stay in flight across the wait (2-deep ring: everything has landed)
    wait_ahead(min(NT_NST - 2, nk - 1 - kt));
    __builtin_amdgcn_s_barrier();   // slab kt visible to all waves; compute(kt-1) done everywhere -> its slot is free
    if (kt + NT_NST - 1 < nk) stage(kt + NT_NST - 1, ring >= 1 ? ring - 1 : NT_NST - 1);
    const int ring_now = ring;
    ring = ring + 1 == NT_NST ? 0 : ring + 1;
#pragma unroll
    for (int sub = 0; sub < SPB; ++sub) {
    const char* cur = lds + (ring_now * SPB + sub) * SLAB;
    if (SPLIT == 2) {
      // bf16 operands / fp32 accumulate (TSPO_SEL_BF16): the slab's fp32 values rounded to bf16 in registers, ONE MFMA per tile and slab
      const int c0 = (((2 * q) ^ sw) << 4), c1 = (((2 * q + 1) ^ sw) << 4);
      bf16x8 ah[MI], bh[WN];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(cur + offA[i] + c0), v = *reinterpret_cast<const f32x4*>(cur + offA[i] + c1);
        const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        ah[i] = round_bf16x8(x);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(cur + offW[j] + c0), v = *reinterpret_cast<const f32x4*>(cur + offW[j] + c1);
        const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        bh[j] = round_bf16x8(x);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    } else if (SPLIT) {
      // one bf16 MFMA spans the whole 32-deep slab: lane (row l15, q) owns k = 8q..8q+7 = 16-byte chunks 2q, 2q+1
      const int c0 = (((2 * q) ^ sw) << 4), c1 = (((2 * q + 1) ^ sw) << 4);
      bf16x8 ah[MI], al[MI], bh[WN], bl[WN];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(cur + offA[i] + c0), v = *reinterpret_cast<const f32x4*>(cur + offA[i] + c1);
        const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        split_bf16x8(x, ah[i], al[i]);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(cur + offW[j] + c0), v = *reinterpret_cast<const f32x4*>(cur + offW[j] + c1);
        const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        split_bf16x8(x, bh[j], bl[j]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q) ^ sw) << 4);
      f32x4 a[MI], b[WN];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(cur + offA[i] + co);
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const f32x4*>(cur + offW[j] + co);
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][st], b[j][st], acc[i][j], 0, 0, 0);
    }
    }
    }   // sub
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n0 + wn * 16 * WN + 16 * j + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 16 * MI + 16 * i + q * 4 + r;
        if (row < M) {
          float v = acc[i][j][r] + bv[j];
          const size_t o = (size_t)row * N + col;
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI == EPI_RESID) v += rv[i][j][r];
          if (EPI == EPI_MASK) v = rv[i][j][r] > 0.f ? v : 0.f;
          C[o] = v;
        }
      }
    }
}

template <int EPI, int WN, int NWM, int NT_NST, int SPLIT, int MI = 4 / NWM, int SPB = 1>
__global__ __launch_bounds__(NWM * 128) void gemm_f32_nt_lds_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ R, float* __restrict__ C,
                                                                    int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[nt_lds_bytes<WN, NT_NST, 16 * NWM * MI, SPB>()];   // the only LDS object
  gemm_f32_nt_lds_body<EPI, WN, NWM, NT_NST, SPLIT, MI, 1, SPB>(lds, A, W, bias, R, C, M, N, K, blockIdx.x, blockIdx.y);
}

template <int EPI, int SPLIT>
int launch_gemm_nt_p(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N, int K,
                     hipStream_t st) {
  const int mt = (M + 63) / 64;
  // small-M form: the 64x96 tiling would leave CUs without a workgroup (BT = 512: 64 / 192 workgroups), so 32x32 tiles (four waves)
  // spread the MFMAs over every CU: DxD at BT = 512 26 -> 12-13 us, Dx3D 27 -> 25.5 us (4.5 wave-tiles per SIMD: 32x64 / 32x96 /
  // 64x96 tiles all end at 26-27 us, the CU with one workgroup more sets the time).  Measured and not kept: fragments of slab kt+1
  // read under the MFMAs of slab kt with the DMA three slabs ahead (12.7-13.7 vs 11.9-13.0 us), a 4-deep ring (same).  Same
  // contraction order per output element as the large form: bitwise the same results.  At BT = 2048 the DxD forwards take 28 us with
  // ANY of the tilings (64x96, 32x96, 32x64, 32x32: +-0.3 % on the step), and a 64-row problem still takes 9.3-10 us: the floor is the
  // serial chain of 24 K-steps (barrier, DMA issue, fragment reads, 8 dependent MFMAs: ~800 cycles each), not the tile shape.
  if (K % 32 == 0 && N % 32 == 0 && (long)(N / 96) * mt < 256) {
    if (K % 64 == 0)   // two slabs per barrier: a 512-row DxD launch 27.1 k -> 23.2 k cycles, a 64-row one 21.4 k -> 16.9 k (probe)
      hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 1, 2, 3, SPLIT, 1, 2>), dim3(N / 32, (M + 31) / 32), dim3(256), 0, st, A, W, bias, R,
                         C, M, N, K);
    else
      hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 1, 2, 3, SPLIT, 1>), dim3(N / 32, (M + 31) / 32), dim3(256), 0, st, A, W, bias, R,
                         C, M, N, K);
    return tspo::check_launch("selector gemm_nt");
  }
  if (K % 64 == 0 && N % 96 == 0 && (long)(N / 96) * mt <= 256) {
    // one workgroup per CU (DxD at BT = 2048): two slabs per barrier, two such stages (80 KB): 64.2 k -> 55.6 k cycles per launch
    // (probe; with three stages 59.3 k); the epilogue's share 8.0 k -> 4.4 k by fetching its operands before the K-loop
    hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 3, 4, 2, SPLIT, 1, 2>), dim3(N / 96, mt), dim3(512), 0, st, A, W, bias, R, C, M, N, K);
    return tspo::check_launch("selector gemm_nt");
  }
  // between one and two 64x96 workgroups per CU (Dx3D at BT = 1024: 384) the CUs that hold two set the time: 32x96 tiles (four waves,
  // 48 KB rings) make it 768 = three per CU - 46 -> 40 us, same box; same contraction order per output element
  if (K % 32 == 0 && N % 96 == 0 && (long)(N / 96) * mt > 256 && (long)(N / 96) * mt < 512) {
    hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 3, 2, 3, SPLIT, 1>), dim3(N / 96, (M + 31) / 32), dim3(256), 0, st, A, W, bias, R, C, M, N, K);
    return tspo::check_launch("selector gemm_nt");
  }
  // ring depth 3 keeps two workgroups (16 waves) per CU; when the grid holds more than two workgroups per CU a 2-deep
  // ring (40 KB) lets three run at once instead of leaving the third for a half-empty second round
  if (K % 32 == 0 && N % 96 == 0) {
    if ((long)(N / 96) * mt > 512)
      hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 3, 4, 2, SPLIT>), dim3(N / 96, mt), dim3(512), 0, st, A, W, bias, R, C, M, N, K);
    else   // (a 6-deep ring for the split-precision case measured slower: 24-26 us vs 21-23 us for the DxD GEMMs)
      hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 3, 4, 3, SPLIT>), dim3(N / 96, mt), dim3(512), 0, st, A, W, bias, R, C, M, N, K);
  } else if (K % 32 == 0)
    hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI, 2, 4, 3, SPLIT>), dim3(N / 64, mt), dim3(512), 0, st, A, W, bias, R, C, M, N, K);
  else
    hipLaunchKernelGGL((gemm_f32_nt_kernel<EPI>), dim3(N / 64, mt), dim3(256), 0, st, A, W, bias, R, C, M, N, K);   // (always exact)
  return tspo::check_launch("selector gemm_nt");
}
template <int EPI>
int launch_gemm_nt(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N, int K,
                   hipStream_t st, int split = 0) {   // 0 exact fp32, 1 bf16x3 (hi/lo split), 2 bf16 operands
  if (split == 2 && K % 32 == 0) return launch_gemm_nt_p<EPI, 2>(A, W, bias, R, C, M, N, K, st);
  return split == 1 ? launch_gemm_nt_p<EPI, 1>(A, W, bias, R, C, M, N, K, st) : launch_gemm_nt_p<EPI, 0>(A, W, bias, R, C, M, N, K, st);
}

// ---------------------------------------------------------------------------
// Banded multi-head attention.  One 32-lane half-wave per (row, head); lane j
// (and j+32 for windows > 32) owns the score of the j-th key of the window,
// every lane owns head_dim/32 (<=4) channels of q / the output.
// qkv [B*T, 3D] (q | k | v), ctx [B*T, D], P [B*T*H, w] (softmax weights, saved for backward).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 32));
  return v;
}

__global__ __launch_bounds__(256) void band_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                            float* __restrict__ P, int B, int T, int D, int H, int w) {
  const int hl = threadIdx.x & 31;
  const long pair = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (pair >= (long)B * T * H) return;
  const int h = (int)(pair % H);
  const long row = pair / H;
  const int t = (int)(row % T);
  const long rowb = row - t;  // first row of this video
  const int hd = D / H;
  const float scale = 1.0f / sqrtf((float)hd);
  const int lo = max(0, t - w / 2), hi = min(T - 1, t - w / 2 + w - 1);
  const int n = hi - lo + 1;
  const size_t ld = (size_t)3 * D;
  float qv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = hl + 32 * c;
    qv[c] = e < hd ? qkv[row * ld + h * hd + e] : 0.f;
  }
  float s0 = -INFINITY, s1 = -INFINITY;
  for (int j0 = 0; j0 < n; j0 += 4) {   // 4 keys per trip: their loads are issued together (rows clamped, extras unused)
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, n - 1);
      const float* kr = qkv + (rowb + lo + j) * ld + D + h * hd;
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = hl + 32 * c;
        if (e < hd) a += qv[c] * kr[e];
      }
      d[u] = a;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      const float dd = half_sum(d[u]) * scale;
      if (j < n && j == hl) s0 = dd;
      if (j < n && j == hl + 32) s1 = dd;
    }
  }
  const float m = half_max(fmaxf(s0, s1));
  const float e0 = hl < n ? expf(s0 - m) : 0.f;
  const float e1 = hl + 32 < n ? expf(s1 - m) : 0.f;
  const float l = half_sum(e0 + e1);
  const float p0 = e0 / l, p1 = e1 / l;
  if (hl < w) P[pair * w + hl] = p0;
  if (hl + 32 < w) P[pair * w + hl + 32] = p1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < n; j0 += 4) {
    float vv[4][4], pj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, n - 1);
      const float pr = j < 32 ? __shfl(p0, j, 32) : __shfl(p1, j - 32, 32);
      pj[u] = j0 + u < n ? pr : 0.f;
      const float* vr = qkv + (rowb + lo + j) * ld + 2 * D + h * hd;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = hl + 32 * c;
        vv[u][c] = e < hd ? vr[e] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += pj[u] * vv[u][c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = hl + 32 * c;
    if (e < hd) ctx[row * D + h * hd + e] = acc[c];
  }
}

// backward part 1: per (query row, head): dS (scaled) and dq
__global__ __launch_bounds__(256) void band_attn_bwd_q_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ P,
                                                              const float* __restrict__ dctx,
                                                              float* __restrict__ dqkv, float* __restrict__ dS, int B,
                                                              int T, int D, int H, int w) {
  const int hl = threadIdx.x & 31;
  const long pair = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (pair >= (long)B * T * H) return;
  const int h = (int)(pair % H);
  const long row = pair / H;
  const int t = (int)(row % T);
  const long rowb = row - t;
  const int hd = D / H;
  const float scale = 1.0f / sqrtf((float)hd);
  const int lo = max(0, t - w / 2), hi = min(T - 1, t - w / 2 + w - 1);
  const int n = hi - lo + 1;
  const size_t ld = (size_t)3 * D;
  float dc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = hl + 32 * c;
    dc[c] = e < hd ? dctx[row * D + h * hd + e] : 0.f;
  }
  const float p0 = hl < n ? P[pair * w + hl] : 0.f;
  const float p1 = hl + 32 < n ? P[pair * w + hl + 32] : 0.f;
  float dp0 = 0.f, dp1 = 0.f;
  for (int j = 0; j < n; ++j) {
    const float* vr = qkv + (rowb + lo + j) * ld + 2 * D + h * hd;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = hl + 32 * c;
      if (e < hd) d += dc[c] * vr[e];
    }
    d = half_sum(d);
    if (j == hl) dp0 = d;
    if (j == hl + 32) dp1 = d;
  }
  const float dot = half_sum(p0 * dp0 + p1 * dp1);
  const float ds0 = p0 * (dp0 - dot) * scale, ds1 = p1 * (dp1 - dot) * scale;
  if (hl < w) dS[pair * w + hl] = hl < n ? ds0 : 0.f;
  if (hl + 32 < w) dS[pair * w + hl + 32] = hl + 32 < n ? ds1 : 0.f;
  float dq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < n; ++j) {
    const float dsj = j < 32 ? __shfl(ds0, j, 32) : __shfl(ds1, j - 32, 32);
    const float* kr = qkv + (rowb + lo + j) * ld + D + h * hd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = hl + 32 * c;
      if (e < hd) dq[c] += dsj * kr[e];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = hl + 32 * c;
    if (e < hd) dqkv[row * ld + h * hd + e] = dq[c];
  }
}

// backward part 2: per (key row, head): gather dk, dv over the queries whose window holds this key
__global__ __launch_bounds__(256) void band_attn_bwd_kv_kernel(const float* __restrict__ qkv,
                                                               const float* __restrict__ P,
                                                               const float* __restrict__ dS,
                                                               const float* __restrict__ dctx,
                                                               float* __restrict__ dqkv, int B, int T, int D, int H,
                                                               int w) {
  const int hl = threadIdx.x & 31;
  const long pair = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (pair >= (long)B * T * H) return;
  const int h = (int)(pair % H);
  const long row = pair / H;
  const int j = (int)(row % T);
  const long rowb = row - j;
  const int hd = D / H;
  const size_t ld = (size_t)3 * D;
  // queries t with j in [max(0,t-w/2), min(T-1,t-w/2+w-1)]  <=>  j-(w-1-w/2) <= t <= j+w/2
  const int tlo = max(0, j - (w - 1 - w / 2)), thi = min(T - 1, j + w / 2);
  float dk[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = tlo; t <= thi; ++t) {
    const int lo_t = max(0, t - w / 2);
    const long pr = ((rowb + t) * H + h) * (long)w + (j - lo_t);
    const float ds = dS[pr], p = P[pr];
    const float* qr = qkv + (rowb + t) * ld + h * hd;
    const float* dc = dctx + (rowb + t) * D + h * hd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = hl + 32 * c;
      if (e < hd) {
        dk[c] += ds * qr[e];
        dv[c] += p * dc[e];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = hl + 32 * c;
    if (e < hd) {
      dqkv[row * ld + D + h * hd + e] = dk[c];
      dqkv[row * ld + 2 * D + h * hd + e] = dv[c];
    }
  }
}

// ---------------------------------------------------------------------------
// MFMA form of the banded attention for the production shapes (head width a multiple of 16 up to 128, window <= 17):
// one wave owns a block of 16 consecutive rows of one (video, head).  All three kernels work on TRANSPOSED score tiles
// S^T[key][query] (2 tiles of 16 keys cover the 16 + w - 1 keys any of the 16 queries can see): in the MFMA result
// layout a lane then holds, for its query column l15, the keys 4q+r (+16) — softmax reductions are 8 in-lane values plus
// two cross-lane steps (xor 16, 32), and the same registers are directly the A operand (row = query l15, k <-> key
// 4q+r) of the second product, whose B operand loads the matching rows of V (or K / Q / dctx) as contiguous float4 per
// lane: no LDS, no transposes.  Everything is fp32 (v_mfma_f32_16x16x4_f32).
template <int HD>
struct HeadCols {   // output columns of one head split into lane-vector groups: 64-wide (float4/lane), then 32 (float2), then 16
  static constexpr int N4 = HD / 64;
  static constexpr bool HAS2 = (HD % 64) >= 32;
  static constexpr bool HAS1 = (HD % 32) >= 16;
  static constexpr int BASE2 = 64 * N4;
  static constexpr int BASE1 = BASE2 + (HAS2 ? 32 : 0);
  static constexpr int TILES = HD / 16;
};

// acc[tt][r] += sum_d Arows[key kb+16tt+4q+r... (as MFMA row l15)][d] * Brows[query (MFMA col l15)][d]
// arow[tt] / brow: clamped absolute row pointers (already offset to the head's first column) for this lane's l15.
template <int HD>
__device__ __forceinline__ void band_scores_T(const float* const (&arow)[2], const float* brow, int q, f32x4 (&acc)[2]) {
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // every operand load of the product first (3 x HD / 16 float4 per lane), then the MFMA chain: the launch is one latency chain
  // per wave, and the compiler's own schedule kept only a chunk or two in flight
  f32x4 bf[HD / 16], a0[HD / 16], a1[HD / 16];
#pragma unroll
  for (int ch = 0; ch < HD / 16; ++ch) {
    bf[ch] = *reinterpret_cast<const f32x4*>(brow + ch * 16 + 4 * q);
    a0[ch] = *reinterpret_cast<const f32x4*>(arow[0] + ch * 16 + 4 * q);
    a1[ch] = *reinterpret_cast<const f32x4*>(arow[1] + ch * 16 + 4 * q);
  }
#pragma unroll
  for (int ch = 0; ch < HD / 16; ++ch)   // (the values as operands of an empty statement: every request is out before the first MFMA)
    asm volatile("" : "+v"(bf[ch]), "+v"(a0[ch]), "+v"(a1[ch]));
#pragma unroll
  for (int ch = 0; ch < HD / 16; ++ch) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ch][st], bf[ch][st], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ch][st], bf[ch][st], acc[1], 0, 0, 0);
    }
  }
}

// The 32 rows rb .. rb+31 (clamped to [0, T-1]) of `mat` (head's first column of row 0 of the video) as MFMA B operands: this
// lane's rows rb + 16 tt + 4 q + r, its 4 (2, 1) columns of every lane-vector group.  Loaded ahead of time (band_rows_load, issued
// before the score product of the same kernel - the rows do not depend on it) and consumed by band_rows_mma: round 5, the
// just-in-time loads of the combined form exposed one L2 round trip per row (8 in a row) in each of the three kernels.
template <int HD>
struct BandRows {
  f32x4 v4[8][HeadCols<HD>::N4 > 0 ? HeadCols<HD>::N4 : 1];
  float2 v2[8];
  float v1[8];
};
template <int HD>
__device__ __forceinline__ void band_rows_load(const float* mat, size_t ld, int rb, int T, int q, int l15, BandRows<HD>& R) {
  using HC = HeadCols<HD>;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = rb + 16 * tt + 4 * q + r;
      row = row < 0 ? 0 : (row > T - 1 ? T - 1 : row);
      const float* mr = mat + (size_t)row * ld;
#pragma unroll
      for (int g = 0; g < HC::N4; ++g) R.v4[tt * 4 + r][g] = *reinterpret_cast<const f32x4*>(mr + 64 * g + 4 * l15);
      if (HC::HAS2) R.v2[tt * 4 + r] = *reinterpret_cast<const float2*>(mr + HC::BASE2 + 2 * l15);
      if (HC::HAS1) R.v1[tt * 4 + r] = mr[HC::BASE1 + l15];
    }
}
// out[tile] = sum over the 32 rows of coef[tt][r] (A: MFMA row l15) x row (B); rows outside [0, T-1] carry coefficient 0.
template <int HD>
__device__ __forceinline__ void band_rows_mma(const f32x4 (&coef)[2], const BandRows<HD>& R, f32x4 (&out)[HD / 16]) {
  using HC = HeadCols<HD>;
#pragma unroll
  for (int c = 0; c < HC::TILES; ++c) out[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float cf = coef[tt][r];
#pragma unroll
      for (int g = 0; g < HC::N4; ++g) {
        const f32x4 v = R.v4[tt * 4 + r][g];
#pragma unroll
        for (int c = 0; c < 4; ++c) out[4 * g + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, v[c], out[4 * g + c], 0, 0, 0);
      }
      if (HC::HAS2) {
        const float2 v = R.v2[tt * 4 + r];
        out[4 * HC::N4] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, v.x, out[4 * HC::N4], 0, 0, 0);
        out[4 * HC::N4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, v.y, out[4 * HC::N4 + 1], 0, 0, 0);
      }
      if (HC::HAS1) out[HC::TILES - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, R.v1[tt * 4 + r], out[HC::TILES - 1], 0, 0, 0);
    }
}

// store out (rows i0+4q+r of the block, head columns in the lane-vector grouping) to dst (head's first column, row 0)
template <int HD>
__device__ __forceinline__ void band_store_rows(const f32x4 (&out)[HD / 16], float* dst, size_t ld, int i0, int T, int q,
                                                int l15) {
  using HC = HeadCols<HD>;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = i0 + 4 * q + r;
    if (t < T) {
      float* dr = dst + (size_t)t * ld;
#pragma unroll
      for (int g = 0; g < HC::N4; ++g) {
        const f32x4 v = {out[4 * g][r], out[4 * g + 1][r], out[4 * g + 2][r], out[4 * g + 3][r]};
        *reinterpret_cast<f32x4*>(dr + 64 * g + 4 * l15) = v;
      }
      if (HC::HAS2) *reinterpret_cast<float2*>(dr + HC::BASE2 + 2 * l15) = make_float2(out[4 * HC::N4][r], out[4 * HC::N4 + 1][r]);
      if (HC::HAS1) dr[HC::BASE1 + l15] = out[HC::TILES - 1][r];
    }
  }
}

struct BandBlk {   // wave -> (video, head, 16-row block)
  int b, h, i0;
  bool live;
};
__device__ __forceinline__ BandBlk band_block(int B, int T, int H) {
  const int nblk = (T + 15) / 16;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  BandBlk k;
  k.live = wave < (long)B * nblk * H;
  const long wv = k.live ? wave : 0;
  k.h = (int)(wv % H);
  const long rest = wv / H;
  k.i0 = (int)(rest % nblk) * 16;
  k.b = (int)(rest / nblk);
  return k;
}
__device__ __forceinline__ float xor_sum_hi(float v) {   // over the 4 lanes that share l15
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float xor_max_hi(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

template <int HD>
__global__ __launch_bounds__(256) void band_attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                                 float* __restrict__ P, int B, int T, int D, int H, int w) {
  const BandBlk k = band_block(B, T, H);
  if (!k.live) return;
  const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
  const size_t ld = (size_t)3 * D;
  const float* base = qkv + (size_t)k.b * T * ld + k.h * HD;   // row 0 of the video, q columns of the head
  const int kb = k.i0 - w / 2;                                   // first key of the tile pair (may be < 0)
  const int t = k.i0 + l15;                                      // this lane's query (MFMA column)
  const int tq = t < T ? t : T - 1;
  const float* arow[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    int kr = kb + 16 * tt + l15;
    kr = kr < 0 ? 0 : (kr > T - 1 ? T - 1 : kr);
    arow[tt] = base + (size_t)kr * ld + D;
  }
  BandRows<HD> vrows;
  band_rows_load<HD>(base + 2 * D, ld, kb, T, q, l15, vrows);   // the V rows of the second product, requested before the first
  asm volatile("" ::: "memory");                                 // (keeps the requests here: hipcc sinks them behind the softmax)
  f32x4 sc[2];
  band_scores_T<HD>(arow, base + (size_t)tq * ld, q, sc);
  const float scale = 1.0f / sqrtf((float)HD);
  const int lo = max(0, tq - w / 2), hi = min(T - 1, tq - w / 2 + w - 1);
  float m = -INFINITY;
  bool ok[2][4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * tt + 4 * q + r;
      ok[tt][r] = key >= lo && key <= hi;
      sc[tt][r] = ok[tt][r] ? sc[tt][r] * scale : -INFINITY;
      m = fmaxf(m, sc[tt][r]);
    }
  m = xor_max_hi(m);
  float l = 0.f;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[tt][r] = ok[tt][r] ? expf(sc[tt][r] - m) : 0.f;
      l += sc[tt][r];
    }
  l = xor_sum_hi(l);
  const size_t pair = ((size_t)k.b * T + tq) * H + k.h;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[tt][r] = sc[tt][r] / l;
      const int slot = kb + 16 * tt + 4 * q + r - lo;
      if (t < T && slot >= 0 && slot < w) P[pair * w + slot] = sc[tt][r];   // slots past the clipped window hold 0
    }
  f32x4 out[HD / 16];
  band_rows_mma<HD>(sc, vrows, out);
  band_store_rows<HD>(out, ctx + (size_t)k.b * T * D + k.h * HD, (size_t)D, k.i0, T, q, l15);
}

// backward part 1 (MFMA): dS (scaled) and dq for a block of 16 queries
template <int HD>
__global__ __launch_bounds__(256) void band_attn_bwd_q_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                                   const float* __restrict__ dctx,
                                                                   float* __restrict__ dqkv, float* __restrict__ dS, int B,
                                                                   int T, int D, int H, int w) {
  const BandBlk k = band_block(B, T, H);
  if (!k.live) return;
  const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
  const size_t ld = (size_t)3 * D;
  const float* base = qkv + (size_t)k.b * T * ld + k.h * HD;
  const int kb = k.i0 - w / 2;
  const int t = k.i0 + l15;
  const int tq = t < T ? t : T - 1;
  const float* arow[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    int kr = kb + 16 * tt + l15;
    kr = kr < 0 ? 0 : (kr > T - 1 ? T - 1 : kr);
    arow[tt] = base + (size_t)kr * ld + 2 * D;   // V rows
  }
  BandRows<HD> krows;
  band_rows_load<HD>(base + D, ld, kb, T, q, l15, krows);   // the K rows of dq = dS . K, requested before the first product
  asm volatile("" ::: "memory");
  const float scale = 1.0f / sqrtf((float)HD);
  const int lo = max(0, tq - w / 2), hi = min(T - 1, tq - w / 2 + w - 1);
  const size_t pair = ((size_t)k.b * T + tq) * H + k.h;
  f32x4 pv[2];   // the softmax weights of this lane's (query, key) pairs: requested with the rows, they do not depend on the product
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * tt + 4 * q + r;
      const bool ok = key >= lo && key <= hi;
      pv[tt][r] = ok ? P[pair * w + (key - lo)] : 0.f;
    }
  __builtin_amdgcn_sched_barrier(0);   // (the requests stay in front of the product; nothing waits for them here)
  f32x4 dp[2];
  band_scores_T<HD>(arow, dctx + ((size_t)k.b * T + tq) * D + k.h * HD, q, dp);
  float dot = 0.f;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dot += pv[tt][r] * dp[tt][r];
  dot = xor_sum_hi(dot);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ds = pv[tt][r] * (dp[tt][r] - dot) * scale;   // 0 outside the window (p == 0)
      dp[tt][r] = ds;
      const int slot = kb + 16 * tt + 4 * q + r - lo;
      if (t < T && slot >= 0 && slot < w) dS[pair * w + slot] = ds;
    }
  f32x4 out[HD / 16];
  band_rows_mma<HD>(dp, krows, out);   // dq = dS . K
  band_store_rows<HD>(out, dqkv + (size_t)k.b * T * ld + k.h * HD, ld, k.i0, T, q, l15);
}

// backward part 2 (MFMA): dk, dv for a block of 16 keys, gathered over the <= 16 + w - 1 queries that see them
template <int HD>
__global__ __launch_bounds__(256) void band_attn_bwd_kv_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                                    const float* __restrict__ dS,
                                                                    const float* __restrict__ dctx,
                                                                    float* __restrict__ dqkv, int B, int T, int D, int H,
                                                                    int w) {
  const BandBlk k = band_block(B, T, H);
  if (!k.live) return;
  const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
  const size_t ld = (size_t)3 * D;
  const int j = k.i0 + l15;                       // this lane's key (MFMA row of the coefficient operand)
  const int tb = k.i0 - (w - 1 - w / 2);          // first query of the tile pair: j-(w-1-w/2) <= t <= j+w/2
  BandRows<HD> qrows, grows;                      // the Q rows (dk) and the dctx rows (dv): requested before the coefficients
  band_rows_load<HD>(qkv + (size_t)k.b * T * ld + k.h * HD, ld, tb, T, q, l15, qrows);
  band_rows_load<HD>(dctx + (size_t)k.b * T * D + k.h * HD, (size_t)D, tb, T, q, l15, grows);
  f32x4 cds[2], cp[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = tb + 16 * tt + 4 * q + r;
      float vds = 0.f, vp = 0.f;
      if (t >= 0 && t < T && j < T) {
        const int lo = max(0, t - w / 2), hi = min(T - 1, t - w / 2 + w - 1);
        if (j >= lo && j <= hi) {
          const size_t pr = (((size_t)k.b * T + t) * H + k.h) * w + (j - lo);
          vds = dS[pr];
          vp = P[pr];
        }
      }
      cds[tt][r] = vds;
      cp[tt][r] = vp;
    }
  f32x4 out[HD / 16];
  float* dst = dqkv + (size_t)k.b * T * ld + k.h * HD;
  band_rows_mma<HD>(cds, qrows, out);   // dk = dS^T . Q
  band_store_rows<HD>(out, dst + D, ld, k.i0, T, q, l15);
  band_rows_mma<HD>(cp, grows, out);   // dv = P^T . dctx
  band_store_rows<HD>(out, dst + 2 * D, ld, k.i0, T, q, l15);
}

// which == 0: forward, 1: backward-q, 2: backward-kv.  Returns false when the shape needs the generic kernels.
template <int HD>
void launch_band_mfma(int which, const float* qkv, float* ctx, float* P, const float* dctx, float* dqkv, float* dS, int B,
                      int T, int D, int H, int w, hipStream_t st) {
  const long waves = (long)B * ((T + 15) / 16) * H;
  const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
  if (which == 0)
    hipLaunchKernelGGL((band_attn_fwd_mfma_kernel<HD>), grid, blk, 0, st, qkv, ctx, P, B, T, D, H, w);
  else if (which == 1)
    hipLaunchKernelGGL((band_attn_bwd_q_mfma_kernel<HD>), grid, blk, 0, st, qkv, (const float*)P, dctx, dqkv, dS, B, T, D, H, w);
  else
    hipLaunchKernelGGL((band_attn_bwd_kv_mfma_kernel<HD>), grid, blk, 0, st, qkv, (const float*)P, (const float*)dS, dctx, dqkv,
                       B, T, D, H, w);
}
bool band_mfma(int which, const float* qkv, float* ctx, float* P, const float* dctx, float* dqkv, float* dS, int B, int T,
               int D, int H, int w, hipStream_t st) {
  if (w > 17 || D % H) return false;
  switch (D / H) {
    case 16: launch_band_mfma<16>(which, qkv, ctx, P, dctx, dqkv, dS, B, T, D, H, w, st); return true;
    case 32: launch_band_mfma<32>(which, qkv, ctx, P, dctx, dqkv, dS, B, T, D, H, w, st); return true;
    case 64: launch_band_mfma<64>(which, qkv, ctx, P, dctx, dqkv, dS, B, T, D, H, w, st); return true;
    case 96: launch_band_mfma<96>(which, qkv, ctx, P, dctx, dqkv, dS, B, T, D, H, w, st); return true;
    case 128: launch_band_mfma<128>(which, qkv, ctx, P, dctx, dqkv, dS, B, T, D, H, w, st); return true;
    default: return false;
  }
}

// ---------------------------------------------------------------------------
// s_t = (mean_m h_t.e_m / (|h_t||e_m| + 1e-6) + clip_t) / tau   (temporal_agent.py:106-114,135-141)
// one wave per (b,t) row; 16-byte loads, |h|^2 together with the first text row's dot products in ONE pass over the row (round 5:
// three dependent passes of 4-byte loads before - the launch is a latency chain, 9.0 -> 5 us at B T = 1024)
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }
// this lane's share of h.h, h.e and e.e over a row of nv float4 (lane + 64 i): the loads of up to 4 chunks of BOTH rows are issued
// together, then the arithmetic (D <= 1024: one round trip per row)
__device__ __forceinline__ void row_dots(const f32x4* __restrict__ hr, const f32x4* __restrict__ er, int nv, int lane, bool want_hh,
                                         float& hh, float& de, float& ee) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = lane; c0 < nv; c0 += 256) {
    f32x4 hv[4], ev[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + 64 * i;
      hv[i] = c < nv ? hr[c] : z;
      ev[i] = c < nv ? er[c] : z;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (want_hh) hh += dot4(hv[i], hv[i]);
      de += dot4(hv[i], ev[i]);
      ee += dot4(ev[i], ev[i]);
    }
  }
}
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ h, const float* __restrict__ txt,
                                                        const float* __restrict__ clip, float* __restrict__ scores,
                                                        int B, int T, int D, int M, float tau) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T);
  const f32x4* hr = reinterpret_cast<const f32x4*>(h + row * D);
  const int nv = D >> 2;   // D % 64 == 0
  float hh = 0.f, acc = 0.f, hn = 0.f;
  for (int m = 0; m < M; ++m) {
    const f32x4* er = reinterpret_cast<const f32x4*>(txt + ((size_t)b * M + m) * D);
    float de = 0.f, ee = 0.f;
    row_dots(hr, er, nv, lane, m == 0, hh, de, ee);
    if (m == 0) hn = sqrtf(wave_sum(hh));
    de = wave_sum(de);
    const float en = sqrtf(wave_sum(ee));
    acc += de / (hn * en + 1e-6f);
  }
  if (lane == 0) {
    float s = acc / (float)M;
    s = s + (clip ? clip[row] : 0.f);
    scores[row] = s / tau;
  }
}

// dh_t = (ds_t / tau / M) * sum_m [ e_m/den_m - (h.e_m) |e_m| / (den_m^2 |h|) h ]
// The launch also carries the two weight transposes the data-gradient GEMMs need (W1^T, W2^T; blocks past n_score, 32x32
// tiles): they are independent of the score gradient and far too small for a launch of their own.
// PG (tspo_policy_backward): dscores is not an input - the wave of row (b, t) derives it from the rollouts
// (tspo_trainer.py:587-609: group-relative advantage of the G rewards of prompt b, then the closed-form policy-gradient
// term of tspo_pg_grad_logits) with the same arithmetic, in the same order, as grpo_pg_grad_kernel.
struct PgIn {
  const float* rewards; const float* logp; const int64_t* idx;   // [B,G], [B,T], [B,G,k] ascending
  int G, k; float eps, scale;
  float* adv; float* loss;                                       // out: [B,G], [B] (nullable)
};
template <bool PG>
__global__ __launch_bounds__(256) void score_bwd_kernel(const float* __restrict__ h, const float* __restrict__ txt,
                                                        const float* __restrict__ dscores, float* __restrict__ dh,
                                                        int B, int T, int D, int M, float tau, int n_score,
                                                        const float* __restrict__ w1, float* __restrict__ w1t,
                                                        const float* __restrict__ w2, float* __restrict__ w2t, PgIn pg) {
  if ((int)blockIdx.x >= n_score) {
    __shared__ float tile[32][33];
    const int tpr = (D + 31) / 32;
    int tb = blockIdx.x - n_score;
    const bool second = tb >= tpr * tpr;
    tb -= second ? tpr * tpr : 0;
    const float* in = second ? w2 : w1;
    float* out = second ? w2t : w1t;
    const int bx = (tb % tpr) * 32, by = (tb / tpr) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8)
      if (by + r < D && bx + tx < D) tile[r][tx] = in[(size_t)(by + r) * D + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
      if (bx + r < D && by + tx < D) out[(size_t)(bx + r) * D + by + tx] = tile[tx][r];
    return;
  }
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T);
  // 16-byte loads / stores; |h|^2 and the first text row's products in one pass; dh written once (M = 1: no read-modify-write)
  const f32x4* hr = reinterpret_cast<const f32x4*>(h + row * D);
  f32x4* dr = reinterpret_cast<f32x4*>(dh + row * D);
  const int nv = D >> 2;
  // PG: what the policy-gradient term needs from memory is requested FIRST, with the row's own loads (round 5: the membership test
  // was a binary search in global memory behind everything else - five dependent round trips, the longest chain of the launch).
  // G * k <= 256 (the training shapes: 8 x 16): the wave reads the prompt's G sorted lists flat, lane l entries l, l + 64, ...
  float pg_r = 0.f, pg_lp = 0.f;
  int64_t pg_e[4] = {-1, -1, -1, -1};
  const bool coop = PG && pg.G * pg.k <= 256;
  if (PG) {
    pg_r = lane < pg.G ? pg.rewards[(size_t)b * pg.G + lane] : 0.f;
    pg_lp = pg.logp[row];
    if (coop) {
      const int64_t* lst = pg.idx + (size_t)b * pg.G * pg.k;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c * 64 + lane < pg.G * pg.k) pg_e[c] = lst[c * 64 + lane];
    }
  }
  float hh = 0.f, de0 = 0.f, ee0 = 0.f;
  row_dots(hr, reinterpret_cast<const f32x4*>(txt + (size_t)b * M * D), nv, lane, true, hh, de0, ee0);
  const float hn = sqrtf(wave_sum(hh));
  float ds;
  if (PG) {   // G <= 64: lane g owns rollout g of this row's prompt
    const int t = (int)(row - (long)b * T), G = pg.G, k = pg.k;
    const bool live = lane < G;
    const float r = pg_r;
    const float mean = wave_sum(r) / (float)G;
    const float dm = live ? r - mean : 0.f;
    const float sd = sqrtf(wave_sum(dm * dm) / (float)(G - 1));
    const float a = live ? (r - mean) / (sd + pg.eps) : 0.f;
    bool hit = false;
    if (coop) {   // is t in rollout lane's list?  ballot of "entry == t" per 64-entry chunk, lane g looks at its own bit range
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned long long m = __ballot(pg_e[c] == (int64_t)t);
        const int lo = max(lane * k - c * 64, 0), hi = min((lane + 1) * k - c * 64, 64);
        if (live && lo < hi) {
          const unsigned long long width = hi - lo >= 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull);
          hit = hit || (((m >> lo) & width) != 0ull);
        }
      }
    } else if (live) {
      const int64_t* my = pg.idx + ((size_t)b * G + lane) * k;
      int lo = 0, hi = k - 1;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int v = (int)my[mid];
        if (v == t) { hit = true; break; }
        if (v < t) lo = mid + 1; else hi = mid - 1;
      }
    }
    const float p = expf(pg_lp);
    const float invk = 1.f / (float)k, invG = 1.f / (float)G;
    const float y = (hit ? invk : 0.f) - p;
    float acc = 0.f, sumA = 0.f;
    for (int gg = 0; gg < G; ++gg) {   // serial over the rollouts, like the stand-alone kernel (same bits, same fma)
      const float ag = __shfl(a, gg, 64);
      acc += ag * __shfl(y, gg, 64);
      sumA += ag;
    }
    ds = -invG * acc * pg.scale;
    if (t == 0) {
      if (live) pg.adv[(size_t)b * G + lane] = a;
      if (lane == 0 && pg.loss) pg.loss[b] = -sumA * invG;
    }
  } else {
    ds = dscores[row];
  }
  const float g = ds / tau / (float)M;
  for (int m = 0; m < M; ++m) {
    const f32x4* er = reinterpret_cast<const f32x4*>(txt + ((size_t)b * M + m) * D);
    float de = de0, ee = ee0;
    if (m > 0) {
      float unused = 0.f;
      de = 0.f; ee = 0.f;
      row_dots(hr, er, nv, lane, false, unused, de, ee);
    }
    de = wave_sum(de);
    const float en = sqrtf(wave_sum(ee));
    const float den = hn * en + 1e-6f;
    const float ca = g / den;
    const float cb = hn > 0.f ? g * de * en / (den * den * hn) : 0.f;
    for (int c0 = lane; c0 < nv; c0 += 256) {   // (h and e are L1 / L2 hits by now; all loads of a round first)
      f32x4 hv[4], ev[4], dv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + 64 * i;
        if (c < nv) { hv[i] = hr[c]; ev[i] = er[c]; if (m > 0) dv[i] = dr[c]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + 64 * i;
        if (c < nv) {
          const f32x4 v = ca * ev[i] - cb * hv[i];
          dr[c] = m == 0 ? v : dv[i] + v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight-gradient GEMM: Cp[s][i][j] = sum_{m in chunk s} dY[m][i] * X[m][j]
// ("TN": the contraction runs over rows).  One wave = 64x64 outputs made of 16
// MFMA tiles with an interleaved feature map (tile c holds features 4*l+c) so
// that every operand load and every store is a contiguous float4.
__global__ __launch_bounds__(256) void gemm_f32_tn_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                          float* __restrict__ Cp, int Mrows, int NI, int NJ,
                                                          int chunk) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.y * 128 + (wid >> 1) * 64;
  const int j0 = blockIdx.x * 128 + (wid & 1) * 64;
  if (i0 >= NI || j0 >= NJ) return;
  const int s = blockIdx.z;
  const int mb = s * chunk, me = min(Mrows, mb + chunk);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* yp = dY + i0 + 4 * l15;
  const float* xp = X + j0 + 4 * l15;
  // rows past the chunk end are clamped and their contribution zeroed by a multiplier (no branch around the loads,
  // so the compiler keeps counted vmcnt waits and the prefetch distance of 2 survives)
  const int rlast = me > mb ? me - 1 : mb;
  auto ld = [&](f32x4& a, f32x4& b, int m) {
    const int r = m + q;
    const int rc = r < rlast ? r : rlast;
    const float keep = (r < me) ? 1.f : 0.f;
    a = *reinterpret_cast<const f32x4*>(yp + (size_t)rc * NI) * keep;
    b = *reinterpret_cast<const f32x4*>(xp + (size_t)rc * NJ);
  };
  auto mm = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ca], b[cb], acc[ca][cb], 0, 0, 0);
  };
  f32x4 a0, b0, a1, b1, a2, b2;
  if (me > mb) {
    ld(a0, b0, mb);
    ld(a1, b1, mb + 4);
    const int nsteps = (me - mb + 3) >> 2;
    int st = 0;
    for (; st + 3 <= nsteps; st += 3) {
      ld(a2, b2, mb + ((st + 2) << 2));
      mm(a0, b0);
      ld(a0, b0, mb + ((st + 3) << 2));
      mm(a1, b1);
      ld(a1, b1, mb + ((st + 4) << 2));
      mm(a2, b2);
    }
    if (st < nsteps) mm(a0, b0);
    if (st + 1 < nsteps) mm(a1, b1);
  }
  float* cp = Cp + (size_t)s * NI * NJ;
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * (q * 4 + r) + ca;  // D row (q*4+r) of tile ca -> feature 4*row + ca
      f32x4 v = {acc[ca][0][r], acc[ca][1][r], acc[ca][2][r], acc[ca][3][r]};
      *reinterpret_cast<f32x4*>(cp + (size_t)i * NJ + j0 + 4 * l15) = v;
    }
}

// LDS-staged version of the weight-gradient GEMM (NI, NJ multiples of 128): one workgroup = 128x128 outputs (4 waves of
// 64x64 as above); operand slabs of 16 contraction rows (2 x 8 KB) travel by LDS-DMA through a 4-deep ring with counted
// vmcnt waits, so ~3 slabs of loads are in flight per workgroup — the direct-load kernel above keeps a single wave per
// SIMD waiting on HBM/L2 latency.  Fragment reads are conflict-free ds_read_b128 (16 lanes = 256 contiguous bytes).
#define TN_ROWS 16
#define TN_NST 4
// NW = 4: 2x2 waves of 64x64 (the stand-alone kernel); NW = 8: 2x4 waves of 64x32 (inside dgrad_wgrad_kernel, whose
// workgroups have 512 threads).  Same contraction order per output element either way.  `cpart` (optional): the
// workgroups of feature-tile column 0 also emit the column sums of their dY rows (= this split's bias-gradient
// partial, cpart[s * cstride + i]) from the fragments they read anyway.
template <int NW, int NST>
__device__ __forceinline__ void gemm_f32_tn_lds_body(char* lds, const float* __restrict__ dY, const float* __restrict__ X,
                                                     float* __restrict__ Cp, float* __restrict__ cpart, int cstride,
                                                     int Mrows, int NI, int NJ, int chunk, int bj, int bi, int s) {
  constexpr int WJ = NW / 2;            // waves along j
  constexpr int CB = 4 / (WJ / 2);      // interleaved 16-column tiles per wave along j: 4 (64 columns) or 2 (32 columns)
  constexpr int PW = 8 / NW;            // DMA instructions per wave, operand and slab (2 rows each, 16 rows per slab)
  typedef float fb_t __attribute__((ext_vector_type(CB)));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wi = wid / WJ, wj = wid % WJ;
  const int i0 = bi * 128, j0 = bj * 128;
  const int mb = s * chunk, me = min(Mrows, mb + chunk);
  f32x4 acc[4][CB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const bool want_cs = cpart != nullptr && bj == 0 && wj == 0;   // wave-uniform
  const int nst = me > mb ? (me - mb + TN_ROWS - 1) / TN_ROWS : 0;
  const int rlast = me > mb ? me - 1 : mb;
  const int prow = lane >> 5, pcol = (lane & 31) << 2;
  // NW = 8 reads its X fragments with ds_read_b64 (two 16-lane groups = rows m, m+1 per cycle, 512 B apart = the same banks):
  // the DMA stores the 128-byte segments of odd rows pairwise swapped, the reads undo it (dgrad_wgrad_kernel: 10.8 % of LDS cycles
  // were bank conflicts before, profiles/r4_h_policy_sq_counters.txt)
  const int pcolB = NW == 8 ? ((((lane & 31) >> 3) ^ prow) << 5) + ((lane & 7) << 2) : pcol;
  auto stage = [&](int t) {   // wave wid moves rows 2*PW*wid .. 2*PW*wid + 2*PW-1 of both slabs
    char* buf = lds + (t % NST) * 16384;
#pragma unroll
    for (int p = 0; p < PW; ++p) {
      const int rr = (wid * PW + p) * 2 + prow;
      int r = mb + t * TN_ROWS + rr;
      r = r < rlast ? r : rlast;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dY + (size_t)r * NI + i0 + pcol),
                                       (__attribute__((address_space(3))) void*)(buf + (wid * PW + p) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)r * NJ + j0 + pcolB),
                                       (__attribute__((address_space(3))) void*)(buf + 8192 + (wid * PW + p) * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nst) stage(t);
  // (rows m = 4 ks + q: m & 1 = q & 1; slab rows rr = 2 * piece + prow: rr & 1 = prow)
  const int offA = (wi * 64 + 4 * l15) * 4, offB = 8192 + ((NW == 8 ? (wj ^ (q & 1)) : wj) * 16 * CB + CB * l15) * 4;
  for (int t = 0; t < nst; ++t) {
    // slabs t+1, t+2 (2*PW DMA instructions each per wave) may stay in flight
    const int ahead = min(NST - 2, nst - 1 - t);
    if (PW == 2) {
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // slab t visible to all waves; everyone is done with slab t-1 -> its slot is free
    if (t + NST - 1 < nst) stage(t + NST - 1);
    const char* cur = lds + (t % NST) * 16384;
#pragma unroll
    for (int ks = 0; ks < TN_ROWS / 4; ++ks) {
      const int m = ks * 4 + q;
      const float keep = (mb + t * TN_ROWS + m < me) ? 1.f : 0.f;
      const f32x4 a = *reinterpret_cast<const f32x4*>(cur + offA + m * 512) * keep;
      const fb_t b = *reinterpret_cast<const fb_t*>(cur + offB + m * 512);
      if (want_cs) csum += a;
#pragma unroll
      for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ca], b[cb], acc[ca][cb], 0, 0, 0);
    }
  }
  float* cp = Cp + (size_t)s * NI * NJ;
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wi * 64 + 4 * (q * 4 + r) + ca;
      fb_t v;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) v[cb] = acc[ca][cb][r];
      *reinterpret_cast<fb_t*>(cp + (size_t)i * NJ + j0 + wj * 16 * CB + CB * l15) = v;
    }
  if (want_cs) {   // rows m = 4*ks + q were summed per lane: add the four q groups (fixed order -> deterministic)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = csum[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      csum[c] = v;
    }
    if (q == 0) *reinterpret_cast<f32x4*>(cpart + (size_t)s * cstride + i0 + wi * 64 + 4 * l15) = csum;
  }
}

// 64x64-tile form of the weight-gradient GEMM for contractions short enough to stay in ONE workgroup (BT = 512, the reference's
// micro-batch: 32 slabs): four waves (2 x 2) of 32x32 = 2 x 2 MFMA tiles, slabs of 16 contraction rows (2 x 4 KB) through an
// NST-deep LDS-DMA ring.  With S == 1 the workgroup holds the FINAL gradient tile: it is written (or added, `accumulate`: the
// second micro-step of an accumulation window) straight into the gradient bucket, the tile-column-0 workgroups emit the final
// bias gradient from the fragments they read anyway, and every workgroup leaves the sum of squares of what it wrote in
// sq_partial[blockIdx] - no partial planes, no reduction launch, no separate norm pass.  With S > 1 it behaves like the 128x128
// form (plane s of Cp / cpart, reduced later).
// LDS: row m of a slab = 64 floats (256 B = all 64 banks), so the two 16-lane groups a ds_read_b64 serves per cycle (rows m, m+1)
// would hit the same banks: the DMA stores the 128-byte halves of odd rows swapped, the reads undo it.
template <int NST, int KG>
__device__ __forceinline__ void gemm_f32_tn64_body(char* lds, const float* __restrict__ dY, const float* __restrict__ X,
                                                   float* __restrict__ Cp, float* __restrict__ cpart, int cstride,
                                                   float* __restrict__ sq_partial, int accumulate, int Mrows, int NI, int NJ,
                                                   int chunk, int bj, int bi, int s, int wg_index) {
  // KG = 2: 8 waves; waves 4..7 (K-group 1) take rows 16..31 of every 32-row step and their accumulators are added to K-group 0's
  // through LDS at the end (fixed order): twice the waves per CU for the same tile count, half the dependent chain
  constexpr int STEP = TN_ROWS * KG, SLAB = 8192 * KG;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int kg = wid >> 2, w4 = wid & 3;
  const int l15 = lane & 15, q = lane >> 4;
  const int wi = w4 >> 1, wj = w4 & 1;
  const int i0 = bi * 64, j0 = bj * 64;
  const int mb = s * chunk, me = min(Mrows, mb + chunk);
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x2 csum = {0.f, 0.f};
  const bool want_cs = cpart != nullptr && bj == 0 && wj == 0;   // wave-uniform
  const int nst = me > mb ? (me - mb + STEP - 1) / STEP : 0;
  const int rlast = me > mb ? me - 1 : mb;
  // DMA: wave w moves rows 4w..4w+3 of the step, both operands (one 1 KB piece each); lane (r4, chunk) fetches the 16 bytes whose LDS
  // place is row r4, chunk `chunk`, i.e. source half (chunk >> 3) ^ (r4 & 1)
  const int r4 = lane >> 4, ch = lane & 15;
  const int scol = ((((ch >> 3) ^ (r4 & 1)) << 5) + ((ch & 7) << 2));
  auto stage = [&](int t) {
    char* buf = lds + (t % NST) * SLAB + kg * 8192;
    int r = mb + t * STEP + wid * 4 + r4;
    r = r < rlast ? r : rlast;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dY + (size_t)r * NI + i0 + scol),
                                     (__attribute__((address_space(3))) void*)(buf + w4 * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)r * NJ + j0 + scol),
                                     (__attribute__((address_space(3))) void*)(buf + 4096 + w4 * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nst) stage(t);
  // rows m = 4 ks + q: m & 1 = q & 1
  const int offA = kg * 8192 + ((wi ^ (q & 1)) << 7) + l15 * 8, offB = kg * 8192 + 4096 + ((wj ^ (q & 1)) << 7) + l15 * 8;
  for (int t = 0; t < nst; ++t) {
    const int ahead = min(NST - 2, nst - 1 - t);   // newer steps of this wave (2 DMA instructions each) that may stay in flight
    if (ahead >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // step t visible to all waves; everyone is done with step t-1 -> its slot is free
    if (t + NST - 1 < nst) stage(t + NST - 1);
    const char* cur = lds + (t % NST) * SLAB;
#pragma unroll
    for (int ks = 0; ks < TN_ROWS / 4; ++ks) {
      const int m = ks * 4 + q;
      const float keep = (mb + t * STEP + kg * TN_ROWS + m < me) ? 1.f : 0.f;
      const f32x2 a = *reinterpret_cast<const f32x2*>(cur + offA + m * 256) * keep;
      const f32x2 b = *reinterpret_cast<const f32x2*>(cur + offB + m * 256);
      if (want_cs) csum += a;
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ca], b[cb], acc[ca][cb], 0, 0, 0);
    }
  }
  if (KG == 2) {   // K-group 1 -> LDS -> added by K-group 0 (18 floats per lane: 16 accumulators + 2 column sums)
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();   // the ring is no longer read
    if (kg == 1) {
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((w4 * 18) + (ca * 2 + cb) * 4 + r) * 64 + lane] = acc[ca][cb][r];
      red[(w4 * 18 + 16) * 64 + lane] = csum[0];
      red[(w4 * 18 + 17) * 64 + lane] = csum[1];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ca][cb][r] += red[((w4 * 18) + (ca * 2 + cb) * 4 + r) * 64 + lane];
      csum[0] += red[(w4 * 18 + 16) * 64 + lane];
      csum[1] += red[(w4 * 18 + 17) * 64 + lane];
    }
  }
  float sq = 0.f;
  if (kg == 0) {
    float* cp = Cp + (size_t)s * NI * NJ;
#pragma unroll
    for (int ca = 0; ca < 2; ++ca)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi * 32 + 2 * (q * 4 + r) + ca;
        float* dst = cp + (size_t)i * NJ + j0 + wj * 32 + 2 * l15;
        f32x2 v = {acc[ca][0][r], acc[ca][1][r]};
        if (accumulate) v += *reinterpret_cast<const f32x2*>(dst);
        *reinterpret_cast<f32x2*>(dst) = v;
        sq += v[0] * v[0] + v[1] * v[1];
      }
    if (want_cs) {   // rows m = 4*ks + q were summed per lane: add the four q groups (fixed order -> deterministic)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float v = csum[c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        csum[c] = v;
      }
      if (q == 0) {
        float* dst = cpart + (size_t)s * cstride + i0 + wi * 32 + 2 * l15;
        f32x2 v = csum;
        if (accumulate) v += *reinterpret_cast<const f32x2*>(dst);
        *reinterpret_cast<f32x2*>(dst) = v;
        sq += v[0] * v[0] + v[1] * v[1];
      }
    }
  }
  if (sq_partial) {   // (uniform) one partial per workgroup, fixed order inside: deterministic
    float* red = reinterpret_cast<float*>(lds) + 4 * 18 * 64;   // (behind the K-group exchange area)
    __syncthreads();
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) sq_partial[wg_index] = sq;
  }
}

__global__ __launch_bounds__(512) void gemm_f32_tn64_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                            float* __restrict__ Cp, float* __restrict__ cpart, int cstride,
                                                            float* __restrict__ sq_partial, int accumulate, int Mrows, int NI,
                                                            int NJ, int chunk) {
  __shared__ __attribute__((aligned(16))) char lds[3 * 16384];
  const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  gemm_f32_tn64_body<3, 2>(lds, dY, X, Cp, cpart, cstride, sq_partial, accumulate, Mrows, NI, NJ, chunk, blockIdx.x, blockIdx.y,
                           blockIdx.z, wg);
}

// (3-deep ring here: 48 KB and <= 168 registers let three workgroups share a CU, so the 108 x S tiles of the q/k/v weight
// gradient - 756 for BT = 2048 - are resident at once instead of leaving a half-empty second round)
__global__ __launch_bounds__(256, 3) void gemm_f32_tn_lds_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                 float* __restrict__ Cp, float* __restrict__ cpart,
                                                                 int cstride, int Mrows, int NI, int NJ, int chunk) {
  __shared__ __attribute__((aligned(16))) char lds[3 * 16384];   // per stage: A 16x512 B | B 16x512 B
  gemm_f32_tn_lds_body<4, 3>(lds, dY, X, Cp, cpart, cstride, Mrows, NI, NJ, chunk, blockIdx.x, blockIdx.y, blockIdx.z);
}

// One launch = the data-gradient GEMM of a linear layer (dX = dY W: the 64x96-tile kernel above, 512-thread workgroups)
// AND the weight-gradient tiles of the layer after it (the two are independent).  Each alone fills the 256 CUs with one
// workgroup per CU and leaves the matrix pipes idle across its barriers; the hardware dispatches the data-gradient
// workgroups first and the weight-gradient ones into the second slot of every CU.
template <int EPI>
__global__ __launch_bounds__(512, 2) void dgrad_wgrad_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                            const float* __restrict__ R, float* __restrict__ C, int M,
                                                            int N, int K, int n_nt,
                                                            const float* __restrict__ dY, const float* __restrict__ X,
                                                            float* __restrict__ Cp, float* __restrict__ cpart, int cstride,
                                                            int NI, int NJ, int chunk) {
  constexpr int LDS = nt_lds_bytes<3, 3>() > TN_NST * 16384 ? nt_lds_bytes<3, 3>() : TN_NST * 16384;
  __shared__ __attribute__((aligned(16))) char lds[LDS];
  const int b = blockIdx.x;
  if (b < n_nt) {
    const int nx = N / 96;
    gemm_f32_nt_lds_body<EPI, 3, 4, 3, 0>(lds, A, W, nullptr, R, C, M, N, K, b % nx, b / nx);
  } else {
    const int t = b - n_nt, tj = NJ >> 7, ti = NI >> 7;
    gemm_f32_tn_lds_body<8, TN_NST>(lds, dY, X, Cp, cpart, cstride, M, NI, NJ, chunk, t % tj, (t / tj) % ti, t / (tj * ti));
  }
}

// Small-M form of the launch above (BT = 512: the 64x96 data-gradient tiling has 64 workgroups): 256-thread workgroups, the data
// gradient in 32x32 tiles (gemm_f32_nt_lds_body<.., WN 1, NWM 2, ring 3, MI 1>), the weight-gradient tiles as four waves of 64x64.
template <int EPI>
__global__ __launch_bounds__(512) void dgrad_wgrad_small_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                               const float* __restrict__ R, float* __restrict__ C, int M,
                                                               int N, int K, int n_nt, int n_tiles,
                                                               const float* __restrict__ dY, const float* __restrict__ X,
                                                               float* __restrict__ Cp, float* __restrict__ cpart, int cstride,
                                                               int NI, int NJ, int chunk, int t64, float* __restrict__ sq_partial,
                                                               int accumulate) {
  constexpr int LDS = TN_NST * 16384 > 3 * 16384 ? TN_NST * 16384 : 3 * 16384;
  __shared__ __attribute__((aligned(16))) char lds[LDS];
  static_assert(2 * nt_lds_bytes<1, 3, 32>() <= LDS, "LDS");
  const int b = blockIdx.x;
  if (b < n_nt) {   // two 32x32 data-gradient tiles per workgroup (the last one may repeat a tile: same values written twice)
    const int half = threadIdx.x >> 8, nx = N / 32;
    const int tile = min(2 * b + half, n_tiles - 1);
    gemm_f32_nt_lds_body<EPI, 1, 2, 3, 0, 1, 2>(lds + half * nt_lds_bytes<1, 3, 32>(), A, W, nullptr, R, C, M, N, K, tile % nx,
                                                tile / nx);
  } else if (t64) {   // 64x64 tiles, whole contraction per workgroup (or `chunk` rows of it): see gemm_f32_tn64_body
    const int t = b - n_nt, tj = NJ >> 6, ti = NI >> 6;
    gemm_f32_tn64_body<3, 2>(lds, dY, X, Cp, cpart, cstride, sq_partial, accumulate, M, NI, NJ, chunk, t % tj, (t / tj) % ti,
                             t / (tj * ti), t);
  } else {
    const int t = b - n_nt, tj = NJ >> 7, ti = NI >> 7;
    gemm_f32_tn_lds_body<8, TN_NST>(lds, dY, X, Cp, cpart, cstride, M, NI, NJ, chunk, t % tj, (t / tj) % ti, t / (tj * ti));
  }
}

// Split-precision (bf16x3) version of the weight-gradient GEMM: same 128x128 tile / 4 waves of 64x64 / interleaved
// feature map, but a slab is 32 contraction rows = one bf16 MFMA k-step: lane (l15, q) owns rows 8q..8q+7, reads them as
// 8 float4 per operand (4 interleaved feature tiles each), splits every value into hi + lo bf16 and issues
// lo*hi + hi*lo + hi*hi for each of the 16 tile pairs (48 MFMAs per slab instead of 128 fp32 ones at half the rate).
#define TS_ROWS 32
#define TS_NST 3   // (4-deep measured no better)
__global__ __launch_bounds__(256) void gemm_f32_tn_split_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                float* __restrict__ Cp, int Mrows, int NI, int NJ,
                                                                int chunk) {
  __shared__ __attribute__((aligned(16))) char lds[TS_NST * 32768];   // per stage: A 32x512 B | B 32x512 B
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wi = wid >> 1, wj = wid & 1;
  const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
  const int s = blockIdx.z;
  const int mb = s * chunk, me = min(Mrows, mb + chunk);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nst = me > mb ? (me - mb + TS_ROWS - 1) / TS_ROWS : 0;
  const int rlast = me > mb ? me - 1 : mb;
  const int prow = lane >> 5, pcol = (lane & 31) << 2;
  auto stage = [&](int t, int ring) {   // wave wid moves rows 8*wid..8*wid+7 of both slabs (2 rows per DMA instruction)
    char* buf = lds + ring * 32768;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rr = wid * 8 + p * 2 + prow;
      int r = mb + t * TS_ROWS + rr;
      r = r < rlast ? r : rlast;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dY + (size_t)r * NI + i0 + pcol),
                                       (__attribute__((address_space(3))) void*)(buf + (wid * 8 + p * 2) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)r * NJ + j0 + pcol),
                                       (__attribute__((address_space(3))) void*)(buf + 16384 + (wid * 8 + p * 2) * 512), 16, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < TS_NST - 1; ++t)
    if (t < nst) stage(t, t);
  const int offA = (wi * 64 + 4 * l15) * 4, offB = 16384 + (wj * 64 + 4 * l15) * 4;
  int ring = 0;
  for (int t = 0; t < nst; ++t) {
    // slabs t+1, t+2 (8 DMA instructions each per wave) may stay in flight
    const int ahead = min(TS_NST - 2, nst - 1 - t);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // slab t visible to all waves; everyone is done with slab t-1 -> its slot is free
    if (t + TS_NST - 1 < nst) stage(t + TS_NST - 1, ring >= 1 ? ring - 1 : TS_NST - 1);
    const char* cur = lds + ring * 32768;
    ring = ring + 1 == TS_NST ? 0 : ring + 1;
    f32x4 a4[8], b4[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = 8 * q + r;
      const float keep = (mb + t * TS_ROWS + m < me) ? 1.f : 0.f;
      a4[r] = *reinterpret_cast<const f32x4*>(cur + offA + m * 512) * keep;
      b4[r] = *reinterpret_cast<const f32x4*>(cur + offB + m * 512);
    }
    bf16x8 bh[4], bl[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const float x[8] = {b4[0][cb], b4[1][cb], b4[2][cb], b4[3][cb], b4[4][cb], b4[5][cb], b4[6][cb], b4[7][cb]};
      split_bf16x8(x, bh[cb], bl[cb]);
    }
#pragma unroll
    for (int ca = 0; ca < 4; ++ca) {
      const float x[8] = {a4[0][ca], a4[1][ca], a4[2][ca], a4[3][ca], a4[4][ca], a4[5][ca], a4[6][ca], a4[7][ca]};
      bf16x8 ah, al;
      split_bf16x8(x, ah, al);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[cb], acc[ca][cb], 0, 0, 0);
        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[cb], acc[ca][cb], 0, 0, 0);
        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[cb], acc[ca][cb], 0, 0, 0);
      }
    }
  }
  float* cp = Cp + (size_t)s * NI * NJ;
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wi * 64 + 4 * (q * 4 + r) + ca;
      f32x4 v = {acc[ca][0][r], acc[ca][1][r], acc[ca][2][r], acc[ca][3][r]};
      *reinterpret_cast<f32x4*>(cp + (size_t)i * NJ + j0 + wj * 64 + 4 * l15) = v;
    }
}

// One launch for all split reductions of a backward pass: segment g holds S planes of n floats (plane stride n) and
// sums them in plane order into out (fixed order -> deterministic).
#define RED_SEGS 6
struct RedSegs {
  const float* part[RED_SEGS];
  float* out[RED_SEGS];
  unsigned long long end[RED_SEGS];   // cumulative element counts
  unsigned long long n[RED_SEGS];
  int S[RED_SEGS];
  int count;
};
// norm_partial != nullptr: block b also writes the sum of squares of the values it produced to norm_partial[b] (the
// optimiser's gradient norm without a second pass over the bucket; fixed grid -> fixed order -> deterministic).
// accumulate: the sums are ADDED to what `out` holds (second.. micro-step of a gradient-accumulation window).
__global__ __launch_bounds__(256) void reduce_segments_kernel(RedSegs L, float* __restrict__ norm_partial, int accumulate) {   // every n is a multiple of 4 (D % 64 == 0)
  __shared__ float red[32];
  float sq = 0.f;
  const unsigned long long total4 = L.end[L.count - 1] >> 2;
  for (unsigned long long i4 = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i4 < total4;
       i4 += (unsigned long long)gridDim.x * 256) {
    const unsigned long long i = i4 << 2;
    int g = 0;
    while (i >= L.end[g]) ++g;
    const unsigned long long e = i - (g ? L.end[g - 1] : 0ull);
    const float* p = L.part[g] + e;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const int S = L.S[g];
    const size_t stride = L.n[g];
    for (int s = 0; s < S; ++s) a += *reinterpret_cast<const f32x4*>(p + (size_t)s * stride);
    if (accumulate) a += *reinterpret_cast<const f32x4*>(L.out[g] + e);   // (the planes first: out + this gradient, one rounding)
    *reinterpret_cast<f32x4*>(L.out[g] + e) = a;
    sq += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
  }
  if (norm_partial) {
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) norm_partial[blockIdx.x] = sq;
  }
}

// bias-gradient column sums of the three dY matrices (dh2 | dh1 | dqkv) in one launch: blockIdx.x walks 64-column blocks
// of the virtual concatenation, blockIdx.y the row slab; part[y][concat col].
__global__ __launch_bounds__(256) void colsum3_kernel(const float* __restrict__ y0, const float* __restrict__ y1,
                                                      const float* __restrict__ y2, float* __restrict__ part, int Mrows,
                                                      int D, int rows_per) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ccol = blockIdx.x * 64 + tx;       // D % 64 == 0, so a block never straddles two matrices
  const float* src;
  int N, col;
  if (ccol < D) { src = y0; N = D; col = ccol; }
  else if (ccol < 2 * D) { src = y1; N = D; col = ccol - D; }
  else { src = y2; N = 3 * D; col = ccol - 2 * D; }
  const int mb = blockIdx.y * rows_per, me = min(Mrows, mb + rows_per);
  float a = 0.f;
  for (int m = mb + ty; m < me; m += 4) a += src[(size_t)m * N + col];
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0) part[(size_t)blockIdx.y * 5 * D + ccol] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// cos(text_b, feat_bt) like torch.nn.CosineSimilarity(dim=-1, eps=1e-8)
__global__ __launch_bounds__(256) void clip_scores_kernel(const float* __restrict__ txt, const float* __restrict__ f,
                                                          float* __restrict__ out, int B, int T, int D, int M) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T);
  const float* fr = f + row * D;
  const float* er = txt + (size_t)b * M * D;
  float de = 0.f, ee = 0.f, ff = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float e = er[d], x = fr[d];
    de += e * x; ee += e * e; ff += x * x;
  }
  de = wave_sum(de); ee = wave_sum(ee); ff = wave_sum(ff);
  if (lane == 0) out[row] = de / sqrtf(fmaxf(ee * ff, 1e-16f));
}

// ---------------------------------------------------------------------------
struct SelWs {  // workspace layout shared by forward and backward
  float *xpe, *qkv, *P, *ctx, *h1, *h2;                      // saved by forward
  float *dh2, *dh1, *dctx, *dqkv, *dS, *w1t, *w2t, *part, *cpart;  // backward scratch
  int S, CS;
  size_t bytes;
};

// Number of contraction splits of the weight-gradient GEMMs.  Each split of a 128x128 tile is one workgroup that runs
// at one-SIMD-per-wave speed, so the makespan is set by how evenly tiles*S workgroups fill 256 CUs: pick the S (chunk
// of >= 128 rows, at most 16 partial planes) that wastes the least of the last "round" for both the DxD and 3DxD GEMMs.
int split_for(int BT, int D) {
  // BT <= 1024 (the reference's two micro-steps as one batch): measured sweep S = 3 ... 8 on the coalesced DP path: 267 (S = 3), 278,
  // 278, 272, 269 (S = 7, what the rule below picks), 270 us per optimizer step - fewer planes for the reduction launch to sum
  if (BT > 768 && BT <= 1024 && D == 768) return 3;
  int smax = BT / 128;
  smax = smax < 1 ? 1 : (smax > 16 ? 16 : smax);
  const int tiles = ((D + 127) / 128) * ((D + 127) / 128);
  int best = 1;
  double best_eff = -1.0;
  for (int S = 1; S <= smax; ++S) {
    double eff = 1.0;
    for (int mul = 1; mul <= 3; mul += 2) {
      const int wgs = tiles * mul * S;
      const double e = (double)wgs / (double)(((wgs + 255) / 256) * 256);
      eff = e < eff ? e : eff;
    }
    if (eff > best_eff + 1e-9) { best_eff = eff; best = S; }
  }
  return best;
}

SelWs carve(void* ws, int B, int T, int D, int H, int M, int w) {
  (void)M;
  tspo::Carver c(ws);
  SelWs s;
  const size_t BT = (size_t)B * T;
  s.xpe = c.take<float>(BT * D);
  s.qkv = c.take<float>(BT * 3 * D);
  s.P = c.take<float>(BT * H * w);
  s.ctx = c.take<float>(BT * D);
  s.h1 = c.take<float>(BT * D);
  s.h2 = c.take<float>(BT * D);
  s.dh2 = c.take<float>(BT * D);
  s.dh1 = c.take<float>(BT * D);
  s.dctx = c.take<float>(BT * D);
  s.dqkv = c.take<float>(BT * 3 * D);
  s.dS = c.take<float>(BT * H * w);
  s.w1t = c.take<float>((size_t)D * D);
  s.w2t = c.take<float>((size_t)D * D);
  s.S = split_for((int)BT, D);
  s.CS = 16;
  s.part = c.take<float>((size_t)s.S * 5 * D * D);   // planes of dW2 | dW1 | dWqkv partials (kept until the single reduce)
  s.cpart = c.take<float>((size_t)s.CS * 5 * D);
  s.bytes = c.bytes();
  return s;
}

int check_dims(const char* fn, int B, int T, int D, int H, int M, int w) {
  TSPO_REQUIRE(B >= 1 && T >= 1 && M >= 1, "%s: bad dims B=%d T=%d M=%d", fn, B, T, M);
  TSPO_REQUIRE(D >= 64 && D % 64 == 0, "%s: D=%d must be a positive multiple of 64", fn, D);
  TSPO_REQUIRE(H >= 1 && D % H == 0 && D / H <= 128, "%s: heads=%d must divide D=%d with head_dim <= 128", fn, H, D);
  TSPO_REQUIRE(w >= 1 && w <= 64, "%s: window_size=%d must be in [1,64]", fn, w);
  return TSPO_OK;
}

}  // namespace

extern "C" size_t tspo_selector_workspace_bytes(int B, int T, int D, int H, int M, int window) {
  if (B < 1 || T < 1 || D < 1 || H < 1 || window < 1) return 0;
  return carve(nullptr, B, T, D, H, M, window).bytes;
}

static int selector_forward_impl(const tspo_selector_weights* w, const float* img, const float* txt,
                                 const float* clip, int B, int T, int D, int H, int M, int window, float tau,
                                 float* scores, float* temporal_attn, void* workspace, size_t workspace_bytes,
                                 tspo_stream_t stream, int flags) {
  TSPO_REQUIRE((flags & ~(TSPO_SEL_BF16X3 | TSPO_SEL_BF16)) == 0, "selector_forward: unknown flags 0x%x", flags);
  TSPO_REQUIRE((flags & (TSPO_SEL_BF16X3 | TSPO_SEL_BF16)) != (TSPO_SEL_BF16X3 | TSPO_SEL_BF16), "selector_forward: TSPO_SEL_BF16X3 and TSPO_SEL_BF16 are exclusive");
  TSPO_REQUIRE(!(flags & TSPO_SEL_BF16) || D % 32 == 0, "selector_forward: TSPO_SEL_BF16 needs D %% 32 == 0 (D=%d)", D);
  const int split = (flags & TSPO_SEL_BF16) ? 2 : ((flags & TSPO_SEL_BF16X3) ? 1 : 0);
  TSPO_REQUIRE(w && img && txt && scores && workspace, "selector_forward: null pointer");
  TSPO_REQUIRE(w->wqkv && w->bqkv && w->w1 && w->b1 && w->w2 && w->b2, "selector_forward: null weight pointer");
  if (int e = check_dims("selector_forward", B, T, D, H, M, window)) return e;
  TSPO_REQUIRE(tau != 0.f, "selector_forward: score_tau must be non-zero");
  SelWs s = carve(workspace, B, T, D, H, M, window);
  if (workspace_bytes < s.bytes)
    return tspo::set_err(TSPO_EWORKSPACE, "selector_forward: workspace %zu < %zu", workspace_bytes, s.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int BT = B * T;
  const size_t tot = (size_t)BT * D;
  int nb = (int)((tot / 2 + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(posenc_add_kernel, dim3(nb), dim3(256), 0, st, img, s.xpe, T, D, tot);
  if (int e = launch_gemm_nt<EPI_NONE>(s.xpe, w->wqkv, w->bqkv, nullptr, s.qkv, BT, 3 * D, D, st, split)) return e;
  const long pairs = (long)BT * H;
  if (!band_mfma(0, s.qkv, s.ctx, s.P, nullptr, nullptr, nullptr, B, T, D, H, window, st))
    hipLaunchKernelGGL(band_attn_fwd_kernel, dim3((unsigned)((pairs + 7) / 8)), dim3(256), 0, st, s.qkv, s.ctx, s.P, B, T,
                       D, H, window);
  if (int e = launch_gemm_nt<EPI_RELU>(s.ctx, w->w1, w->b1, nullptr, s.h1, BT, D, D, st, split)) return e;
  if (int e = launch_gemm_nt<EPI_RESID>(s.h1, w->w2, w->b2, img, s.h2, BT, D, D, st, split)) return e;
  hipLaunchKernelGGL(score_fwd_kernel, dim3((BT + 3) / 4), dim3(256), 0, st, s.h2, txt, clip, scores, B, T, D, M, tau);
  if (temporal_attn) {
    hipError_t e = hipMemcpyAsync(temporal_attn, s.h2, tot * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return tspo::set_err(TSPO_ELAUNCH, "selector_forward: copy: %s", hipGetErrorString(e));
  }
  return tspo::check_launch("selector_forward");
}

extern "C" int tspo_selector_forward(const tspo_selector_weights* w, const float* img, const float* txt,
                                     const float* clip, int B, int T, int D, int H, int M, int window, float tau,
                                     float* scores, float* temporal_attn, void* workspace, size_t workspace_bytes,
                                     tspo_stream_t stream) {
  return selector_forward_impl(w, img, txt, clip, B, T, D, H, M, window, tau, scores, temporal_attn, workspace,
                               workspace_bytes, stream, 0);
}
extern "C" int tspo_selector_forward_ex(const tspo_selector_weights* w, const float* img, const float* txt,
                                        const float* clip, int B, int T, int D, int H, int M, int window, float tau,
                                        float* scores, float* temporal_attn, void* workspace, size_t workspace_bytes,
                                        tspo_stream_t stream, int flags) {
  return selector_forward_impl(w, img, txt, clip, B, T, D, H, M, window, tau, scores, temporal_attn, workspace,
                               workspace_bytes, stream, flags);
}

namespace {
// cpart != nullptr: the LDS kernel also writes this split's bias-gradient partials; returns through *fused whether it did
// `fin` (64x64 tiles, the whole contraction in one workgroup: the kernel writes the FINAL gradient): where it goes and whether it is added
struct FinalGrad {
  float* w = nullptr;      // != nullptr: final mode; weight gradient [NI][NJ]
  float* b = nullptr;      // bias gradient [NI]
  float* sq = nullptr;     // one sum of squares per workgroup (or nullptr)
  int accumulate = 0;
};
int weight_grad(const float* dY, const float* X, float* part, float* cpart, int cstride, int BT, int NI, int NJ,
                const SelWs& s, hipStream_t st, bool split, const FinalGrad& fin = FinalGrad()) {
  if (fin.w) {
    hipLaunchKernelGGL(gemm_f32_tn64_kernel, dim3(NJ / 64, NI / 64, 1), dim3(512), 0, st, dY, X, fin.w, fin.b, 0, fin.sq,
                       fin.accumulate, BT, NI, NJ, (BT + 3) / 4 * 4);
    return tspo::check_launch("selector weight_grad (64x64, final)");
  }
  const int chunk = ((BT + s.S - 1) / s.S + 3) / 4 * 4;
  dim3 grid((NJ + 127) / 128, (NI + 127) / 128, s.S);
  if (split && NI % 128 == 0 && NJ % 128 == 0)
    hipLaunchKernelGGL(gemm_f32_tn_split_kernel, grid, dim3(256), 0, st, dY, X, part, BT, NI, NJ, chunk);
  else if (NI % 128 == 0 && NJ % 128 == 0)
    hipLaunchKernelGGL(gemm_f32_tn_lds_kernel, grid, dim3(256), 0, st, dY, X, part, cpart, cstride, BT, NI, NJ, chunk);
  else
    hipLaunchKernelGGL(gemm_f32_tn_kernel, grid, dim3(256), 0, st, dY, X, part, BT, NI, NJ, chunk);
  return tspo::check_launch("selector weight_grad");
}

// dX = mask?(dY W) together with the weight gradient dYw^T Xw of another layer in one launch (see dgrad_wgrad_kernel)
template <int EPI>
int dgrad_with_wgrad(const float* dY, const float* Wt, const float* R, float* dX, int BT, int N, int K, const float* dYw,
                     const float* Xw, float* part, float* cpart, int cstride, int NI, int NJ, const SelWs& s,
                     hipStream_t st, const FinalGrad& fin = FinalGrad()) {
  const int chunk = ((BT + s.S - 1) / s.S + 3) / 4 * 4;
  const int n_nt = (N / 96) * ((BT + 63) / 64), n_tn = (NI / 128) * (NJ / 128) * s.S;
  // small-M form (BT = 512: 28 -> 20.5 us per launch) while the 64x96 data-gradient tiling has fewer workgroups than HALF the CUs.
  // At BT = 1024 (round 5: the reference's two micro-steps coalesced into one B = 2 batch; 128 such workgroups) the large form wins:
  // 37 us against 54 us per launch, same box - the whole-contraction 64x64 weight-gradient tiles of the small form grow linearly
  // with BT, the split tiles of the large form do not.
  if (n_nt < 128 && N % 32 == 0) {
    const int n_tiles = (N / 32) * ((BT + 31) / 32), n_nt_s = (n_tiles + 1) / 2;
    if (fin.w)
      hipLaunchKernelGGL((dgrad_wgrad_small_kernel<EPI>), dim3(n_nt_s + (NI / 64) * (NJ / 64)), dim3(512), 0, st, dY, Wt, R, dX, BT, N,
                         K, n_nt_s, n_tiles, dYw, Xw, fin.w, fin.b, 0, NI, NJ, (BT + 3) / 4 * 4, 1, fin.sq, fin.accumulate);
    else
      hipLaunchKernelGGL((dgrad_wgrad_small_kernel<EPI>), dim3(n_nt_s + n_tn), dim3(512), 0, st, dY, Wt, R, dX, BT, N, K, n_nt_s,
                         n_tiles, dYw, Xw, part, cpart, cstride, NI, NJ, chunk, 0, (float*)nullptr, 0);
    return tspo::check_launch("selector dgrad+wgrad");
  }
  hipLaunchKernelGGL((dgrad_wgrad_kernel<EPI>), dim3(n_nt + n_tn), dim3(512), 0, st, dY, Wt, R, dX, BT, N, K, n_nt, dYw, Xw,
                     part, cpart, cstride, NI, NJ, chunk);
  return tspo::check_launch("selector dgrad+wgrad");
}
}  // namespace

static int selector_backward_impl(const tspo_selector_weights* w, const float* img, const float* txt,
                                  const float* dscores, int B, int T, int D, int H, int M, int window, float tau,
                                  const tspo_selector_grads* g, void* workspace, size_t workspace_bytes,
                                  tspo_stream_t stream, int flags, const PgIn* pg = nullptr, float* norm_partials = nullptr,
                                  int* n_partials = nullptr) {
  const bool split = (flags & TSPO_SEL_BF16X3) != 0;
  const int accum = (flags & TSPO_SEL_ACCUMULATE) ? 1 : 0;
  TSPO_REQUIRE((flags & ~(TSPO_SEL_BF16X3 | TSPO_SEL_ACCUMULATE)) == 0, "selector_backward: unknown flags 0x%x", flags);
  TSPO_REQUIRE(w && img && txt && (dscores || pg) && g && workspace, "selector_backward: null pointer");
  TSPO_REQUIRE(g->wqkv && g->bqkv && g->w1 && g->b1 && g->w2 && g->b2, "selector_backward: null grad pointer");
  if (int e = check_dims("selector_backward", B, T, D, H, M, window)) return e;
  SelWs s = carve(workspace, B, T, D, H, M, window);
  if (workspace_bytes < s.bytes)
    return tspo::set_err(TSPO_EWORKSPACE, "selector_backward: workspace %zu < %zu", workspace_bytes, s.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int BT = B * T;
  const size_t DD = (size_t)D * D;
  float* part_w2 = s.part;
  float* part_w1 = s.part + (size_t)s.S * DD;
  float* part_qkv = s.part + (size_t)s.S * 2 * DD;
  // score -> dh2
  const int n_score = (BT + 3) / 4, tpr = (D + 31) / 32;
  if (pg)
    hipLaunchKernelGGL(score_bwd_kernel<true>, dim3(n_score + 2 * tpr * tpr), dim3(256), 0, st, s.h2, txt, nullptr, s.dh2, B, T,
                       D, M, tau, n_score, w->w1, s.w1t, w->w2, s.w2t, *pg);
  else
    hipLaunchKernelGGL(score_bwd_kernel<false>, dim3(n_score + 2 * tpr * tpr), dim3(256), 0, st, s.h2, txt, dscores, s.dh2, B,
                       T, D, M, tau, n_score, w->w1, s.w1t, w->w2, s.w2t, PgIn{});
  // exact-fp32 path with 128-divisible D: bias-gradient column sums come out of the weight-gradient kernels, and each of
  // the two DxD weight gradients shares a launch with the data-gradient GEMM that does not depend on it
  const bool fused = !split && D % 128 == 0 && D % 96 == 0;
  const int cstride = 5 * D;
  // short contractions (the small-M forms: BT = 512, the reference's micro-batch): 64x64 weight-gradient tiles that hold the whole
  // contraction write the FINAL gradients (and their sums of squares) - no partial planes, no reduction launch
  const int t64_tiles = (D / 64) * (D / 64);
  const bool t64 = fused && (D / 96) * ((BT + 63) / 64) < 128 && (!norm_partials || 5 * t64_tiles <= 2048);   // (same rule as dgrad_with_wgrad)
  FinalGrad f2, f1, fq;
  if (t64) {
    f2.w = g->w2; f2.b = g->b2; f2.sq = norm_partials; f2.accumulate = accum;
    f1.w = g->w1; f1.b = g->b1; f1.sq = norm_partials ? norm_partials + t64_tiles : nullptr; f1.accumulate = accum;
    fq.w = g->wqkv; fq.b = g->bqkv; fq.sq = norm_partials ? norm_partials + 2 * t64_tiles : nullptr; fq.accumulate = accum;
  }
  if (fused) {
    // mlp.2 data gradient (dh1) || mlp.2 weight gradient (dh2^T h1)
    if (int e = dgrad_with_wgrad<EPI_MASK>(s.dh2, s.w2t, s.h1, s.dh1, BT, D, D, s.dh2, s.h1, part_w2, s.cpart, cstride, D, D, s,
                                           st, f2)) return e;
    // mlp.0 data gradient (dctx) || mlp.0 weight gradient (dh1^T ctx)
    if (int e = dgrad_with_wgrad<EPI_NONE>(s.dh1, s.w1t, nullptr, s.dctx, BT, D, D, s.dh1, s.ctx, part_w1, s.cpart + D, cstride,
                                           D, D, s, st, f1)) return e;
  } else {
    // mlp.2
    if (int e = weight_grad(s.dh2, s.h1, part_w2, nullptr, 0, BT, D, D, s, st, split)) return e;
    if (int e = launch_gemm_nt<EPI_MASK>(s.dh2, s.w2t, nullptr, s.h1, s.dh1, BT, D, D, st, split)) return e;
    // mlp.0
    if (int e = weight_grad(s.dh1, s.ctx, part_w1, nullptr, 0, BT, D, D, s, st, split)) return e;
    if (int e = launch_gemm_nt<EPI_NONE>(s.dh1, s.w1t, nullptr, nullptr, s.dctx, BT, D, D, st, split)) return e;
  }
  // banded attention
  const long pairs = (long)BT * H;
  const unsigned pb = (unsigned)((pairs + 7) / 8);
  if (!band_mfma(1, s.qkv, nullptr, s.P, s.dctx, s.dqkv, s.dS, B, T, D, H, window, st))
    hipLaunchKernelGGL(band_attn_bwd_q_kernel, dim3(pb), dim3(256), 0, st, s.qkv, s.P, s.dctx, s.dqkv, s.dS, B, T, D, H,
                       window);
  if (!band_mfma(2, s.qkv, nullptr, s.P, s.dctx, s.dqkv, s.dS, B, T, D, H, window, st))
    hipLaunchKernelGGL(band_attn_bwd_kv_kernel, dim3(pb), dim3(256), 0, st, s.qkv, s.P, s.dS, s.dctx, s.dqkv, B, T, D, H,
                       window);
  // q/k/v projections
  if (int e = weight_grad(s.dqkv, s.xpe, part_qkv, fused ? s.cpart + 2 * D : nullptr, cstride, BT, 3 * D, D, s, st, split, fq))
    return e;
  if (t64) {   // every gradient is final
    if (norm_partials && n_partials) *n_partials = 5 * t64_tiles;
    (void)img;
    return tspo::check_launch("selector_backward");
  }
  // bias grads: column sums of dh2 | dh1 | dqkv (fused above, or CS row slabs here), then every split reduction
  // (3 weights + 3 biases) in one launch
  const int bias_planes = fused ? s.S : s.CS;
  if (!fused) {
    const int rows_per = (BT + s.CS - 1) / s.CS;
    hipLaunchKernelGGL(colsum3_kernel, dim3(5 * D / 64, s.CS), dim3(256), 0, st, s.dh2, s.dh1, s.dqkv, s.cpart, BT, D,
                       rows_per);
  }
  RedSegs L;
  const float* parts[RED_SEGS] = {part_w2, part_w1, part_qkv, s.cpart, s.cpart + D, s.cpart + 2 * D};
  float* outs[RED_SEGS] = {g->w2, g->w1, g->wqkv, g->b2, g->b1, g->bqkv};
  const unsigned long long ns[RED_SEGS] = {DD, DD, 3 * DD, (unsigned long long)D, (unsigned long long)D, 3ull * D};
  unsigned long long run = 0;
  for (int i = 0; i < RED_SEGS; ++i) {
    L.part[i] = parts[i];
    L.out[i] = outs[i];
    run += ns[i];
    L.end[i] = run;
    // weight partial planes are n apart; bias partial planes are 5*D apart (one row of the concatenated column sums)
    L.n[i] = i < 3 ? ns[i] : 5ull * D;
    L.S[i] = i < 3 ? s.S : bias_planes;
  }
  L.count = RED_SEGS;
  int nb = (int)((run / 4 + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (norm_partials) {   // one partial sum of squares per block, at most the 2048 tspo_adamw_clip_step_ex accepts
    if (nb > 2048) nb = 2048;
    if (n_partials) *n_partials = nb;
  }
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(nb), dim3(256), 0, st, L, norm_partials, accum);
  (void)img;
  return tspo::check_launch("selector_backward");
}

extern "C" int tspo_policy_backward(const tspo_selector_weights* w, const float* img, const float* txt, const float* rewards,
                                    const float* logp, const int64_t* idx, int B, int T, int D, int H, int M, int window,
                                    float tau, int G, int k, float adv_eps, float scale, const tspo_selector_grads* grads,
                                    float* adv, float* loss, void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                                    int flags) {
  TSPO_REQUIRE(rewards && logp && idx && adv, "policy_backward: null pointer");
  TSPO_REQUIRE(G >= 1 && G <= 64 && k >= 1 && k <= T, "policy_backward: bad dims G=%d (1..64) k=%d T=%d", G, k, T);
  const PgIn pg{rewards, logp, idx, G, k, adv_eps, scale, adv, loss};
  return selector_backward_impl(w, img, txt, nullptr, B, T, D, H, M, window, tau, grads, workspace, workspace_bytes, stream,
                                flags, &pg);
}

extern "C" int tspo_policy_backward_ex(const tspo_selector_weights* w, const float* img, const float* txt, const float* rewards,
                                       const float* logp, const int64_t* idx, int B, int T, int D, int H, int M, int window,
                                       float tau, int G, int k, float adv_eps, float scale, const tspo_selector_grads* grads,
                                       float* adv, float* loss, void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                                       int flags, float* norm_partials, int* n_partials) {
  TSPO_REQUIRE(rewards && logp && idx && adv, "policy_backward: null pointer");
  TSPO_REQUIRE(G >= 1 && G <= 64 && k >= 1 && k <= T, "policy_backward: bad dims G=%d (1..64) k=%d T=%d", G, k, T);
  TSPO_REQUIRE(!norm_partials || n_partials, "policy_backward_ex: norm_partials without n_partials");
  const PgIn pg{rewards, logp, idx, G, k, adv_eps, scale, adv, loss};
  return selector_backward_impl(w, img, txt, nullptr, B, T, D, H, M, window, tau, grads, workspace, workspace_bytes, stream,
                                flags, &pg, norm_partials, n_partials);
}

extern "C" int tspo_selector_backward(const tspo_selector_weights* w, const float* img, const float* txt,
                                      const float* dscores, int B, int T, int D, int H, int M, int window, float tau,
                                      const tspo_selector_grads* g, void* workspace, size_t workspace_bytes,
                                      tspo_stream_t stream) {
  return selector_backward_impl(w, img, txt, dscores, B, T, D, H, M, window, tau, g, workspace, workspace_bytes, stream, 0);
}
extern "C" int tspo_selector_backward_ex(const tspo_selector_weights* w, const float* img, const float* txt,
                                         const float* dscores, int B, int T, int D, int H, int M, int window, float tau,
                                         const tspo_selector_grads* g, void* workspace, size_t workspace_bytes,
                                         tspo_stream_t stream, int flags) {
  return selector_backward_impl(w, img, txt, dscores, B, T, D, H, M, window, tau, g, workspace, workspace_bytes, stream,
                                flags);
}

extern "C" int tspo_clip_scores(const float* txt, const float* feat, int B, int T, int D, int M, float* clip,
                                tspo_stream_t stream) {
  TSPO_REQUIRE(txt && feat && clip, "clip_scores: null pointer");
  TSPO_REQUIRE(B >= 1 && T >= 1 && D >= 1 && M >= 1, "clip_scores: bad dims");
  hipLaunchKernelGGL(clip_scores_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, txt, feat, clip, B, T,
                     D, M);
  return tspo::check_launch("clip_scores");
}
