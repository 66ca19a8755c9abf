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

namespace {

// ---------------------------------------------------------------------------
// x_pe = x + pe, pe[t,2i] = sin((t/T) * exp(2i * -ln(1e4)/C)), pe[t,2i+1] = cos(..)
// (model/temporal_agent.py:10-19, 128-129); all fp32 like torch.
__global__ __launch_bounds__(256) void posenc_add_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int T, int D, size_t total) {
  const float c = -9.21034049987793f / (float)D;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % T);
    const float div = expf((float)(d & ~1) * c);
    const float pos = (float)t / (float)T;
    const float a = pos * div;
    out[i] = x[i] + ((d & 1) ? cosf(a) : sinf(a));
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
// (16 rows x 64 B per wave instruction) are address-processing bound; here each 64x32-float operand tile is brought in
// as full 128-byte rows by LDS-DMA (XOR-swizzled like the bf16 GEMM tiles) and fragments come from ds_read_b128.
// 64x64 outputs per workgroup, 4 waves (2x2) of 32x32, BK = 32, two stages (32 KB of LDS -> 5 workgroups per CU).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_nt_lds_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const float* __restrict__ R,
                                                              float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 16384];   // 2 stages x (A 64x128 B | W 64x128 B); the only LDS object
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int nk = K >> 5;
  const int rin = lane >> 3, slot = lane & 7;
  auto stage = [&](int kt, char* buf) {   // 16 pieces of 1 KB (8 rows x 128 B); wave wid: A pieces 2*wid, 2*wid+1 and the same of W
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int piece = wid * 2 + p;
      const int r = piece * 8 + rin;
      int gr = m0 + r;
      gr = gr < M ? gr : M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)gr * K + kt * 32 + ((slot ^ rin) << 2)),
                                       (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (size_t)(n0 + r) * K + kt * 32 + ((slot ^ rin) << 2)),
                                       (__attribute__((address_space(3))) void*)(buf + 8192 + piece * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offA[2], offW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    offA[i] = (wm * 32 + i * 16 + l15) * 128;
    offW[i] = 8192 + (wn * 32 + i * 16 + l15) * 128;
  }
  const int sw = l15 & 7;
  stage(0, lds);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();   // stage kt landed (vmcnt(0) folded in by the compiler), compute(kt-1) done everywhere
    const char* cur = lds + (kt & 1) * 16384;
    if (kt + 1 < nk) stage(kt + 1, lds + ((kt + 1) & 1) * 16384);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q) ^ sw) << 4);
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(cur + offA[i] + co);
        b[i] = *reinterpret_cast<const f32x4*>(cur + offW[i] + co);
      }
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][st], b[j][st], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 32 + 16 * j + l15;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + 16 * i + q * 4 + r;
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

template <int EPI>
int launch_gemm_nt(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N, int K,
                   hipStream_t st) {
  dim3 grid(N / 64, (M + 63) / 64);
  if (K % 32 == 0)
    hipLaunchKernelGGL((gemm_f32_nt_lds_kernel<EPI>), grid, dim3(256), 0, st, A, W, bias, R, C, M, N, K);
  else
    hipLaunchKernelGGL((gemm_f32_nt_kernel<EPI>), grid, dim3(256), 0, st, A, W, bias, R, C, M, N, K);
  return tspo::check_launch("selector gemm_nt");
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
  for (int j = 0; j < n; ++j) {
    const float* kr = qkv + (rowb + lo + j) * ld + D + h * hd;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = hl + 32 * c;
      if (e < hd) d += qv[c] * kr[e];
    }
    d = half_sum(d) * scale;
    if (j == hl) s0 = d;
    if (j == hl + 32) s1 = d;
  }
  const float m = half_max(fmaxf(s0, s1));
  const float e0 = hl < n ? expf(s0 - m) : 0.f;
  const float e1 = hl + 32 < n ? expf(s1 - m) : 0.f;
  const float l = half_sum(e0 + e1);
  const float p0 = e0 / l, p1 = e1 / l;
  if (hl < w) P[pair * w + hl] = p0;
  if (hl + 32 < w) P[pair * w + hl + 32] = p1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < n; ++j) {
    const float pj = j < 32 ? __shfl(p0, j, 32) : __shfl(p1, j - 32, 32);
    const float* vr = qkv + (rowb + lo + j) * ld + 2 * D + h * hd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = hl + 32 * c;
      if (e < hd) acc[c] += pj * vr[e];
    }
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
// s_t = (mean_m h_t.e_m / (|h_t||e_m| + 1e-6) + clip_t) / tau   (temporal_agent.py:106-114,135-141)
// one wave per (b,t) row
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ h, const float* __restrict__ txt,
                                                        const float* __restrict__ clip, float* __restrict__ scores,
                                                        int B, int T, int D, int M, float tau) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T);
  const float* hr = h + row * D;
  float hh = 0.f;
  for (int d = lane; d < D; d += 64) hh += hr[d] * hr[d];
  const float hn = sqrtf(wave_sum(hh));
  float acc = 0.f;
  for (int m = 0; m < M; ++m) {
    const float* er = txt + ((size_t)b * M + m) * D;
    float de = 0.f, ee = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float e = er[d];
      de += hr[d] * e;
      ee += e * e;
    }
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
__global__ __launch_bounds__(256) void score_bwd_kernel(const float* __restrict__ h, const float* __restrict__ txt,
                                                        const float* __restrict__ dscores, float* __restrict__ dh,
                                                        int B, int T, int D, int M, float tau) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T);
  const float* hr = h + row * D;
  float hh = 0.f;
  for (int d = lane; d < D; d += 64) hh += hr[d] * hr[d];
  const float hn = sqrtf(wave_sum(hh));
  const float g = dscores[row] / tau / (float)M;
  for (int d = lane; d < D; d += 64) dh[row * D + d] = 0.f;
  for (int m = 0; m < M; ++m) {
    const float* er = txt + ((size_t)b * M + m) * D;
    float de = 0.f, ee = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float e = er[d];
      de += hr[d] * e;
      ee += e * e;
    }
    de = wave_sum(de);
    const float en = sqrtf(wave_sum(ee));
    const float den = hn * en + 1e-6f;
    const float ca = g / den;
    const float cb = hn > 0.f ? g * de * en / (den * den * hn) : 0.f;
    for (int d = lane; d < D; d += 64) dh[row * D + d] += ca * er[d] - cb * hr[d];
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

// out[i] = sum_s part[s][i]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           size_t n, int S) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(size_t)s * n + i];
    out[i] = a;
  }
}

// column sums (bias grads): part[y][i] = sum over the rows of slab y of dY[m][i]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dY, float* __restrict__ part, int Mrows,
                                                     int N, int rows_per) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int mb = blockIdx.y * rows_per, me = min(Mrows, mb + rows_per);
  float a = 0.f;
  if (col < N)
    for (int m = mb + ty; m < me; m += 4) a += dY[(size_t)m * N + col];
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && col < N) part[(size_t)blockIdx.y * N + col] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int Cc) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8)
    if (by + r < R && bx + tx < Cc) tile[r][tx] = in[(size_t)(by + r) * Cc + bx + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bx + r < Cc && by + tx < R) out[(size_t)(bx + r) * R + by + tx] = tile[tx][r];
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

int split_for(int BT) {
  int S = (BT + 255) / 256;
  return S < 1 ? 1 : (S > 16 ? 16 : S);
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
  s.S = split_for((int)BT);
  s.CS = 16;
  s.part = c.take<float>((size_t)s.S * 3 * D * D);
  s.cpart = c.take<float>((size_t)s.CS * 3 * D);
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

extern "C" int tspo_selector_forward(const tspo_selector_weights* w, const float* img, const float* txt,
                                     const float* clip, int B, int T, int D, int H, int M, int window, float tau,
                                     float* scores, float* temporal_attn, void* workspace, size_t workspace_bytes,
                                     tspo_stream_t stream) {
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
  int nb = (int)((tot + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(posenc_add_kernel, dim3(nb), dim3(256), 0, st, img, s.xpe, T, D, tot);
  if (int e = launch_gemm_nt<EPI_NONE>(s.xpe, w->wqkv, w->bqkv, nullptr, s.qkv, BT, 3 * D, D, st)) return e;
  const long pairs = (long)BT * H;
  hipLaunchKernelGGL(band_attn_fwd_kernel, dim3((unsigned)((pairs + 7) / 8)), dim3(256), 0, st, s.qkv, s.ctx, s.P, B, T,
                     D, H, window);
  if (int e = launch_gemm_nt<EPI_RELU>(s.ctx, w->w1, w->b1, nullptr, s.h1, BT, D, D, st)) return e;
  if (int e = launch_gemm_nt<EPI_RESID>(s.h1, w->w2, w->b2, img, s.h2, BT, D, D, st)) return e;
  hipLaunchKernelGGL(score_fwd_kernel, dim3((BT + 3) / 4), dim3(256), 0, st, s.h2, txt, clip, scores, B, T, D, M, tau);
  if (temporal_attn) {
    hipError_t e = hipMemcpyAsync(temporal_attn, s.h2, tot * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return tspo::set_err(TSPO_ELAUNCH, "selector_forward: copy: %s", hipGetErrorString(e));
  }
  return tspo::check_launch("selector_forward");
}

namespace {
int weight_grad(const float* dY, const float* X, float* dW, float* db, int BT, int NI, int NJ, const SelWs& s,
                hipStream_t st) {
  const int chunk = ((BT + s.S - 1) / s.S + 3) / 4 * 4;
  dim3 grid((NJ + 127) / 128, (NI + 127) / 128, s.S);
  hipLaunchKernelGGL(gemm_f32_tn_kernel, grid, dim3(256), 0, st, dY, X, s.part, BT, NI, NJ, chunk);
  const size_t n = (size_t)NI * NJ;
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(nb), dim3(256), 0, st, s.part, dW, n, s.S);
  const int rows_per = (BT + s.CS - 1) / s.CS;
  hipLaunchKernelGGL(colsum_kernel, dim3((NI + 63) / 64, s.CS), dim3(256), 0, st, dY, s.cpart, BT, NI, rows_per);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((NI + 255) / 256), dim3(256), 0, st, s.cpart, db, (size_t)NI, s.CS);
  return tspo::check_launch("selector weight_grad");
}
}  // namespace

extern "C" int tspo_selector_backward(const tspo_selector_weights* w, const float* img, const float* txt,
                                      const float* dscores, int B, int T, int D, int H, int M, int window, float tau,
                                      const tspo_selector_grads* g, void* workspace, size_t workspace_bytes,
                                      tspo_stream_t stream) {
  TSPO_REQUIRE(w && img && txt && dscores && g && workspace, "selector_backward: null pointer");
  TSPO_REQUIRE(g->wqkv && g->bqkv && g->w1 && g->b1 && g->w2 && g->b2, "selector_backward: null grad pointer");
  if (int e = check_dims("selector_backward", B, T, D, H, M, window)) return e;
  SelWs s = carve(workspace, B, T, D, H, M, window);
  if (workspace_bytes < s.bytes)
    return tspo::set_err(TSPO_EWORKSPACE, "selector_backward: workspace %zu < %zu", workspace_bytes, s.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int BT = B * T;
  dim3 tg((D + 31) / 32, (D + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, tg, dim3(256), 0, st, w->w1, s.w1t, D, D);
  hipLaunchKernelGGL(transpose_kernel, tg, dim3(256), 0, st, w->w2, s.w2t, D, D);
  // score -> dh2
  hipLaunchKernelGGL(score_bwd_kernel, dim3((BT + 3) / 4), dim3(256), 0, st, s.h2, txt, dscores, s.dh2, B, T, D, M, tau);
  // mlp.2
  if (int e = weight_grad(s.dh2, s.h1, g->w2, g->b2, BT, D, D, s, st)) return e;
  if (int e = launch_gemm_nt<EPI_MASK>(s.dh2, s.w2t, nullptr, s.h1, s.dh1, BT, D, D, st)) return e;
  // mlp.0
  if (int e = weight_grad(s.dh1, s.ctx, g->w1, g->b1, BT, D, D, s, st)) return e;
  if (int e = launch_gemm_nt<EPI_NONE>(s.dh1, s.w1t, nullptr, nullptr, s.dctx, BT, D, D, st)) return e;
  // banded attention
  const long pairs = (long)BT * H;
  const unsigned pb = (unsigned)((pairs + 7) / 8);
  hipLaunchKernelGGL(band_attn_bwd_q_kernel, dim3(pb), dim3(256), 0, st, s.qkv, s.P, s.dctx, s.dqkv, s.dS, B, T, D, H,
                     window);
  hipLaunchKernelGGL(band_attn_bwd_kv_kernel, dim3(pb), dim3(256), 0, st, s.qkv, s.P, s.dS, s.dctx, s.dqkv, B, T, D, H,
                     window);
  // q/k/v projections
  if (int e = weight_grad(s.dqkv, s.xpe, g->wqkv, g->bqkv, BT, 3 * D, D, s, st)) return e;
  (void)img;
  return tspo::check_launch("selector_backward");
}

extern "C" int tspo_clip_scores(const float* txt, const float* feat, int B, int T, int D, int M, float* clip,
                                tspo_stream_t stream) {
  TSPO_REQUIRE(txt && feat && clip, "clip_scores: null pointer");
  TSPO_REQUIRE(B >= 1 && T >= 1 && D >= 1 && M >= 1, "clip_scores: bad dims");
  hipLaunchKernelGGL(clip_scores_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, txt, feat, clip, B, T,
                     D, M);
  return tspo::check_launch("clip_scores");
}
