// On-device CLIP image preprocessing (K1 of SURVEY 2.3): uint8 frames -> resize (shortest edge -> 224, PIL
// antialiased bicubic, 8-bit fixed point exactly as Pillow's ImagingResample) -> centre crop 224 -> uint8 CHW.
// Replaces the per-frame PIL loop of the reference (model/temporal_agent.py:156-164, CLIPImageProcessor);
// the (x/255-mean)/std step is fused into the encoder's patch gather (tspo_clip_vit_forward, TSPO_U8).
// Two separable passes like Pillow: horizontal into a uint8 intermediate (rounded), then vertical.  The integer
// coefficient tables (22 fractional bits) come from the host (tspo_amd/preprocess.py), already restricted to the
// crop window, so results are bit-identical to PIL.
#include "common.h"

namespace {

// 32-bit accumulation like Pillow's ImagingResample (INT32 ss = 1 << 21; ss += pixel * k): |sum| <= 255 * sum|k| with
// sum|k| < 1.4 * 2^22 for the antialiased bicubic, i.e. < 2^31.
// taps are 22-bit fixed point with |k| < 2^23 and pixels are 8-bit: v_mul_i32_i24 / v_mad_i32_i24 (full rate) give the exact
// product; a plain 32-bit multiply (v_mul_lo_u32) runs at a quarter of that rate
// The clamp goes through an opaque v_med3_i32: for `clip8(a) | clip8(b) << 8 | ...` hipcc (ROCm 7.2) selects the gfx950
// instruction v_ashr_pk_u8_i32 and treats the upper 16 bits of its result as zero, but on the hardware they carry bits of the
// first source, which are then OR-ed into bytes 2-3 of the packed dword (measured with tools/dbg_pp.py: a constant-128 image
// came out as 128 128 160 160 ...).
__device__ __forceinline__ uint32_t clip8(int v) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v >> 22), "v"(255));
  return (uint32_t)r;
}

// tmp[t][c][yy][x] for yy in [0, nrows): input row ylo + yy, output column x of the crop window
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ in, int layout, int T, int H, int W,
                                                         const int* __restrict__ coef, const int* __restrict__ bound,
                                                         int ow, int ksize, int ylo, int nrows,
                                                         uint8_t* __restrict__ tmp) {
  const size_t total = (size_t)T * nrows * ow;
  for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
    const int x = (int)(id % ow);
    const int yy = (int)((id / ow) % nrows);
    const size_t t = id / ((size_t)ow * nrows);
    const int xmin = bound[2 * x], xn = bound[2 * x + 1];
    const int* k = coef + (size_t)x * ksize;
    const int y = ylo + yy;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    if (layout == 0) {  // [T,H,W,3]
      const uint8_t* p = in + ((t * H + y) * (size_t)W + xmin) * 3;
      for (int i = 0; i < xn; ++i) {
        const int w = k[i];
        a0 += __mul24(w, (int)p[3 * i]); a1 += __mul24(w, (int)p[3 * i + 1]); a2 += __mul24(w, (int)p[3 * i + 2]);
      }
    } else {  // [T,3,H,W]
      const uint8_t* p = in + ((t * 3) * (size_t)H + y) * W + xmin;
      const size_t cs = (size_t)H * W;
      for (int i = 0; i < xn; ++i) {
        const int w = k[i];
        a0 += __mul24(w, (int)p[i]); a1 += __mul24(w, (int)p[cs + i]); a2 += __mul24(w, (int)p[2 * cs + i]);
      }
    }
    const size_t o = ((t * 3) * (size_t)nrows + yy) * ow + x;
    const size_t cs = (size_t)nrows * ow;
    tmp[o] = (uint8_t)clip8(a0); tmp[o + cs] = (uint8_t)clip8(a1); tmp[o + 2 * cs] = (uint8_t)clip8(a2);
  }
}

__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, int T, int nrows, int ow,
                                                         const int* __restrict__ coef, const int* __restrict__ bound,
                                                         int oh, int ksize, int ylo, uint8_t* __restrict__ out) {
  const size_t total = (size_t)T * 3 * oh * ow;
  for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
    const int x = (int)(id % ow);
    const int y = (int)((id / ow) % oh);
    const size_t tc = id / ((size_t)ow * oh);
    const int ymin = bound[2 * y] - ylo, yn = bound[2 * y + 1];
    const int* k = coef + (size_t)y * ksize;
    const uint8_t* p = tmp + (tc * nrows + ymin) * (size_t)ow + x;
    int a = 1 << 21;
    for (int i = 0; i < yn; ++i) a += __mul24(k[i], (int)p[(size_t)i * ow]);
    out[id] = (uint8_t)clip8(a);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged forms (round 3).  The byte-granular kernels above issue one global byte load per tap and channel (42 per output
// pixel at 720p): 0.6 TB/s of input.  Here a workgroup owns 64 input rows x 32 output columns: the contiguous input span
// those columns need (32*scale + 2*support + 1 pixels, ~350 bytes per row at 720p) is brought in ONCE with coalesced
// dword loads, and the horizontal pass then runs with LANES = ROWS: all 64 lanes of a wave work on the same output column,
// so tap count, tap offsets and the 22-bit coefficients are wave-uniform (scalar loads) and every lane reads the same byte
// offset of its own LDS row - rows are pitched an odd number of dwords apart, i.e. conflict-free.  Four output columns
// are packed into one dword store.  The vertical pass handles four adjacent columns per thread (dword loads of the uint8
// intermediate, coalesced along x).  Arithmetic (32-bit accumulators, 1 << 21 rounding, clip) is unchanged: bit-exact.
#define PP_ROWS 64
#define PP_XC 32
template <int LAYOUT>
__global__ __launch_bounds__(256) void resample_h_lds_kernel(const uint8_t* __restrict__ in, size_t in_bytes, int T, int H, int W,
                                                             const int* __restrict__ coef, const int* __restrict__ bound,
                                                             int ow, int ksize, int ylo, int nrows, int nrb, int nchunk,
                                                             int pitch, uint8_t* __restrict__ tmp) {
  extern __shared__ uint32_t pp_lds[];   // LAYOUT 0 (THWC): [64][pitch]; LAYOUT 1 (TCHW): [3][64][pitch]   (pitch in bytes)
  constexpr int NP = LAYOUT == 0 ? 1 : 3, TS = LAYOUT == 0 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x % nchunk, rb = (blockIdx.x / nchunk) % nrb;
  const size_t t = blockIdx.x / ((size_t)nchunk * nrb);
  const int x0 = chunk * PP_XC, x1 = x0 + PP_XC < ow ? x0 + PP_XC : ow;
  const int xs = bound[2 * x0], xe = bound[2 * (x1 - 1)] + bound[2 * (x1 - 1) + 1];
  const int span = (xe - xs) * TS;
  const int yy0 = rb * PP_ROWS;
  const int ndw = (span + 6) / 4 + 1;             // dwords per row: the span plus up to 3 bytes of alignment shift in front
  const int pdw = pitch >> 2;
  auto row_addr = [&](int r, int pl) -> uintptr_t {   // global address of byte 0 of the span of row r (plane pl)
    int yy = yy0 + r;
    yy = yy < nrows ? yy : nrows - 1;
    const size_t y = (size_t)(ylo + yy);
    const size_t off = LAYOUT == 0 ? ((t * H + y) * (size_t)W + xs) * 3 : ((t * 3 + pl) * (size_t)H + y) * W + xs;
    return reinterpret_cast<uintptr_t>(in) + off;
  };
  const uintptr_t in_end = reinterpret_cast<uintptr_t>(in) + in_bytes;
  {   // staging: four threads per row, 16 bytes per load (global loads only need dword alignment), rows realigned per
      // row to a dword boundary (the shift is added back to the lane's LDS offset below)
    const int r = tid >> 2, q = tid & 3;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const uintptr_t a0 = row_addr(r, pl) & ~(uintptr_t)3;
      uint32_t* dst = pp_lds + (pl * PP_ROWS + r) * pdw;
      for (int d = q * 4; d < ndw; d += 16) {
        const uintptr_t a = a0 + (uintptr_t)d * 4;
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        if (a + 16 <= in_end) {
          const uint4 t4 = *reinterpret_cast<const uint4*>(a);
          v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
        } else {   // the tail of the last rows: never read past the buffer
          for (int b = 0; b < 16; ++b)
            if (a + b < in_end) v[b >> 2] |= (uint32_t)(*reinterpret_cast<const uint8_t*>(a + b)) << (8 * (b & 3));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (d + e < pdw) dst[d + e] = v[e];
      }
    }
  }
  // this chunk's taps and bounds next to the rows: read back as LDS broadcasts (a scalar load per tap would put a
  // memory latency into every iteration of the tap loop)
  int* kl = reinterpret_cast<int*>(pp_lds + NP * PP_ROWS * pdw);
  int* bl = kl + PP_XC * ksize;
  for (int idx = tid; idx < (x1 - x0) * ksize; idx += 256) kl[idx] = coef[(size_t)x0 * ksize + idx];
  if (tid < 2 * (x1 - x0)) bl[tid] = bound[2 * x0 + tid];
  __syncthreads();
  const uint8_t* lb = reinterpret_cast<const uint8_t*>(pp_lds);
  int rowoff[NP];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) rowoff[pl] = (pl * PP_ROWS + lane) * pitch + (int)(row_addr(lane, pl) & 3);
  const int yy = yy0 + lane;
  const int xw0 = x0 + wv * 8;
  uint32_t pk[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
  for (int j = 0; j < 8; ++j) {
    const int x = xw0 + j;       // wave-uniform
    if (x >= x1) break;
    const int xmin = bl[2 * (x - x0)], xn = bl[2 * (x - x0) + 1];
    const int* k = kl + (x - x0) * ksize;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    if (LAYOUT == 0) {
      const uint8_t* p = lb + rowoff[0] + (xmin - xs) * 3;
#pragma unroll 4
      for (int i = 0; i < xn; ++i) {
        const int w = k[i];
        a0 += __mul24(w, (int)p[3 * i]); a1 += __mul24(w, (int)p[3 * i + 1]); a2 += __mul24(w, (int)p[3 * i + 2]);
      }
    } else {
      const uint8_t* p0 = lb + rowoff[0] + (xmin - xs);
      const uint8_t* p1 = lb + rowoff[NP > 1 ? 1 : 0] + (xmin - xs);
      const uint8_t* p2 = lb + rowoff[NP > 2 ? 2 : 0] + (xmin - xs);
      for (int i = 0; i < xn; ++i) {
        const int w = k[i];
        a0 += __mul24(w, (int)p0[i]); a1 += __mul24(w, (int)p1[i]); a2 += __mul24(w, (int)p2[i]);
      }
    }
    const int sh = 8 * (j & 3), q = j >> 2;
    pk[0][q] |= (uint32_t)clip8(a0) << sh; pk[1][q] |= (uint32_t)clip8(a1) << sh; pk[2][q] |= (uint32_t)clip8(a2) << sh;
  }
  if (yy < nrows && xw0 < x1) {
    const size_t cs = (size_t)nrows * ow;
    uint8_t* o = tmp + ((t * 3) * (size_t)nrows + yy) * ow + xw0;
    const int nx = x1 - xw0 < 8 ? x1 - xw0 : 8;    // multiple of 4 (ow % 4 == 0)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      *reinterpret_cast<uint32_t*>(o + c * cs) = pk[c][0];
      if (nx > 4) *reinterpret_cast<uint32_t*>(o + c * cs + 4) = pk[c][1];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Horizontal pass on the matrix pipe (round 3).  A row's resize is a banded matrix product: out[x] = sum_i K[x, i] in[i].
// For 16 consecutive output columns the taps touch at most 15 * scale + ksize + 1 input columns, i.e. one or two
// 64-wide K blocks, so a tile "16 output columns x 16 image rows of one channel" is NKB x 3 v_mfma_i32_16x16x64_i8:
//   * the 22-bit taps are split on the host into three SIGNED byte digits k = d0 + 2^8 d1 + 2^16 d2 (one A matrix per
//     digit, zero outside the tap range), pixels enter as p - 128 (XOR 0x80 while staging), and the constant
//     128 * sum(k) + 2^21 comes back in as a per-column bias: d0 + (d1 << 8) + (d2 << 16) + bias == Pillow's 32-bit
//     accumulator exactly (wrap-around arithmetic; the true value fits);
//   * src0 = taps (row = output column), src1 = pixels (row = image row), so a lane ends up with 4 consecutive output
//     columns of one image row: one packed dword store;
//   * a workgroup stages 32 rows x the span of 64 output columns as three byte planes (R | G | B de-interleaved in
//     registers with v_perm_b32 on the way in), each wave owns one 16-column block, its tap fragments live in registers.
// Same results bit for bit as the scalar kernels (tests/test_preprocess.py); ~5x fewer issued instructions per output.
typedef int pp_v4i __attribute__((ext_vector_type(4)));
#define PM_ROWS 32
template <int LAYOUT, int NKB>
__global__ __launch_bounds__(256) void resample_h_mfma_kernel(const uint8_t* __restrict__ in, size_t in_bytes, int T, int H, int W,
                                                              const pp_v4i* __restrict__ atab, const int* __restrict__ bias,
                                                              const int* __restrict__ xs_tab, int nblk, int ow,
                                                              int ylo, int nrows, int nrb, int nchunk, int pitch,
                                                              uint8_t* __restrict__ tmp) {
  extern __shared__ uint32_t pm_lds[];   // [3 planes][32 rows][pitch bytes], pixels as signed bytes (p ^ 0x80)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x % nchunk, rb32 = (blockIdx.x / nchunk) % nrb;
  const size_t t = blockIdx.x / ((size_t)nchunk * nrb);
  constexpr int nkb = NKB;
  const int b0 = chunk * 4, b1 = b0 + 4 < nblk ? b0 + 4 : nblk;
  const int xs0 = xs_tab[b0];
  const int span = xs_tab[b1 - 1] + nkb * 64 - xs0;      // pixels any block of the chunk may touch (taps are 0 past a block's range)
  const int yy0 = rb32 * PM_ROWS;
  const int pdw = pitch >> 2;
  const uintptr_t in_end = reinterpret_cast<uintptr_t>(in) + in_bytes;
  const size_t frame_px = (size_t)H * W;
  // ---- staging: groups of 4 pixels -> one dword per plane ----
  const int ngrp = (span + 3) >> 2;
  const int srow = tid >> 3;               // 8 threads per row, 32 rows: the row's base address is computed once
  int syy = yy0 + srow;
  syy = syy < nrows ? syy : nrows - 1;
  const size_t y = (size_t)(ylo + syy);
  const uintptr_t rowbase0 = reinterpret_cast<uintptr_t>(in) + ((t * H + y) * (size_t)W + xs0) * 3;                // THWC
  const uintptr_t rowbase1 = reinterpret_cast<uintptr_t>(in) + (t * 3) * frame_px + y * (size_t)W + xs0;           // TCHW, plane 0
  for (int gi = tid & 7; gi < ngrp; gi += 8) {
    const int r = srow;
    uint32_t out3[3];                       // (columns past the row end read the next row's bytes: their taps are 0)
    if (LAYOUT == 0) {
      const uintptr_t a = rowbase0 + (uintptr_t)gi * 12;
      const uintptr_t al = a & ~(uintptr_t)3;
      const int sh = (int)(a & 3);
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (al + 16 <= in_end) {
        const uint4 v = *reinterpret_cast<const uint4*>(al);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      } else {
        for (int bb = 0; bb < 16; ++bb)
          if (al + bb < in_end) w[bb >> 2] |= (uint32_t)(*reinterpret_cast<const uint8_t*>(al + bb)) << (8 * (bb & 3));
      }
      const uint32_t d0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh), d1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh),
                     d2 = __builtin_amdgcn_alignbyte(w[3], w[2], sh);
      // d0 = [R0 G0 B0 R1], d1 = [G1 B1 R2 G2], d2 = [B2 R3 G3 B3]  (byte 0 first).  v_perm_b32(S0, S1, sel): byte
      // selectors 0-3 take S1's bytes, 4-7 take S0's.
      const uint32_t r01 = __builtin_amdgcn_perm(d1, d0, 0x00060300u);   // [R0 R1 R2 x ] from d0 (0,3) and d1 (2 -> selector 6)
      const uint32_t g01 = __builtin_amdgcn_perm(d1, d0, 0x00070401u);   // [G0 G1 G2 x ] d0 byte 1, d1 bytes 0 (4) and 3 (7)
      const uint32_t b01 = __builtin_amdgcn_perm(d1, d0, 0x00000502u);   // [B0 B1 x  x ] d0 byte 2, d1 byte 1 (5)
      out3[0] = __builtin_amdgcn_perm(d2, r01, 0x05020100u);             // R3 = d2 byte 1 (5)
      out3[1] = __builtin_amdgcn_perm(d2, g01, 0x06020100u);             // G3 = d2 byte 2 (6)
      out3[2] = __builtin_amdgcn_perm(d2, b01, 0x07040100u);             // B2 = d2 byte 0 (4), B3 = d2 byte 3 (7)
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const uintptr_t a = rowbase1 + (uintptr_t)c * frame_px + (uintptr_t)gi * 4;
        const uintptr_t al = a & ~(uintptr_t)3;
        const int sh = (int)(a & 3);
        uint32_t w0 = 0u, w1 = 0u;
        if (al + 8 <= in_end) {
          const uint2 v = *reinterpret_cast<const uint2*>(al);
          w0 = v.x; w1 = v.y;
        } else {
          for (int bb = 0; bb < 8; ++bb)
            if (al + bb < in_end) (bb < 4 ? w0 : w1) |= (uint32_t)(*reinterpret_cast<const uint8_t*>(al + bb)) << (8 * (bb & 3));
        }
        out3[c] = __builtin_amdgcn_alignbyte(w1, w0, sh);
      }
    }
    if (gi < pdw) {
#pragma unroll
      for (int c = 0; c < 3; ++c) pm_lds[(c * PM_ROWS + r) * pdw + gi] = out3[c] ^ 0x80808080u;
    }
  }
  __syncthreads();
  // ---- this wave's 16-column block ----
  const int blk = b0 + wv;
  if (blk >= b1) return;
  pp_v4i afrag[NKB][3];     // [K block][digit]
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int d = 0; d < 3; ++d) afrag[kb][d] = atab[(((size_t)blk * nkb + kb) * 3 + d) * 64 + lane];
  const pp_v4i bq = *reinterpret_cast<const pp_v4i*>(bias + blk * 16 + q4 * 4);
  const int ob = xs_tab[blk] - xs0;                           // byte offset of the block's first input column in a plane row
  const int sh = ob & 3;
  const char* lb = reinterpret_cast<const char*>(pm_lds);
  uint8_t* const obase = tmp + ((t * 3) * (size_t)nrows + yy0 + l15) * ow + blk * 16 + q4 * 4;      // (plane 0, row block 0)
  const size_t oplane = (size_t)nrows * ow;
#pragma unroll
  for (int rbk = 0; rbk < 2; ++rbk) {
    const int yy = yy0 + rbk * 16 + l15;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pp_v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      const char* rowp = lb + ((c * PM_ROWS + rbk * 16 + l15) * pitch) + (ob & ~3) + q4 * 16;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        uint32_t w[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) w[i] = *reinterpret_cast<const uint32_t*>(rowp + kb * 64 + i * 4);
        pp_v4i bf;
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = (int)__builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[kb][d], bf, acc[d], 0, 0, 0);
      }
      uint32_t pk = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = (int)((uint32_t)acc[0][r] + ((uint32_t)acc[1][r] << 8) + ((uint32_t)acc[2][r] << 16) + (uint32_t)bq[r]);
        pk |= clip8(v) << (8 * r);
      }
      if (yy < nrows) *reinterpret_cast<uint32_t*>(obase + c * oplane + (size_t)rbk * 16 * ow) = pk;
    }
  }
}

// vertical pass: PV_R output rows x 4 adjacent output columns per thread (ow % 4 == 0, buffers dword-aligned).  A wave owns
// PV_R = 4 consecutive output rows: at a down-scale of s their tap ranges overlap by ksize - s input rows from one row to
// the next, so every dword of the uint8 intermediate is loaded once for all four (2.4x fewer loads at 720p) and feeds
// up to 16 multiply-adds (PV_R = 8 was measured slower: 92 vs 75 us for 256 720p frames - too few workgroups per plane).  The taps of the rows are laid out in LDS against the wave's common input-row range, zero
// outside a row's own range, so the loop has no branches; tap reads are LDS broadcasts (a wave's rows are uniform).
#define PV_R 4
__global__ __launch_bounds__(256) void resample_v4_kernel(const uint8_t* __restrict__ tmp, int T, int nrows, int ow,
                                                          const int* __restrict__ coef, const int* __restrict__ bound,
                                                          int oh, int ksize, int ylo, int vspan, uint8_t* __restrict__ out) {
  extern __shared__ int pv_lds[];   // [4 waves][PV_R rows][vspan] taps against the wave's input-row range
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ow4 = ow >> 2, xblocks = (ow4 + 63) >> 6, ygroups = (oh + 4 * PV_R - 1) / (4 * PV_R);
  const int xb = blockIdx.x % xblocks, yg = (blockIdx.x / xblocks) % ygroups;
  const size_t tc = blockIdx.x / ((size_t)xblocks * ygroups);
  for (int idx = tid; idx < 4 * PV_R * vspan; idx += 256) pv_lds[idx] = 0;
  __syncthreads();
  {   // thread -> (row of the workgroup, tap): 16 threads per row scatter the row's taps to their offset in its wave's range
    static_assert(256 / (4 * PV_R) == 16, "thread-to-row mapping of the tap scatter: 256 threads = 4 waves x PV_R rows x 16 threads");
    const int rr = tid >> 4, y = yg * 4 * PV_R + rr;          // 4 * PV_R rows x 16 threads: every rr is a row this workgroup owns
    if (y < oh) {
      const int y0 = yg * 4 * PV_R + (rr / PV_R) * PV_R;
      const int base = bound[2 * y0], ymin = bound[2 * y], yn = bound[2 * y + 1];
      for (int ti = tid & 15; ti < yn; ti += 16) {
        const int o = ymin - base + ti;
        if (o < vspan) pv_lds[rr * vspan + o] = coef[(size_t)y * ksize + ti];
      }
    }
  }
  __syncthreads();
  const int y0 = yg * 4 * PV_R + wv * PV_R, x4 = xb * 64 + lane;
  if (y0 >= oh || x4 >= ow4) return;
  const int ylast = y0 + PV_R - 1 < oh ? y0 + PV_R - 1 : oh - 1;
  const int rbase = bound[2 * y0] - ylo;
  int rend = bound[2 * ylast] + bound[2 * ylast + 1] - ylo;
  if (rend - rbase > vspan) rend = rbase + vspan;
  const int* k = pv_lds + wv * PV_R * vspan;
  const uint8_t* p = tmp + (tc * nrows + rbase) * (size_t)ow + (size_t)x4 * 4;
  int acc[PV_R][4];
#pragma unroll
  for (int rr = 0; rr < PV_R; ++rr)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[rr][c] = 1 << 21;
#pragma unroll 4
  for (int i = 0; i < rend - rbase; ++i) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p + (size_t)i * ow);
    const int b0 = (int)(u & 255u), b1 = (int)((u >> 8) & 255u), b2 = (int)((u >> 16) & 255u), b3 = (int)(u >> 24);
#pragma unroll
    for (int rr = 0; rr < PV_R; ++rr) {
      const int w = k[rr * vspan + i];
      acc[rr][0] += __mul24(w, b0); acc[rr][1] += __mul24(w, b1); acc[rr][2] += __mul24(w, b2); acc[rr][3] += __mul24(w, b3);
    }
  }
#pragma unroll
  for (int rr = 0; rr < PV_R; ++rr)
    if (y0 + rr < oh)
      *reinterpret_cast<uint32_t*>(out + (tc * oh + y0 + rr) * (size_t)ow + (size_t)x4 * 4) =
          clip8(acc[rr][0]) | (clip8(acc[rr][1]) << 8) | (clip8(acc[rr][2]) << 16) | (clip8(acc[rr][3]) << 24);
}

}  // namespace

extern "C" size_t tspo_preprocess_workspace_bytes(int T, int nrows, int out_w) {
  if (T < 1 || nrows < 1 || out_w < 1) return 0;
  return tspo::align_up((size_t)T * 3 * nrows * out_w, 256);
}

static int preprocess_impl(const uint8_t* frames, int layout, int T, int H, int W, const int32_t* hcoef,
                           const int32_t* hbound, int out_w, int hk, const int32_t* vcoef,
                           const int32_t* vbound, int out_h, int vk, int ylo, int nrows, uint8_t* out,
                           void* workspace, size_t workspace_bytes, tspo_stream_t stream, const int8_t* mfma_taps,
                           const int32_t* mfma_bias, const int32_t* mfma_xs, int mfma_nkb, int mfma_span) {
  TSPO_REQUIRE(frames && hcoef && hbound && vcoef && vbound && out && workspace, "preprocess_frames: null pointer");
  TSPO_REQUIRE(T >= 1 && H >= 1 && W >= 1 && out_w >= 1 && out_h >= 1 && hk >= 1 && vk >= 1, "preprocess_frames: bad dims");
  TSPO_REQUIRE(layout == 0 || layout == 1, "preprocess_frames: layout must be 0 (THWC) or 1 (TCHW)");
  TSPO_REQUIRE(ylo >= 0 && nrows >= 1 && ylo + nrows <= H, "preprocess_frames: row window [%d,%d) outside H=%d", ylo, ylo + nrows, H);
  const size_t need = tspo_preprocess_workspace_bytes(T, nrows, out_w);
  if (workspace_bytes < need)
    return tspo::set_err(TSPO_EWORKSPACE, "preprocess_frames: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  size_t n1 = (size_t)T * nrows * out_w, n2 = (size_t)T * 3 * out_h * out_w;
  unsigned g1 = (unsigned)((n1 + 255) / 256 > 65536 ? 65536 : (n1 + 255) / 256);
  unsigned g2 = (unsigned)((n2 + 255) / 256 > 65536 ? 65536 : (n2 + 255) / 256);
  // LDS-staged path: needs dword-aligned rows of the intermediate / output (out_w % 4 == 0) and the widest input span of
  // a 32-column chunk x 64 rows to fit the 64 KB of LDS a workgroup may use (scale factors up to ~14); the horizontal
  // tables live on the device, so the span bound is computed from the geometry: 32 * scale + taps
  const bool aligned = out_w % 4 == 0 && ((uintptr_t)workspace & 3) == 0 && ((uintptr_t)out & 3) == 0;
  // input rows that PV_R consecutive output rows can touch: the last row's taps start at most (PV_R - 1) * scale + 1 rows after
  // the first's and add vk rows; vk = 2 * ceil(2 * scale) + 1 bounds the scale: scale <= (vk - 1) / 4
  const int vspan = vk > 1 ? ((PV_R - 1) * (vk - 1) + 3) / 4 + vk + 2 : PV_R;
  // matrix-pipe horizontal pass: the caller supplies the digit-split tap matrices (tspo_amd/preprocess.py builds them with
  // the tap tables); needs 16-column blocks, at most 4 K blocks of 64 input columns per block and planes that fit in LDS
  if (mfma_taps && aligned && out_w % 16 == 0 && mfma_nkb >= 1 && mfma_nkb <= 4 && hk > 1) {
    TSPO_REQUIRE(mfma_bias && mfma_xs && mfma_span >= 1, "preprocess_frames_ex: incomplete matrix-pass tables");
    int mp = ((mfma_span + 16 + 3) / 4) * 4;
    if (((mp >> 2) & 1) == 0) mp += 4;
    const size_t mlds = (size_t)3 * PM_ROWS * mp;
    if (mlds <= 64 * 1024) {
      const int nblk = out_w / 16, nchunk = (nblk + 3) / 4, nrb = (nrows + PM_ROWS - 1) / PM_ROWS;
      const size_t in_bytes = (size_t)T * H * W * 3;
      const unsigned grid = (unsigned)((size_t)T * nrb * nchunk);
#define PM_LAUNCH(L, K)                                                                                                   \
  hipLaunchKernelGGL((resample_h_mfma_kernel<L, K>), dim3(grid), dim3(256), mlds, st, frames, in_bytes, T, H, W,               \
                     (const pp_v4i*)mfma_taps, mfma_bias, mfma_xs, nblk, out_w, ylo, nrows, nrb, nchunk, mp, (uint8_t*)workspace)
      if (layout == 0) {
        if (mfma_nkb == 1) PM_LAUNCH(0, 1); else if (mfma_nkb == 2) PM_LAUNCH(0, 2); else if (mfma_nkb == 3) PM_LAUNCH(0, 3); else PM_LAUNCH(0, 4);
      } else {
        if (mfma_nkb == 1) PM_LAUNCH(1, 1); else if (mfma_nkb == 2) PM_LAUNCH(1, 2); else if (mfma_nkb == 3) PM_LAUNCH(1, 3); else PM_LAUNCH(1, 4);
      }
#undef PM_LAUNCH
      const size_t g4 = (size_t)T * 3 * ((out_h + 4 * PV_R - 1) / (4 * PV_R)) * ((out_w / 4 + 63) / 64);
      hipLaunchKernelGGL(resample_v4_kernel, dim3((unsigned)g4), dim3(256), (size_t)4 * PV_R * vspan * 4, st, (const uint8_t*)workspace, T,
                         nrows, out_w, vcoef, vbound, out_h, vk, ylo, vspan, out);
      return tspo::check_launch("preprocess_frames");
    }
  }
  const int ts = layout == 0 ? 3 : 1, planes = layout == 0 ? 1 : 3;
  // span of a chunk in pixels: xmin advances by at most scale per column (+1 for the truncation) and the last column adds
  // its taps; ksize = 2 * ceil(2 * scale) + 1 bounds the scale from above: scale <= (hk - 1) / 4
  const int span_px = (31 * (hk - 1) + 3) / 4 + hk + 2;
  int pitch = ((span_px * ts + 8 + 3) / 4 + 1) * 4;
  if (((pitch >> 2) & 1) == 0) pitch += 4;                                      // odd number of dwords: conflict-free rows
  const size_t lds_bytes = (size_t)planes * PP_ROWS * pitch + (size_t)PP_XC * (hk + 2) * 4;
  if (aligned && lds_bytes <= 64 * 1024 && hk > 1) {
    const int nrb = (nrows + PP_ROWS - 1) / PP_ROWS, nchunk = (out_w + PP_XC - 1) / PP_XC;
    const size_t in_bytes = (size_t)T * H * W * 3;
    const unsigned grid = (unsigned)((size_t)T * nrb * nchunk);
    if (layout == 0)
      hipLaunchKernelGGL(resample_h_lds_kernel<0>, dim3(grid), dim3(256), lds_bytes, st, frames, in_bytes, T, H, W, hcoef, hbound,
                         out_w, hk, ylo, nrows, nrb, nchunk, pitch, (uint8_t*)workspace);
    else
      hipLaunchKernelGGL(resample_h_lds_kernel<1>, dim3(grid), dim3(256), lds_bytes, st, frames, in_bytes, T, H, W, hcoef, hbound,
                         out_w, hk, ylo, nrows, nrb, nchunk, pitch, (uint8_t*)workspace);
    const size_t g4 = (size_t)T * 3 * ((out_h + 4 * PV_R - 1) / (4 * PV_R)) * ((out_w / 4 + 63) / 64);
    hipLaunchKernelGGL(resample_v4_kernel, dim3((unsigned)g4), dim3(256), (size_t)4 * PV_R * vspan * 4, st, (const uint8_t*)workspace, T,
                       nrows, out_w, vcoef, vbound, out_h, vk, ylo, vspan, out);
    return tspo::check_launch("preprocess_frames");
  }
  hipLaunchKernelGGL(resample_h_kernel, dim3(g1), dim3(256), 0, st, frames, layout, T, H, W, hcoef, hbound, out_w, hk, ylo,
                     nrows, (uint8_t*)workspace);
  hipLaunchKernelGGL(resample_v_kernel, dim3(g2), dim3(256), 0, st, (const uint8_t*)workspace, T, nrows, out_w, vcoef,
                     vbound, out_h, vk, ylo, out);
  return tspo::check_launch("preprocess_frames");
}

extern "C" int tspo_preprocess_frames(const uint8_t* frames, int layout, int T, int H, int W, const int32_t* hcoef,
                                      const int32_t* hbound, int out_w, int hk, const int32_t* vcoef,
                                      const int32_t* vbound, int out_h, int vk, int ylo, int nrows, uint8_t* out,
                                      void* workspace, size_t workspace_bytes, tspo_stream_t stream) {
  return preprocess_impl(frames, layout, T, H, W, hcoef, hbound, out_w, hk, vcoef, vbound, out_h, vk, ylo, nrows, out, workspace,
                         workspace_bytes, stream, nullptr, nullptr, nullptr, 0, 0);
}

extern "C" int tspo_preprocess_frames_ex(const uint8_t* frames, int layout, int T, int H, int W, const int32_t* hcoef,
                                         const int32_t* hbound, int out_w, int hk, const int32_t* vcoef,
                                         const int32_t* vbound, int out_h, int vk, int ylo, int nrows, uint8_t* out,
                                         void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                                         const int8_t* mfma_taps, const int32_t* mfma_bias, const int32_t* mfma_xs, int mfma_nkb,
                                         int mfma_span) {
  return preprocess_impl(frames, layout, T, H, W, hcoef, hbound, out_w, hk, vcoef, vbound, out_h, vk, ylo, nrows, out, workspace,
                         workspace_bytes, stream, mfma_taps, mfma_bias, mfma_xs, mfma_nkb, mfma_span);
}
