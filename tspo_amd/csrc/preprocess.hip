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
__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= 22;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
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
    tmp[o] = clip8(a0); tmp[o + cs] = clip8(a1); tmp[o + 2 * cs] = clip8(a2);
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
    out[id] = clip8(a);
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

// vertical pass: four adjacent output columns per thread (ow % 4 == 0, buffers dword-aligned), one output ROW per wave, so
// the row's taps and bounds are wave-uniform: staged in LDS once per workgroup (4 rows) and read back as broadcasts; the
// only per-lane memory traffic left is one coalesced dword load of the uint8 intermediate per tap
__global__ __launch_bounds__(256) void resample_v4_kernel(const uint8_t* __restrict__ tmp, int T, int nrows, int ow,
                                                          const int* __restrict__ coef, const int* __restrict__ bound,
                                                          int oh, int ksize, int ylo, uint8_t* __restrict__ out) {
  extern __shared__ int pv_lds[];   // [4][ksize] taps, then [4][2] bounds
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ow4 = ow >> 2, xblocks = (ow4 + 63) >> 6, ygroups = (oh + 3) >> 2;
  const int xb = blockIdx.x % xblocks, yg = (blockIdx.x / xblocks) % ygroups;
  const size_t tc = blockIdx.x / ((size_t)xblocks * ygroups);
  int* bl = pv_lds + 4 * ksize;
  for (int idx = tid; idx < 4 * ksize; idx += 256) {
    const int yy = yg * 4 + idx / ksize;
    pv_lds[idx] = yy < oh ? coef[(size_t)yy * ksize + idx % ksize] : 0;
  }
  if (tid < 8) { const int yy = yg * 4 + (tid >> 1); bl[tid] = yy < oh ? bound[2 * yy + (tid & 1)] : 0; }
  __syncthreads();
  const int y = yg * 4 + wv, x4 = xb * 64 + lane;
  if (y >= oh || x4 >= ow4) return;
  const int ymin = bl[2 * wv] - ylo, yn = bl[2 * wv + 1];
  const int* k = pv_lds + wv * ksize;
  const uint8_t* p = tmp + (tc * nrows + ymin) * (size_t)ow + (size_t)x4 * 4;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
#pragma unroll 4
  for (int i = 0; i < yn; ++i) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p + (size_t)i * ow);
    const int w = k[i];
    a0 += __mul24(w, (int)(u & 255u)); a1 += __mul24(w, (int)((u >> 8) & 255u)); a2 += __mul24(w, (int)((u >> 16) & 255u));
    a3 += __mul24(w, (int)(u >> 24));
  }
  *reinterpret_cast<uint32_t*>(out + (tc * oh + y) * (size_t)ow + (size_t)x4 * 4) =
      (uint32_t)clip8(a0) | ((uint32_t)clip8(a1) << 8) | ((uint32_t)clip8(a2) << 16) | ((uint32_t)clip8(a3) << 24);
}

}  // namespace

extern "C" size_t tspo_preprocess_workspace_bytes(int T, int nrows, int out_w) {
  if (T < 1 || nrows < 1 || out_w < 1) return 0;
  return tspo::align_up((size_t)T * 3 * nrows * out_w, 256);
}

extern "C" int tspo_preprocess_frames(const uint8_t* frames, int layout, int T, int H, int W, const int32_t* hcoef,
                                      const int32_t* hbound, int out_w, int hk, const int32_t* vcoef,
                                      const int32_t* vbound, int out_h, int vk, int ylo, int nrows, uint8_t* out,
                                      void* workspace, size_t workspace_bytes, tspo_stream_t stream) {
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
    const size_t g4 = (size_t)T * 3 * ((out_h + 3) / 4) * ((out_w / 4 + 63) / 64);
    hipLaunchKernelGGL(resample_v4_kernel, dim3((unsigned)g4), dim3(256), (size_t)(4 * vk + 8) * 4, st, (const uint8_t*)workspace, T,
                       nrows, out_w, vcoef, vbound, out_h, vk, ylo, out);
    return tspo::check_launch("preprocess_frames");
  }
  hipLaunchKernelGGL(resample_h_kernel, dim3(g1), dim3(256), 0, st, frames, layout, T, H, W, hcoef, hbound, out_w, hk, ylo,
                     nrows, (uint8_t*)workspace);
  hipLaunchKernelGGL(resample_v_kernel, dim3(g2), dim3(256), 0, st, (const uint8_t*)workspace, T, nrows, out_w, vcoef,
                     vbound, out_h, vk, ylo, out);
  return tspo::check_launch("preprocess_frames");
}
