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
        a0 += w * (int)p[3 * i]; a1 += w * (int)p[3 * i + 1]; a2 += w * (int)p[3 * i + 2];
      }
    } else {  // [T,3,H,W]
      const uint8_t* p = in + ((t * 3) * (size_t)H + y) * W + xmin;
      const size_t cs = (size_t)H * W;
      for (int i = 0; i < xn; ++i) {
        const int w = k[i];
        a0 += w * (int)p[i]; a1 += w * (int)p[cs + i]; a2 += w * (int)p[2 * cs + i];
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
    for (int i = 0; i < yn; ++i) a += k[i] * (int)p[(size_t)i * ow];
    out[id] = clip8(a);
  }
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
  hipLaunchKernelGGL(resample_h_kernel, dim3(g1), dim3(256), 0, st, frames, layout, T, H, W, hcoef, hbound, out_w, hk, ylo,
                     nrows, (uint8_t*)workspace);
  hipLaunchKernelGGL(resample_v_kernel, dim3(g2), dim3(256), 0, st, (const uint8_t*)workspace, T, nrows, out_w, vcoef,
                     vbound, out_h, vk, ylo, out);
  return tspo::check_launch("preprocess_frames");
}
