// Frame samplers + policy-gradient reductions + optimiser for gfx950.
// HBM/latency-bound integer & reduction work: one workgroup per (prompt[,rollout])
// row, the row resident in LDS, wave64 shuffles for reductions.  No T x T or
// sorted intermediates are materialised.
#include "common.h"
#include <stdarg.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace tspo {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(TSPO_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TSPO_OK;
}
}  // namespace tspo

extern "C" int tspo_version(void) { return TSPO_ABI_VERSION; }
extern "C" const char* tspo_last_error(void) { return tspo::err_buf(); }

// ---------------------------------------------------------------------------
// order-preserving float -> uint32 key (larger float <=> larger key; NaN largest;
// -0.0 == +0.0)
__device__ __forceinline__ uint32_t order_key(float f) {
  if (f != f) return 0xFFFFFFFFu;
  if (f == 0.f) f = 0.f;
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (counter-based RNG for in-kernel Gumbel noise)
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ uint32_t philox_x0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}
__device__ __forceinline__ float gumbel_from_bits(uint32_t x) {
  float u = (float)(((double)x + 0.5) * 2.3283064365386963e-10);  // 2^-32
  u = fminf(u, 0.99999994f);                                       // 1 - 2^-24
  return -logf(-logf(u));
}

// ---------------------------------------------------------------------------
// Radix select + index-ordered compaction over keys held in LDS (or recomputed
// from global through `keyfn` when the row does not fit).
#define SEL_THREADS 256
#define SEL_LDS_KEYS 16384

struct SelShared {
  uint32_t hist[256];
  uint32_t prefix, need;
  uint32_t wsum_gt[SEL_THREADS / 64], wsum_eq[SEL_THREADS / 64];
};

// keys(t): order-preserving key of element t - an LDS array when the row fits (T <= SEL_LDS_KEYS), otherwise recomputed
// from the scores in global memory on each of the 6 passes (long videos: the evaluation harness samples up to 50000
// frames, gen_id_tspo.py:70).
template <typename KeyFn>
__device__ void select_topk_sorted(KeyFn keys, int T, int k, int64_t* __restrict__ out, SelShared& sh) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0, mask = 0, need = (uint32_t)k;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    if (tid < 256) sh.hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += SEL_THREADS) {
      const uint32_t key = keys(t);
      if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0 walks the histogram from the top: lane l owns bins 4l..4l+3; a suffix scan over the lanes finds the
      // one lane whose bins contain the need-th largest candidate, which then resolves the bin serially (4 steps).
      const uint32_t h0 = sh.hist[4 * tid], h1 = sh.hist[4 * tid + 1], h2 = sh.hist[4 * tid + 2], h3 = sh.hist[4 * tid + 3];
      const uint32_t tot = h0 + h1 + h2 + h3;
      uint32_t suf = tot;   // inclusive suffix sum over lanes >= tid
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(suf, o, 64);
        if (tid + o < 64) suf += v;
      }
      uint32_t cum = suf - tot;   // candidates in bins above this lane's
      const bool mine = cum < need && need <= suf;
      // fewer candidates than needed cannot happen (k <= T); lane 0 is the fallback like the serial walk's b = 0
      const bool fallback = tid == 0 && need > suf;
      if (mine || fallback) {
        int b = 4 * tid;
        if (cum + h3 >= need) b += 3;
        else if (cum + h3 + h2 >= need) { b += 2; cum += h3; }
        else if (cum + h3 + h2 + h1 >= need) { b += 1; cum += h3 + h2; }
        else cum += h3 + h2 + h1;
        sh.prefix = prefix | ((uint32_t)b << shift);
        sh.need = need - cum;
      }
    }
    __syncthreads();
    prefix = sh.prefix;
    need = sh.need;
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  // prefix = key of the k-th largest element; `need` elements equal to it are taken (lowest indices first)
  const uint32_t thr = prefix;
  const int chunk = (T + SEL_THREADS - 1) / SEL_THREADS;
  const int t0 = tid * chunk, t1 = min(T, t0 + chunk);
  int ngt = 0, neq = 0;
  for (int t = t0; t < t1; ++t) {
    const uint32_t key = keys(t);
    ngt += key > thr;
    neq += key == thr;
  }
  // exclusive block scan of (ngt, neq)
  const int lane = tid & 63, wid = tid >> 6;
  int sgt = ngt, seq = neq;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(sgt, o, 64), b = __shfl_up(seq, o, 64);
    if (lane >= o) { sgt += a; seq += b; }
  }
  if (lane == 63) { sh.wsum_gt[wid] = sgt; sh.wsum_eq[wid] = seq; }
  __syncthreads();
  int base_gt = 0, base_eq = 0;
  for (int w = 0; w < wid; ++w) { base_gt += sh.wsum_gt[w]; base_eq += sh.wsum_eq[w]; }
  int pgt = base_gt + sgt - ngt, peq = base_eq + seq - neq;  // exclusive prefixes
  for (int t = t0; t < t1; ++t) {
    const uint32_t key = keys(t);
    if (key > thr) {
      out[pgt + min(peq, (int)need)] = (int64_t)t;
      ++pgt;
    } else if (key == thr) {
      if (peq < (int)need) out[pgt + peq] = (int64_t)t;
      ++peq;
    }
  }
}

__global__ __launch_bounds__(SEL_THREADS) void topk_sorted_kernel(const float* __restrict__ scores, int T, int k,
                                                                  int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_keys[];
  __shared__ SelShared sh;
  const int b = blockIdx.x;
  const float* s = scores + (size_t)b * T;
  if (T <= SEL_LDS_KEYS) {
    for (int t = threadIdx.x; t < T; t += SEL_THREADS) lds_keys[t] = order_key(s[t]);
    __syncthreads();
    const uint32_t* kp = lds_keys;
    select_topk_sorted([kp](int t) { return kp[t]; }, T, k, idx + (size_t)b * k, sh);
  } else {
    select_topk_sorted([s](int t) { return order_key(s[t]); }, T, k, idx + (size_t)b * k, sh);
  }
}

extern "C" int tspo_topk_sorted(const float* scores, int B, int T, int k, int64_t* idx, tspo_stream_t stream) {
  TSPO_REQUIRE(scores && idx, "topk_sorted: null pointer");
  TSPO_REQUIRE(B >= 0 && T >= 1 && k >= 1, "topk_sorted: bad dims B=%d T=%d k=%d", B, T, k);
  if (B == 0) return TSPO_OK;
  const int ke = k < T ? k : T;
  hipLaunchKernelGGL(topk_sorted_kernel, dim3(B), dim3(SEL_THREADS), T <= SEL_LDS_KEYS ? (size_t)T * 4 : 0,
                     (hipStream_t)stream, scores, T, ke, idx);
  return tspo::check_launch("topk_sorted");
}

// ---------------------------------------------------------------------------
// bin-max
__device__ __forceinline__ long long binmax_anchor(int j, int k, int T) {
  if (k == 1) return (long long)(T - 1);
  const double step = (double)(T - 1) / (double)(k - 1);  // model/utils.py:15
  return (long long)rint((double)j * step);               // python round(): half-to-even on the double
}

__global__ __launch_bounds__(256) void binmax_kernel(const float* __restrict__ scores, int T, int k,
                                                     int64_t* __restrict__ idx) {
  const int b = blockIdx.x;
  const float* s = scores + (size_t)b * T;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int j = wid; j < k; j += nw) {
    const long long a = binmax_anchor(j, k, T);
    const long long lo = (j == 0) ? 0 : ((binmax_anchor(j - 1, k, T) + a) / 2 + 1);
    const long long hi = (j == k - 1) ? (long long)(T - 1) : ((a + binmax_anchor(j + 1, k, T)) / 2);
    float best = -INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    bool nan_seen = false;
    for (long long t = lo + lane; t <= hi; t += 64) {
      const float v = s[t];
      const bool isn = v != v;
      // torch.argmax: NaN counts as the maximum, first occurrence wins
      if (isn) { if (!nan_seen) { nan_seen = true; bi = t; } }
      else if (!nan_seen && (v > best || bi == 0x7fffffffffffffffLL)) { best = v; bi = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const long long oi = __shfl_xor(bi, o, 64);
      const int on = __shfl_xor((int)nan_seen, o, 64);
      bool take;
      if (on != (int)nan_seen) take = on;                              // NaN side wins
      else if (nan_seen) take = oi < bi;                               // both NaN: first
      else take = (oi != 0x7fffffffffffffffLL) && (bi == 0x7fffffffffffffffLL || ob > best || (ob == best && oi < bi));
      if (take) { best = ob; bi = oi; nan_seen = on; }
    }
    if (lane == 0) idx[(size_t)b * k + j] = bi;
  }
}

extern "C" int tspo_binmax(const float* scores, int B, int T, int k, int64_t* idx, tspo_stream_t stream) {
  TSPO_REQUIRE(scores && idx, "binmax: null pointer");
  TSPO_REQUIRE(B >= 0 && T >= 1 && k >= 1, "binmax: bad dims B=%d T=%d k=%d", B, T, k);
  if (B == 0) return TSPO_OK;
  const int ke = k < T ? k : T;
  hipLaunchKernelGGL(binmax_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, T, ke, idx);
  return tspo::check_launch("binmax");
}

// ---------------------------------------------------------------------------
// Gumbel-top-k for all (prompt, rollout) pairs in one launch.  LONG (T > SEL_LDS_KEYS): the row does not fit in LDS, so the
// perturbed logit z(t) - a pure function of (logit, injected noise or the Philox counter) - is recomputed on each of the
// selection's passes, like the global-key path of topk_sorted_kernel; results are identical to the LDS-resident form.
template <bool LONG>
__global__ __launch_bounds__(SEL_THREADS) void gumbel_topk_kernel(
    const float* __restrict__ logits, const float* __restrict__ noise, uint32_t key0, uint32_t seed_hi, uint64_t offset,
    int per_offset, int G, int T, int k, float tau, int64_t* __restrict__ idx, float* __restrict__ logp,
    float* __restrict__ probs, float* __restrict__ noise_out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_keys[];  // T keys, then T floats (z) when probs wanted
  __shared__ SelShared sh;
  __shared__ float red[32];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* l = logits + (size_t)b * T;
  if (g == G) {   // the extra workgroup of prompt b: log(softmax(logits)) - no noise (model/utils.py:78).  Its own workgroup since
                  // round 5 (rollout 0's workgroup did it after its selection: the longest chain of the launch, 8.8 -> 7 us)
    float lmax = -INFINITY;
    for (int t = tid; t < T; t += SEL_THREADS) lmax = fmaxf(lmax, l[t]);
    lmax = block_max(lmax, red);
    float se = 0.f;
    for (int t = tid; t < T; t += SEL_THREADS) se += expf(l[t] - lmax);
    se = block_sum(se, red);
    for (int t = tid; t < T; t += SEL_THREADS) logp[(size_t)b * T + t] = logf(expf(l[t] - lmax) / se);
    return;
  }
  const size_t row = ((size_t)b * G + g) * T;
  float* zbuf = reinterpret_cast<float*>(lds_keys + (LONG ? 0 : T));
  // per_offset = p > 0: prompts come in groups of p that would have been separate calls (micro-steps of one optimizer step):
  // prompt b draws what prompt b % p of a call with offset + b / p draws
  const uint64_t off = per_offset > 0 ? offset + (uint64_t)(b / per_offset) : offset;
  const uint32_t cb = (uint32_t)(per_offset > 0 ? b % per_offset : b);
  const uint32_t key1 = seed_hi ^ (uint32_t)(off >> 32), off_lo = (uint32_t)off;
  auto gnoise = [&](int t) {
    return noise ? noise[row + t] : gumbel_from_bits(philox_x0((uint32_t)t, (uint32_t)g, cb, off_lo, key0, key1));
  };
  auto zf = [&](int t) { return (l[t] + gnoise(t)) / tau; };
  float zmax = -INFINITY;
  for (int t = tid; t < T; t += SEL_THREADS) {
    const float gn = gnoise(t);
    if (noise_out) noise_out[row + t] = gn;
    const float z = (l[t] + gn) / tau;
    if (!LONG) {
      lds_keys[t] = order_key(z);
      if (probs) zbuf[t] = z;
    }
    zmax = fmaxf(zmax, z);
  }
  __syncthreads();
  if (LONG) {
    select_topk_sorted([&](int t) { return order_key(zf(t)); }, T, k, idx + ((size_t)b * G + g) * k, sh);
  } else {
    const uint32_t* kp = lds_keys;
    select_topk_sorted([kp](int t) { return kp[t]; }, T, k, idx + ((size_t)b * G + g) * k, sh);
  }

  if (probs) {  // (one_hot - y) + y with y = softmax(z)  (model/utils.py:74-75)
    zmax = block_max(zmax, red);
    float se = 0.f;
    for (int t = tid; t < T; t += SEL_THREADS) se += expf((LONG ? zf(t) : zbuf[t]) - zmax);
    se = block_sum(se, red);
    __syncthreads();  // idx writes of this block visible to the block (global, same workgroup)
    const int64_t* my = idx + ((size_t)b * G + g) * k;
    for (int t = tid; t < T; t += SEL_THREADS) {
      const float y = expf((LONG ? zf(t) : zbuf[t]) - zmax) / se;
      int lo = 0, hi = k - 1;
      float oh = 0.f;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int64_t v = my[mid];
        if (v == t) { oh = 1.f; break; }
        if (v < t) lo = mid + 1; else hi = mid - 1;
      }
      probs[row + t] = (oh - y) + y;
    }
  }
}

extern "C" int tspo_gumbel_topk_ex(const float* logits, const float* noise, uint64_t seed, uint64_t offset, int B, int G,
                                   int T, int k, float tau, int64_t* idx, float* logp, float* probs, float* noise_out,
                                   tspo_stream_t stream, int prompts_per_offset) {
  TSPO_REQUIRE(logits && idx, "gumbel_topk: null pointer");
  TSPO_REQUIRE(B >= 0 && G >= 1 && T >= 1 && k >= 1, "gumbel_topk: bad dims B=%d G=%d T=%d k=%d", B, G, T, k);
  TSPO_REQUIRE(k <= T, "gumbel_topk: selected index k out of range (k=%d > T=%d)", k, T);
  TSPO_REQUIRE(tau > 0.f, "gumbel_topk: tau must be > 0");
  TSPO_REQUIRE(prompts_per_offset >= 0 && (prompts_per_offset == 0 || B % prompts_per_offset == 0),
               "gumbel_topk: prompts_per_offset=%d does not divide B=%d", prompts_per_offset, B);
  if (B == 0) return TSPO_OK;
  const uint32_t key0 = (uint32_t)(seed & 0xFFFFFFFFu), seed_hi = (uint32_t)(seed >> 32);
  if (T <= SEL_LDS_KEYS) {
    const size_t lds = (size_t)T * 4 * (probs ? 2 : 1);
    hipLaunchKernelGGL(gumbel_topk_kernel<false>, dim3(G + (logp ? 1 : 0), B), dim3(SEL_THREADS), lds, (hipStream_t)stream, logits, noise, key0,
                       seed_hi, offset, prompts_per_offset, G, T, k, tau, idx, logp, probs, noise_out);
  } else {
    hipLaunchKernelGGL(gumbel_topk_kernel<true>, dim3(G + (logp ? 1 : 0), B), dim3(SEL_THREADS), 0, (hipStream_t)stream, logits, noise, key0,
                       seed_hi, offset, prompts_per_offset, G, T, k, tau, idx, logp, probs, noise_out);
  }
  return tspo::check_launch("gumbel_topk");
}

extern "C" int tspo_gumbel_topk(const float* logits, const float* noise, uint64_t seed, uint64_t offset, int B, int G,
                                int T, int k, float tau, int64_t* idx, float* logp, float* probs, float* noise_out,
                                tspo_stream_t stream) {
  return tspo_gumbel_topk_ex(logits, noise, seed, offset, B, G, T, k, tau, idx, logp, probs, noise_out, stream, 0);
}

// ---------------------------------------------------------------------------
// group-relative advantage (one wave per prompt)
__global__ __launch_bounds__(64) void grpo_advantage_kernel(const float* __restrict__ r, int G, float eps,
                                                            float* __restrict__ adv) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* rb = r + (size_t)b * G;
  float s = 0.f;
  for (int g = lane; g < G; g += 64) s += rb[g];
  const float mean = wave_sum(s) / (float)G;
  float q = 0.f;
  for (int g = lane; g < G; g += 64) { const float d = rb[g] - mean; q += d * d; }
  const float var = wave_sum(q) / (float)(G - 1);  // unbiased (torch.std default); G==1 -> NaN like torch
  const float sd = sqrtf(var);
  for (int g = lane; g < G; g += 64) adv[(size_t)b * G + g] = (rb[g] - mean) / (sd + eps);
}

extern "C" int tspo_grpo_advantage(const float* rewards, int B, int G, float eps, float* adv, tspo_stream_t stream) {
  TSPO_REQUIRE(rewards && adv, "grpo_advantage: null pointer");
  TSPO_REQUIRE(B >= 0 && G >= 1, "grpo_advantage: bad dims B=%d G=%d", B, G);
  if (B == 0) return TSPO_OK;
  hipLaunchKernelGGL(grpo_advantage_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, rewards, G, eps, adv);
  return tspo::check_launch("grpo_advantage");
}

// ---------------------------------------------------------------------------
// closed-form policy-gradient w.r.t. the logits (deterministic: membership by
// binary search in each rollout's ascending index list, no atomics)
__global__ __launch_bounds__(256) void pg_grad_kernel(const float* __restrict__ logp, const int64_t* __restrict__ idx,
                                                      const float* __restrict__ adv, int G, int T, int k, float scale,
                                                      float* __restrict__ dlogits, float* __restrict__ loss) {
  extern __shared__ __attribute__((aligned(16))) int lds_idx[];  // G*k ints then G floats
  float* a = reinterpret_cast<float*>(lds_idx + (size_t)G * k);
  const int b = blockIdx.x, tid = threadIdx.x;   // blockIdx.y: 256-frame slice of the row
  for (int i = tid; i < G * k; i += 256) lds_idx[i] = (int)idx[(size_t)b * G * k + i];
  for (int g = tid; g < G; g += 256) a[g] = adv[(size_t)b * G + g];
  __syncthreads();
  float sumA = 0.f;
  for (int g = 0; g < G; ++g) sumA += a[g];
  const float invk = 1.f / (float)k, invG = 1.f / (float)G;
  const int t = blockIdx.y * 256 + tid;
  if (t < T) {
    const float p = expf(logp[(size_t)b * T + t]);
    float acc = 0.f;
    for (int g = 0; g < G; ++g) {
      const int* my = lds_idx + g * k;
      int lo = 0, hi = k - 1;
      bool hit = false;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int v = my[mid];
        if (v == t) { hit = true; break; }
        if (v < t) lo = mid + 1; else hi = mid - 1;
      }
      acc += a[g] * ((hit ? invk : 0.f) - p);
    }
    dlogits[(size_t)b * T + t] = -invG * acc * scale;
  }
  if (loss && tid == 0 && blockIdx.y == 0) loss[b] = -sumA * invG;
}

extern "C" int tspo_pg_grad_logits(const float* logp, const int64_t* idx, const float* adv, int B, int G, int T, int k,
                                   float scale, float* dlogits, float* loss, tspo_stream_t stream) {
  TSPO_REQUIRE(logp && idx && adv && dlogits, "pg_grad_logits: null pointer");
  TSPO_REQUIRE(B >= 0 && G >= 1 && T >= 1 && k >= 1 && k <= T, "pg_grad_logits: bad dims B=%d G=%d T=%d k=%d", B, G, T, k);
  const size_t lds = (size_t)G * k * 4 + (size_t)G * 4;
  TSPO_REQUIRE(lds <= 64 * 1024, "pg_grad_logits: G*k=%d too large for LDS", G * k);
  if (B == 0) return TSPO_OK;
  hipLaunchKernelGGL(pg_grad_kernel, dim3(B, (T + 255) / 256), dim3(256), lds, (hipStream_t)stream, logp, idx, adv, G, T, k, scale,
                     dlogits, loss);
  return tspo::check_launch("pg_grad_logits");
}

// The two steps above in one launch (what PolicyTrainer.backward issues): every block recomputes its prompt's G
// advantages (first wave, same reduction as grpo_advantage_kernel -> same bits) instead of reading them back.
__global__ __launch_bounds__(256) void grpo_pg_grad_kernel(const float* __restrict__ rewards, const float* __restrict__ logp,
                                                           const int64_t* __restrict__ idx, int G, int T, int k, float eps,
                                                           float scale, float* __restrict__ adv_out,
                                                           float* __restrict__ dlogits, float* __restrict__ loss) {
  extern __shared__ __attribute__((aligned(16))) int lds_idx[];  // G*k ints then G floats
  float* a = reinterpret_cast<float*>(lds_idx + (size_t)G * k);
  const int b = blockIdx.x, tid = threadIdx.x;   // blockIdx.y: 256-frame slice of the row
  for (int i = tid; i < G * k; i += 256) lds_idx[i] = (int)idx[(size_t)b * G * k + i];
  if (tid < 64) {
    const float* rb = rewards + (size_t)b * G;
    float s = 0.f;
    for (int g = tid; g < G; g += 64) s += rb[g];
    const float mean = wave_sum(s) / (float)G;
    float q = 0.f;
    for (int g = tid; g < G; g += 64) { const float d = rb[g] - mean; q += d * d; }
    const float sd = sqrtf(wave_sum(q) / (float)(G - 1));
    for (int g = tid; g < G; g += 64) {
      const float v = (rb[g] - mean) / (sd + eps);
      a[g] = v;
      if (blockIdx.y == 0) adv_out[(size_t)b * G + g] = v;
    }
  }
  __syncthreads();
  float sumA = 0.f;
  for (int g = 0; g < G; ++g) sumA += a[g];
  const float invk = 1.f / (float)k, invG = 1.f / (float)G;
  const int t = blockIdx.y * 256 + tid;
  if (t < T) {
    const float p = expf(logp[(size_t)b * T + t]);
    float acc = 0.f;
    for (int g = 0; g < G; ++g) {
      const int* my = lds_idx + g * k;
      int lo = 0, hi = k - 1;
      bool hit = false;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int v = my[mid];
        if (v == t) { hit = true; break; }
        if (v < t) lo = mid + 1; else hi = mid - 1;
      }
      acc += a[g] * ((hit ? invk : 0.f) - p);
    }
    dlogits[(size_t)b * T + t] = -invG * acc * scale;
  }
  if (loss && tid == 0 && blockIdx.y == 0) loss[b] = -sumA * invG;
}

extern "C" int tspo_grpo_pg_grad(const float* rewards, const float* logp, const int64_t* idx, int B, int G, int T, int k,
                                 float eps, float scale, float* adv, float* dlogits, float* loss, tspo_stream_t stream) {
  TSPO_REQUIRE(rewards && logp && idx && adv && dlogits, "grpo_pg_grad: null pointer");
  TSPO_REQUIRE(B >= 0 && G >= 1 && T >= 1 && k >= 1 && k <= T, "grpo_pg_grad: bad dims B=%d G=%d T=%d k=%d", B, G, T, k);
  const size_t lds = (size_t)G * k * 4 + (size_t)G * 4;
  TSPO_REQUIRE(lds <= 64 * 1024, "grpo_pg_grad: G*k=%d too large for LDS", G * k);
  if (B == 0) return TSPO_OK;
  hipLaunchKernelGGL(grpo_pg_grad_kernel, dim3(B, (T + 255) / 256), dim3(256), lds, (hipStream_t)stream, rewards, logp, idx, G,
                     T, k, eps, scale, adv, dlogits, loss);
  return tspo::check_launch("grpo_pg_grad");
}

// ---------------------------------------------------------------------------
// grad norm (two-stage, deterministic) and AdamW
#define NORM_BLOCKS 512
__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g, size_t n,
                                                            float* __restrict__ partial) {
  __shared__ float red[32];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = g[i];
    s += v * v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, int np, float pre_scale,
                                                         float max_norm, float* __restrict__ out2) {
  __shared__ float red[32];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(s);
    out2[0] = nrm;
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(1.f, max_norm / (nrm + 1e-6f));
    out2[1] = c * pre_scale;
  }
}

extern "C" int tspo_grad_norm_scale(const float* grad, size_t n, float pre_scale, float max_norm, float* out2,
                                    void* workspace, size_t workspace_bytes, tspo_stream_t stream) {
  TSPO_REQUIRE(grad && out2 && workspace, "grad_norm_scale: null pointer");
  if (workspace_bytes < NORM_BLOCKS * sizeof(float))
    return tspo::set_err(TSPO_EWORKSPACE, "grad_norm_scale: workspace %zu < %zu", workspace_bytes,
                         NORM_BLOCKS * sizeof(float));
  int nb = (int)((n + 255) / 256);
  if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, grad, n, (float*)workspace);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nb,
                     pre_scale, max_norm, out2);
  return tspo::check_launch("grad_norm_scale");
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float rsq_bc2,
                                                    float gscale, const float* __restrict__ d_gscale) {
  const float gs = gscale * (d_gscale ? d_gscale[1] : 1.f);
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rsq_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// AdamW that finishes the gradient-norm reduction itself: every block adds the <= NORM_BLOCKS partial sums of squares
// (same order as norm_final_kernel -> same bits), derives the clip coefficient and applies it; block 0 also publishes
// (norm, coefficient * pre_scale) for the caller's logging.
__global__ __launch_bounds__(256) void adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                         float b1, float b2, float eps, float wd, float bc1, float rsq_bc2,
                                                         const float* __restrict__ partial, int np, float pre_scale,
                                                         float max_norm, float* __restrict__ out2) {
  __shared__ float red[32];
  __shared__ float gs_sh;
  // this thread's first element group is requested BEFORE the norm reduction (it does not depend on it): the reduction's two
  // barriers and its serial tail used to run with nothing in flight (round 5; same arithmetic, same results)
  const size_t n4 = n >> 2;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool has0 = i0 < n4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 g0 = has0 ? reinterpret_cast<const f32x4*>(g)[i0] : z4, p0 = has0 ? reinterpret_cast<f32x4*>(p)[i0] : z4;
  const f32x4 m0 = has0 ? reinterpret_cast<f32x4*>(m)[i0] : z4, v0 = has0 ? reinterpret_cast<f32x4*>(v)[i0] : z4;
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(s);
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(1.f, max_norm / (nrm + 1e-6f));
    gs_sh = c * pre_scale;
    if (blockIdx.x == 0) { out2[0] = nrm; out2[1] = c * pre_scale; }
  }
  __syncthreads();
  const float gs = gs_sh;
  const float step_size = lr / bc1;
  auto upd = [&](size_t i, f32x4 g4, f32x4 p4, f32x4 m4, f32x4 v4) {
    g4 = g4 * gs;
    p4 = p4 * (1.f - lr * wd);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      m4[c] = b1 * m4[c] + (1.f - b1) * g4[c];
      v4[c] = b2 * v4[c] + (1.f - b2) * g4[c] * g4[c];
      p4[c] -= step_size * (m4[c] / (sqrtf(v4[c]) * rsq_bc2 + eps));
    }
    reinterpret_cast<f32x4*>(p)[i] = p4;
    reinterpret_cast<f32x4*>(m)[i] = m4;
    reinterpret_cast<f32x4*>(v)[i] = v4;
  };
  if (has0) upd(i0, g0, p0, m0, v0);
  for (size_t i = i0 + (size_t)gridDim.x * 256; i < n4; i += (size_t)gridDim.x * 256)
    upd(i, reinterpret_cast<const f32x4*>(g)[i], reinterpret_cast<f32x4*>(p)[i], reinterpret_cast<f32x4*>(m)[i],
        reinterpret_cast<f32x4*>(v)[i]);
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= step_size * (mi / (sqrtf(vi) * rsq_bc2 + eps));
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// 16-byte loads; same partial layout as sqsum_partial_kernel
__global__ __launch_bounds__(256) void sqsum_partial4_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float red[32];
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

static int adamw_clip_impl(float* param, const float* grad, float* m, float* v, size_t n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, float pre_scale, float max_norm,
                           float* out2, void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                           const float* norm_partials, int n_partials) {
  TSPO_REQUIRE(param && grad && m && v && out2 && (workspace || norm_partials), "adamw_clip_step: null pointer");
  TSPO_REQUIRE(!norm_partials || (n_partials >= 1 && n_partials <= 4 * NORM_BLOCKS), "adamw_clip_step_ex: n_partials=%d (1..%d)",
               n_partials, 4 * NORM_BLOCKS);
  TSPO_REQUIRE(step >= 1, "adamw_clip_step: step must be >= 1");
  TSPO_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "adamw_clip_step: buffers must be 16-byte aligned");
  if (!norm_partials && workspace_bytes < NORM_BLOCKS * sizeof(float))
    return tspo::set_err(TSPO_EWORKSPACE, "adamw_clip_step: workspace %zu < %zu", workspace_bytes,
                         NORM_BLOCKS * sizeof(float));
  if (n == 0) return TSPO_OK;
  int nb = (int)((n / 4 + 255) / 256);
  if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
  if (nb < 1) nb = 1;
  if (norm_partials) nb = n_partials;     // the producer of the gradient already left its partial sums of squares
  else hipLaunchKernelGGL(sqsum_partial4_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, grad, n, (float*)workspace);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  size_t ab = (n / 4 + 255) / 256;
  if (ab > 2048) ab = 2048;
  if (ab < 1) ab = 1;
  hipLaunchKernelGGL(adamw_clip_kernel, dim3((unsigned)ab), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)),
                     norm_partials ? norm_partials : (const float*)workspace, nb, pre_scale, max_norm, out2);
  return tspo::check_launch("adamw_clip_step");
}

extern "C" int tspo_adamw_clip_step(float* param, const float* grad, float* m, float* v, size_t n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, float pre_scale, float max_norm,
                                    float* out2, void* workspace, size_t workspace_bytes, tspo_stream_t stream) {
  return adamw_clip_impl(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, step, pre_scale, max_norm, out2, workspace,
                         workspace_bytes, stream, nullptr, 0);
}
extern "C" int tspo_adamw_clip_step_ex(float* param, const float* grad, float* m, float* v, size_t n, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, int step, float pre_scale, float max_norm,
                                       float* out2, const float* norm_partials, int n_partials, tspo_stream_t stream) {
  TSPO_REQUIRE(norm_partials, "adamw_clip_step_ex: null norm_partials");
  return adamw_clip_impl(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, step, pre_scale, max_norm, out2, nullptr, 0,
                         stream, norm_partials, n_partials);
}

extern "C" int tspo_adamw_step(float* param, const float* grad, float* m, float* v, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale,
                               const float* d_grad_scale, tspo_stream_t stream) {
  TSPO_REQUIRE(param && grad && m && v, "adamw_step: null pointer");
  TSPO_REQUIRE(step >= 1, "adamw_step: step must be >= 1");
  if (n == 0) return TSPO_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  size_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr,
                     beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, d_grad_scale);
  return tspo::check_launch("adamw_step");
}
