// optim.hip -- optimizer step of the SM3Det training loop for gfx950: global gradient-norm clipping + AdamW over ALL
// parameter tensors in two launches, with a per-tensor learning-rate / weight-decay vector.
//
// What it replaces: OptimizerHook.after_train_iter (mmcv/mmcv/runner/hooks/optimizer.py:55-73:
// clip_grad_norm_(max_norm=35) then optimizer.step()) on an AdamW built by DefaultOptimizerConstructor with ONE param
// group per parameter (mmcv/mmcv/runner/optimizer/default_constructor.py:180-227) because the paper's dynamic
// learning-rate adjustment writes a different lr into every group each step (mmrotate/core/hook/dynamic_lr.py:197-218,
// `assert len(param_group['params']) == 1` at :180).  That layout defeats torch's foreach/fused paths (~700 tiny
// launches); here the per-group lr is just an element of a device vector.
//
// Layout: a device table of tensors (param / grad / exp_avg / exp_avg_sq pointers, numel) and a table of fixed-size
// chunks (tensor id, chunk index) -- one workgroup per chunk, so launch shape is static (hipGraph friendly).  HBM-bound:
// 4 reads + 3 writes of 4 B per parameter.  The step counter and the clip coefficient live on the device: no host sync.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int OPT_CHUNK = 16384;  // elements per workgroup
constexpr int OPT_THREADS = 256;

struct TensorTable {
  const uint64_t* p;   // float* addresses
  const uint64_t* g;
  const uint64_t* m;
  const uint64_t* v;
  const int64_t* numel;
  const uint64_t* h;   // _Float16* fp16 shadows of the parameters (0 = none), or NULL: AMP operand copies, written here
};

// partial[c] = sum of squares of chunk c of the gradient of tensor chunk_tab[c].x
__global__ __launch_bounds__(OPT_THREADS) void grad_sumsq_kernel(TensorTable tt, const int32_t* __restrict__ chunk_tab,
                                                                float* __restrict__ partial) {
  const int tid = chunk_tab[2 * blockIdx.x], ck = chunk_tab[2 * blockIdx.x + 1];
  const float* __restrict__ g = reinterpret_cast<const float*>(tt.g[tid]);
  const int64_t n = tt.numel[tid];
  const int64_t base = (int64_t)ck * OPT_CHUNK;
  const int64_t end = min(n, base + OPT_CHUNK);
  float s = 0.f;
  const int64_t len = end - base;
  const bool vec = ((reinterpret_cast<uintptr_t>(g + base) & 15) == 0);  // 16-byte lanes when the chunk is aligned
  const int64_t nv = vec ? (len >> 2) : 0;
  const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(g + base);
  for (int64_t i = threadIdx.x; i < nv; i += OPT_THREADS) {
    const f32x4 x = g4[i];
    s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
  }
  for (int64_t i = base + 4 * nv + threadIdx.x; i < end; i += OPT_THREADS) {
    const float x = g[i];
    s += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float red[OPT_THREADS / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// norm = sqrt(sum partial); coef = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_ semantics);
// also advances the step counter (one thread).
// With a loss scaler (scaler = [scale, growth_tracker, found_inf], torch.cuda.amp.GradScaler semantics as used by the
// reference's Fp16OptimizerHook, mmcv/mmcv/runner/hooks/optimizer.py:283-300): the gradients carry the factor `scale`;
// a non-finite sum of squares = some gradient overflowed -> found_inf = 1, the AdamW kernel leaves everything untouched,
// the step counter does not advance, scale *= backoff, tracker = 0.  Otherwise the norm is unscaled, the coefficient
// folds 1/scale and the clip factor, tracker += 1 and every `growth_interval` clean steps scale *= growth.
// growth_interval <= 0: static scale (never updated).
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ partial, int nparts, float max_norm,
                                                        float* __restrict__ norm_out, float* __restrict__ coef_out,
                                                        float* __restrict__ step, float* __restrict__ scaler,
                                                        float growth, float backoff, int growth_interval) {
  __shared__ double red[1024 / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 1024) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 1024 / 64; i++) t += red[i];
    float inv_scale = 1.f;
    bool found_inf = false;
    if (scaler) {
      inv_scale = 1.f / scaler[0];
      found_inf = !isfinite(t);
    }
    const float norm = (float)sqrt(t) * inv_scale;
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      coef = coef < 1.f ? coef : 1.f;
    }
    coef *= inv_scale;
    if (norm_out) *norm_out = norm;
    if (scaler) {
      scaler[2] = found_inf ? 1.f : 0.f;
      if (growth_interval > 0) {
        if (found_inf) {
          scaler[0] *= backoff;
          scaler[1] = 0.f;
        } else {
          scaler[1] += 1.f;
          if (scaler[1] >= (float)growth_interval) {
            scaler[0] *= growth;
            scaler[1] = 0.f;
          }
        }
      }
    }
    *coef_out = found_inf ? 0.f : coef;
    if (step && !found_inf) *step += 1.f;
  }
}

__global__ void step_inc_kernel(float* step, float* coef) {
  *step += 1.f;
  if (coef) *coef = 1.f;
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad, maximize=False):
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(OPT_THREADS) void adamw_multi_kernel(TensorTable tt,
                                                                 const int32_t* __restrict__ chunk_tab,
                                                                 const float* __restrict__ lr,
                                                                 const float* __restrict__ wd, float beta1,
                                                                 float beta2, float eps,
                                                                 const float* __restrict__ step_p,
                                                                 const float* __restrict__ coef_p,
                                                                 const float* __restrict__ scaler) {
  if (scaler && scaler[2] != 0.f) return;  // a gradient overflowed under the loss scale: skip the whole step
  const int tid = chunk_tab[2 * blockIdx.x], ck = chunk_tab[2 * blockIdx.x + 1];
  float* __restrict__ p = reinterpret_cast<float*>(tt.p[tid]);
  const float* __restrict__ g = reinterpret_cast<const float*>(tt.g[tid]);
  float* __restrict__ m = reinterpret_cast<float*>(tt.m[tid]);
  float* __restrict__ v = reinterpret_cast<float*>(tt.v[tid]);
  _Float16* __restrict__ h = tt.h ? reinterpret_cast<_Float16*>(tt.h[tid]) : nullptr;
  const int64_t n = tt.numel[tid];
  const int64_t base = (int64_t)ck * OPT_CHUNK;
  const int64_t end = min(n, base + OPT_CHUNK);
  const float step = *step_p;
  const float coef = coef_p ? *coef_p : 1.f;
  const float l = lr[tid], w = wd[tid];
  const float bc1 = 1.f - powf(beta1, step);
  const float bc2 = 1.f - powf(beta2, step);
  const float step_size = l / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const float decay = 1.f - l * w;
  // 16-byte lanes over the aligned body of the chunk (tensors come 256-byte aligned from the allocator and chunks are
  // 64 KiB, so only views at odd offsets and the tail of the last chunk take the scalar loop below)
  const bool vec = (((reinterpret_cast<uintptr_t>(p + base) | reinterpret_cast<uintptr_t>(g + base) |
                      reinterpret_cast<uintptr_t>(m + base) | reinterpret_cast<uintptr_t>(v + base)) & 15) == 0);
  const int64_t nv = vec ? ((end - base) >> 2) : 0;
  f32x4* __restrict__ p4 = reinterpret_cast<f32x4*>(p + base);
  const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(g + base);
  f32x4* __restrict__ m4 = reinterpret_cast<f32x4*>(m + base);
  f32x4* __restrict__ v4 = reinterpret_cast<f32x4*>(v + base);
  for (int64_t i = threadIdx.x; i < nv; i += OPT_THREADS) {
    const f32x4 gq = g4[i], mq = m4[i], vq = v4[i], pq = p4[i];
    f32x4 mo, vo, po;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float gi = gq[e] * coef;
      const float mi = beta1 * mq[e] + (1.f - beta1) * gi;
      const float vi = beta2 * vq[e] + (1.f - beta2) * gi * gi;
      mo[e] = mi;
      vo[e] = vi;
      const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
      po[e] = pq[e] * decay - step_size * (mi / denom);
    }
    m4[i] = mo;
    v4[i] = vo;
    p4[i] = po;
    if (h) {  // the fp16 operand copy of the new value (round to nearest even, what sm3_cast_f32_f16 writes): 8 more bytes
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));  // (chunks are 16384 elements: 8-byte aligned halves)
      *reinterpret_cast<f16x4*>(h + base + 4 * i) = f16x4{(_Float16)po[0], (_Float16)po[1], (_Float16)po[2], (_Float16)po[3]};
    }
  }
  for (int64_t i = base + 4 * nv + threadIdx.x; i < end; i += OPT_THREADS) {
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    const float pn = p[i] * decay - step_size * (mi / denom);
    p[i] = pn;
    if (h) h[i] = (_Float16)pn;
  }
}

// DynamicLrUpdaterHook.after_train_iter (mmrotate/core/hook/dynamic_lr.py:192-217) + get_dynamic_lr (:107-175) as ONE launch
// on device-resident loss scalars: the reference reads every loss with .item() (11 host syncs per iteration), here the loss
// EMAs, the update counter and the iteration counter live in `state` (doubles: [0, n) EMA, [n] number of EMA updates,
// [n + 1] iteration index) and the per-tensor lr vector the AdamW launch reads is written directly.  Double precision as in
// the reference (python floats / float64 tensors).  One workgroup; thread 0 does the O(n) scalar work.
struct DlaCfg {
  int n, n_subnets, n_params;
  int head_policy;      // 0 normal (history / current), 1 reverse, 2 'None'
  int backbone_policy;  // 0 min, 1 avg, 2 max, 3 kl, 4 sigmoid_kl, 5 other (1.0)
  int warmup_iters;     // LrUpdaterHook.warmup_iters; > 0: warm-up for that many iterations; < 0: no warm-up, but the head-weight
                        // gate `history.steps < warmup_iters` (dynamic_lr.py:124) still reads |warmup_iters|
  float warmup_ratio, T, b, beta;
};
__global__ __launch_bounds__(256) void dla_lr_kernel(DlaCfg c, const float* __restrict__ losses,
                                                    const int32_t* __restrict__ loss_subnet,
                                                    const int32_t* __restrict__ param_subnet,
                                                    const float* __restrict__ base_lr, const float* __restrict__ sched,
                                                    double* __restrict__ state, float* __restrict__ lr) {
  __shared__ double s_w[64];  // per-subnet multiplier, [63] = shared (backbone / neck) multiplier
  __shared__ double s_all;    // warm-up: one factor for every tensor (< 0: not in warm-up)
  constexpr int MAXN = 32;
  if (threadIdx.x == 0) {
    const int n = c.n;
    double cur[MAXN], hist[MAXN], bw[MAXN];
    const double steps = state[n], it = state[n + 1];
    for (int i = 0; i < n; i++) {
      cur[i] = (double)losses[i];
      hist[i] = steps > 0 ? state[i] : 1e-3;  // EMA_meter.get()
    }
    s_all = -1.0;
    const double gate_iters = (double)(c.warmup_iters < 0 ? -c.warmup_iters : c.warmup_iters);
    if (c.warmup_iters > 0 && it < (double)c.warmup_iters) {
      // :203-216 -- warm-up: the EMAs keep updating; the lr is base * (1 - k) (mmcv LrUpdaterHook.get_warmup_lr).  The host
      // passes ratio 1 (k = 0: the lr stays the initial lr) for the reference's as-run behaviour, see sm3det_amd/optim.py
      const double k = (1.0 - it / c.warmup_iters) * (1.0 - (double)c.warmup_ratio);
      s_all = 1.0 - k;
    } else {
      if (steps < gate_iters || c.head_policy == 2) {
        for (int i = 0; i < n; i++) bw[i] = 1.0;
      } else {
        double mx = -1e300, sum = 0;
        for (int i = 0; i < n; i++) {
          bw[i] = (c.head_policy == 1 ? cur[i] / hist[i] : hist[i] / cur[i]) / (double)c.T;
          mx = fmax(mx, bw[i]);
        }
        for (int i = 0; i < n; i++) { bw[i] = exp(bw[i] - mx); sum += bw[i]; }
        for (int i = 0; i < n; i++) bw[i] = n * bw[i] / sum;
      }
      double vmin = 1e300, vmax = -1e300, vsum = 0;
      for (int sn = 0; sn < c.n_subnets; sn++) {
        double a = 0; int m = 0;
        for (int i = 0; i < n; i++) if (loss_subnet[i] == sn) { a += bw[i]; m++; }
        // a sub-network none of whose losses is present this step: the reference divides 0 by 0 (a python error); 1 here
        const double w = m ? a / m : 1.0;
        s_w[sn] = w;
        vmin = fmin(vmin, w); vmax = fmax(vmax, w); vsum += w;
      }
      double shared = 1.0;
      if (c.backbone_policy == 0) shared = vmin;
      else if (c.backbone_policy == 1) shared = vsum / c.n_subnets;
      else if (c.backbone_policy == 2) shared = vmax;
      else if (c.backbone_policy == 3 || c.backbone_policy == 4) {
        // F.kl_div(softmax(cur).log(), softmax(history), reduction='batchmean') of 1-D tensors: sum / n
        double mh = -1e300, mc = -1e300, sh = 0, sc = 0;
        for (int i = 0; i < n; i++) { mh = fmax(mh, hist[i]); mc = fmax(mc, cur[i]); }
        for (int i = 0; i < n; i++) { sh += exp(hist[i] - mh); sc += exp(cur[i] - mc); }
        double kl = 0;
        for (int i = 0; i < n; i++) {
          const double lh = hist[i] - mh - log(sh), lc = cur[i] - mc - log(sc);
          kl += exp(lh) * (lh - lc);
        }
        kl /= n;
        shared = c.backbone_policy == 3 ? 1.0 + (1.0 - kl) / sqrt((double)c.T)
                                        : 2.0 / (1.0 + exp(-((1.0 - kl - (double)c.b) * (double)c.T)));
      }
      s_w[63] = shared;
    }
    for (int i = 0; i < n; i++)  // EMA_meter.update, after the weights were taken from the OLD history
      state[i] = steps > 0 ? (1.0 - (double)c.beta) * state[i] + (double)c.beta * cur[i] : cur[i];
    state[n] = steps + 1.0;
    state[n + 1] = it + 1.0;
  }
  __syncthreads();
  const double sch = (double)sched[0];  // step-decay factor gamma^exp of get_lr (:92-105), maintained by the host
  for (int p = threadIdx.x; p < c.n_params; p += blockDim.x) {
    const int sn = param_subnet[p];
    const double mult = s_all >= 0.0 ? s_all : (sn >= 0 ? s_w[sn] : s_w[63]);
    lr[p] = (float)((double)base_lr[p] * sch * mult);  // (warm-up ends long before the first decay step: sch == 1 there)
  }
}

}  // namespace

extern "C" {

int sm3_optim_chunk_elems(void) { return OPT_CHUNK; }

int sm3_dla_lr(const float* losses, int n, const int32_t* loss_subnet, int n_subnets, const int32_t* param_subnet,
               const float* base_lr, int n_params, const float* sched, double* state, int head_policy,
               int backbone_policy, int warmup_iters, float warmup_ratio, float T, float b, float ema_beta, float* lr,
               sm3_stream_t stream) {
  if (!losses || !loss_subnet || !param_subnet || !base_lr || !sched || !state || !lr) return SM3_ERR_INVALID_ARG;
  if (n <= 0 || n > 32 || n_subnets <= 0 || n_subnets > 63 || n_params <= 0) return SM3_ERR_INVALID_ARG;
  if (head_policy < 0 || head_policy > 2 || backbone_policy < 0 || backbone_policy > 5) return SM3_ERR_INVALID_ARG;
  DlaCfg c{n, n_subnets, n_params, head_policy, backbone_policy, warmup_iters, warmup_ratio, T, b, ema_beta};
  dla_lr_kernel<<<1, 256, 0, (hipStream_t)stream>>>(c, losses, loss_subnet, param_subnet, base_lr, sched, state, lr);
  return launch_status();
}

int sm3_adamw_multi(const uint64_t* p_ptrs, const uint64_t* g_ptrs, const uint64_t* m_ptrs, const uint64_t* v_ptrs,
                    const uint64_t* h_ptrs, const int64_t* numel, const int32_t* chunk_tab, int n_chunks, const float* lr, const float* wd,
                    float beta1, float beta2, float eps, float max_grad_norm, float* step, float* clip_coef,
                    float* grad_norm, float* partials, float* scaler, float growth_factor, float backoff_factor,
                    int growth_interval, sm3_stream_t stream) {
  if (!p_ptrs || !g_ptrs || !m_ptrs || !v_ptrs || !numel || !chunk_tab || !lr || !wd || !step || !clip_coef)
    return SM3_ERR_INVALID_ARG;
  if (n_chunks <= 0) return SM3_OK;
  hipStream_t st = (hipStream_t)stream;
  TensorTable tt{p_ptrs, g_ptrs, m_ptrs, v_ptrs, numel, h_ptrs};
  if (max_grad_norm > 0.f || scaler) {  // the overflow check of the loss scaler needs the same pass as the clip norm
    if (!partials) return SM3_ERR_WORKSPACE;
    grad_sumsq_kernel<<<n_chunks, OPT_THREADS, 0, st>>>(tt, chunk_tab, partials);
    clip_coef_kernel<<<1, 1024, 0, st>>>(partials, n_chunks, max_grad_norm, grad_norm, clip_coef, step, scaler,
                                         growth_factor, backoff_factor, growth_interval);
  } else {
    step_inc_kernel<<<1, 1, 0, st>>>(step, clip_coef);
  }
  adamw_multi_kernel<<<n_chunks, OPT_THREADS, 0, st>>>(tt, chunk_tab, lr, wd, beta1, beta2, eps, step, clip_coef,
                                                       scaler);
  return launch_status();
}

}  // extern "C"
