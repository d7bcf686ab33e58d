// The training-side half of the flow models: one network evaluation per sample with a PER-SAMPLE
// time input, and the elementwise forward-process pieces around it.
//
//   frl_net_eval              self(xt, t, condition)        ddpm.py:249-254,286-293; continuous_ddpm.py:234-241; cnf.py:155-162
//   frl_add_noise             DDPM.add_noise                ddpm.py:164-171   (xt from a table gathered at integer t)
//                             ContinuousDDPM.add_noise      continuous_ddpm.py:107-115 (alpha, sigma = noise_schedule_func(t))
//                             CNF.linear_interpolation      cnf.py:85-90
//   frl_ddpm_onestep_update   DDPM.onestep_sample's update  ddpm.py:256-274   (per-sample t, num_samples repeats)
//   frl_mse_loss              ((pred - target) ** 2).mean() ddpm.py:294, cnf.py:163, continuous_ddpm.py:242 (+ its cotangent)
//
// In a sampler call every chain shares the step's time value, so the time branch collapses to a [T, H] table
// (csrc/prepare.hip).  Here t differs per sample: frl_time_rows_kernel evaluates the time branch per sample
// (embedding -> MLP_0 with mish after both layers, ddpm.py:83-88) and writes u[b] = [tfeat(t_b) | cond(state(b))]; the
// rows x | time | cond of Dense_0 are contiguous (ddpm.py:92), so one pass of frl_state_bias_kernel contracts u against
// kernel_0[A:] and the fused MFMA network kernel (MODE_EVAL, csrc/sampler_rs.hip / sampler_lat.hip) adds that row to its
// layer-0 accumulators per chain and contracts the x rows itself.
#include "frl_internal.h"

namespace frl {

__device__ __forceinline__ float mish_exact_ne(float x) {
  const double xd = (double)x;
  const double sp = (xd > 30.0) ? xd : log1p(exp(xd));
  return (float)(xd * tanh(sp));
}

struct TRArgs {
  const float* t;            // (B,)
  const float* cond;         // (n_states, C) or NULL
  const float* time_kernel;  // Fourier kernel (D/2, 1) or NULL
  const float* tk0; const float* tb0; const float* tk1; const float* tb1;   // MLP_0
  float* u;                  // (B, Dp + C)
  long long B;
  int particles, C, D, Dp, time_embed;
};

// 64 threads per sample, four samples per workgroup.  float32 in the reference's op order; transcendentals through double
// so that each float32 op is reproduced correctly rounded (as csrc/prepare.hip does for the table form).
__global__ void __launch_bounds__(256) frl_time_rows_kernel(const TRArgs a) {
  extern __shared__ float sh[];      // [4 samples][3][D]
  const int g = threadIdx.x >> 6, gt = threadIdx.x & 63;
  const long long b = (long long)blockIdx.x * 4 + g;
  const int D = a.D, W = a.Dp + a.C;
  float* temb = sh + (size_t)g * 3 * (D > 0 ? D : 1);
  float* h1 = temb + (D > 0 ? D : 1);
  const bool valid = b < a.B;
  const float tv = valid ? a.t[b] : 0.0f;
  float* urow = a.u + (valid ? b : 0) * W;
  if (a.time_embed == FRL_TEMB_NONE) {
    if (valid && gt == 0) urow[0] = tv;          // raw scalar time is input column A of Dense_0 (cnf.py:38-41)
  } else {
    const int half = D / 2;
    for (int d = gt; d < D; d += 64) {
      const int dd = (d < half) ? d : d - half;
      float arg;
      if (a.time_embed == FRL_TEMB_POSITIONAL) {
        const float lf = (float)log(10000.0) / (float)(half - 1);
        const float fr = (float)exp((double)((float)dd * -lf));
        arg = tv * fr;                                     // time_embedding.py:11-16
      } else {
        arg = (6.283185307179586f * tv) * a.time_kernel[dd];   // time_embedding.py:23-27
      }
      temb[d] = (d < half) ? (float)cos((double)arg) : (float)sin((double)arg);
    }
    __syncthreads();
    for (int d = gt; d < D; d += 64) {
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(temb[k], a.tk0[(size_t)k * D + d], acc);
      h1[d] = mish_exact_ne(acc + a.tb0[d]);
    }
    __syncthreads();
    for (int d = gt; d < D; d += 64) {
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(h1[k], a.tk1[(size_t)k * D + d], acc);
      if (valid) urow[d] = mish_exact_ne(acc + a.tb1[d]);
    }
  }
  if (valid && a.C > 0) {
    const float* crow = a.cond + ((a.particles == 1) ? b : b / a.particles) * a.C;
    for (int k = gt; k < a.C; k += 64) urow[a.Dp + k] = crow[k];
  }
}

int launch_time_rows(const FrlDesc& d, const FrlParams& p, const float* t, const float* cond, int C, float* u, long long B, hipStream_t stream) {
  if (B <= 0) return FRL_OK;
  TRArgs a;
  a.t = t; a.cond = cond; a.time_kernel = p.time_kernel;
  a.tk0 = p.tmlp_kernel[0]; a.tb0 = p.tmlp_bias[0]; a.tk1 = p.tmlp_kernel[1]; a.tb1 = p.tmlp_bias[1];
  a.u = u; a.B = B; a.particles = d.particles; a.C = C;
  a.D = d.time_embed == FRL_TEMB_NONE ? 0 : d.time_dim;
  a.Dp = d.time_embed == FRL_TEMB_NONE ? 1 : d.time_dim;
  a.time_embed = d.time_embed;
  const size_t lds = sizeof(float) * 4 * 3 * (size_t)(a.D > 0 ? a.D : 1);
  hipLaunchKernelGGL(frl_time_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("time rows launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- per-sample time-branch backward (the single-evaluation VJP) ---------------------------------------------------------
// dL/dtfeat[b] (the first D entries of du[b]) -> MLP_0 (mish, mish) and, for the Fourier embedding, its kernel.  Two stages
// like the cond_embedding backward: a per-sample chain rule (64 threads per sample, double precision: O(D^2) per sample) that
// leaves temb | h1 | dL/dz1 | dL/dz2 | 2 pi t gf in scratch, then one thread per gradient element adding the samples in index
// order (bitwise reproducible).
__device__ __forceinline__ void mish_grad_d(double x, double& h, double& d) {
  const double sp = (x > 30.0) ? x : log1p(exp(x));
  const double th = tanh(sp);
  const double sg = 1.0 / (1.0 + exp(-x));
  h = x * th;
  d = th + x * (1.0 - th * th) * sg;
}

__global__ void __launch_bounds__(256) frl_time_bwd_sample_kernel(const TimeBwdSampleArgs a) {
  extern __shared__ double shd[];     // [4 samples][4][D]: temb | h1 | dz2 | dz1
  const int g = threadIdx.x >> 6, gt = threadIdx.x & 63;
  const long long b = (long long)blockIdx.x * 4 + g;
  const int D = a.D, half = D / 2;
  double* temb = shd + (size_t)g * 4 * D;
  double* h1 = temb + D;
  double* dz2 = h1 + D;
  double* dz1 = dz2 + D;
  const bool valid = b < a.B;
  const float tv = valid ? a.t[b] : 0.0f;
  for (int d = gt; d < D; d += 64) {
    const int dd = (d < half) ? d : d - half;
    float arg;
    if (a.time_embed == FRL_TEMB_POSITIONAL) {
      const float lf = (float)log(10000.0) / (float)(half - 1);
      arg = tv * (float)exp((double)((float)dd * -lf));
    } else {
      arg = (6.283185307179586f * tv) * a.time_kernel[dd];
    }
    temb[d] = (d < half) ? cos((double)arg) : sin((double)arg);
  }
  __syncthreads();
  double a1[2] = {0.0, 0.0}, a2[2] = {0.0, 0.0};     // mish'(z1), mish'(z2) of this thread's features (D <= 128)
  for (int d = gt, i = 0; d < D; d += 64, ++i) {
    double z = a.tb0[d];
    for (int k = 0; k < D; ++k) z += temb[k] * (double)a.tk0[(size_t)k * D + d];
    double h, dh;
    mish_grad_d(z, h, dh);
    h1[d] = h; a1[i] = dh;
  }
  __syncthreads();
  for (int d = gt, i = 0; d < D; d += 64, ++i) {
    double z = a.tb1[d];
    for (int k = 0; k < D; ++k) z += h1[k] * (double)a.tk1[(size_t)k * D + d];
    double h, dh;
    mish_grad_d(z, h, dh);
    a2[i] = dh;
    dz2[d] = valid ? (double)a.du[(size_t)b * a.du_stride + d] * dh : 0.0;
  }
  __syncthreads();
  for (int d = gt, i = 0; d < D; d += 64, ++i) {
    double s = 0.0;
    for (int o = 0; o < D; ++o) s += (double)a.tk1[(size_t)d * D + o] * dz2[o];
    dz1[d] = s * a1[i];
  }
  __syncthreads();
  if (!valid) return;
  float* sc = a.scratch + (size_t)b * (4 * D + half);
  for (int d = gt; d < D; d += 64) {
    sc[d] = (float)temb[d]; sc[D + d] = (float)h1[d]; sc[2 * D + d] = (float)dz1[d]; sc[3 * D + d] = (float)dz2[d];
  }
  if (a.time_embed == FRL_TEMB_FOURIER) {
    for (int d = gt; d < half; d += 64) {
      double gc = 0.0, gs = 0.0;          // dL/dtemb[d] (cos part), dL/dtemb[half + d] (sin part) = W_t0 rows . dz1
      for (int o = 0; o < D; ++o) { gc += (double)a.tk0[(size_t)d * D + o] * dz1[o]; gs += (double)a.tk0[(size_t)(half + d) * D + o] * dz1[o]; }
      const double gf = -temb[half + d] * gc + temb[d] * gs;      // f = 2 pi t w: d cos f = -sin f, d sin f = cos f
      sc[4 * D + d] = (float)(6.283185307179586 * (double)tv * gf);
    }
  }
}

// Stage 1 (STAGE 0): thread (leaf element e, chunk c) adds the chunk's samples in index order; stage 2 (STAGE 1): thread e adds
// the chunk partials in index order and writes the leaf.  Both keep 16 loads in flight (one load per iteration left the loop
// at one L2 round trip per sample: 93 us at B = 256, and linear in B) and four interleaved accumulators; two calls give the
// same bits.
template <int STAGE>
__global__ void __launch_bounds__(256) frl_time_bwd_reduce_kernel(const TimeBwdSampleArgs a, long long chunk, long long nch) {
  const int D = a.D, half = D / 2, S = 4 * D + half;
  const long long n0 = (long long)D * D, total = 2 * n0 + 2 * D + (a.time_embed == FRL_TEMB_FOURIER ? half : 0);
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  double* part = (double*)(a.scratch + (((size_t)a.B * S + 1) & ~(size_t)1));      // [nch][total]
  int kind, i = 0, o;     // 0: dW0[i][o] = temb[i] dz1[o]   1: db0[o]   2: dW1[i][o] = h1[i] dz2[o]   3: db1[o]   4: d time kernel[o]
  long long r = e;
  if (r < n0) { kind = 0; i = (int)(r / D); o = (int)(r % D); }
  else if ((r -= n0) < D) { kind = 1; o = (int)r; }
  else if ((r -= D) < n0) { kind = 2; i = (int)(r / D); o = (int)(r % D); }
  else if ((r -= n0) < D) { kind = 3; o = (int)r; }
  else { kind = 4; o = (int)(r - D); }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (STAGE == 0) {
    // every sample contributes sc[oa] * sc[ob] (oa < 0: the bias / frequency leaves, a plain sum)
    const int oa = (kind == 0) ? i : (kind == 2) ? D + i : -1;
    const int ob = (kind <= 1) ? 2 * D + o : (kind <= 3) ? 3 * D + o : 4 * D + o;
    const long long c = blockIdx.y, s_end = (c + 1) * chunk < a.B ? (c + 1) * chunk : a.B;
    long long s = c * chunk;
    for (; s + 16 <= s_end; s += 16) {
      float va[16], vb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* sc = a.scratch + (size_t)(s + u) * S;
        vb[u] = sc[ob];
        va[u] = (oa >= 0) ? sc[oa] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] += (double)va[u] * (double)vb[u];
    }
    for (int u = 0; s < s_end; ++s, ++u) {
      const float* sc = a.scratch + (size_t)s * S;
      acc[u & 3] += (double)((oa >= 0) ? sc[oa] : 1.0f) * (double)sc[ob];
    }
    part[c * total + e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  } else {
    long long c = 0;
    for (; c + 16 <= nch; c += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[(c + u) * total + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
    }
    for (int u = 0; c < nch; ++c, ++u) acc[u & 3] += part[c * total + e];
    const float sum = (float)((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (kind == 0) a.g_tk0[(size_t)i * D + o] = sum;
    else if (kind == 1) a.g_tb0[o] = sum;
    else if (kind == 2) a.g_tk1[(size_t)i * D + o] = sum;
    else if (kind == 3) a.g_tb1[o] = sum;
    else a.g_time_kernel[o] = sum;
  }
}

int launch_time_bwd_samples(const TimeBwdSampleArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.time_embed == FRL_TEMB_NONE) return FRL_OK;
  if (a.D > 128 || a.D < 4) { set_error("single-evaluation VJP: time_dim %d outside [4, 128]", a.D); return FRL_EUNSUPPORTED; }
  const size_t lds = sizeof(double) * 4 * 4 * (size_t)a.D;
  hipLaunchKernelGGL(frl_time_bwd_sample_kernel, dim3((unsigned)((a.B + 3) / 4)), dim3(256), lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("time bwd (samples) launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  const long long total = 2LL * a.D * a.D + 2 * a.D + (a.time_embed == FRL_TEMB_FOURIER ? a.D / 2 : 0);
  const long long chunk = time_bwd_samples_chunk(a.B), nch = (a.B + chunk - 1) / chunk;
  hipLaunchKernelGGL(frl_time_bwd_reduce_kernel<0>, dim3((unsigned)((total + 255) / 256), (unsigned)nch), dim3(256), 0, stream, a, chunk, nch);
  hipLaunchKernelGGL(frl_time_bwd_reduce_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, chunk, nch);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("time bwd (reduce) launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- forward process -----------------------------------------------------------------------------------------------
// table[t] as jnp indexes it (ddpm.py:169,257-259 index the (steps + 1,) tables with the caller's per-sample t): a negative
// index wraps once, numpy style, and what is still out of range is clamped — never an out-of-bounds read
__device__ __forceinline__ int jnp_index(int t, int n) {
  if (t < 0) t += n;
  return t < 0 ? 0 : (t >= n ? n - 1 : t);
}

__global__ void __launch_bounds__(256)
frl_add_noise_kernel(int kind, long long B, int A, int steps, const float* __restrict__ x0, const float* __restrict__ e, const void* __restrict__ tv,
                     const float* __restrict__ alpha_hats, float p0, float p1, float* __restrict__ xt, float* __restrict__ vel) {
  const long long n = B * A;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / A;
    if (kind == FRL_NOISE_DDPM) {
      const float ah = alpha_hats[jnp_index(((const int32_t*)tv)[b], steps + 1)];
      xt[i] = sqrtf(ah) * x0[i] + sqrtf(1.0f - ah) * e[i];                         // ddpm.py:170
    } else if (kind == FRL_NOISE_FLOW) {
      const float t = ((const float*)tv)[b];
      xt[i] = (1.0f - t) * x0[i] + t * e[i];                                        // cnf.py:88 (e = x1)
      if (vel != nullptr) vel[i] = e[i] - x0[i];                                    // cnf.py:89
    } else {
      const float t = ((const float*)tv)[b];
      float alpha;
      if (kind == FRL_NOISE_CDDPM_LINEAR) {                                          // continuous_ddpm.py:20-23, beta0 = p0, beta1 = p1
        const float la = -((p1 - p0) / 4.0f) * (t * t) - (p0 / 2.0f) * t;
        alpha = (float)exp((double)la);
      } else {                                                                       // continuous_ddpm.py:27-30, s = p0
        const float hp = 1.5707963267948966f;
        const float tc = fminf(fmaxf(t, 0.0f), 0.9946f);
        const float num = (float)cos((double)(hp * (tc + p0) / (1.0f + p0)));
        const float den = (float)cos((double)(hp * p0 / (1.0f + p0)));
        alpha = num / den;
      }
      const float sigma = sqrtf(1.0f - alpha * alpha);
      xt[i] = alpha * x0[i] + sigma * e[i];                                         // continuous_ddpm.py:114
    }
  }
}

int launch_add_noise(int kind, long long B, int A, int steps, const float* x0, const float* e, const void* t, const float* alpha_hats, float p0,
                     float p1, float* xt, float* vel, hipStream_t stream) {
  if (B <= 0) return FRL_OK;
  const long long n = B * A, want = (n + 255) / 256, cap = 16ll * device_cus();
  hipLaunchKernelGGL(frl_add_noise_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, stream, kind, B, A, steps, x0, e, t, alpha_hats, p0, p1, xt, vel);
  hipError_t er = hipGetLastError();
  if (er != hipSuccess) { set_error("add_noise launch: %s", hipGetErrorString(er)); return FRL_EHIP; }
  return FRL_OK;
}

// DDPM.onestep_sample's update for per-sample t (ddpm.py:256-274): xt_1 from (xt, eps_theta, t), repeated num_samples times,
// ddpm: + (t > 1) sqrt(1 - alphas[t]) noise (note: sqrt(beta_t), not the posterior variance the sampler uses); ddim: clip.
__global__ void __launch_bounds__(256)
frl_onestep_update_kernel(int solver, long long B, int A, int N, int steps, const float* __restrict__ xt, const float* __restrict__ eps,
                          const int32_t* __restrict__ tt, const float* __restrict__ alphas, const float* __restrict__ alpha_hats, int clip,
                          float x_min, float x_max, const float* __restrict__ noise, float* __restrict__ out) {
  const long long n = B * N * A;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ai = (int)(i % A);
    const long long b = i / ((long long)N * A);
    const int t = tt[b];
    const float x = xt[b * A + ai], e = eps[b * A + ai];
    const float ah = alpha_hats[jnp_index(t, steps + 1)], ah1 = alpha_hats[jnp_index(t - 1, steps + 1)], al = alphas[jnp_index(t, steps + 1)];
    float r;
    if (solver == FRL_SOLVER_DDPM) {
      float x0h = 1.0f / sqrtf(ah) * (x - sqrtf(1.0f - ah) * e);
      if (clip) x0h = fminf(fmaxf(x0h, x_min), x_max);
      r = 1.0f / (1.0f - ah) * (sqrtf(ah1) * (1.0f - al) * x0h + sqrtf(al) * (1.0f - ah1) * x);
      r += (t > 1 ? 1.0f : 0.0f) * sqrtf(1.0f - al) * noise[i];
    } else {
      const float a1 = 1.0f / sqrtf(al), a2 = sqrtf(1.0f - ah), a3 = sqrtf(1.0f - ah1);
      r = a1 * (x - a2 * e) + a3 * e;
      if (clip) r = fminf(fmaxf(r, x_min), x_max);
    }
    out[i] = r;
  }
}

int launch_onestep_update(int solver, long long B, int A, int N, int steps, const float* xt, const float* eps, const int32_t* t, const float* alphas,
                          const float* alpha_hats, int clip, float x_min, float x_max, const float* noise, float* out, hipStream_t stream) {
  if (B <= 0) return FRL_OK;
  const long long n = B * N * A, want = (n + 255) / 256, cap = 16ll * device_cus();
  hipLaunchKernelGGL(frl_onestep_update_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, stream, solver, B, A, N, steps, xt, eps, t,
                     alphas, alpha_hats, clip, x_min, x_max, noise, out);
  hipError_t er = hipGetLastError();
  if (er != hipSuccess) { set_error("onestep update launch: %s", hipGetErrorString(er)); return FRL_EHIP; }
  return FRL_OK;
}

// ((pred - target) ** 2).mean(), deterministic: MSE_BLOCKS workgroups each add a contiguous range into one partial (tree
// inside the workgroup), the last stage adds the partials in index order; optionally the cotangent 2 (pred - target) / n.
constexpr int MSE_BLOCKS = 256;

__global__ void __launch_bounds__(256)
frl_mse_partial_kernel(long long n, const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ grad, double* __restrict__ part) {
  __shared__ double red[256];
  const long long per = (n + gridDim.x - 1) / gridDim.x, i0 = per * blockIdx.x, i1 = (i0 + per < n) ? i0 + per : n;
  double s = 0.0;
  const float inv = 2.0f / (float)n;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = pred[i] - target[i];
    s += (double)(d * d);
    if (grad != nullptr) grad[i] = inv * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void frl_mse_final_kernel(long long n, int nb, const double* __restrict__ part, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nb; ++i) s += part[i];
  *loss = (float)(s / (double)n);
}

int launch_mse(long long n, const float* pred, const float* target, float* loss, float* grad, void* ws, hipStream_t stream) {
  const int nb = (int)((n + 255) / 256 < MSE_BLOCKS ? (n + 255) / 256 : MSE_BLOCKS);
  hipLaunchKernelGGL(frl_mse_partial_kernel, dim3(nb), dim3(256), 0, stream, n, pred, target, grad, (double*)ws);
  hipError_t er = hipGetLastError();
  if (er == hipSuccess) { hipLaunchKernelGGL(frl_mse_final_kernel, dim3(1), dim3(64), 0, stream, n, nb, (const double*)ws, loss); er = hipGetLastError(); }
  if (er != hipSuccess) { set_error("mse launch: %s", hipGetErrorString(er)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
