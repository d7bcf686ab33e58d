// VJP of the batch-invariant time branch (DDPMBackbone / FlowBackbone, flowrl/flow/ddpm.py:83-91, cnf.py:30-41):
//     temb = time_embedding(t)                     time_embedding.py:6-27
//     tfeat = mish(Dense_1(mish(Dense_0(temb))))   MLP(hidden_dims=[D, D], activation=mish), mlp.py:25-33
//     tb[j] = bias_0 + tfeat[j] @ kernel_0[A : A + D']        (the sampler kernels' hoisted layer-0 bias table)
// given dL/dtb [NE][H] from the weight-gradient kernel.  O(T (D^2 + D H)) work: one workgroup, double precision (what
// jax.grad computes here in float32 is reproduced to float32 rounding).  Writes dL/d MLP_0 leaves, dL/d
// time_embedding/kernel, dL/d Dense_0 bias and the time rows [A, A + D') of dL/d Dense_0 kernel.
#include "frl_internal.h"

namespace frl {

__device__ __forceinline__ double mish_d(double x) {
  const double sp = (x > 30.0) ? x : log1p(exp(x));
  return x * tanh(sp);
}
__device__ __forceinline__ double mish_grad_d(double x) {
  const double sp = (x > 30.0) ? x : log1p(exp(x));
  const double t = tanh(sp);
  const double sg = 1.0 / (1.0 + exp(-x));
  return t + x * (1.0 - t * t) * sg;
}

constexpr int TBW_PART = 4 * 1280;      // doubles
__host__ __device__ inline bool tbw_stage_w0(int D, int H) { return (size_t)TBW_PART * 8 + (size_t)D * (H + 1) * 4 <= 150 * 1024; }

__global__ void __launch_bounds__(1024) frl_time_bwd_kernel(const TBArgs a) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int T = a.T, D = a.D, H = a.H, A = a.A, half = a.D / 2;
#ifdef FRL_TBW_PROFILE
  long long tk[12]; int nk = 0;
#define TICK() do { __syncthreads(); tk[nk++] = wall_clock64(); } while (0)
#else
#define TICK() do { } while (0)
#endif
  TICK();
  const float* W0 = a.p.kernel[0];
  // LDS: s_part [TBW_PART doubles] (h2 copy for the Dense_0 time rows, then the four-segment partial dots) | s_w0 [D][H + 1]
  // floats (the time rows of Dense_0, staged once so that the per-(j, d) dots read them conflict-free) when it fits
  extern __shared__ __attribute__((aligned(16))) double s_part[];
  float* s_w0 = (float*)(s_part + TBW_PART);
  const bool small = T * D <= TBW_PART / 4 && (H & 3) == 0;      // the LDS-backed forms below; else the plain loops
  const bool stage_w0 = small && nt % H == 0 && tbw_stage_w0(D, H);
  double* tv = a.scratch;                 // [T]
  // ---- time input of iteration j, as frl_prepare_kernel derives it ----
  for (int j = tid; j < T; j += nt) {
    double v;
    if (a.kind == FRL_KIND_CNF) v = (double)((float)j / (float)T);
    else if (a.kind == FRL_KIND_DDPM) v = (double)(T - j);
    else v = (double)a.s.ts[T - j];
    tv[j] = v;
  }
  __syncthreads();
  // dL/d bias_0 = sum over steps of dL/dtb
  for (int f = tid; f < H; f += nt) {
    double s = 0.0;
    for (int j = 0; j < T; ++j) s += (double)a.grad_tb[j * H + f];
    a.g_bias0[f] = (float)s;
  }
  TICK();
  if (a.time_embed == FRL_TEMB_NONE) {
    // raw scalar time is input row A of Dense_0 (cnf.py:38-41): dL/dW0[A][f] = sum_j t_j dL/dtb[j][f]
    for (int f = tid; f < H; f += nt) {
      double s = 0.0;
      for (int j = 0; j < T; ++j) s += tv[j] * (double)a.grad_tb[j * H + f];
      a.g_kernel0[(size_t)A * H + f] = (float)s;
    }
    return;
  }
  double* temb = tv + T;                  // [T][D] each
  double* a1 = temb + (size_t)T * D;
  double* h1 = a1 + (size_t)T * D;
  double* a2 = h1 + (size_t)T * D;
  double* h2 = a2 + (size_t)T * D;
  double* g2 = h2 + (size_t)T * D;        // dL/da2, then dL/da1 in g1
  double* g1 = g2 + (size_t)T * D;
  // ---- forward, kept ----
  for (int idx = tid; idx < T * D; idx += nt) {
    const int j = idx / D, d = idx - j * D, dd = (d < half) ? d : d - half;
    double arg;
    if (a.time_embed == FRL_TEMB_POSITIONAL) {
      const float lf = (float)log(10000.0) / (float)(half - 1);       // the float32 frequencies of time_embedding.py:12-14
      arg = tv[j] * (double)(float)exp((double)((float)dd * -lf));
    } else {
      arg = 6.283185307179586 * tv[j] * (double)a.p.time_kernel[dd];
    }
    temb[idx] = (d < half) ? cos(arg) : sin(arg);
  }
  __syncthreads();
  TICK();
  for (int idx = tid; idx < T * D; idx += nt) {
    const int j = idx / D, d = idx - j * D;
    double acc = (double)a.p.tmlp_bias[0][d];
#pragma unroll 16
    for (int k = 0; k < D; ++k) acc += temb[j * D + k] * (double)a.p.tmlp_kernel[0][(size_t)k * D + d];
    a1[idx] = acc;
  }
  TICK();
  for (int idx = tid; idx < T * D; idx += nt) h1[idx] = mish_d(a1[idx]);      // (own loop: see h2 below)
  __syncthreads();
  TICK();
  for (int idx = tid; idx < T * D; idx += nt) {
    const int j = idx / D, d = idx - j * D;
    double acc = (double)a.p.tmlp_bias[1][d];
#pragma unroll 16
    for (int k = 0; k < D; ++k) acc += h1[j * D + k] * (double)a.p.tmlp_kernel[1][(size_t)k * D + d];
    a2[idx] = acc;
  }
  for (int idx = tid; idx < T * D; idx += nt) h2[idx] = mish_d(a2[idx]);      // (own loop: beside the transcendental's code the dot's loads are not batched)
  __syncthreads();
  TICK();
  // ---- backward ----
  // time rows of Dense_0: dL/dW0[A + d][f] = sum_j tfeat[j][d] dL/dtb[j][f]
  if (small && nt % H == 0) {
    // thread = (feature f, row group): dL/dtb[j][f] is loaded once per step and meets 16 rows of tfeat (LDS) in registers
    for (int idx = tid; idx < T * D; idx += nt) s_part[idx] = h2[idx];
    const int f = tid % H, gi = tid / H, ng = nt / H;
    if (stage_w0) {
      for (int i0 = 0; gi + ng * i0 < D; i0 += 16) {      // 16 independent loads in flight, then the LDS stores
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int d = gi + ng * (i0 + u); w[u] = (d < D) ? W0[(size_t)(A + d) * H + f] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int d = gi + ng * (i0 + u); if (d < D) s_w0[d * (H + 1) + f] = w[u]; }
      }
    }
    __syncthreads();
    TICK();
    for (int i0 = 0; gi + ng * i0 < D; i0 += 16) {
      double acc[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = 0.0;
#pragma unroll 4
      for (int j = 0; j < T; ++j) {
        const double g = (double)a.grad_tb[j * H + f];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int d = gi + ng * (i0 + u); if (d < D) acc[u] += s_part[j * D + d] * g; }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int d = gi + ng * (i0 + u); if (d < D) a.g_kernel0[(size_t)(A + d) * H + f] = (float)acc[u]; }
    }
    __syncthreads();      // s_part is reused below
  } else {
    for (int idx = tid; idx < D * H; idx += nt) {
      const int d = idx / H, f = idx - d * H;
      double s = 0.0;
#pragma unroll 4
      for (int j = 0; j < T; ++j) s += h2[j * D + d] * (double)a.grad_tb[j * H + f];
      a.g_kernel0[(size_t)(A + d) * H + f] = (float)s;
    }
  }
  TICK();
  // dL/da2[j][d] = mish'(a2) * sum_f W0[A + d][f] dL/dtb[j][f]
  if (small) {
    // every (j, d) dot over the H features is cut in four segments, one thread each (4 T D threads at work, not T D)
    for (int i = tid; i < 4 * T * D; i += nt) {
      const int seg = i / (T * D), idx = i - seg * (T * D), j = idx / D, d = idx - j * D, f0 = seg * (H / 4);
      double s = 0.0;
      if (stage_w0) {
#pragma unroll 16
        for (int f = f0; f < f0 + H / 4; ++f) s += (double)s_w0[d * (H + 1) + f] * (double)a.grad_tb[j * H + f];
      } else {
#pragma unroll 16
        for (int f = f0; f < f0 + H / 4; ++f) s += (double)W0[(size_t)(A + d) * H + f] * (double)a.grad_tb[j * H + f];
      }
      s_part[i] = s;
    }
    __syncthreads();
    for (int idx = tid; idx < T * D; idx += nt)
      g2[idx] = ((s_part[idx] + s_part[T * D + idx]) + (s_part[2 * T * D + idx] + s_part[3 * T * D + idx])) * mish_grad_d(a2[idx]);
  } else {
    for (int idx = tid; idx < T * D; idx += nt) {
      const int j = idx / D, d = idx - j * D;
      double s = 0.0;
#pragma unroll 16
      for (int f = 0; f < H; ++f) s += (double)W0[(size_t)(A + d) * H + f] * (double)a.grad_tb[j * H + f];
      g2[idx] = s * mish_grad_d(a2[idx]);
    }
  }
  __syncthreads();
  TICK();
  for (int idx = tid; idx < D * D; idx += nt) {          // MLP_0/Dense_1 kernel [k][d]
    const int k = idx / D, d = idx - k * D;
    double s = 0.0;
#pragma unroll 4
    for (int j = 0; j < T; ++j) s += h1[j * D + k] * g2[j * D + d];
    a.g_tmlp_kernel[1][idx] = (float)s;
  }
  for (int d = tid; d < D; d += nt) {
    double s = 0.0;
    for (int j = 0; j < T; ++j) s += g2[j * D + d];
    a.g_tmlp_bias[1][d] = (float)s;
  }
  TICK();
  for (int idx = tid; idx < T * D; idx += nt) {          // dL/da1[j][k] = mish'(a1) * sum_d K1[k][d] dL/da2[j][d]
    const int j = idx / D, k = idx - j * D;
    double s = 0.0;
#pragma unroll 16
    for (int d = 0; d < D; ++d) s += (double)a.p.tmlp_kernel[1][(size_t)k * D + d] * g2[j * D + d];
    g1[idx] = s * mish_grad_d(a1[idx]);
  }
  __syncthreads();
  TICK();
  for (int idx = tid; idx < D * D; idx += nt) {          // MLP_0/Dense_0 kernel [k][d]
    const int k = idx / D, d = idx - k * D;
    double s = 0.0;
#pragma unroll 4
    for (int j = 0; j < T; ++j) s += temb[j * D + k] * g1[j * D + d];
    a.g_tmlp_kernel[0][idx] = (float)s;
  }
  for (int d = tid; d < D; d += nt) {
    double s = 0.0;
    for (int j = 0; j < T; ++j) s += g1[j * D + d];
    a.g_tmlp_bias[0][d] = (float)s;
  }
  TICK();
  if (a.time_embed == FRL_TEMB_FOURIER && a.g_time_kernel != nullptr) {
    // temb = [cos f, sin f], f = 2 pi t w (time_embedding.py:23-27): dL/dw[dd] = sum_j 2 pi t_j (-sin f gtemb_cos + cos f gtemb_sin)
    if (small) {
      // one thread per (step, frequency): the two D-long dots with MLP_0/Dense_0's rows, then a fixed-order sum over the steps
      __syncthreads();      // s_part's previous readers are done
      for (int i = tid; i < T * half; i += nt) {
        const int j = i / half, dd = i - j * half;
        double gc = 0.0, gs = 0.0;                           // dL/dtemb[j][dd], dL/dtemb[j][half + dd] = K0 dL/da1
#pragma unroll 16
        for (int d = 0; d < D; ++d) {
          gc += (double)a.p.tmlp_kernel[0][(size_t)dd * D + d] * g1[j * D + d];
          gs += (double)a.p.tmlp_kernel[0][(size_t)(half + dd) * D + d] * g1[j * D + d];
        }
        s_part[i] = 6.283185307179586 * tv[j] * (-temb[j * D + half + dd] * gc + temb[j * D + dd] * gs);
      }
      __syncthreads();
      for (int dd = tid; dd < half; dd += nt) {
        double s = 0.0;
        for (int j = 0; j < T; ++j) s += s_part[j * half + dd];
        a.g_time_kernel[dd] = (float)s;
      }
    } else {
    for (int dd = tid; dd < half; dd += nt) {
      double s = 0.0;
      for (int j = 0; j < T; ++j) {
        double gc = 0.0, gs = 0.0;                           // dL/dtemb[j][dd], dL/dtemb[j][half + dd] = K0 dL/da1
        for (int d = 0; d < D; ++d) {
          gc += (double)a.p.tmlp_kernel[0][(size_t)dd * D + d] * g1[j * D + d];
          gs += (double)a.p.tmlp_kernel[0][(size_t)(half + dd) * D + d] * g1[j * D + d];
        }
        // sin f = temb[half + dd], cos f = temb[dd]
        s += 6.283185307179586 * tv[j] * (-temb[j * D + half + dd] * gc + temb[j * D + dd] * gs);
      }
      a.g_time_kernel[dd] = (float)s;
    }
    }
  }
#ifdef FRL_TBW_PROFILE
  TICK();
  if (tid == 0) { printf("tbw ticks (100 MHz):"); for (int i = 1; i < nk; ++i) printf(" %lld", tk[i] - tk[i - 1]); printf("\n"); }
#endif
}

int launch_time_bwd(const TBArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)TBW_PART * 8 + (tbw_stage_w0(a.D, a.H) ? (size_t)a.D * (a.H + 1) * 4 : 0);
  hipError_t e0 = hipFuncSetAttribute((const void*)frl_time_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e0 != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e0)); return FRL_EHIP; }
  hipLaunchKernelGGL(frl_time_bwd_kernel, dim3(1), dim3(1024), lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("time-branch bwd launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
