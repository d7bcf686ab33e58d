// Layer-by-layer single-evaluation VJP: the parameter gradient of the eps-MSE / flow-matching losses for every predictor
// the fused 256-wide sweep (csrc/sampler_bwd.hip) does not carry — 512-wide MLPs (FQL's actor_bc: flowrl/flow/cnf.py:145-170
// called from flowrl/agent/offline/fql/fql.py:83-123 on the network of fql.py:189-211), LayerNorm predictors in both orders
// (flowrl/module/mlp.py:26-29, flowrl/agent/offline/fql/network.py:26-30), ResidualMLP (mlp.py:37-98; DAC with
// diffusion.resnet, flowrl/flow/ddpm.py:280-304 from flowrl/agent/offline/dac.py:200-207) and Simba (simba.py:10-63).
//
// What jax.grad computes through `model.apply({"params": p}, xt, t, cond)` is restated as a sequence of launches over
// (B, width) f32 row-major activations kept in the workspace:
//   frl_lgemm_kernel     C = op(A) op(B) [+ bias row] [+ addend] on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): Dense forward
//                        (h W), input gradient (dz W^T) and weight gradient (X^T dz: contraction over the batch, optional
//                        split over the batch with a fixed-order second stage) are three stride patterns of one kernel
//   frl_lrow_fwd_kernel  one wave per sample: [LayerNorm] -> act -> [LayerNorm] (flax LayerNorm, eps 1e-6, variance about the mean)
//   frl_lrow_bwd_kernel  its derivative (LayerNorm statistics recomputed from the saved pre-activations), optionally added to
//                        a residual stream, and the per-sample products the LayerNorm scale / bias gradients sum
//   frl_lcolsum_kernel   column sums over the batch in two fixed-order stages (bias and LayerNorm gradients)
// No float atomics: two calls give the same bits.  Launch-bound at training batch sizes (a few dozen small launches per
// call); the 256-wide plain MLPs keep the fused path.
#include <string.h>

#include "frl_device.h"
#include "frl_actgrad.h"

namespace frl {

// ---- GEMM -----------------------------------------------------------------------------------------------------------
struct LGemm {
  const float* A; const float* B; float* C;
  const float* bias;        // (N,) added to every row, or NULL
  const float* add;         // (M, N) row-major with leading dimension ldadd, added elementwise, or NULL
  int M, N, K;
  long long sam, sak;       // element (m, k) of op(A) sits at A[m * sam + k * sak]
  long long sbk, sbn;       // element (k, n) of op(B) sits at B[k * sbk + n * sbn]
  int ldc, ldadd;
  int splits;               // > 1: split s owns k in [s K / splits, (s + 1) K / splits) and writes part[s][M][N]
  float* part;
};

constexpr int LG_TM = 64, LG_TN = 64, LG_TK = 16, LG_LD = 65;

__global__ void __launch_bounds__(256) frl_lgemm_kernel(const LGemm g) {
  __shared__ float sA[LG_TK * LG_LD];
  __shared__ float sB[LG_TK * LG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * LG_TM, n0 = blockIdx.x * LG_TN;
  const int s = blockIdx.z;
  const int k_lo = (int)((long long)g.K * s / g.splits), k_hi = (int)((long long)g.K * (s + 1) / g.splits);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  sfor<16>([&](auto RI) { acc[decltype(RI)::value] = 0.0f; });
  const bool a_kfast = (g.sak == 1), b_nfast = (g.sbn == 1);
  for (int kc = k_lo; kc < k_hi; kc += LG_TK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      int kk, mm;
      if (a_kfast) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      const int m = m0 + mm, k = kc + kk;
      sA[kk * LG_LD + mm] = (m < g.M && k < k_hi) ? g.A[(long long)m * g.sam + (long long)k * g.sak] : 0.0f;
      int kb, nn;
      if (b_nfast) { nn = idx & 63; kb = idx >> 6; } else { kb = idx & 15; nn = idx >> 4; }
      const int n = n0 + nn, k2 = kc + kb;
      sB[kb * LG_LD + nn] = (n < g.N && k2 < k_hi) ? g.B[(long long)k2 * g.sbk + (long long)n * g.sbn] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < LG_TK; k2 += 2) {
      const float a = sA[(k2 + lh) * LG_LD + 32 * wm + li];
      const float b = sB[(k2 + lh) * LG_LD + 32 * wn + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + 32 * wn + li;
  if (n >= g.N) return;
  sfor<16>([&](auto RI) {
    constexpr int r = decltype(RI)::value;
    const int m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < g.M) {
      if (g.splits > 1) {
        g.part[((long long)s * g.M + m) * g.N + n] = acc[r];
      } else {
        float v = acc[r];
        if (g.bias != nullptr) v += g.bias[n];
        if (g.add != nullptr) v += g.add[(long long)m * g.ldadd + n];
        g.C[(long long)m * g.ldc + n] = v;
      }
    }
  });
}

__global__ void __launch_bounds__(256) frl_lgemm_reduce_kernel(const LGemm g) {
  const long long total = (long long)g.M * g.N;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(e / g.N), n = (int)(e - (long long)m * g.N);
    float v = 0.0f;
    for (int s = 0; s < g.splits; ++s) v += g.part[(long long)s * total + e];      // split order: fixed
    if (g.bias != nullptr) v += g.bias[n];
    if (g.add != nullptr) v += g.add[(long long)m * g.ldadd + n];
    g.C[(long long)m * g.ldc + n] = v;
  }
}

static int run_gemm(LGemm g, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0) return FRL_OK;
  if (g.splits < 1 || g.part == nullptr) g.splits = 1;
  const dim3 grid((unsigned)((g.N + LG_TN - 1) / LG_TN), (unsigned)((g.M + LG_TM - 1) / LG_TM), (unsigned)g.splits);
  hipLaunchKernelGGL(frl_lgemm_kernel, grid, dim3(256), 0, stream, g);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && g.splits > 1) {
    const long long total = (long long)g.M * g.N;
    hipLaunchKernelGGL(frl_lgemm_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, stream, g);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { set_error("layered VJP: GEMM launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- row operations ---------------------------------------------------------------------------------------------------
enum { ROW_ACT = 0, ROW_LN_ACT = 1, ROW_ACT_LN = 2, ROW_LN = 3 };
constexpr int ROW_MAXE = 16;      // elements per lane: widths up to 1024

__device__ __forceinline__ void act_grad_rt(int act, float z, float& h, float& d) {
  f32x2 hh, dd;
  const f32x2 zz = {z, z};
  switch (act) {
    case FRL_ACT_MISH: act_grad2<FRL_ACT_MISH>(zz, hh, dd); break;
    case FRL_ACT_SILU: act_grad2<FRL_ACT_SILU>(zz, hh, dd); break;
    case FRL_ACT_RELU: act_grad2<FRL_ACT_RELU>(zz, hh, dd); break;
    case FRL_ACT_GELU_TANH: act_grad2<FRL_ACT_GELU_TANH>(zz, hh, dd); break;
    case FRL_ACT_TANH: act_grad2<FRL_ACT_TANH>(zz, hh, dd); break;
    default: act_grad2<FRL_ACT_ELU>(zz, hh, dd); break;
  }
  h = hh[0]; d = dd[0];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// mean and 1 / sqrt(var + eps) of a row held as v[i] <-> column lane + 64 i (flax LayerNorm: eps 1e-6; the variance is
// taken about the mean — the same quantity as flax's E[x^2] - E[x]^2 without its float32 cancellation)
__device__ __forceinline__ void row_stats(const float (&v)[ROW_MAXE], int W, int lane, float& mean, float& rstd) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < W) s += v[i];
  mean = wave_sum(s) / (float)W;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < W) { const float c = v[i] - mean; q = fmaf(c, c, q); }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + 1e-6f);
}

struct LRow {
  const float* z;      // (B, W) input of the row operation
  float* h;            // forward: (B, W) output
  const float* g;      // backward: (B, W) dL/d output
  float* dz;           // backward: (B, W) dL/d input (+ addend)
  const float* addend; // backward: (B, W) added to dz (residual stream), or NULL
  float* pdy; float* pxh;   // backward, LayerNorm modes: (B, W) per-sample terms of dL/dbias and dL/dscale, or NULL
  const float* scale; const float* bias;   // LayerNorm parameters (W,)
  long long B; int W, mode, act;
};

__global__ void __launch_bounds__(256) frl_lrow_fwd_kernel(const LRow a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B) return;
  const float* zr = a.z + row * a.W;
  float v[ROW_MAXE];
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) v[i] = (lane + 64 * i < a.W) ? zr[lane + 64 * i] : 0.0f;
  if (a.mode == ROW_LN_ACT || a.mode == ROW_LN) {
    float mean, rstd;
    row_stats(v, a.W, lane, mean, rstd);
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) v[i] = (v[i] - mean) * (rstd * a.scale[lane + 64 * i]) + a.bias[lane + 64 * i];
  }
  if (a.mode != ROW_LN) {
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) v[i] = act_rt(a.act, v[i]);
  }
  if (a.mode == ROW_ACT_LN) {
    float mean, rstd;
    row_stats(v, a.W, lane, mean, rstd);
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) v[i] = (v[i] - mean) * (rstd * a.scale[lane + 64 * i]) + a.bias[lane + 64 * i];
  }
  float* hr = a.h + row * a.W;
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) hr[lane + 64 * i] = v[i];
}

// y = xhat * scale + bias, xhat = (x - mean) rstd:  dL/dx = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)),  dxh = dy * scale
__device__ __forceinline__ void ln_bwd(const float (&x)[ROW_MAXE], float (&dy)[ROW_MAXE], float (&xh)[ROW_MAXE], const float* scale, int W, int lane) {
  float mean, rstd;
  row_stats(x, W, lane, mean, rstd);
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < W) {
    xh[i] = (x[i] - mean) * rstd;
    const float d = dy[i] * scale[lane + 64 * i];
    s1 += d; s2 = fmaf(d, xh[i], s2);
  }
  s1 = wave_sum(s1) / (float)W; s2 = wave_sum(s2) / (float)W;
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < W) {
    const float d = dy[i] * scale[lane + 64 * i];
    dy[i] = rstd * (d - s1 - xh[i] * s2);       // dy now holds dL/dx
  }
}

__global__ void __launch_bounds__(256) frl_lrow_bwd_kernel(const LRow a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B) return;
  const float* zr = a.z + row * a.W;
  const float* gr = a.g + row * a.W;
  float z[ROW_MAXE], g[ROW_MAXE], xh[ROW_MAXE];
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) {
    const bool in = lane + 64 * i < a.W;
    z[i] = in ? zr[lane + 64 * i] : 0.0f;
    g[i] = in ? gr[lane + 64 * i] : 0.0f;
    xh[i] = 0.0f;
  }
  float* pdy = a.pdy != nullptr ? a.pdy + row * a.W : nullptr;
  float* pxh = a.pxh != nullptr ? a.pxh + row * a.W : nullptr;
  if (a.mode == ROW_ACT) {
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) { float h, d; act_grad_rt(a.act, z[i], h, d); g[i] *= d; }
  } else if (a.mode == ROW_LN_ACT) {
    // n = LN(z); h = act(n):  dn = g act'(n);  dz = LN'(dn)
    float mean, rstd;
    row_stats(z, a.W, lane, mean, rstd);
    float dn[ROW_MAXE];
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) {
      dn[i] = 0.0f;
      if (lane + 64 * i < a.W) {
        const float n = (z[i] - mean) * (rstd * a.scale[lane + 64 * i]) + a.bias[lane + 64 * i];
        float h, d;
        act_grad_rt(a.act, n, h, d);
        g[i] *= d;
        dn[i] = g[i];
        if (pdy != nullptr) pdy[lane + 64 * i] = g[i];
      }
    }
    ln_bwd(z, g, xh, a.scale, a.W, lane);
    if (pxh != nullptr) {
#pragma unroll
      for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) pxh[lane + 64 * i] = dn[i] * xh[i];
    }
  } else if (a.mode == ROW_ACT_LN) {
    // a = act(z); h = LN(a):  da = LN'(g);  dz = da act'(z)
    float av[ROW_MAXE], dv[ROW_MAXE];
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) { av[i] = 0.0f; dv[i] = 0.0f; if (lane + 64 * i < a.W) act_grad_rt(a.act, z[i], av[i], dv[i]); }
    if (pdy != nullptr) {
#pragma unroll
      for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) pdy[lane + 64 * i] = g[i];
    }
    float g0[ROW_MAXE];
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) g0[i] = g[i];
    ln_bwd(av, g, xh, a.scale, a.W, lane);
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) {
      if (pxh != nullptr) pxh[lane + 64 * i] = g0[i] * xh[i];
      g[i] *= dv[i];
    }
  } else {   // ROW_LN
    if (pdy != nullptr) {
#pragma unroll
      for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) pdy[lane + 64 * i] = g[i];
    }
    float g0[ROW_MAXE];
#pragma unroll
    for (int i = 0; i < ROW_MAXE; ++i) g0[i] = g[i];
    ln_bwd(z, g, xh, a.scale, a.W, lane);
    if (pxh != nullptr) {
#pragma unroll
      for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) pxh[lane + 64 * i] = g0[i] * xh[i];
    }
  }
  float* dr = a.dz + row * a.W;
  const float* ad = a.addend != nullptr ? a.addend + row * a.W : nullptr;
#pragma unroll
  for (int i = 0; i < ROW_MAXE; ++i) if (lane + 64 * i < a.W) dr[lane + 64 * i] = g[i] + (ad != nullptr ? ad[lane + 64 * i] : 0.0f);
}

static int run_row(const LRow& a, bool backward, hipStream_t stream) {
  if (a.B <= 0) return FRL_OK;
  if (a.W > 64 * ROW_MAXE) { set_error("layered VJP: row width %d > %d", a.W, 64 * ROW_MAXE); return FRL_EUNSUPPORTED; }
  const dim3 grid((unsigned)((a.B + 3) / 4));
  if (backward) hipLaunchKernelGGL(frl_lrow_bwd_kernel, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(frl_lrow_fwd_kernel, grid, dim3(256), 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("layered VJP: row kernel launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- column sums over the batch, two fixed-order stages ---------------------------------------------------------------
constexpr int CS_CHUNKS = 64;
__global__ void __launch_bounds__(256) frl_lcolsum_kernel(const float* src, int ld, long long B, int W, float* part) {
  __shared__ float sp[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int nch = gridDim.y, ch = blockIdx.y;
  const long long r0 = B * ch / nch, r1 = B * (ch + 1) / nch;
  float s = 0.0f;
  if (c < W) for (long long r = r0 + q; r < r1; r += 4) s += src[r * ld + c];
  sp[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && c < W) part[(long long)ch * W + c] = (sp[0][threadIdx.x] + sp[1][threadIdx.x]) + (sp[2][threadIdx.x] + sp[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256) frl_lcolsum2_kernel(const float* part, int nch, int W, float* dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= W) return;
  float s = 0.0f;
  for (int ch = 0; ch < nch; ++ch) s += part[(long long)ch * W + c];
  dst[c] = s;
}
static int run_colsum(const float* src, int ld, long long B, int W, float* part, float* dst, hipStream_t stream) {
  if (W <= 0) return FRL_OK;
  int nch = (int)((B + 255) / 256);
  if (nch < 1) nch = 1;
  if (nch > CS_CHUNKS) nch = CS_CHUNKS;
  hipLaunchKernelGGL(frl_lcolsum_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)nch), dim3(256), 0, stream, src, ld, B, W, part);
  hipLaunchKernelGGL(frl_lcolsum2_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, stream, (const float*)part, nch, W, dst);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("layered VJP: column-sum launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- small copies -------------------------------------------------------------------------------------------------------
// inp[b] = [x[b] (A) | u[b] (Cp)]  (the predictor's input row, ddpm.py:91-94)
__global__ void __launch_bounds__(256) frl_lconcat_kernel(const float* x, int A, const float* u, int Cp, long long B, float* inp) {
  const int K0 = A + Cp;
  const long long total = B * K0;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / K0;
    const int k = (int)(e - b * K0);
    inp[e] = k < A ? x[b * A + k] : u[b * Cp + (k - A)];
  }
}
__global__ void __launch_bounds__(256) frl_lslice_kernel(const float* src, int ld, int cols, long long B, float* dst) {
  const long long total = B * cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / cols;
    dst[e] = src[b * ld + (e - b * cols)];
  }
}
static unsigned grid_for(long long total) { const long long g = (total + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// ---- the network as a sequence of launches ------------------------------------------------------------------------------
// Workspace layout (floats), computed by layered_plan: inp (B, K0) | per stage the saved activations | temporaries.
int layered_plan(const FrlDesc& d, int Cp, LayeredPlan& q) {
  memset(&q, 0, sizeof(q));
  q.B = (long long)d.n_states * d.particles;
  q.A = d.x_dim; q.Cp = Cp; q.K0 = q.A + Cp;
  q.H = d.hidden[0];
  q.res = d.backbone != FRL_BACKBONE_MLP;
  q.L = d.n_hidden;
  q.Wmax = q.res ? q.H * d.res_multiplier : q.H;
  if (q.Wmax > 64 * ROW_MAXE) { set_error("single-evaluation VJP (layered): width %d > %d", q.Wmax, 64 * ROW_MAXE); return FRL_EUNSUPPORTED; }
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
  const size_t BH = (size_t)q.B * q.H, BW = (size_t)q.B * q.Wmax;
  q.off_inp = take((size_t)q.B * q.K0);
  q.off_dinp = take((size_t)q.B * q.K0);
  if (!q.res) {
    for (int l = 0; l < q.L; ++l) { q.off_z[l] = take(BH); q.off_h[l] = take(BH); }
  } else {
    for (int i = 0; i <= q.L; ++i) q.off_z[i] = take(BH);           // trunk before block i (i = L: after the last block)
    for (int i = 0; i < q.L; ++i) { q.off_h[i] = take(BH); q.off_e[i] = take(BW); q.off_a[i] = take(BW); }   // y_i = [LN](x_i), e_i, a_i = act(e_i)
    q.off_hf = take(BH);                                             // head input: act(x_L) or LN(x_L)
  }
  q.off_g0 = take(BW); q.off_g1 = take(BW); q.off_g2 = take(BW);
  q.off_pdy = take(BW); q.off_pxh = take(BW);
  q.off_cs = take((size_t)CS_CHUNKS * q.Wmax);
  // weight gradients contract over the batch: split it beyond 4096 samples (fixed-order second stage)
  q.splits = (int)((q.B + 4095) / 4096);
  if (q.splits < 1) q.splits = 1;
  if (q.splits > 16) q.splits = 16;
  const size_t maxw = (size_t)(q.K0 > q.Wmax ? q.K0 : q.Wmax) * q.Wmax;
  q.off_part = take(q.splits > 1 ? (size_t)q.splits * maxw : 0);
  q.total_floats = o;
  return FRL_OK;
}

namespace {
struct Ctx {
  const LayeredPlan& q; float* ws; hipStream_t st; int act; int rc;
  float* at(size_t off) const { return ws + off; }
  // C (M, N) = X (M, K) W (K, N) + bias [+ add]
  void dense(const float* X, int ldx, const float* W, const float* bias, int K, int N, float* C, const float* add = nullptr) {
    if (rc != FRL_OK) return;
    LGemm g; memset(&g, 0, sizeof(g));
    g.A = X; g.B = W; g.C = C; g.bias = bias; g.add = add; g.M = (int)q.B; g.N = N; g.K = K;
    g.sam = ldx; g.sak = 1; g.sbk = N; g.sbn = 1; g.ldc = N; g.ldadd = N; g.splits = 1;
    rc = run_gemm(g, st);
  }
  // dX (M, K) = dZ (M, N) W^T [+ add]
  void dense_dx(const float* dZ, const float* W, int K, int N, float* dX, const float* add = nullptr) {
    if (rc != FRL_OK) return;
    LGemm g; memset(&g, 0, sizeof(g));
    g.A = dZ; g.B = W; g.C = dX; g.add = add; g.M = (int)q.B; g.N = K; g.K = N;
    g.sam = N; g.sak = 1; g.sbk = 1; g.sbn = N; g.ldc = K; g.ldadd = K; g.splits = 1;
    rc = run_gemm(g, st);
  }
  // dW (K, N) = X^T dZ, db (N) = column sums of dZ
  void dense_dw(const float* X, int ldx, const float* dZ, int K, int N, float* dW, float* db) {
    if (rc != FRL_OK) return;
    LGemm g; memset(&g, 0, sizeof(g));
    g.A = X; g.B = dZ; g.C = dW; g.M = K; g.N = N; g.K = (int)q.B;
    g.sam = 1; g.sak = ldx; g.sbk = N; g.sbn = 1; g.ldc = N; g.splits = q.splits; g.part = at(q.off_part);
    rc = run_gemm(g, st);
    if (rc == FRL_OK && db != nullptr) rc = run_colsum(dZ, N, q.B, N, at(q.off_cs), db, st);
  }
  void row_fwd(int mode, const float* z, float* h, int W, const float* scale, const float* bias) {
    if (rc != FRL_OK) return;
    LRow a; memset(&a, 0, sizeof(a));
    a.z = z; a.h = h; a.scale = scale; a.bias = bias; a.B = q.B; a.W = W; a.mode = mode; a.act = act;
    rc = run_row(a, false, st);
  }
  // dz = d(row op)/dz applied to g (+ addend); LayerNorm modes also produce dL/dscale, dL/dbias
  void row_bwd(int mode, const float* z, const float* g, float* dz, int W, const float* scale, const float* bias, const float* addend,
               float* g_scale, float* g_bias) {
    if (rc != FRL_OK) return;
    LRow a; memset(&a, 0, sizeof(a));
    const bool ln = mode != ROW_ACT;
    a.z = z; a.g = g; a.dz = dz; a.addend = addend; a.scale = scale; a.bias = bias; a.B = q.B; a.W = W; a.mode = mode; a.act = act;
    if (ln) { a.pdy = at(q.off_pdy); a.pxh = at(q.off_pxh); }
    rc = run_row(a, true, st);
    if (ln && rc == FRL_OK) rc = run_colsum(a.pxh, W, q.B, W, at(q.off_cs), g_scale, st);
    if (ln && rc == FRL_OK) rc = run_colsum(a.pdy, W, q.B, W, at(q.off_cs), g_bias, st);
  }
};
}  // namespace

static int mlp_row_mode(const FrlDesc& d) {
  return d.layer_norm == FRL_LN_PRE_ACT ? ROW_LN_ACT : d.layer_norm == FRL_LN_POST_ACT ? ROW_ACT_LN : ROW_ACT;
}

// forward: out = predictor([x | u]), keeping every pre-activation; u = [tfeat(t) | cond] (B, Cp) per sample
int launch_layered_fwd(const FrlDesc& d, const FrlParams& p, const LayeredPlan& q, float* ws, const float* x, const float* u, float* out,
                       hipStream_t stream) {
  if (q.B <= 0) return FRL_OK;
  hipLaunchKernelGGL(frl_lconcat_kernel, dim3(grid_for(q.B * q.K0)), dim3(256), 0, stream, x, q.A, u, q.Cp, q.B, ws + q.off_inp);
  Ctx c{q, ws, stream, d.act, FRL_OK};
  const int H = q.H;
  if (!q.res) {
    const int mode = mlp_row_mode(d);
    const float* hin = c.at(q.off_inp);
    int kin = q.K0;
    for (int l = 0; l < q.L; ++l) {
      c.dense(hin, kin, p.kernel[l], p.bias[l], kin, H, c.at(q.off_z[l]));
      c.row_fwd(mode, c.at(q.off_z[l]), c.at(q.off_h[l]), H, p.ln_scale[l], p.ln_bias[l]);
      hin = c.at(q.off_h[l]); kin = H;
    }
    c.dense(hin, kin, p.kernel[q.L], p.bias[q.L], kin, q.A, out);
  } else {
    const bool simba = d.backbone == FRL_BACKBONE_SIMBA;
    const bool ln = simba || d.layer_norm != 0;
    const int W = H * d.res_multiplier;
    c.dense(c.at(q.off_inp), q.K0, p.kernel[0], p.bias[0], q.K0, H, c.at(q.off_z[0]));
    for (int i = 0; i < q.L; ++i) {
      const float* xi = c.at(q.off_z[i]);
      const float* yi = xi;
      if (ln) { c.row_fwd(ROW_LN, xi, c.at(q.off_h[i]), H, p.res_ln_scale[i], p.res_ln_bias[i]); yi = c.at(q.off_h[i]); }
      c.dense(yi, H, p.res_kernel[i][0], p.res_bias[i][0], H, W, c.at(q.off_e[i]));
      c.row_fwd(ROW_ACT, c.at(q.off_e[i]), c.at(q.off_a[i]), W, nullptr, nullptr);
      c.dense(c.at(q.off_a[i]), W, p.res_kernel[i][1], p.res_bias[i][1], W, H, c.at(q.off_z[i + 1]), xi);      // x + Dense_1(...)
    }
    c.row_fwd(simba ? ROW_LN : ROW_ACT, c.at(q.off_z[q.L]), c.at(q.off_hf), H, p.final_ln_scale, p.final_ln_bias);
    c.dense(c.at(q.off_hf), H, p.kernel[1], p.bias[1], H, q.A, out);
  }
  hipError_t e = hipGetLastError();
  if (c.rc == FRL_OK && e != hipSuccess) { set_error("layered VJP forward: %s", hipGetErrorString(e)); return FRL_EHIP; }
  if (c.rc == FRL_OK) note_kernel("layered single-evaluation forward (frl_lgemm_kernel + frl_lrow_fwd_kernel per layer)");
  return c.rc;
}

// backward: every predictor leaf, dL/dinp = [dL/dx | dL/du] left in the workspace (q.off_dinp, row stride K0)
int launch_layered_bwd(const FrlDesc& d, const FrlParams& p, const LayeredPlan& q, float* ws, const float* g_out, const FrlEvalBwdBufs& gb,
                       float* g_x, hipStream_t stream) {
  if (q.B <= 0) return FRL_OK;
  Ctx c{q, ws, stream, d.act, FRL_OK};
  const int H = q.H;
  float* g0 = c.at(q.off_g0); float* g1 = c.at(q.off_g1); float* g2 = c.at(q.off_g2);
  if (!q.res) {
    const int mode = mlp_row_mode(d);
    const int L = q.L;
    c.dense_dw(c.at(q.off_h[L - 1]), H, g_out, H, q.A, gb.grad_kernel[L], gb.grad_bias[L]);
    c.dense_dx(g_out, p.kernel[L], H, q.A, g0);                                 // dL/dh_{L-1}
    for (int l = L - 1; l >= 0; --l) {
      c.row_bwd(mode, c.at(q.off_z[l]), g0, g1, H, p.ln_scale[l], p.ln_bias[l], nullptr, gb.grad_ln_scale[l], gb.grad_ln_bias[l]);   // dL/dz_l
      const float* xin = l > 0 ? c.at(q.off_h[l - 1]) : c.at(q.off_inp);
      const int kin = l > 0 ? H : q.K0;
      c.dense_dw(xin, kin, g1, kin, H, gb.grad_kernel[l], gb.grad_bias[l]);
      c.dense_dx(g1, p.kernel[l], kin, H, l > 0 ? g0 : c.at(q.off_dinp));
    }
  } else {
    const bool simba = d.backbone == FRL_BACKBONE_SIMBA;
    const bool ln = simba || d.layer_norm != 0;
    const int W = H * d.res_multiplier;
    // head: out = Dense_1(hf), hf = act(x_L) | LN(x_L)
    c.dense_dw(c.at(q.off_hf), H, g_out, H, q.A, gb.grad_kernel[1], gb.grad_bias[1]);
    c.dense_dx(g_out, p.kernel[1], H, q.A, g0);
    c.row_bwd(simba ? ROW_LN : ROW_ACT, c.at(q.off_z[q.L]), g0, g1, H, p.final_ln_scale, p.final_ln_bias, nullptr,
              gb.grad_final_ln_scale, gb.grad_final_ln_bias);               // g1 = dL/dx_L
    float* gx = g1; float* gt = g0;                                         // gx: dL/dx_{i+1}; gt: the other trunk-sized buffer
    for (int i = q.L - 1; i >= 0; --i) {
      const float* yi = ln ? c.at(q.off_h[i]) : c.at(q.off_z[i]);
      c.dense_dw(c.at(q.off_a[i]), W, gx, W, H, gb.grad_res_kernel[i][1], gb.grad_res_bias[i][1]);
      c.dense_dx(gx, p.res_kernel[i][1], W, H, g2);                          // dL/da_i  (B, W)
      c.row_bwd(ROW_ACT, c.at(q.off_e[i]), g2, g2, W, nullptr, nullptr, nullptr, nullptr, nullptr);   // dL/de_i in place (a lane reads its own elements before writing them)
      c.dense_dw(yi, H, g2, H, W, gb.grad_res_kernel[i][0], gb.grad_res_bias[i][0]);
      if (ln) {
        c.dense_dx(g2, p.res_kernel[i][0], H, W, gt);                        // dL/dy_i
        c.row_bwd(ROW_LN, c.at(q.off_z[i]), gt, gt, H, p.res_ln_scale[i], p.res_ln_bias[i], gx, gb.grad_res_ln_scale[i], gb.grad_res_ln_bias[i]);   // + skip
      } else {
        c.dense_dx(g2, p.res_kernel[i][0], H, W, gt, gx);                    // dL/dx_i = dL/dy_i + skip
      }
      float* t = gx; gx = gt; gt = t;
    }
    c.dense_dw(c.at(q.off_inp), q.K0, gx, q.K0, H, gb.grad_kernel[0], gb.grad_bias[0]);
    c.dense_dx(gx, p.kernel[0], q.K0, H, c.at(q.off_dinp));
  }
  if (c.rc == FRL_OK && g_x != nullptr)
    hipLaunchKernelGGL(frl_lslice_kernel, dim3(grid_for(q.B * q.A)), dim3(256), 0, stream, (const float*)c.at(q.off_dinp), q.K0, q.A, q.B, g_x);
  hipError_t e = hipGetLastError();
  if (c.rc == FRL_OK && e != hipSuccess) { set_error("layered VJP backward: %s", hipGetErrorString(e)); return FRL_EHIP; }
  if (c.rc == FRL_OK) note_kernel("layered single-evaluation backward (frl_lrow_bwd_kernel + 2 x frl_lgemm_kernel + frl_lcolsum_kernel per layer)");
  return c.rc;
}

}  // namespace frl
