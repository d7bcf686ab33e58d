// Stage 1 of the fused sampler: everything that is batch-invariant.
//
//  * per-step update coefficients, evaluated in float32 in the reference's op order
//      DDPM            flowrl/flow/ddpm.py:199-213
//      ContinuousDDPM  flowrl/flow/continuous_ddpm.py:165-177
//      CNF             flowrl/flow/cnf.py:108
//  * the time branch of DDPMBackbone / FlowBackbone (ddpm.py:83-88, cnf.py:30-35):
//      time embedding (time_embedding.py:6-27) -> MLP[D, D] with mish after BOTH Dense layers
//      (mlp.py:25-33, output_dim = 0) -> its slice of predictor Dense_0.  The time value is the
//      same for every chain of a sampler call, so the slice collapses to a [T, H] bias table
//          tb[j] = bias_0 + tfeat[j] @ kernel_0[A : A + D', :]
//  * zero-padded, MFMA-friendly images of the two edge layers (Dense_0's x / cond rows, Dense_L).
//
// Workgroup e < NE runs the coefficient row and the time branch of network evaluation e (a chain of small dependent
// stages); the other workgroups pack the edge-layer images (independent grid-stride copies).  O(T * (D^2 + D*H)) work — microseconds.  Transcendentals here go through double
// precision so that each f32 op of the reference is reproduced correctly rounded.
#include "frl_internal.h"

namespace frl {

__device__ __forceinline__ float mish_exact(float x) {
  // x * tanh(softplus(x)), functional/activation.py:17-18, correctly rounded via double
  double xd = (double)x;
  double sp = (xd > 30.0) ? xd : log1p(exp(xd));
  return (float)(xd * tanh(sp));
}

// Zero-padded, MFMA-friendly images of the two edge layers: independent element-wise copies, strided over `nt` threads.
__device__ __forceinline__ void pack_edge_layers(const PArgs& a, int tid, int nt) {
  const Plan& pl = a.plan;
  const int H = pl.H, A = pl.A, C = pl.C, AO = pl.AO;
  float* ws = a.ws;
  float* w0p = ws + pl.off_w0p;
  float* wlp = ws + pl.off_wlp;
  float* wlq = ws + pl.off_wlq;
  float* blp = ws + pl.off_blp;
  const float* W0 = a.p.kernel[0];
  // ---- packed layer-0 image: rows [0, XR) = x rows (zero beyond A), then cond rows (zero beyond C) --
  const int w0_rows = (int)pl.w0p_rows;
  for (int idx = tid; idx < w0_rows * H; idx += nt) {
    const int r = idx / H, f = idx - r * H;
    float v = 0.0f;
    if (r < pl.XR) {
      if (r < A) v = W0[(size_t)(pl.x_row0 + r) * H + f];
    } else {
      const int k = r - pl.XR;
      if (k < C) v = W0[(size_t)(pl.c_row0 + k) * H + f];
    }
    w0p[idx] = v;
  }
  // ---- last layer padded to LAST_PAD output columns ---------------------------------------------------
  const int HL = pl.H;   // width of the last hidden layer
  const float* WL = a.p.kernel[pl.L];
  for (int idx = tid; idx < HL * LAST_PAD; idx += nt) {
    const int k = idx / LAST_PAD, i = idx - k * LAST_PAD;
    wlp[idx] = (i < AO) ? WL[(size_t)k * AO + i] : 0.0f;
  }
  for (int i = tid; i < LAST_PAD; i += nt) blp[i] = (i < AO) ? a.p.bias[pl.L][i] : 0.0f;
  // ---- last layer, per-lane packed: load q (16 B per lane) = A operands of k-steps 4q..4q+3, k = block*NR + r,
  //      lane (i = lane % TR, s = lane / TR) contracts feature GW*g + j + 4*(ROWR(r) + 4 s), block = 4 g + j ----
  {
    const int TR = pl.TR, NR = pl.NR, GW = 4 * TR;
    for (int idx = tid; idx < HL * 64; idx += nt) {
      const int q = idx >> 8, lane = (idx >> 2) & 63, e = idx & 3;
      const int k = 4 * q + e, blk = k / NR, r = k - blk * NR;
      const int rowr = (TR == 32) ? ((r & 3) + 8 * (r >> 2)) : ((r & 3) + 16 * (r >> 2));
      const int i = lane % TR, sl = lane / TR;
      const int kf = GW * (blk >> 2) + (blk & 3) + 4 * (rowr + 4 * sl);
      wlq[idx] = (i < AO && kf < HL) ? WL[(size_t)kf * AO + i] : 0.0f;
    }
  }
}

__global__ void __launch_bounds__(256) frl_prepare_kernel(const PArgs a) {
  extern __shared__ float sh[];  // 3 * Dmax floats: this workgroup's own row of temb, h1, h2
  const Plan& pl = a.plan;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int T = pl.T, H = pl.H, A = pl.A, C = pl.C, D = pl.D, Dp = pl.Dp, AO = pl.AO;
  float* ws = a.ws;
  float* coef = ws + pl.off_coef;
  float* tb = ws + pl.off_tb;
  float* w0p = ws + pl.off_w0p;
  float* wlp = ws + pl.off_wlp;
  float* wlq = ws + pl.off_wlq;
  float* blp = ws + pl.off_blp;
  float* temb = sh;                 // [D], row e0 only (indexed by the column d below)
  float* h1 = sh + (D > 0 ? D : 1);
  float* h2 = h1 + (D > 0 ? D : 1);
  __shared__ float tvals[256];      // time input of network evaluation e (NE <= 256 checked on host)

  const int NE = pl.NE;
  const float* W0 = a.p.kernel[0];
  const float* b0 = a.p.bias[0];
  if ((int)blockIdx.x >= NE) {
    pack_edge_layers(a, (int)(blockIdx.x - NE) * nt + tid, (int)(gridDim.x - NE) * nt);
    return;
  }
  // workgroup e < NE owns network evaluation row e: its coefficients, its time features, its row of tb
  const int e0 = blockIdx.x, e1 = e0 + 1;
  // ---- coefficients + time values, one row per network evaluation -----------------------------------
  for (int e = e0 + tid; e < e1; e += nt) {
    float c[COEF_STRIDE] = {0, 0, 0, 0, 0, 0, 0, 0};
    float tv;
    const int j = (pl.mode == MODE_CDDPM_DPM2) ? e >> 1 : e;     // loop iteration of the reference's scan
    if (pl.mode == MODE_EVAL) {
      tv = 0.0f;                             // plain evaluation: no time input, no update rule
    } else if (pl.mode == MODE_CNF_EULER) {
      tv = (float)j / (float)T;              // step2t(arange(steps)), cnf.py:80-83,139
      c[0] = (float)T;                       // x_next = xt + vel / steps, cnf.py:108
    } else {
      const int t = T - j;                   // scan over arange(steps, 0, -1), ddpm.py:219
      const float ah = a.s.alpha_hats[t], ah1 = a.s.alpha_hats[t - 1], al = a.s.alphas[t];
      if (pl.mode == MODE_DDPM_DDPM) {
        tv = (float)t;
        c[0] = 1.0f / sqrtf(ah);             // x0_hat = c0 * (xt - c1 * eps)
        c[1] = sqrtf(1.0f - ah);
        c[2] = 1.0f / (1.0f - ah);           // xt_1 = c2 * (c3 * x0_hat + c4 * xt)
        c[3] = sqrtf(ah1) * (1.0f - al);
        c[4] = sqrtf(al) * (1.0f - ah1);
        c[5] = (t > 1 ? 1.0f : 0.0f) * sqrtf(a.s.postvars[t]);   // (t>1) * std_t
      } else if (pl.mode == MODE_DDPM_DDIM) {
        tv = (float)t;
        c[0] = 1.0f / sqrtf(al);             // alpha_1
        c[1] = sqrtf(1.0f - ah);             // alpha_2
        c[2] = sqrtf(1.0f - ah1);            // alpha_3
      } else {
        const float be = a.s.betas[t];
        tv = a.s.ts[t];
        c[0] = sqrtf(ah);                    // x0_hat = (xt - c1 * eps) / c0
        c[1] = sqrtf(1.0f - ah);
        if (pl.mode == MODE_CDDPM_DDPM) {
          c[2] = sqrtf(ah1) * be / (1.0f - ah);            // mean_coef1
          c[3] = sqrtf(al) * (1.0f - ah1) / (1.0f - ah);   // mean_coef2
          c[5] = (t > 1 ? 1.0f : 0.0f) * sqrtf(be);
        } else if (pl.mode == MODE_CDDPM_DDIM) {
          c[2] = sqrtf(ah1);
          c[3] = sqrtf(1.0f - ah1);
        } else if (pl.mode == MODE_CDDPM_DPM2) {
          // continuous_ddpm.py:179-196.  Evaluation 2j: eps at (xt, ts[t]) -> x_s = alpha_s x0_hat + sigma_s eps;
          // evaluation 2j+1: eps_s at (x_s, midpoint) -> xt_1 = sqrt(ah1) x0_hat_corrected + sqrt(1 - ah1) eps_s,
          // the corrected x0_hat taken from the ORIGINAL xt.  Both have the shape  c2 * x0_hat + c3 * eps.
          if ((e & 1) == 0) {
            c[2] = a.s.mid_alphas[t - 1];
            c[3] = a.s.mid_sigmas[t - 1];
          } else {
            tv = a.s.mid_ts[t - 1];
            c[2] = sqrtf(ah1);
            c[3] = sqrtf(1.0f - ah1);
            c[6] = 1.0f;                                    // second evaluation of the step
          }
        } else {
          // DPM-Solver++(2M), continuous_ddpm.py:141-142,198-212.  All of h, prev_h, r are batch-invariant:
          //   d = c4 * x0_hat - c5 * prev_x0_hat;   xt_1 = c2 * xt - c3 * d
          auto lam = [&](int k) {
            const float v = a.s.alpha_hats[k];
            return logf(sqrtf(v) / sqrtf(fmaxf(1.0f - v, 1e-12f)));
          };
          const float h = lam(t - 1) - lam(t);
          const float prev_h = (j == 0) ? 0.0f : lam(t) - lam(t + 1);    // h of the previous iteration (carry init 0)
          const float r = prev_h / fmaxf(h, 1e-8f);
          const bool second = prev_h > 1e-8f;
          c[2] = sqrtf(1.0f - ah1) / sqrtf(1.0f - ah);      // sigma_next / sigma_curr
          c[3] = sqrtf(ah1) * (float)expm1((double)-h);     // alpha_next * expm1(-h)
          c[4] = second ? 1.0f + 0.5f / r : 1.0f;
          c[5] = second ? 0.5f / r : 0.0f;
        }
      }
    }
    tvals[e] = tv;
    for (int k = 0; k < COEF_STRIDE; ++k) coef[e * COEF_STRIDE + k] = c[k];
  }
  __syncthreads();

  if (a.time_embed == FRL_TEMB_NONE) {
    // raw scalar time is input column A of Dense_0 (cnf.py:38-41)
    for (int idx = e0 * H + tid; idx < e1 * H; idx += nt) {
      const int j = idx / H, f = idx - j * H;
      tb[idx] = (Dp == 0) ? b0[f] : fmaf(tvals[j], W0[(size_t)A * H + f], b0[f]);
    }
  } else {
    const int half = D / 2;
    // ---- embedding ------------------------------------------------------------------------------
    for (int idx = e0 * D + tid; idx < e1 * D; idx += nt) {
      const int j = idx / D, d = idx - j * D;
      const int dd = (d < half) ? d : d - half;
      float arg;
      if (a.time_embed == FRL_TEMB_POSITIONAL) {
        // f = log(max_positions)/(half-1); f = exp(arange(half) * -f); f = x * f   (all f32)
        const float lf = (float)log(10000.0) / (float)(half - 1);
        const float fr = (float)exp((double)((float)dd * -lf));
        arg = tvals[j] * fr;
      } else {
        // f = 2*pi*x @ w.T, w (D/2, 1)
        const float two_pi_x = 6.283185307179586f * tvals[j];
        arg = two_pi_x * a.p.time_kernel[dd];
      }
      temb[d] = (d < half) ? (float)cos((double)arg) : (float)sin((double)arg);   // cos first
    }
    __syncthreads();
    // ---- MLP_0/Dense_0 + mish ---------------------------------------------------------------------
    for (int idx = e0 * D + tid; idx < e1 * D; idx += nt) {
      const int j = idx / D, d = idx - j * D;
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(temb[k], a.p.tmlp_kernel[0][(size_t)k * D + d], acc);
      h1[d] = mish_exact(acc + a.p.tmlp_bias[0][d]);
    }
    __syncthreads();
    // ---- MLP_0/Dense_1 + mish ---------------------------------------------------------------------
    for (int idx = e0 * D + tid; idx < e1 * D; idx += nt) {
      const int j = idx / D, d = idx - j * D;
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(h1[k], a.p.tmlp_kernel[1][(size_t)k * D + d], acc);
      h2[d] = mish_exact(acc + a.p.tmlp_bias[1][d]);
    }
    __syncthreads();
    // ---- time slice of predictor Dense_0 -> bias table --------------------------------------------
    for (int idx = e0 * H + tid; idx < e1 * H; idx += nt) {
      const int j = idx / H, f = idx - j * H;
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(h2[k], W0[(size_t)(A + k) * H + f], acc);
      tb[idx] = acc + b0[f];
    }
  }

}

int launch_prepare(const PArgs& a, hipStream_t stream) {
  const int Dm = a.plan.D > 0 ? a.plan.D : 1;
  const size_t shbytes = (size_t)3 * Dm * sizeof(float);   // one row of temb / h1 / h2 per workgroup
  if (shbytes > 60000) { set_error("prepare: time_dim %d too large (%zu B of LDS)", a.plan.D, shbytes); return FRL_EUNSUPPORTED; }
  hipLaunchKernelGGL(frl_prepare_kernel, dim3(a.plan.NE + 32), dim3(256), shbytes, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("prepare launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
