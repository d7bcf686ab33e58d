// bf16-MFMA variant of the fused T-step sampler (FrlDesc.dtype = FRL_DTYPE_BF16; hidden width 256).
// BASELINE north star: "fp32 for bit-faithful runs, bf16 as a stated-tolerance variant".  Same reference semantics
// as sampler_rs.hip (DDPM.sample ddpm.py:184-221, ContinuousDDPM.sample continuous_ddpm.py:149-222, CNF.sample
// cnf.py:122-141; network ddpm.py:83-95 + mlp.py:25-33); what changes is the arithmetic of the Dense layers:
// operands (weights, layer inputs) are rounded to bf16, products are accumulated in f32 by
// v_mfma_f32_32x32x16_bf16; biases, the hoisted time branch, activations and the chain state / update stay f32.
//
// Layout.  Still TRANSPOSED ([32 output features] x [32 chains] per MFMA tile, a chain on a lane).  The 32x32
// accumulator layout is that of the f32 kernel (register r of lane half s = feature row (r&3) + 8(r>>2) + 4s), and
// the B operand of the x16 instruction wants 8 consecutive k on a lane (k = 8s + e): registers 8q..8q+7 of an
// accumulator block, activated and packed pairwise with v_cvt_pk_bf16_f32, ARE the B operand of k-step (block, q)
// of the next layer — activations still never leave the register file.  The k <-> feature map this implies,
//     feature(block, q, s, e) = 32 block + (e&3) + 8 (2q + (e>>2)) + 4 s,
// is baked into the weight image: frl_prepare_bf16_kernel writes, per step, a stream of 1 KB A-operand "pieces"
// (64 lanes x 8 bf16: lane (i, s) holds W[feature(.., s, e)][32 ob + i], e = 0..7), 8 pieces (ob = 0..7) per k-step,
// in exactly the order the kernel consumes them.
//
// One buffer_load per MFMA is NOT viable here (measured: 64 cycles per 32-cycle MFMA, scripts/ubench), so the
// stream goes through LDS: 32 KB chunks (4 k-steps) by LDS-DMA into a 4-slot ring, three chunks in flight, one
// s_barrier per chunk, read back with one ds_read_b128 per MFMA (measured free beside the MFMA).  The kernel is
// VALU-bound (activation: 7 VALU + cvt per element vs 1 MFMA per 8 elements): plain f32 ops only, because
// v_pk_*_f32 beside bf16 MFMAs is an anti-lever (measured +8.5 cycles each) — this TU is built with
// -fno-slp-vectorize so the compiler does not pack them either.
#include "frl_device.h"

namespace frl {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define BF_ROWR(r) (((r) & 3) + 8 * ((r) >> 2))

__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even (NaN not expected in weights)
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// ---- weight stream image ------------------------------------------------------------------------------------
struct PBfArgs {
  const float* kernel[FRL_MAX_HIDDEN + 1];
  unsigned short* img;
  int A, C, Dp, L, nxk, nck, l0c, ns;
};

__global__ void __launch_bounds__(256) frl_prepare_bf16_kernel(const PBfArgs p) {
  constexpr int H = 256;
  const long long total = (long long)p.ns * (BF_CHUNK_BYTES / 2);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / (BF_CHUNK_BYTES / 2));
    const int within = (int)(idx - (long long)c * (BF_CHUNK_BYTES / 2));
    const int piece = within >> 9, lane = (within >> 3) & 63, e = within & 7;
    const int i = lane & 31, s = lane >> 5;
    float v = 0.0f;
    if (c < p.l0c) {                                   // Dense_0: x k-steps, then cond k-steps (time rows are in tb)
      const int ks = c * 4 + (piece >> 3), ob = piece & 7;
      if (ks < p.nxk) {
        const int row = BF_ROWR(8 * ks + e) + 4 * s;
        if (row < p.A) v = p.kernel[0][(size_t)row * H + 32 * ob + i];
      } else if (ks < p.nxk + p.nck) {
        const int ci = 16 * (ks - p.nxk) + 8 * s + e;
        if (ci < p.C) v = p.kernel[0][(size_t)(p.A + p.Dp + ci) * H + 32 * ob + i];
      }
    } else if (c < p.l0c + 4 * (p.L - 1)) {            // hidden Dense_l
      const int hc = c - p.l0c, l = 1 + (hc >> 2);
      const int ks = (hc & 3) * 4 + (piece >> 3), ob = piece & 7;
      const int feat = 32 * (ks >> 1) + BF_ROWR(8 * (ks & 1) + e) + 4 * s;
      v = p.kernel[l][(size_t)feat * H + 32 * ob + i];
    } else if (piece < 16) {                           // Dense_L, one output block (rows >= A are zero)
      const int ks = piece;
      const int feat = 32 * (ks >> 1) + BF_ROWR(8 * (ks & 1) + e) + 4 * s;
      if (i < p.A) v = p.kernel[p.L][(size_t)feat * p.A + i];
    }
    p.img[idx] = f2bf(v);
  }
}

int launch_prepare_bf16(const FrlParams& prm, const Plan& pl, float* ws, hipStream_t stream) {
  PBfArgs p;
  for (int l = 0; l <= pl.L; ++l) p.kernel[l] = prm.kernel[l];
  p.img = (unsigned short*)(ws + pl.off_bf16);
  p.A = pl.A; p.C = pl.C; p.Dp = pl.Dp; p.L = pl.L;
  p.nxk = pl.bf_nxk; p.nck = pl.bf_nck; p.l0c = pl.bf_l0c; p.ns = pl.bf_ns;
  hipLaunchKernelGGL(frl_prepare_bf16_kernel, dim3(64), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 prepare launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- activations: cheapest f32 forms whose error is far below one bf16 ulp (2^-9) ---------------------------
template <int ACT>
__device__ __forceinline__ float act_bf(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (ACT == FRL_ACT_MISH) {
    // x tanh(softplus x) = x n / (n + 2) = x - 2 x / (n + 2),  n = e (e + 2);  e = inf gives x, e = 0 gives 0
    const float e = __builtin_amdgcn_exp2f(x * LOG2E);
    const float d = __builtin_fmaf(e, e + 2.0f, 2.0f);
    const float m = x * __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(-2.0f, m, x);
  } else if constexpr (ACT == FRL_ACT_SILU) {
    const float e = __builtin_amdgcn_exp2f(x * -LOG2E);
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
  } else if constexpr (ACT == FRL_ACT_RELU) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));   // fmaxf() adds a canonicalising v_max first
    return y;
  } else if constexpr (ACT == FRL_ACT_GELU_TANH) {
    // x sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): exponent -2u log2e as a polynomial in x
    constexpr float K1 = -2.0f * 0.7978845608028654f * LOG2E, K3 = K1 * 0.044715f;
    const float pq = __builtin_fmaf(x * x, K3, K1);
    const float e = __builtin_amdgcn_exp2f(pq * x);
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
  } else {
    return act_fn<ACT>(x);
  }
}

// registers 8q .. 8q+7 of an accumulator block -> the packed bf16 B operand of k-step (block, q)
template <int ACT, int Q8>
__device__ __forceinline__ u32x4 pack_act(const f32x16& p) {
  u32x4 o;
  sfor<4>([&](auto WI) {
    constexpr int w = decltype(WI)::value;
    const f32x2 v = {act_bf<ACT>(p[Q8 + 2 * w]), act_bf<ACT>(p[Q8 + 2 * w + 1])};
    o[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  });
  return o;
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int OFF>
__device__ __forceinline__ void dma16_bf(const void* gsrc, void* lds_dst) {
  // 64 lanes x 16 B: per-lane global source, LDS destination = wave-uniform base + lane * 16; the instruction's
  // immediate offset OFF applies to both sides (pieces are 1 KB apart in the image and in the ring alike)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, OFF, 0);
}

// One hidden Dense layer: 16 k-steps x 8 output blocks.  Operand fetch (8 x ds_read_b128) and the activated,
// packed B operand of k-step n+1 are produced in the shadow of k-step n's 8 MFMAs; a chunk boundary (barrier +
// next DMA) every 4 k-steps.
template <int ACT, typename NextChunk>
__device__ __forceinline__ void hidden_layer_bf(f32x16 (&Q)[8], const f32x16 (&P)[8], NextChunk& next_chunk, unsigned lane16) {
  const char* sl = next_chunk() + lane16;
  u32x4 aw[8], nw[8];
  sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; aw[ob] = *(const u32x4*)(sl + ob * 1024); });
  u32x4 bop = pack_act<ACT, 0>(P[0]), nb = bop;
  sfor<16>([&](auto KI) {
    constexpr int ks = decltype(KI)::value;
    if constexpr (ks + 1 < 16) {
      constexpr int n = ks + 1;
      if constexpr ((n & 3) == 0) sl = next_chunk() + lane16;
      sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; nw[ob] = *(const u32x4*)(sl + ((n & 3) * 8 + ob) * 1024); });
      nb = pack_act<ACT, 8 * (n & 1)>(P[n >> 1]);
    }
    sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; Q[ob] = mfma_bf(aw[ob], bop, Q[ob]); });
    if constexpr (ks + 1 < 16) {
      // the 8 operand reads go out in the first 4 gaps, so the last one has >= 4 MFMAs (128+ cycles) to land before
      // k-step n+1 starts (one read per gap left its full LDS latency exposed at every k-step boundary)
      sfor<4>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 x ds_read (A operands of k-step n+1)
        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);   // VALU (activation + pack of k-step n+1)
      });
      sfor<4>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; aw[ob] = nw[ob]; });
    bop = nb;
  });
}

template <int ACT, int NRX, int CKS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_bf16_kernel(const KArgs a) {
  constexpr int H = 256, NXK = NRX / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, ls = lane >> 5;

  // bias tables, permuted so that a lane half reads the 16 accumulator inits of a block as 4 x 16 B:
  // tbl[(s * 8 + ob) * 16 + r] = bias[32 ob + ROWR(r) + 4 s]
  float* s_tb = (float*)(smem + BF_NBUF * BF_CHUNK_BYTES);   // [T][256]
  float* s_bh = s_tb + a.T * H;                               // [L-1][256]
  float* s_bl = s_bh + (a.L - 1) * H;                         // [32] padded last-layer bias (natural order)
  float* s_coef = s_bl + LAST_PAD;                            // [T][8]
  auto perm_src = [](int p) { const int s = p >> 7, ob = (p >> 4) & 7, r = p & 15; return 32 * ob + BF_ROWR(r) + 4 * s; };
  for (int idx = tid; idx < a.T * H; idx += 256) s_tb[idx] = a.ws_tb[(idx & ~(H - 1)) + perm_src(idx & (H - 1))];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr_rs(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += 256) s_bh[(l - 1) * H + idx] = bp[perm_src(idx)];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.T * COEF_STRIDE; idx += 256) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  // ---- weight stream: the same bf_ns chunks every step of every tile, three in flight ----------------------
  const int NS = a.bf_ns;
  const char* img = (const char*)a.ws_bf16 + wave * 8192 + lane * 16;   // this wave DMAs pieces 8w .. 8w+7 of a chunk
  auto issue = [&](int seq, unsigned boff) {
    const char* src = img + (size_t)seq * BF_CHUNK_BYTES;
    char* dst = smem + boff + wave * 8192;
    sfor<8>([&](auto II) {
      constexpr int i = decltype(II)::value;
      dma16_bf<(i & 3) * 1024>(src + (i >> 2) * 4096, dst + (i >> 2) * 4096);
    });
  };
  auto wrap_seq = [&](int s) { return s >= NS ? s - NS : s; };
  issue(0, 0);
  issue(wrap_seq(1), BF_CHUNK_BYTES);
  issue(wrap_seq(wrap_seq(1) + 1), 2 * BF_CHUNK_BYTES);
  int seq_issue = wrap_seq(wrap_seq(wrap_seq(1) + 1) + 1);
  unsigned boff_issue = 3 * BF_CHUNK_BYTES, boff_cur = 0;
  // start of a chunk's compute: its DMA has landed (all but the two newer chunks' 16 DMAs of this wave are done),
  // every wave is past the previous chunk (barrier) so its slot can take the chunk three ahead.
  auto next_chunk = [&]() -> const char* {
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(seq_issue, boff_issue);
    seq_issue = (seq_issue + 1 == NS) ? 0 : seq_issue + 1;
    boff_issue = (boff_issue + BF_CHUNK_BYTES) & (BF_NBUF * BF_CHUNK_BYTES - 1);
    const char* p = smem + boff_cur;
    boff_cur = (boff_cur + BF_CHUNK_BYTES) & (BF_NBUF * BF_CHUNK_BYTES - 1);
    return p;
  };

  const unsigned lane16 = lane * 16;
  auto bias_init = [&](f32x16 (&Q)[8], const float* sb) {
    sfor<8>([&](auto OI) {
      constexpr int ob = decltype(OI)::value;
      sfor<4>([&](auto VI) {
        constexpr int v = decltype(VI)::value;
        const f32x4 bv = *(const f32x4*)(sb + (ls * 8 + ob) * 16 + 4 * v);
        sfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; Q[ob][4 * v + j] = bv[j]; });
      });
    });
  };

  f32x16 P[8], Q[8];
  const int ntiles = (int)((a.B + 127) / 128);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long chain = (long long)tile * 128 + wave * 32 + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    float xs[NRX];      // chain state x^T (f32): register r holds action row ROWR(r) + 4 s
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = BF_ROWR(r) + 4 * ls;
      xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f;
    });
    u32x4 cbp[CKS];     // cond operands, packed once per tile: k-step kc, lane half s, element e <-> cond[16 kc + 8 s + e]
    sfor<CKS>([&](auto CI) {
      constexpr int kc = decltype(CI)::value;
      float cv[8];
      sfor<8>([&](auto EI) {
        constexpr int e = decltype(EI)::value;
        const int c = 16 * kc + 8 * ls + e;
        cv[e] = (valid && c < a.C) ? a.cond[state * a.C + c] : 0.0f;
      });
      sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; cbp[kc][w] = pack2(cv[2 * w], cv[2 * w + 1]); });
    });

    for (int j = 0; j < a.T; ++j) {
      float zr[NRX];    // noise of this step: issued now, consumed after the last layer
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = BF_ROWR(r) + 4 * ls;
        zr[r] = (a.noise != nullptr && valid && ai < a.A) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
      });

      // ---------------- Dense_0 : P = tb[j] + W0x^T x + W0c^T cond ----------------------------------
      bias_init(P, s_tb + j * H);
      {
        const char* sl = next_chunk() + lane16;
        sfor<NXK>([&](auto QI) {
          constexpr int q = decltype(QI)::value;
          u32x4 bx;
          sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; bx[w] = pack2(xs[8 * q + 2 * w], xs[8 * q + 2 * w + 1]); });
          u32x4 aw[8];
          sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; aw[ob] = *(const u32x4*)(sl + (q * 8 + ob) * 1024); });
          sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; P[ob] = mfma_bf(aw[ob], bx, P[ob]); });
        });
        sfor<CKS>([&](auto CI) {
          constexpr int kc = decltype(CI)::value;
          constexpr int ks = NXK + kc;
          if (kc < a.bf_nck) {
            if constexpr ((ks & 3) == 0) sl = next_chunk() + lane16;
            u32x4 aw[8];
            sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; aw[ob] = *(const u32x4*)(sl + ((ks & 3) * 8 + ob) * 1024); });
            sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; P[ob] = mfma_bf(aw[ob], cbp[kc], P[ob]); });
          }
        });
      }

      // ---------------- hidden Dense_l : Q = b_l + W_l^T act(P), roles alternate (no copy) ----------
      int l = 1;
      for (; l + 1 < a.L; l += 2) {
        bias_init(Q, s_bh + (l - 1) * H);
        hidden_layer_bf<ACT>(Q, P, next_chunk, lane16);
        bias_init(P, s_bh + l * H);
        hidden_layer_bf<ACT>(P, Q, next_chunk, lane16);
      }
      if (l < a.L) {
        bias_init(Q, s_bh + (l - 1) * H);
        hidden_layer_bf<ACT>(Q, P, next_chunk, lane16);
        sfor<8>([&](auto OI) { constexpr int ob = decltype(OI)::value; P[ob] = Q[ob]; });
      }

      // ---------------- Dense_L : O = b_L + W_L^T act(P), one output block --------------------------
      f32x16 O;
      sfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; O[r] = s_bl[BF_ROWR(r) + 4 * ls]; });
      {
        const char* sl = next_chunk() + lane16;
        u32x4 awl[8];   // operand ring, 8 k-steps ahead
        sfor<8>([&](auto KI) { constexpr int ks = decltype(KI)::value; awl[ks] = *(const u32x4*)(sl + ks * 1024); });
        u32x4 bop = pack_act<ACT, 0>(P[0]), nb = bop;
        sfor<16>([&](auto KI) {
          constexpr int ks = decltype(KI)::value;
          if constexpr (ks + 1 < 16) nb = pack_act<ACT, 8 * ((ks + 1) & 1)>(P[(ks + 1) >> 1]);
          O = mfma_bf(awl[ks & 7], bop, O);
          if constexpr (ks + 8 < 16) awl[ks & 7] = *(const u32x4*)(sl + (ks + 8) * 1024);
          bop = nb;
        });
      }

      // ---------------- epilogue: one reverse-diffusion / Euler step (f32) ---------------------------
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = BF_ROWR(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        const float eps = O[r], xt = xs[r];
        if (a.xt_hist != nullptr && in) {
          const long long o = ((long long)j * a.B + chain) * a.A + ai;
          a.xt_hist[o] = xt;
          a.eps_hist[o] = eps;
        }
        float xn;
        if (a.mode == MODE_DDPM_DDPM) {
          float x0h = c0 * (xt - c1 * eps);                       // ddpm.py:199
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);   // ddpm.py:200-201
          xn = c2 * (c3 * x0h + c4 * xt);                           // ddpm.py:202-203
          xn += c5 * zr[r];                                         // ddpm.py:206
        } else if (a.mode == MODE_DDPM_DDIM) {
          xn = c0 * (xt - c1 * eps) + c2 * eps;                     // ddpm.py:211
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        } else if (a.mode == MODE_CDDPM_DDPM) {
          float x0h = (xt - c1 * eps) / c0;                         // continuous_ddpm.py:165
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * xt;                                  // :173
          xn += c5 * zr[r];                                         // :174
        } else if (a.mode == MODE_CDDPM_DDIM) {
          float x0h = (xt - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;                                 // :177
        } else {
          xn = xt + eps / c0;                                       // cnf.py:108
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        }
        xs[r] = in ? xn : 0.0f;
      });
    }

    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = BF_ROWR(r) + 4 * ls;
      if (valid && ai < a.A) a.x0[chain * a.A + ai] = xs[r];
    });
  }
  // the stream runs three chunks ahead: drain the prefetches before the wave ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- launcher ------------------------------------------------------------------------------------------------
template <int ACT, int NRX, int CKS>
static int launch_inst_bf(const KArgs& a, int grid, size_t lds, hipStream_t stream) {
  auto kern = frl_sampler_bf16_kernel<ACT, NRX, CKS>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

template <int ACT>
static int launch_act_bf(const KArgs& a, int grid, size_t lds, hipStream_t stream) {
  if (a.bf_nxk == 1 && a.bf_nck <= 2) return launch_inst_bf<ACT, 8, 2>(a, grid, lds, stream);
  if (a.bf_nck <= 8) {
    if (a.bf_nxk == 1) { set_error("internal: bf16 wide instantiation expects 2 x k-steps"); return FRL_EINVAL; }
    return launch_inst_bf<ACT, 16, 8>(a, grid, lds, stream);
  }
  set_error("bf16 kernel: x_dim %d / cond_dim %d outside the compiled operand budgets (x_dim <= 32, cond_dim <= 128)", a.A, a.C);
  return FRL_EUNSUPPORTED;
}

size_t bf16_lds_bytes(int T, int L) {
  return (size_t)BF_NBUF * BF_CHUNK_BYTES + sizeof(float) * ((size_t)T * 256 + (size_t)(L - 1) * 256 + LAST_PAD + (size_t)T * COEF_STRIDE);
}

int launch_sampler_bf16(const FrlDesc& d, const KArgs& a, hipStream_t stream) {
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
  const long long ntiles = (a.B + 127) / 128;
  const int grid = (int)(ntiles < ncu ? ntiles : ncu);
  const size_t lds = bf16_lds_bytes(a.T, a.L);
  if (lds > 160 * 1024) { set_error("bf16 kernel: steps=%d needs %zu B of LDS (> 160 KiB)", a.T, lds); return FRL_EUNSUPPORTED; }
  switch (d.act) {
#if !defined(FRL_BF_ONLY_MISH)
    case FRL_ACT_SILU: return launch_act_bf<FRL_ACT_SILU>(a, grid, lds, stream);
    case FRL_ACT_RELU: return launch_act_bf<FRL_ACT_RELU>(a, grid, lds, stream);
    case FRL_ACT_GELU_TANH: return launch_act_bf<FRL_ACT_GELU_TANH>(a, grid, lds, stream);
#endif
    case FRL_ACT_MISH: return launch_act_bf<FRL_ACT_MISH>(a, grid, lds, stream);
    default: break;
  }
  set_error("bf16 kernel: activation %d not compiled in (mish / silu / relu / gelu_tanh)", d.act);
  return FRL_EUNSUPPORTED;
}

}  // namespace frl
