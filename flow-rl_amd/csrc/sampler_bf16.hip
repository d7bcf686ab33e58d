// bf16-MFMA variant of the fused T-step sampler (FrlDesc.dtype = FRL_DTYPE_BF16; hidden width 256, or 512 — see the end
// of this comment).
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
// (64 lanes x 8 bf16: lane (i, s) holds W[feature(.., s, e)][32 ob + i], e = 0..7) in exactly the order the kernel
// consumes them: OUTPUT-BLOCK-MAJOR (for ob: for k-step), so that a wave finishes one 32-feature output block at a
// time (16 MFMAs into ONE accumulator), activates + packs it (-> two k-steps of the next layer's B operand) while
// the next block's MFMAs run, and never holds a whole layer of f32 accumulators: 64 registers of packed inputs,
// 64 of packed outputs, 2 x 16 accumulators.  That fits 256 registers, i.e. TWO waves per SIMD — which is the
// point: measured (scripts/ubench/act_mix.hip and the first version of this kernel), VALU work issued by the wave
// that owns the MFMA stream is NOT hidden behind it (k-step = 256 cycles of MFMA + ~3.6 cycles per VALU
// instruction, additive); a second wave on the SIMD is what overlaps one wave's activation with the other's MFMAs.
//
// One buffer_load per MFMA is NOT viable (measured: 64 cycles per 32-cycle MFMA), so the stream goes through LDS:
// 32 KB chunks (2 output blocks) by LDS-DMA into a 4-slot ring, three chunks in flight, one s_barrier per chunk,
// read back with one ds_read_b128 per MFMA.  Plain f32 VALU only: v_pk_*_f32 beside bf16 MFMAs is an anti-lever
// (measured +8.5 cycles each) — this TU is built with -fno-slp-vectorize so the compiler does not pack either.
//
// 512 wide (cfg3's FQL net; SURVEY 7 "hard parts": bf16 weights are what lets the 3 x 512 x 512 stream sit in an XCD's L2):
// the same body with 16 output blocks of 32 k-steps.  The packed inputs and outputs of a layer are then 2 x 128 registers,
// so this build runs ONE wave per SIMD (512 registers) on 128-chain workgroup tiles: four waves share the stream, each
// DMAs 8 of a chunk's 32 pieces.  A hidden layer is 16 chunks (one output block each), Dense_L fills one chunk.
#include "frl_bf16.h"
#include "frl_rng.h"

#ifndef FRL_BF_RING_H512
#define FRL_BF_RING_H512 8   // A/B at cfg3 (gelu-tanh): 4 -> 20.96 ms, 8 -> 19.4 .. 19.9, 12 / 16 no further gain
#endif
#ifndef FRL_BF_TICK
#define FRL_BF_TICK 3          // one refill DMA per three MFMAs (2 and 1: no better)
#endif

namespace frl {

// ---- weight stream image ------------------------------------------------------------------------------------
struct PBfArgs {
  const float* kernel[FRL_MAX_HIDDEN + 1];
  unsigned short* img;
  int A, C, Dp, L, nxk, k0, l0c, ns;   // k0 = layer-0 k-steps per output block (x k-steps + compiled cond k-steps)
  int H;
};

__global__ void __launch_bounds__(256) frl_prepare_bf16_kernel(const PBfArgs p) {
  const int H = p.H, NOB = H >> 5, NKH = H >> 4, HC = (NOB * NKH) >> 5;   // output blocks, hidden k-steps, chunks per hidden layer
  const long long total = (long long)p.ns * (BF_CHUNK_BYTES / 2);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / (BF_CHUNK_BYTES / 2));
    const int within = (int)(idx - (long long)c * (BF_CHUNK_BYTES / 2));
    const int piece = within >> 9, lane = (within >> 3) & 63, e = within & 7;
    const int i = lane & 31, s = lane >> 5;
    float v = 0.0f;
    if (c < p.l0c) {                                   // Dense_0: per output block, x k-steps then cond k-steps
      const int pi = c * 32 + piece;                   //   (time rows are in tb; k-steps beyond the data are zero)
      const int ob = pi / p.k0, ks = pi - ob * p.k0;
      if (ob < NOB) {
        if (ks < p.nxk) {
          const int row = BF_ROWR(8 * ks + e) + 4 * s;
          if (row < p.A) v = p.kernel[0][(size_t)row * H + 32 * ob + i];
        } else {
          const int ci = 16 * (ks - p.nxk) + 8 * s + e;
          if (ci < p.C) v = p.kernel[0][(size_t)(p.A + p.Dp + ci) * H + 32 * ob + i];
        }
      }
    } else if (c < p.l0c + HC * (p.L - 1)) {           // hidden Dense_l: NOB x NKH pieces, output-block-major (at 256: chunk =
      const int hc = c - p.l0c, l = 1 + hc / HC;       //   output blocks 2 hc, 2 hc + 1; at 512: one block per chunk)
      const int pl = (hc - (l - 1) * HC) * 32 + piece;
      const int ob = pl / NKH, ks = pl - ob * NKH;
      const int feat = 32 * (ks >> 1) + BF_ROWR(8 * (ks & 1) + e) + 4 * s;
      v = p.kernel[l][(size_t)feat * H + 32 * ob + i];
    } else if (piece < NKH) {                          // Dense_L, one output block (rows >= A are zero)
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
  p.nxk = pl.bf_nxk; p.k0 = pl.bf_nxk + pl.bf_cks; p.l0c = pl.bf_l0c; p.ns = pl.bf_ns; p.H = pl.H;
  hipLaunchKernelGGL(frl_prepare_bf16_kernel, dim3(pl.H == 256 ? 64 : 256), dim3(256), 0, stream, p);
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

// One Dense layer, output-block-major: for each of the NOB output blocks, NK MFMAs into one accumulator; the
// activation + packing of the previous block (-> bout[2 ob], bout[2 ob + 1]) is issued alongside.  Pieces are read
// through a 4-deep register ring; a chunk boundary (barrier + next DMA) every 32 pieces.
template <int ACT, int NK, int NOB, typename BFun, typename NextChunk, typename Tick>
__device__ __forceinline__ void layer_bf(u32x4 (&bout)[2 * NOB], BFun&& bfun, const float* sb, NextChunk& next_chunk, Tick& dma_tick,
                                         unsigned lane16, int ls) {
  constexpr int NP = NOB * NK, R = (NOB == 16) ? FRL_BF_RING_H512 : 4;    // pieces the register ring runs ahead of the MFMAs
  f32x16 acc[2];
  u32x4 ring[R];
  const char* sl = next_chunk() + lane16;
  const char* sl_next = sl;
  sfor<R>([&](auto PI) { constexpr int p = decltype(PI)::value; ring[p] = *(const u32x4*)(sl + p * 1024); });
  sfor<NOB>([&](auto OI) {
    constexpr int ob = decltype(OI)::value;
    sfor<4>([&](auto VI) {                               // accumulator init = bias (permuted table: 4 x 16 B per block)
      constexpr int v = decltype(VI)::value;
      const f32x4 bv = *(const f32x4*)(sb + (ls * NOB + ob) * 16 + 4 * v);
      sfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; acc[ob & 1][4 * v + j] = bv[j]; });
    });
    if constexpr (ob > 0) {
      bout[2 * (ob - 1)] = pack_act<ACT, 0>(acc[(ob - 1) & 1]);
      bout[2 * (ob - 1) + 1] = pack_act<ACT, 8>(acc[(ob - 1) & 1]);
    }
    sfor<NK>([&](auto KI) {
      constexpr int ks = decltype(KI)::value;
      constexpr int p = ob * NK + ks;
      acc[ob & 1] = mfma_bf(ring[p % R], bfun(KI), acc[ob & 1]);
      if constexpr (p % FRL_BF_TICK == 0) dma_tick();    // (512-wide build) one refill piece per FRL_BF_TICK MFMAs
      if constexpr (p + R < NP) {
        constexpr int q = p + R;
        if constexpr ((q & 31) == 0) sl_next = next_chunk() + lane16;   // the ring runs R pieces ahead of the MFMAs
        ring[p % R] = *(const u32x4*)(sl_next + (q & 31) * 1024);
      }
    });
    if constexpr (ob > 0) {
      sfor<NK>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // ds_read (piece p + R)
        __builtin_amdgcn_sched_group_barrier(0x002, (16 * 9 + NK - 1) / NK, 0);   // VALU (activation of block ob - 1)
      });
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  bout[2 * NOB - 2] = pack_act<ACT, 0>(acc[(NOB - 1) & 1]);
  bout[2 * NOB - 1] = pack_act<ACT, 8>(acc[(NOB - 1) & 1]);
}

template <int H, int NW, int ACT, int NRX, int CKS>
__device__ __forceinline__ void sampler_bf16_body(const KArgs& a) {
  constexpr int NXK = NRX / 8, K0 = NXK + CKS, NT = 64 * NW;
  constexpr int NOB = H / 32, NKH = H / 16;     // output blocks per layer, k-steps of a hidden layer
  constexpr int PPW = 32 / NW;                  // pieces of a chunk each wave DMAs (4 or 8)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, ls = lane >> 5;

  // bias tables, permuted so that a lane half reads the 16 accumulator inits of a block as 4 x 16 B:
  // tbl[(s * 8 + ob) * 16 + r] = bias[32 ob + ROWR(r) + 4 s]
  float* s_tb = (float*)(smem + BF_NBUF * BF_CHUNK_BYTES);   // [T][H]
  float* s_bh = s_tb + a.T * H;                               // [L-1][H]
  float* s_bl = s_bh + (a.L - 1) * H;                         // [32] padded last-layer bias (natural order)
  float* s_coef = s_bl + LAST_PAD;                            // [T][8]
  auto perm_src = [](int p) { const int s = p / (16 * NOB), ob = (p >> 4) & (NOB - 1), r = p & 15; return 32 * ob + BF_ROWR(r) + 4 * s; };
  for (int idx = tid; idx < a.T * H; idx += NT) s_tb[idx] = a.ws_tb[(idx & ~(H - 1)) + perm_src(idx & (H - 1))];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr_rs(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += NT) s_bh[(l - 1) * H + idx] = bp[perm_src(idx)];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.T * COEF_STRIDE; idx += NT) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  // ---- weight stream: the same bf_ns chunks every step of every tile, three in flight ----------------------
  const int NS = a.bf_ns;
  const char* img = (const char*)a.ws_bf16 + wave * (PPW * 1024) + lane * 16;   // this wave DMAs pieces PPW w .. PPW w + PPW - 1 of a chunk
  auto issue = [&](int seq, unsigned boff) {
    const char* src = img + (size_t)seq * BF_CHUNK_BYTES;
    char* dst = smem + boff + wave * (PPW * 1024);
    sfor<PPW / 4>([&](auto GI) {     // the instruction's immediate offset reaches 4 pieces; groups of 4 move both bases
      constexpr int g = decltype(GI)::value;
      sfor<4>([&](auto II) { constexpr int i = decltype(II)::value; dma16_bf<i * 1024>(src + g * 4096, dst + g * 4096); });
    });
  };
  // One wave per SIMD (the 512-wide build) feels every DMA instruction: eight in a row behind the chunk barrier cost 3.5 of
  // 21 ms at cfg3 (measured with the DMA compiled out).  There the refill of the freed slot is SPREAD: next_chunk only
  // records it, dma_tick() issues one piece per three MFMAs, and the next chunk boundary flushes what is left.
  constexpr bool SPREAD = (NW == 4);
  const char* dma_src = img;
  char* dma_dst = smem;
  int dma_left = 0;
  auto dma_tick = [&]() {
    if constexpr (SPREAD) {
      if (dma_left > 0) {
        dma16_bf<0>(dma_src, dma_dst);
        dma_src += 1024; dma_dst += 1024; --dma_left;
      }
    }
  };
  auto wrap_seq = [&](int s) { return s >= NS ? s - NS : s; };
  issue(0, 0);
  issue(wrap_seq(1), BF_CHUNK_BYTES);
  issue(wrap_seq(wrap_seq(1) + 1), 2 * BF_CHUNK_BYTES);
  int seq_issue = wrap_seq(wrap_seq(wrap_seq(1) + 1) + 1);
  unsigned boff_issue = 3 * BF_CHUNK_BYTES, boff_cur = 0;
  // start of a chunk's use: its DMA has landed (all but the two newer chunks' 8 DMAs of this wave are done), every
  // wave is past the previous chunk (barrier), so that chunk's slot can take the chunk three ahead.
  auto next_chunk = [&]() -> const char* {
    if constexpr (SPREAD) { sfor<PPW>([&](auto) { dma_tick(); }); }     // the previous refill's remainder, before the wait counts it
#if !defined(FRL_BF_EXP_NOBAR)   // (timing experiments only: results are wrong without the wait / barrier / DMA)
    if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
#if !defined(FRL_BF_EXP_NODMA)
    if constexpr (SPREAD) {
      dma_src = img + (size_t)seq_issue * BF_CHUNK_BYTES;
      dma_dst = smem + boff_issue + wave * (PPW * 1024);
      dma_left = PPW;
    } else {
      issue(seq_issue, boff_issue);
    }
#endif
    seq_issue = (seq_issue + 1 == NS) ? 0 : seq_issue + 1;
    boff_issue = (boff_issue + BF_CHUNK_BYTES) & (BF_NBUF * BF_CHUNK_BYTES - 1);
    const char* p = smem + boff_cur;
    boff_cur = (boff_cur + BF_CHUNK_BYTES) & (BF_NBUF * BF_CHUNK_BYTES - 1);
    return p;
  };

  const unsigned lane16 = lane * 16;
  const int ntiles = (int)((a.B + NW * 32 - 1) / (NW * 32));
  u32x4 bin[NKH], bout[NKH];    // packed bf16 layer inputs / outputs: k-step ks <-> features of block ks >> 1, half ks & 1

#ifdef FRL_BF_EXP_TIMING
  long long tsec[6] = {0, 0, 0, 0, 0, 0}; long long tmark = __builtin_amdgcn_s_memtime();
#define FRL_TMARK(k) { const long long tn = __builtin_amdgcn_s_memtime(); tsec[k] += tn - tmark; tmark = tn; }
#else
#define FRL_TMARK(k)
#endif
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long chain = (long long)tile * (NW * 32) + wave * 32 + li;
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
    FRL_TMARK(0)

    for (int j = 0; j < a.T; ++j) {
      FRL_TMARK(1)
      // ---------------- Dense_0 : h_0 = act(tb[j] + W0x^T x + W0c^T cond) -> bin ----------------------
      {
        u32x4 bx[NXK];
        sfor<NXK>([&](auto QI) {
          constexpr int q = decltype(QI)::value;
          sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; bx[q][w] = pack2(xs[8 * q + 2 * w], xs[8 * q + 2 * w + 1]); });
        });
        auto b0 = [&](auto KI) -> u32x4 {
          constexpr int ks = decltype(KI)::value;
          if constexpr (ks < NXK) return bx[ks]; else return cbp[ks - NXK];
        };
        layer_bf<ACT, K0, NOB>(bin, b0, s_tb + j * H, next_chunk, dma_tick, lane16, ls);
      }
      FRL_TMARK(2)

      // ---------------- hidden Dense_l : h_l = act(b_l + W_l^T h_{l-1}) -------------------------------
      for (int l = 1; l < a.L; ++l) {
        auto bh = [&](auto KI) -> u32x4 { return bin[decltype(KI)::value]; };
        layer_bf<ACT, NKH, NOB>(bout, bh, s_bh + (l - 1) * H, next_chunk, dma_tick, lane16, ls);
        sfor<NKH>([&](auto KI) { constexpr int k = decltype(KI)::value; bin[k] = bout[k]; });
      }
      FRL_TMARK(3)

      // ---------------- Dense_L : O = b_L + W_L^T h_{L-1}, one output block ----------------------------
      f32x16 O;
      sfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; O[r] = s_bl[BF_ROWR(r) + 4 * ls]; });
      // this step's noise: requested at the head of Dense_L (16 MFMAs before its use) instead of in the epilogue, where the
      // wait for it also drained the stream's DMA pieces issued since (vector-memory results return in issue order).
      // A/B'd and dropped in round 3 (same box, cfg2 mish, this kernel 0.534 ms): (1) issuing the refill pieces one per 8
      // MFMAs instead of four at the chunk boundary — no gain; (2) a "ping-pong" build: every wave alternates PURE segments
      // (16 MFMAs of a block | its activation), one s_barrier per segment, waves 4..7 one segment behind so that the two
      // waves of a SIMD are always in opposite phases — 0.664 ms (relu 0.451 vs 0.363; pairing waves (w, w ^ 1) or
      // (w, w ^ 2) instead: 0.94 ms, so w and w + 4 do share a SIMD): a wave alone issues its 16 dependent MFMAs + ds_reads
      // at ~140 cycles each, and twice the barriers cost more than the phase overlap returns.
      float zn[NRX];
      {
        const char* sl = next_chunk() + lane16;
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = BF_ROWR(r) + 4 * ls;
          zn[r] = (a.noise != nullptr && valid && ai < a.A) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
        u32x4 awl[4];
        sfor<4>([&](auto KI) { constexpr int ks = decltype(KI)::value; awl[ks] = *(const u32x4*)(sl + ks * 1024); });
        sfor<NKH>([&](auto KI) {
          constexpr int ks = decltype(KI)::value;
          O = mfma_bf(awl[ks & 3], bin[ks], O);
          if constexpr (ks % FRL_BF_TICK == 0) dma_tick();
          if constexpr (ks + 4 < NKH) awl[ks & 3] = *(const u32x4*)(sl + (ks + 4) * 1024);
        });
      }
      FRL_TMARK(4)

      // ---------------- epilogue: one reverse-diffusion / Euler step (f32) ---------------------------
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = BF_ROWR(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        const float eps = O[r], xt = xs[r];
        float z = zn[r];
        if (a.rng_impl != FRL_RNG_NONE && in && c5 != 0.0f) {   // in-kernel jax.random.normal(key_j, xt.shape)[chain, ai]; the
                                                                   // last iteration multiplies its draw by 0 (ddpm.py:206): skipped
          z = rng_step_normal(a.rng_keys, a.rng_impl, j, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai);
          __builtin_amdgcn_sched_barrier(0);       // one draw at a time: interleaved draws cost registers the hot loop needs
        }
        if (a.xt_hist != nullptr && in) {
          const long long o = ((long long)j * a.B + chain) * a.A + ai;
          a.xt_hist[o] = xt;
          a.eps_hist[o] = eps;
        }
        const float xn = bf_solver_update(a.mode, a.clip, a.x_min, a.x_max, c0, c1, c2, c3, c4, c5, xt, eps, z);
        xs[r] = in ? xn : 0.0f;
      });
    }

    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = BF_ROWR(r) + 4 * ls;
      if (valid && ai < a.A) a.x0[chain * a.A + ai] = xs[r];
    });
  }
#ifdef FRL_BF_EXP_TIMING
  FRL_TMARK(5)
  if (blockIdx.x == 0 && tid == 0) for (int k = 0; k < 6; ++k) a.x0[k] = (float)tsec[k];
#endif
  // the stream runs three chunks ahead: drain the prefetches before the wave ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// 256 wide: eight waves per workgroup, two per SIMD (256 registers each), 256-chain workgroup tiles
template <int ACT, int NRX, int CKS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
frl_sampler_bf16_kernel(const KArgs a) {
  sampler_bf16_body<256, 8, ACT, NRX, CKS>(a);
}

// 512 wide: four waves per workgroup, one per SIMD (512 registers), 128-chain workgroup tiles; x_dim <= 16, cond_dim <= 64
template <int ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_bf16_h512_kernel(const KArgs a) {
  sampler_bf16_body<512, 4, ACT, 8, 4>(a);
}

// ---- launcher ------------------------------------------------------------------------------------------------
template <int ACT, int NRX, int CKS>
static int launch_inst_bf(const KArgs& a, int grid, size_t lds, hipStream_t stream) {
  auto kern = frl_sampler_bf16_kernel<ACT, NRX, CKS>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_sampler_bf16_kernel<act %d, NRX %d, CKS %d> [grid %d]", ACT, NRX, CKS, grid);
  return FRL_OK;
}

template <int ACT>
static int launch_h512_bf(const KArgs& a, int grid, size_t lds, hipStream_t stream) {
  if (a.bf_nxk != 1 || a.bf_cks != 4) { set_error("bf16 512-wide kernel: x_dim %d / cond_dim %d outside its operand budget (x_dim <= 16, cond_dim <= 64)", a.A, a.C); return FRL_EUNSUPPORTED; }
  auto kern = frl_sampler_bf16_h512_kernel<ACT>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_sampler_bf16_h512_kernel<act %d> [grid %d]", ACT, grid);
  return FRL_OK;
}

template <int ACT>
static int launch_act_bf(const KArgs& a, int grid, size_t lds, hipStream_t stream) {
  if (a.H == 512) return launch_h512_bf<ACT>(a, grid, lds, stream);
  if (a.bf_nxk == 1 && a.bf_cks == 2) return launch_inst_bf<ACT, 8, 2>(a, grid, lds, stream);
  if (a.bf_nxk == 2 && a.bf_cks == 8) return launch_inst_bf<ACT, 16, 8>(a, grid, lds, stream);
  set_error("bf16 kernel: x_dim %d / cond_dim %d outside the compiled operand budgets (x_dim <= 32, cond_dim <= 128)", a.A, a.C);
  return FRL_EUNSUPPORTED;
}

size_t bf16_lds_bytes(int T, int L, int H) {
  return (size_t)BF_NBUF * BF_CHUNK_BYTES + sizeof(float) * ((size_t)T * H + (size_t)(L - 1) * H + LAST_PAD + (size_t)T * COEF_STRIDE);
}

int launch_sampler_bf16(const FrlDesc& d, const KArgs& a, hipStream_t stream) {
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
  if (a.H != 256 && a.H != 512) { set_error("bf16 kernel: hidden width %d has no build (256 / 512)", a.H); return FRL_EUNSUPPORTED; }
  const int tile = (a.H == 256) ? 256 : 128;     // chains per workgroup
  const long long ntiles = (a.B + tile - 1) / tile;
  const int grid = (int)(ntiles < ncu ? ntiles : ncu);
  const size_t lds = bf16_lds_bytes(a.T, a.L, a.H);
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
