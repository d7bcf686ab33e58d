// native_bench — torch-free driver of the cfg2 sampler against include/flowrl_hip.h only.
//
//   bench/native_bench [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-gather]
//
// One process, one HIP stream per GPU, chains sharded by state index (weak scaling: B chains per GPU), weights
// replicated, the only collective the final ncclAllGather of x0 over RCCL (ncclCommInitAll, one ncclGroup per step).
// Device memory comes from hipMalloc, the library is called through its C ABI exactly as a jax.ffi handler would call
// it.  Workload = BASELINE.json configs[1] (DQL/DAC hopper: obs 11, act 3, 256x3 mish, positional
// time embedding, vp schedule, T = 5, clip +-1, 262 144 chains per GPU), synthetic N(0,1) inputs, scaled-normal weights
// (throughput does not depend on their values).  Prints one JSON line; on 1 GPU it reproduces bench.py's ms_per_step.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "flowrl_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, ncclGetErrorString(r_)); exit(1); } } while (0)
#define FRLCHECK(x) do { int r_ = (x); if (r_ != FRL_OK) { fprintf(stderr, "%s:%d: libflowrl_hip error %d: %s\n", __FILE__, __LINE__, r_, frl_last_error()); exit(1); } } while (0)

static float* upload(const std::vector<float>& h) {
  float* d = nullptr;
  HIPCHECK(hipMalloc((void**)&d, h.size() * sizeof(float)));
  HIPCHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
static std::vector<float> normal(std::mt19937_64& g, size_t n, float std) {
  std::normal_distribution<float> nd(0.0f, std);
  std::vector<float> v(n);
  for (auto& x : v) x = nd(g);
  return v;
}

struct Gpu {
  hipStream_t stream;
  FrlParams params;
  FrlSchedule sched;
  FrlFwdBufs bufs;
  void* ws; size_t ws_bytes;
  float* gathered;
};

int main(int argc, char** argv) {
  int gpus = 1, steps = 100, warmup = 20, gather = 1;
  long long B = 262144;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--batch") && i + 1 < argc) B = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--no-gather")) gather = 0;
    else { fprintf(stderr, "usage: %s [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-gather]\n", argv[0]); return 2; }
  }
  int have = 0;
  HIPCHECK(hipGetDeviceCount(&have));
  if (gpus < 1 || have < gpus) { fprintf(stderr, "native_bench: --gpus %d asked for, %d GPU(s) visible\n", gpus, have); return 2; }
  if (frl_abi_version() != FRL_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }

  constexpr int A = 3, S = 11, T = 5, D = 64, H = 256, L = 3;
  FrlDesc d;
  memset(&d, 0, sizeof(d));
  d.abi_version = FRL_ABI_VERSION; d.kind = FRL_KIND_DDPM; d.solver = FRL_SOLVER_DDPM; d.steps = T;
  d.n_states = B; d.particles = 1; d.x_dim = A; d.cond_dim = S; d.time_embed = FRL_TEMB_POSITIONAL; d.time_dim = D;
  d.n_hidden = L; for (int l = 0; l < L; ++l) d.hidden[l] = H;
  d.act = FRL_ACT_MISH; d.clip = 1; d.x_min = -1.0f; d.x_max = 1.0f; d.dtype = FRL_DTYPE_F32; d.tile_policy = FRL_TILE_AUTO;
  FRLCHECK(frl_validate(&d));

  // vp schedule, float32, the reference's op order (flowrl/flow/ddpm.py:36-42,132-162; exact posterior variance, DQL)
  std::vector<float> betas(T + 1), alphas(T + 1), ahat(T + 1), postvar(T + 1);
  betas[0] = 0.0f;
  for (int t = 1; t <= T; ++t) {
    const float al = expf(-0.1f / T - 0.5f * (10.0f - 0.1f) * (2.0f * t - 1.0f) / (float)(T * T));
    betas[t] = 1.0f - al;
  }
  float cp = 1.0f;
  for (int t = 0; t <= T; ++t) { alphas[t] = 1.0f - betas[t]; cp *= alphas[t]; ahat[t] = cp; }
  postvar[0] = 0.0f;
  for (int t = 1; t <= T; ++t) postvar[t] = betas[t] * (1.0f - ahat[t - 1]) / (1.0f - ahat[t]);

  std::vector<Gpu> g(gpus);
  std::vector<int> devs(gpus);
  for (int r = 0; r < gpus; ++r) devs[r] = r;
  std::vector<ncclComm_t> comms(gpus);
  if (gpus > 1) NCCLCHECK(ncclCommInitAll(comms.data(), gpus, devs.data()));

  const int K0 = A + D + S;
  for (int r = 0; r < gpus; ++r) {
    HIPCHECK(hipSetDevice(r));
    HIPCHECK(hipStreamCreate(&g[r].stream));
    std::mt19937_64 wg(0);                      // the same weights on every GPU (replicated)
    FrlParams& p = g[r].params;
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < 2; ++k) { p.tmlp_kernel[k] = upload(normal(wg, (size_t)D * D, 1.0f / sqrtf((float)D))); p.tmlp_bias[k] = upload(normal(wg, D, 0.1f)); }
    const int din[L + 1] = {K0, H, H, H}, dout[L + 1] = {H, H, H, A};
    for (int l = 0; l <= L; ++l) {
      p.kernel[l] = upload(normal(wg, (size_t)din[l] * dout[l], sqrtf(2.0f / din[l])));
      p.bias[l] = upload(normal(wg, dout[l], 0.1f));
    }
    FrlSchedule& s = g[r].sched;
    memset(&s, 0, sizeof(s));
    s.betas = upload(betas); s.alphas = upload(alphas); s.alpha_hats = upload(ahat); s.postvars = upload(postvar);
    std::mt19937_64 ig(1234 + r);               // this GPU's shard of the states
    FrlFwdBufs& b = g[r].bufs;
    memset(&b, 0, sizeof(b));
    b.xT = upload(normal(ig, (size_t)B * A, 1.0f));
    b.cond = upload(normal(ig, (size_t)B * S, 1.0f));
    b.noise = upload(normal(ig, (size_t)T * B * A, 1.0f));
    g[r].gathered = nullptr;
    if (gpus > 1 && gather) {
      HIPCHECK(hipMalloc((void**)&g[r].gathered, (size_t)gpus * B * A * sizeof(float)));
      b.x0 = g[r].gathered + (size_t)r * B * A;  // in-place all-gather: this rank's block of the gathered buffer
    } else {
      HIPCHECK(hipMalloc((void**)&b.x0, (size_t)B * A * sizeof(float)));
    }
    g[r].ws_bytes = frl_workspace_bytes(&d);
    HIPCHECK(hipMalloc(&g[r].ws, g[r].ws_bytes));
  }

  auto step = [&]() {
    for (int r = 0; r < gpus; ++r) {
      HIPCHECK(hipSetDevice(r));
      FRLCHECK(frl_sampler_fwd(&d, &g[r].params, &g[r].sched, &g[r].bufs, g[r].ws, g[r].ws_bytes, g[r].stream));
    }
    if (gpus > 1 && gather) {
      NCCLCHECK(ncclGroupStart());
      for (int r = 0; r < gpus; ++r)
        NCCLCHECK(ncclAllGather(g[r].bufs.x0, g[r].gathered, (size_t)B * A, ncclFloat, comms[r], g[r].stream));
      NCCLCHECK(ncclGroupEnd());
    }
  };
  auto sync_all = [&]() { for (int r = 0; r < gpus; ++r) { HIPCHECK(hipSetDevice(r)); HIPCHECK(hipStreamSynchronize(g[r].stream)); } };

  for (int i = 0; i < warmup; ++i) step();
  sync_all();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < steps; ++i) step();
  sync_all();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // sanity: every x0 finite and inside the clip range; checksum of GPU 0's shard
  HIPCHECK(hipSetDevice(0));
  std::vector<float> h((size_t)B * A);
  HIPCHECK(hipMemcpy(h.data(), g[0].bufs.x0, h.size() * sizeof(float), hipMemcpyDeviceToHost));
  double sum = 0.0; int bad = 0;
  for (float v : h) { if (!std::isfinite(v) || fabsf(v) > 8.0f) ++bad; sum += v; }
  const char* kern = frl_last_kernels();
  double ex = 0.0;
  const double fa = frl_algorithmic_flops(&d, &ex);
  const double ms = dt / steps * 1e3;
  printf("{\"metric\": \"action samples/sec x T steps (chain-steps/s), fused sampler, native driver (no torch)\", \"value\": %.6e, "
         "\"unit\": \"chain-steps/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, \"higher_is_better\": true, "
         "\"scaling\": \"weak\", \"dtype\": \"f32\", \"data\": \"synthetic\", \"config\": {\"workload\": \"cfg2 (DQL/DAC hopper, 256x3 mish, vp, T=5), "
         "%lld chains per GPU\", \"collective\": \"%s\"}, \"tflops_algorithmic_per_gpu\": %.3f, \"frac_f32_mfma_peak\": %.4f, "
         "\"kernel\": \"%s\", \"x0_checksum_gpu0\": %.9e, \"x0_bad\": %d}\n",
         (double)gpus * B * T / (dt / steps), gpus, steps, warmup, ms, B, (gpus > 1 && gather) ? "ncclAllGather(x0)" : "none",
         fa / (ms * 1e-3) / 1e12, fa / (ms * 1e-3) / 1e12 / 157.3, kern, sum, bad);
  if (gpus > 1) for (int r = 0; r < gpus; ++r) ncclCommDestroy(comms[r]);
  return bad ? 1 : 0;
}
