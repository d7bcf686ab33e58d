"""bench.py — headline benchmark of the fused action sampler (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2]

One "step" = one full sampler call (frl_sampler_prepare + the fused T-step kernel) over one batch of
synthetic chains that is already resident in HBM.  Default workload = BASELINE.json configs[1]
("cfg2": DQL/DAC hopper shape, obs 11, act 3, 256x3 mish MLP, vp schedule, T=5, clip +-1,
B = 262,144 chains per GPU).  Multi-GPU: one process per GPU (torch.distributed / RCCL), chains
sharded by state index (weak scaling: every rank owns B chains), the only collective is the final
all-gather of x0 (BASELINE.json north_star), inside the timed region.

Launching: `python bench.py --gpus N` with no WORLD_SIZE in the environment is the PARENT: before any GPU call it
starts N rank processes of this same file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set,
one device each), forwards rank 0's JSON line and exits with the first non-zero rank status; it fails loudly when the
box has fewer than N GPUs.  Under `python -m torch.distributed.run ... bench.py --gpus N` the environment already
names the rank, and the process is a rank itself (WORLD_SIZE must equal --gpus).

Prints ONE JSON line on rank 0.  value = chain-steps/s = (ranks * B * T) / (time per step).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT / "flow-rl_amd")]

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (v_mfma_f32_32x32x16_bf16, 32 cycles)

WORKLOADS = {
    # BASELINE.json configs[0..3]; B = chains per GPU (n_states * particles)
    "cfg1": dict(desc="toy2d DDPM (obs 1, act 2, 256x3 mish, linear, T=20, B=4096)", kind="ddpm",
                 A=2, S=1, T=20, B=4096, particles=1, hidden=[256, 256, 256], act="mish", temb="positional",
                 schedule="linear", clip=False, x_min=-4.0, x_max=4.0, cond_embed=False),
    "cfg2": dict(desc="DQL/DAC DDPM hopper (obs 11, act 3, 256x3 mish, vp, T=5, clip +-1, B=262144/GPU)", kind="ddpm",
                 A=3, S=11, T=5, B=262144, particles=1, hidden=[256, 256, 256], act="mish", temb="positional",
                 schedule="vp", clip=True, x_min=-1.0, x_max=1.0, cond_embed=False),
    "cfg3": dict(desc="FQL flow Euler (obs 39, act 8, 512x4 gelu-tanh, raw t, 10 steps, B=1048576/GPU)", kind="cnf",
                 A=8, S=39, T=10, B=1048576, particles=1, hidden=[512, 512, 512, 512], act="gelu_tanh", temb=None,
                 schedule=None, clip=False, x_min=None, x_max=None, cond_embed=False),
    "cfg4": dict(desc="DACER ContinuousDDPM humanoid (obs 376 -> cond-embed 128, act 17, 256x3 mish, Fourier-64, cosine, "
                      "T=20, clip +-1, 8192 states x 64 particles/GPU)", kind="cddpm",
                 A=17, S=376, T=20, B=8192 * 64, particles=64, hidden=[256, 256, 256], act="mish", temb="learnable_fourier",
                 schedule="cosine", clip=True, x_min=-1.0, x_max=1.0, cond_embed=True),
    # cfg5 (--dtype bf16 = the variant BASELINE names; f32 = the bit-faithful one): forward with history + chain VJP + (N>1) one all-reduce of the grads
    "cfg5": dict(desc="DQL backprop-through-chain VJP (cfg2 shapes: obs 11, act 3, 256x3 mish, vp, T=5, clip +-1, B=262144/GPU), "
                      "forward+backward", kind="ddpm", A=3, S=11, T=5, B=262144, particles=1, hidden=[256, 256, 256],
                 act="mish", temb="positional", schedule="vp", clip=True, x_min=-1.0, x_max=1.0, cond_embed=False, vjp=True),
    # residual backbones, not BASELINE configs: DAC / BDPO with diffusion.resnet (dac.py:200-207) at the
    # cfg2 geometry, and backbone_cls: simba behind DACER's embeddings (dacer.py:247-266)
    "res_dac": dict(desc="DAC resnet DDPM hopper (obs 11, act 3, ResidualMLP 2 x [256 -> 1024 -> 256] mish + LayerNorm, vp, T=5, "
                         "clip +-1, B=262144/GPU)", kind="ddpm", A=3, S=11, T=5, B=262144, particles=1, hidden=[256, 256], act="mish",
                    temb="positional", schedule="vp", clip=True, x_min=-1.0, x_max=1.0, cond_embed=False, backbone="residual", ln=True),
    "simba_dacer": dict(desc="DACER simba ContinuousDDPM humanoid (obs 376 -> cond-embed 128, act 17, Simba 2 blocks relu, Fourier-64, "
                             "cosine, T=20, clip +-1, 1024 states x 64 particles/GPU)", kind="cddpm", A=17, S=376, T=20, B=1024 * 64,
                        particles=64, hidden=[256, 256], act="relu", temb="learnable_fourier", schedule="cosine", clip=True,
                        x_min=-1.0, x_max=1.0, cond_embed=True, backbone="simba"),
}


def build_model(w, device, seed=0):
    import flowrl_hip as fh
    from flowrl_hip.flow import DDPM, ContinuousDDPM, ContinuousNormalizingFlow
    te = {"positional": fh.PositionalEmbedding(output_dim=64), "learnable_fourier": fh.LearnableFourierEmbedding(output_dim=64),
          None: None}[w["temb"]]
    if w.get("backbone") == "residual":
        mlp = fh.ResidualMLP(hidden_dims=w["hidden"], output_dim=w["A"], activation=w["act"], layer_norm=bool(w.get("ln")))
    elif w.get("backbone") == "simba":
        mlp = fh.Simba(hidden_dims=w["hidden"], output_dim=w["A"], activation=w["act"])
    else:
        mlp = fh.MLP(hidden_dims=w["hidden"], output_dim=w["A"], activation=w["act"], layer_norm=bool(w.get("ln")))
    ce = fh.MLP(hidden_dims=[128, 128], activation=w["act"]) if w["cond_embed"] else None
    inputs = (np.ones((1, w["A"])), np.zeros((1, 1)), np.ones((1, w["S"])))
    if w["kind"] == "cnf":
        net = fh.FlowBackbone(vel_predictor=mlp, time_embedding=te, cond_embedding=ce)
        params = fh.init_params(net, w["A"], w["S"], seed=seed, bias_std=0.1, device=device)
        return ContinuousNormalizingFlow.create(net, 0, inputs, x_dim=w["A"], steps=w["T"], clip_sampler=w["clip"],
                                                x_min=w["x_min"], x_max=w["x_max"], params=params, device=device)
    net = fh.DDPMBackbone(noise_predictor=mlp, time_embedding=te, cond_embedding=ce)
    params = fh.init_params(net, w["A"], w["S"], seed=seed, bias_std=0.1, device=device)
    if w["kind"] == "ddpm":
        return DDPM.create(net, 0, inputs, x_dim=w["A"], steps=w["T"], noise_schedule=w["schedule"],
                           clip_sampler=w["clip"], x_min=w["x_min"], x_max=w["x_max"], params=params, device=device)
    return ContinuousDDPM.create(net, 0, inputs, x_dim=w["A"], steps=w["T"], noise_schedule=w["schedule"],
                                 clip_sampler=w["clip"], x_min=w["x_min"], x_max=w["x_max"], params=params, device=device)


def make_inputs(w, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    B, A, S, T = w["B"], w["A"], w["S"], w["T"]
    xT = torch.randn((B, A), generator=g, device=device)
    cond = torch.randn((B // w["particles"], S), generator=g, device=device)     # un-repeated: one row per state
    noise = torch.randn((T, B, A), generator=g, device=device) if w["kind"] != "cnf" else None
    return xT, cond, noise


def sample(model, w, xT, cond, noise):
    p = w["particles"] if w["particles"] > 1 else None
    cd = w.get("dtype", "f32")
    if w["kind"] == "cnf":
        return model.sample(None, xT, cond, particles=p, compute_dtype=cd, tile_policy=w.get("tile_policy", "auto"), cache_prepare=w.get("cache_prepare", False))[1]
    if w.get("rng", "none") != "none":
        # the acting loop's shape: `rng, x0 = sample(rng, ...)` — the key is a device array, the next call's key is the rng this
        # one returns (a PRNGKey built from a host integer per step would put a pageable H2D copy in front of every call)
        if w.get("_key") is None:
            from flowrl_hip import random as fr
            w["_key"] = fr.PRNGKey(0, xT.device)
        out = model.sample(w["_key"], xT, cond, False, "ddpm", particles=p, compute_dtype=cd, tile_policy=w.get("tile_policy", "auto"), rng_impl=w["rng"], cache_prepare=w.get("cache_prepare", False), rng_shard=w.get("rng_shard"))
        w["_key"] = out[0]
        return out[1]
    return model.sample(None, xT, cond, False, "ddpm", noise=noise, particles=p, compute_dtype=cd, tile_policy=w.get("tile_policy", "auto"), cache_prepare=w.get("cache_prepare", False))[1]


def vjp_step(model, w, xT, cond, noise, g):
    """One DQL actor-update worth of sampler work: forward keeping history[0], then the chain VJP."""
    cd = w.get("dtype", "f32")
    stash = not w.get("recompute")     # the training forward keeps the stash, the backward only sweeps
    _, x0, hist = model.sample(None, xT, cond, True, "ddpm", params=model.params, noise=noise, history=True, compute_dtype=cd, keep_for_vjp=stash)
    grads, _ = model.sample_vjp(hist[0], g, cond, "ddpm", want_gxT=False, compute_dtype=cd, ctx=hist[2] if stash else None)
    return x0, grads


def vjp_parts_ms(model, w, xT, cond, noise, g, iters):
    """Average durations of the two calls of a VJP step, events on the launch stream: the forward that keeps what the
    backward needs (frl_sampler_fwd_train; --recompute: frl_sampler_fwd with history) and the backward call
    (frl_sampler_bwd: [recompute +] sweep + dW + reduce + time-branch kernels)."""
    cd = w.get("dtype", "f32")
    stash = not w.get("recompute")

    def fwd():
        return model.sample(None, xT, cond, True, "ddpm", params=model.params, noise=noise, history=True, compute_dtype=cd, keep_for_vjp=stash)[2]

    def bwd(hist):
        model.sample_vjp(hist[0], g, cond, "ddpm", want_gxT=False, compute_dtype=cd, ctx=hist[2] if stash else None)

    from flowrl_hip import _lib
    lib = _lib.load()
    hist = fwd()
    names = [lib.frl_last_kernels().decode()]
    for _ in range(2):
        bwd(hist)
    names.append(lib.frl_last_kernels().decode())
    vjp_parts_ms.kernels = "forward call: " + names[0] + " | backward call: " + names[1]
    out = []
    for fn in (fwd, lambda: bwd(hist)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(xT.device)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize(xT.device)
        out.append(e0.elapsed_time(e1) / iters)
    return out[0], out[1]


def kernel_only_ms(model, w, xT, cond, noise, iters):
    """Average duration of the dominant kernel (the fused T-step sampler) alone: prepare once, then `iters`
    launches of frl_sampler_run bracketed by events on the launch stream (for cfg4 this includes the
    cond-embedding kernel, < 0.1 % of the FLOPs)."""
    from flowrl_hip import _lib, _sampler
    from flowrl_hip import schedules as sch
    lib = _lib.load()
    dev = xT.device
    keep = []
    kind = {"ddpm": _lib.FRL_KIND_DDPM, "cddpm": _lib.FRL_KIND_CDDPM, "cnf": _lib.FRL_KIND_CNF}[w["kind"]]
    desc = _sampler.build_desc(model.network, kind=kind, solver=_lib.FRL_SOLVER_EULER if w["kind"] == "cnf" else _lib.FRL_SOLVER_DDPM,
                               steps=w["T"], n_states=w["B"] // w["particles"], particles=w["particles"], x_dim=w["A"],
                               cond_dim=w["S"], clip_sampler=w["clip"], x_min=w["x_min"], x_max=w["x_max"], want_history=False,
                               compute_dtype=w.get("dtype", "f32"), tile_policy=w.get("tile_policy", "auto"),
                               rng_impl=_lib.FRL_RNG[w.get("rng", "none")])
    fp = _sampler.build_params(model.network, model.params, dev, keep)
    fs = _lib.FrlSchedule()
    if w["kind"] == "ddpm":
        tabs = dict(betas=model.betas, alphas=model.alphas, alpha_hats=model.alpha_hats, postvars=model.postvars)
    elif w["kind"] == "cddpm":
        ts, ah, al, be = sch.continuous_tables(w["T"], model.noise_schedule_func, model.t_diffusion, model.t_schedule_n)
        tabs = {k: torch.from_numpy(v).to(dev) for k, v in dict(ts=ts, alpha_hats=ah, alphas=al, betas=be).items()}
    else:
        tabs = {}
    for k, v in tabs.items():
        keep.append(v)
        setattr(fs, k, v.data_ptr())
    x0 = torch.empty_like(xT)
    bufs = _lib.FrlFwdBufs()
    bufs.xT, bufs.cond, bufs.x0 = xT.data_ptr(), cond.data_ptr(), x0.data_ptr()
    if w.get("rng", "none") != "none":
        from flowrl_hip import random as fr
        key = fr.PRNGKey(0, dev)
        rng_out = torch.empty_like(key)
        keep += [key, rng_out]
        bufs.rng_key, bufs.rng_out = key.data_ptr(), rng_out.data_ptr()
    elif noise is not None:
        bufs.noise = noise.data_ptr()
    need = lib.frl_workspace_bytes(C.byref(desc))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.frl_sampler_prepare(C.byref(desc), C.byref(fp), C.byref(fs), ws.data_ptr(), need, stream))
    for _ in range(2):
        _lib.check(lib.frl_sampler_run(C.byref(desc), C.byref(fp), C.byref(bufs), ws.data_ptr(), need, stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(iters):
        _lib.check(lib.frl_sampler_run(C.byref(desc), C.byref(fp), C.byref(bufs), ws.data_ptr(), need, stream))
    e1.record()
    torch.cuda.synchronize(dev)
    if os.environ.get("FRL_DUMP_X0"):      # section cycle counters of an instrumented build (development aid)
        print("x0[:6] =", x0.reshape(-1)[:6].tolist(), file=sys.stderr)
    flops_exec = C.c_double(0.0)
    flops_alg = lib.frl_algorithmic_flops(C.byref(desc), C.byref(flops_exec))
    # work the kernel skips by hoisting the condition rows of Dense_0 (contracted once per state) earns no credit
    kernel_only_ms.hoisted = lib.frl_hoisted_flops(C.byref(desc))
    return e0.elapsed_time(e1) / iters, flops_alg - kernel_only_ms.hoisted, flops_exec.value


# --scaling strong: the job-wide number of STATES; workloads not listed keep their per-GPU batch as the total
STRONG_TOTAL_STATES = {"cfg3": 8388608, "cfg4": 65536}

# committed rocprofv3 PMC passes (scripts/prof_r2.sh) per workload at its full batch: (summary file, sampler kernel name)
TRAFFIC_PROFILES = {
    "cfg1": ("profiles/r02_prof_cfg1/pmc_summary.json", "frl_sampler_lat_kernel"),
    "cfg2": ("profiles/r03_prof_cfg2/pmc_summary.json", "frl_sampler_rs_kernel"),
    "cfg3": ("profiles/r03_prof_cfg3/pmc_summary.json", "frl_sampler_rs_kernel"),
    "cfg4": ("profiles/r03_prof_cfg4/pmc_summary.json", "frl_sampler_rs_kernel"),
    "res_dac": ("profiles/r02_prof_res_dac/pmc_summary.json", "frl_sampler_res_kernel"),
}


def pmc_traffic_bytes(workload, w):
    """(HBM bytes per launch of the sampler kernel, source file) from the COMMITTED rocprofv3 PMC passes of this workload
    at its full batch: 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, both reported in
    KiB.  Not measured in this run (PMC collection needs the profiler): the bench line names the file in
    roofline.traffic_source.  (None, None) when no profile matches."""
    if workload not in TRAFFIC_PROFILES or w["B"] != WORKLOADS[workload]["B"]:
        return None, None
    path, kern = TRAFFIC_PROFILES[workload]
    f = ROOT / path
    if not f.exists():
        return None, None
    pm = json.loads(f.read_text())
    k = [v for name, v in pm.items() if kern in name]
    if not k or "FETCH_SIZE" not in k[0] or "WRITE_SIZE" not in k[0]:
        return None, None
    return (2.0 * k[0]["FETCH_SIZE"]["mean"] + k[0]["WRITE_SIZE"]["mean"]) * 1024.0, path


def cpu_baseline(w, budget_s=15.0):
    """The numpy float32 oracle (a port of the reference's arithmetic; the JAX reference itself cannot
    run here) timed on this box's host cores over a bounded sample of the same workload."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import flowrl_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    spec = orc.BackboneSpec(hidden_dims=w["hidden"], x_dim=w["A"], activation=w["act"], time_embedding=w["temb"],
                            cond_hidden=[128, 128] if w["cond_embed"] else None,
                            predictor_name="vel_predictor" if w["kind"] == "cnf" else "noise_predictor",
                            backbone=w.get("backbone", "mlp"), layer_norm=bool(w.get("ln")))
    params = orc.make_params(spec, w["S"], seed=0, cond_in_dim=w["S"])
    rng = np.random.default_rng(0)
    kw = dict(steps=w["T"], clip_sampler=w["clip"], x_min=w["x_min"], x_max=w["x_max"])

    def run(Bs):
        Bs = (Bs // w["particles"]) * w["particles"]
        xT = rng.standard_normal((Bs, w["A"])).astype(np.float32)
        # the reference materialises obs_repeat (dql.py:145) and runs the whole net per particle
        cond = np.repeat(rng.standard_normal((Bs // w["particles"], w["S"])).astype(np.float32), w["particles"], axis=0)
        nz = rng.standard_normal((w["T"], Bs, w["A"])).astype(np.float32)
        t0 = time.perf_counter()
        if w.get("vjp"):
            import flowrl_oracle_vjp as ov
            ov.ddpm_sample_vjp(spec, params, orc.ddpm_tables(w["T"], w["schedule"]), xT, cond, nz,
                               rng.standard_normal((Bs, w["A"])), solver="ddpm", **kw)
        elif w["kind"] == "ddpm":
            orc.ddpm_sample(spec, params, orc.ddpm_tables(w["T"], w["schedule"]), xT, cond, nz, **kw)
        elif w["kind"] == "cddpm":
            orc.continuous_ddpm_sample(spec, params, orc.continuous_tables(w["T"], w["schedule"]), xT, cond, nz, **kw)
        else:
            orc.cnf_sample(spec, params, xT, cond, **kw)
        return time.perf_counter() - t0

    probe_B = 4096 if w["hidden"][0] <= 256 else 2048
    if w.get("vjp"):
        probe_B = 1024
    run(probe_B)                      # warm-up (BLAS threads, page faults)
    t_probe = run(probe_B)
    Bs = int(min(w["B"], max(probe_B, probe_B * budget_s / max(t_probe, 1e-3))))
    Bs = max(probe_B, (Bs // probe_B) * probe_B)
    t = run(Bs)
    return {"value": Bs * w["T"] / t, "unit": "chain-steps/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy {'float64 manual-backprop oracle (oracle/flowrl_oracle_vjp.py), forward + VJP' if w.get('vjp') else 'float32 oracle (oracle/flowrl_oracle.py)'}, {Bs} of {w['B']} chains x T={w['T']}, "
                      f"one pass, {t:.2f} s, {cores} BLAS threads"}


def launch_ranks(n: int, argv, stub: bool) -> int:
    """Parent of a multi-GPU run: nothing here touches the GPU (torch.cuda.device_count() only counts devices on this
    image), so the children are ordinary fresh processes — never a re-exec of a process that initialised HIP."""
    import shutil
    import socket
    import subprocess
    if not stub:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} asked for, but this box has {have} GPU(s) visible; refusing to run "
                  f"{n} ranks on fewer devices (a scaling number measured that way would be meaningless)", file=sys.stderr, flush=True)
            return 2
    # Rendezvous of self-launched ranks through a FILE store (a fresh temporary path): probing for a free TCP port and
    # closing it again would leave a window in which another process takes it.  MASTER_ADDR / MASTER_PORT are still
    # exported for code that reads them, but the children initialise from FRL_BENCH_INIT.
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rdv_dir = tempfile.mkdtemp(prefix="frl_bench_rdv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FRL_BENCH_CHILD="1", FRL_BENCH_INIT=f"file://{rdv_dir}/store")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained by a thread; the parent polls EVERY child, so that the early death of any rank ends the
    # run at once (the others would otherwise sit in the rendezvous until its timeout)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    bad = []
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc == 0 for rc in rcs):
            break
        time.sleep(0.05)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr, flush=True)
        shutil.rmtree(rdv_dir, ignore_errors=True)
        return bad[0][1] or 1
    shutil.rmtree(rdv_dir, ignore_errors=True)
    reader.join(timeout=10)
    sys.stdout.write("".join(chunks))
    sys.stdout.flush()
    return 0


def stub_main(args, world: int, rank: int) -> None:
    """CPU stand-in of the rank body (tests/test_host.py): same launcher, same rendezvous, same fence / MAX-over-ranks
    timing and all-gather pattern over gloo, with a trivial step instead of the HIP sampler.  Not a measurement."""
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=os.environ.get("FRL_BENCH_INIT", "env://"))
    B, A, T = 64, 3, 5
    x = torch.full((B, A), float(rank))
    gathered = torch.empty((world * B, A))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = x * 2.0 + 1.0
        dist.all_gather_into_tensor(gathered, y)
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = all(bool((gathered[r * B:(r + 1) * B] == 2.0 * r + 1.0).all()) for r in range(world))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * B * T / (float(tt.item()) / args.steps), "unit": "chain-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "data": "stub (CPU, gloo): launcher test only",
                          "gather_ok": ok}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the final RCCL all-gather of x0 (N>1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = bit-faithful kernel (the headline number); bf16 = stated-tolerance bf16-MFMA variant (256- and 512-wide MLP workloads)")
    ap.add_argument("--tile-policy", default="auto", choices=["auto", "throughput", "latency", "latency32", "latency16"],
                    help="f32 forward kernel build (FrlDesc.tile_policy); auto = latency build for small batches")
    ap.add_argument("--cache-prepare", action="store_true",
                    help="let the host wrapper reuse the prepared tables / packed weights across calls (acting between two "
                         "optimizer steps); off by default so that every timed step runs the batch-invariant prepare kernel too")
    ap.add_argument("--rng", default="none", choices=["none", "partitionable", "legacy"],
                    help="draw the per-step noise in-kernel from a threefry key (FrlDesc.rng_impl) instead of reading a (T, B, A) tensor")
    ap.add_argument("--act", default="", help="ablation: override the activation")
    ap.add_argument("--layers", type=int, default=0, help="ablation: override the number of hidden layers")
    ap.add_argument("--layer-norm", action="store_true", help="ablation: LayerNorm after every hidden Dense (MLP(layer_norm=True))")
    ap.add_argument("--kernel-only", action="store_true", help="ablation: print kernel ms only")
    ap.add_argument("--recompute", action="store_true", help="cfg5: plain forward with history + a backward that recomputes every step (round-1 path) instead of the training forward's stash")
    ap.add_argument("--stub", action="store_true", help="launcher test: CPU / gloo stand-in of the rank body (no GPU, not a measurement)")
    ap.add_argument("--select", default="none", choices=["none", "fused", "separate"],
                    help="multi-particle workloads: a step is sample + critic scoring + best-of-N selection (jit_sample_and_select / "
                         "jit_sample_actions) with an Ensemblize(ScalarCritic(MLP 256x2 relu)) x 2 critic; fused = one launch "
                         "(frl_sampler_select_fwd), separate = frl_sampler_fwd + frl_critic_select")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) after the warm-up and time its replays: what a "
                         "launch-bound acting loop (small --batch) pays per step without the host-side enqueue of 3..8 kernels")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank owns the workload's per-GPU batch; strong: the workload's TOTAL batch (SURVEY 8(d): cfg4 = "
                         "65536 states x 64 particles, cfg3 = 8388608 chains; others: the per-GPU batch) is split over the ranks by state index "
                         "(flowrl_hip.parallel.shard_bounds)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.stub))      # parent: starts the ranks, touches no GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} in the environment but --gpus {args.gpus}: launch N ranks with --gpus N "
              f"(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N, or plain python bench.py --gpus N)", file=sys.stderr, flush=True)
        sys.exit(2)
    if args.stub:
        stub_main(args, world, rank)
        return
    if world > 1 and torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} wants cuda:{local_rank}, but {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr, flush=True)
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world,
                                init_method=os.environ.get("FRL_BENCH_INIT", "env://"))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    w = dict(WORKLOADS[args.workload])
    if args.batch > 0:
        w["B"] = args.batch
    total_chains = None
    if args.scaling == "strong":
        # the job's size is fixed; rank r owns the contiguous block of states shard_bounds gives it (a state's particles
        # stay together), ragged when the total does not divide
        from flowrl_hip import parallel as _par
        total_states = STRONG_TOTAL_STATES.get(args.workload, WORKLOADS[args.workload]["B"] // w["particles"]) if args.batch <= 0 else args.batch // w["particles"]
        lo, hi = _par.shard_bounds(total_states, rank, world)
        w["B"] = (hi - lo) * w["particles"]
        total_chains = total_states * w["particles"]
        shard_counts = [(_par.shard_bounds(total_states, r, world)[1] - _par.shard_bounds(total_states, r, world)[0]) * w["particles"] for r in range(world)]
    w["dtype"] = args.dtype
    w["tile_policy"] = args.tile_policy
    w["rng"] = args.rng
    w["rng_shard"] = None   # set below for N > 1: every rank draws its block of the job-wide normal(key_j, (N B, A))
    w["cache_prepare"] = bool(args.cache_prepare)
    w["recompute"] = bool(args.recompute)
    if args.act:
        w["act"] = args.act
    if args.layers > 0:
        w["hidden"] = [w["hidden"][0]] * args.layers
    if args.layer_norm:
        w["ln"] = True
    B, T, A, S = w["B"], w["T"], w["A"], w["S"]
    model = build_model(w, dev)
    if args.kernel_only:
        xT, cond, noise = make_inputs(w, dev, 1234)
        k_ms, fa, fe = kernel_only_ms(model, w, xT, cond, noise, max(5, args.steps))
        print(json.dumps({"kernel_ms": k_ms, "dtype": args.dtype, "act": w["act"], "layers": len(w["hidden"]), "alg_tflops": fa / k_ms / 1e9,
                          "exec_tflops": fe / k_ms / 1e9}))
        return
    xT, cond, noise = make_inputs(w, dev, 1234 + rank)       # this rank's shard of the states
    gathered = torch.empty((world * B, A), device=dev) if (world > 1 and total_chains is None) else None
    if world > 1 and args.rng != "none":
        w["rng_shard"] = (rank * B, world * B) if total_chains is None else (sum(shard_counts[:rank]), total_chains)

    vjp = bool(w.get("vjp"))
    critic = cparams = None
    if args.select != "none":
        import flowrl_hip as fh
        if vjp or args.rng != "none" or args.dtype != "f32":
            ap.error("--select: forward f32 workloads with caller-supplied noise")
        critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=[256, 256], activation="relu")), ensemble_size=2)
        cparams = fh.init_critic_params(critic, S, A, seed=7, bias_std=0.1, device=dev)

    def sample_select():
        import flowrl_hip as fh
        parts = w["particles"] if w["particles"] > 1 else None
        x3 = xT.reshape(B // w["particles"], w["particles"], A) if parts else xT
        pos = (None, x3, cond) if w["kind"] == "cnf" else (None, x3, cond, False, "ddpm")
        kw = dict(particles=parts) if w["kind"] == "cnf" else dict(particles=parts, noise=noise)
        if args.select == "fused":
            sel = fh.FusedSelect(critic, cparams, cond, temperature=0.0)
            model.sample(*pos, select=sel, **kw)
            return sel.best
        x0 = model.sample(*pos, tile_policy=args.tile_policy, **kw)[1]
        return fh.critic_select(critic, cparams, cond, x0, w["particles"], temperature=0.0, want_q=False)[0]

    g_x0 = torch.randn((B, A), generator=torch.Generator(device=dev).manual_seed(99 + rank), device=dev) / (total_chains if total_chains is not None else world * B) if vjp else None

    def step():
        if vjp:
            from flowrl_hip import parallel
            x0, grads = vjp_step(model, w, xT, cond, noise, g_x0)
            if world > 1 and not args.no_gather:
                parallel.all_reduce_grads(grads)
            return x0
        if critic is not None:
            return sample_select()           # (n_states, A) chosen actions; the gather of a multi-GPU run is left out
        x0 = sample(model, w, xT, cond, noise)
        if world > 1 and not args.no_gather:
            if total_chains is None:
                dist.all_gather_into_tensor(gathered, x0)
            else:
                from flowrl_hip import parallel
                parallel.all_gather_ragged(x0, shard_counts)
        return x0

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    run = step
    if args.graph:
        if world > 1 or vjp:
            ap.error("--graph: single-GPU forward workloads")
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            x0_graph = step()

        def run():
            graph.replay()
            return x0_graph
        run()
        fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x0 = run()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(x0).all()
    ms_per_step = dt / args.steps * 1e3
    job_chains = total_chains if total_chains is not None else world * B
    value = job_chains * T / (dt / args.steps)

    out = None
    if rank == 0:
        k_ms, flops_alg, flops_exec = kernel_only_ms(model, w, xT, cond, noise, max(5, args.steps))
        # the library names the kernels its last call enqueued (frl_last_kernels: written by the launchers themselves)
        from flowrl_hip import _lib as _flib
        kname = _flib.load().frl_last_kernels().decode()
        if vjp:
            # FLOPs_alg,bwd = 2 x forward (dX + dW); the sweep's forward recompute is overhead, not credit
            f_ms, b_ms = vjp_parts_ms(model, w, xT, cond, noise, g_x0, max(5, args.steps))
            stash = not args.recompute
            kname = vjp_parts_ms.kernels
            flops_exec = (3.0 if stash else 4.0) * flops_exec
            flops_alg, k_ms = 3.0 * flops_alg, f_ms + b_ms
            vjp_parts = {"forward_ms": f_ms, "backward_ms": b_ms}
        achieved = flops_alg / (k_ms * 1e-3) / 1e12
        traffic, traffic_src = (None, None) if (vjp or args.act or args.layers or args.layer_norm or args.dtype != "f32" or args.rng != "none" or args.tile_policy != "auto") else pmc_traffic_bytes(args.workload, w)
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        out = {
            "metric": "action samples/sec x T steps (chain-steps/s), fused sampler",
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}" + (f" [strong scaling: {job_chains} chains in total, {B} on rank 0]" if total_chains is not None else ""), "chains_per_gpu": B, "T": T,
                       "chains_per_s": value / T, "prepare": "cached across calls" if args.cache_prepare else "every call", "noise": "caller tensor (T, B, A)" if args.rng == "none" else f"in-kernel threefry ({args.rng})", "launch": "HIP graph replay" if args.graph else "eager", "collective": "none" if world == 1 or args.no_gather else ("all_reduce(flat grads)" if vjp else "all_gather(x0)")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": f"committed rocprofv3 PMC passes ({traffic_src}), not measured in this run" if traffic_src else None,
                         "algorithmic_hbm_bytes": B * (4 * (2 * A + (T * A if (w["kind"] != "cnf" and args.rng == "none") else 0)) + 4 * S // w["particles"]),
                         "kernel": kname, "kernel_ms": k_ms,
                         "flops_algorithmic": flops_alg, "flops_hoisted_not_credited": getattr(kernel_only_ms, "hoisted", 0.0) * (3.0 if vjp else 1.0),
                         "flops_executed_mfma": flops_exec,
                         "achieved_executed": flops_exec / (k_ms * 1e-3) / 1e12},
        }
        if vjp:
            out["roofline"].update(vjp_parts)
        if critic is not None:
            from flowrl_hip import _lib as _fl2
            sample_select()
            out["config"]["select"] = {"mode": args.select, "critic": "Ensemblize(ScalarCritic(MLP 256x2 relu)) x 2, argmax",
                                       "kernels": _fl2.load().frl_last_kernels().decode(),
                                       "note": "ms_per_step / value cover sampler + scoring + selection; roofline.* is the sampler kernel alone (frl_sampler_run)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
