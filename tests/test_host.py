"""CPU tests of the host-side logic: schedule tables vs the oracle, condition / particle handling,
descriptor marshalling, and the N>1 sharding path with world_size-2 gloo processes."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import flowrl_oracle as orc
import flowrl_hip as fh
from flowrl_hip import _lib, _sampler, parallel, schedules

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("sched,T,approx", [("vp", 5, False), ("vp", 5, True), ("linear", 20, False), ("cosine", 10, False),
                                            ("linear", 1, False), ("vp", 100, False)])
def test_ddpm_tables_match_oracle(sched, T, approx):
    got = schedules.ddpm_tables(T, sched, None, approx)
    want = orc.ddpm_tables(T, sched, approx_postvar=approx, dtype=np.float32)
    for g, w in zip(got, want):
        assert g.dtype == np.float32 and g.shape == (T + 1,)
        np.testing.assert_array_equal(g, w)
    w64 = orc.ddpm_tables(T, sched, approx_postvar=approx, dtype=np.float64)
    np.testing.assert_allclose(got[2], w64[2], rtol=3e-6, atol=1e-8)   # atol: cosine ends at cos(~pi/2) (cancellation in f32)


@pytest.mark.parametrize("sched,T,n", [("cosine", 20, 1.0), ("linear", 10, 1.0), ("cosine", 5, 2.0)])
def test_continuous_tables_match_oracle(sched, T, n):
    from functools import partial
    fn = schedules.cosine_noise_schedule if sched == "cosine" else schedules.linear_noise_schedule
    td = [1e-3, 0.9946] if sched == "cosine" else [1e-3, 1.0]
    got = schedules.continuous_tables(T, fn, td, n)
    want = orc.continuous_tables(T, sched, t_schedule_n=n, dtype=np.float32)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_split_condition_detects_broadcast_particles():
    obs = torch.randn(7, 11)
    rep = obs[:, None, :].expand(7, 10, 11)            # stride-0 particle axis: never materialised
    xT = torch.randn(7, 10, 3)
    n, p, c2 = _sampler._split_condition(xT, rep, None)
    assert (n, p) == (7, 10) and c2.shape == (7, 11) and c2.data_ptr() == obs.data_ptr()
    mat = obs[:, None, :].repeat(1, 10, 1)             # materialised obs_repeat (dql.py:145): treated as 70 states
    n, p, c2 = _sampler._split_condition(xT, mat, None)
    assert (n, p) == (70, 1) and c2.shape == (70, 11)
    n, p, c2 = _sampler._split_condition(xT.reshape(70, 3), obs, 10)   # explicit particles
    assert (n, p) == (7, 10)
    assert _sampler._split_condition(xT, None, None) == (70, 1, None)
    with pytest.raises(ValueError):
        _sampler._split_condition(xT, torch.randn(7, 11), None)
    with pytest.raises(ValueError):
        _sampler._split_condition(xT.reshape(70, 3), obs, 9)


def test_build_desc_and_param_tree_layout():
    net = fh.DDPMBackbone(noise_predictor=fh.MLP(hidden_dims=[256, 256, 256], output_dim=3, activation="mish"),
                          time_embedding=fh.PositionalEmbedding(output_dim=64))
    p = fh.init_params(net, 3, 11, seed=0)
    assert sorted(p) == ["MLP_0", "noise_predictor"]
    assert tuple(p["noise_predictor"]["Dense_0"]["kernel"].shape) == (3 + 64 + 11, 256)      # x | time | cond
    assert tuple(p["noise_predictor"]["Dense_3"]["kernel"].shape) == (256, 3)
    assert float(p["noise_predictor"]["Dense_1"]["bias"].abs().max()) == 0.0                 # zeros_init
    k = p["noise_predictor"]["Dense_1"]["kernel"].double()
    np.testing.assert_allclose((k.T @ k).numpy(), 2.0 * np.eye(256), atol=1e-5)              # orthogonal(sqrt 2)
    _sampler.check_param_shapes(net, p, 3, 11)
    with pytest.raises(ValueError):
        _sampler.check_param_shapes(net, p, 3, 12)
    d = _sampler.build_desc(net, kind=_lib.FRL_KIND_DDPM, solver=_lib.FRL_SOLVER_DDPM, steps=5, n_states=10, particles=4,
                            x_dim=3, cond_dim=11, clip_sampler=True, x_min=-1, x_max=1, want_history=False)
    assert (d.n_hidden, d.hidden[0], d.act, d.time_embed, d.time_dim, d.clip) == (3, 256, 0, 1, 64, 1)
    flow = fh.FlowBackbone(vel_predictor=fh.MLP(hidden_dims=[256, 256], output_dim=8, activation="gelu"))
    q = fh.init_params(flow, 8, 39, seed=1)
    assert tuple(q["vel_predictor"]["Dense_0"]["kernel"].shape) == (8 + 1 + 39, 256) and "MLP_0" not in q
    ln = fh.DDPMBackbone(noise_predictor=fh.MLP(hidden_dims=[256], output_dim=3, activation="mish", layer_norm=True))
    dl = _sampler.build_desc(ln, kind=0, solver=0, steps=5, n_states=1, particles=1, x_dim=3, cond_dim=0,
                             clip_sampler=False, x_min=None, x_max=None, want_history=False)
    assert dl.layer_norm == 1 and d.layer_norm == 0
    pl = fh.init_params(ln, 3, 0, seed=2)      # flax auto-naming: LayerNorm_i beside Dense_i, scale ones / bias zeros
    assert set(pl["noise_predictor"]) == {"Dense_0", "Dense_1", "LayerNorm_0"}
    assert bool((pl["noise_predictor"]["LayerNorm_0"]["scale"] == 1).all()) and not bool(pl["noise_predictor"]["LayerNorm_0"]["bias"].any())
    drop = fh.DDPMBackbone(noise_predictor=fh.MLP(hidden_dims=[256], output_dim=3, activation="mish", dropout=0.1))
    with pytest.raises(_lib.FlowRLHipUnsupported):
        _sampler.build_desc(drop, kind=0, solver=0, steps=5, n_states=1, particles=1, x_dim=3, cond_dim=0,
                            clip_sampler=False, x_min=None, x_max=None, want_history=False)


def test_unsupported_solver_names():
    from flowrl_hip.flow import DDPM, ContinuousDDPM
    net = fh.DDPMBackbone(noise_predictor=fh.MLP(hidden_dims=[256], output_dim=2, activation="mish"),
                          time_embedding=fh.LearnableFourierEmbedding(output_dim=64))
    m = ContinuousDDPM.create(net, 0, (np.ones((1, 2)), np.zeros((1, 1)), None), x_dim=2, steps=4, noise_schedule="cosine",
                              device="cpu")
    assert m.t_diffusion == [0.001, 0.9946]
    with pytest.raises(RuntimeError, match="GPU only"):     # dpm2 / dpm++2m are fused solvers now: only the device check stops a CPU tensor
        m.sample(None, torch.zeros(3, 2), solver="dpm2")
    with pytest.raises(NotImplementedError):
        m.sample(None, torch.zeros(3, 2), solver="heun")
    d = DDPM.create(fh.DDPMBackbone(noise_predictor=fh.MLP(hidden_dims=[256], output_dim=2, activation="mish"),
                                    time_embedding=fh.PositionalEmbedding(64)), 0,
                    (np.ones((1, 2)), np.zeros((1, 1)), None), x_dim=2, steps=4, noise_schedule="vp", device="cpu")
    with pytest.raises(NotImplementedError):
        d.sample(None, torch.zeros(3, 2), solver="dpm2")


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 65536, 1000003):
        for w in (1, 2, 4, 8):
            b = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 2, 2)


def _gloo_worker(rank, world, port, n_states, particles, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        A, C, T = 3, 5, 4
        xT = torch.randn(n_states * particles, A, generator=g)
        cond = torch.randn(n_states, C, generator=g)
        noise = torch.randn(T, n_states * particles, A, generator=g)

        def fake_sampler(x, c, z):     # stands in for the HIP kernel: any per-chain function of (x, cond row, noise)
            return x * 2 + c.repeat_interleave(particles, 0)[:, :A] + z.sum(0)

        full = parallel.sample_sharded(fake_sampler, xT, cond, noise, particles=particles)
        want = fake_sampler(xT, cond, noise)
        assert full.shape == want.shape
        assert torch.equal(full, want), f"rank {rank}: sharded result differs"
        mine = parallel.sample_sharded(fake_sampler, xT, cond, noise, particles=particles, gather=False)
        lo, hi = parallel.shard_bounds(n_states, rank, world)
        assert torch.equal(mine, want[lo * particles: hi * particles])
        # keyed sharding: the third argument becomes rng_shard = (first chain, total chains); a stand-in that draws the
        # elements [first, first + n) of the unsharded normal(key, (B, A)) — what FrlFwdBufs.rng_chain_offset /
        # rng_total_chains make the kernel do — must gather to the single-call result
        import numpy as np
        import threefry_oracle as tf
        key = tf.prng_key(7)

        def keyed_sampler(x, c, shard):
            first, total = shard
            assert total == n_states * particles
            z = tf.normal(key, (total, A))[first:first + x.shape[0]]
            return x + torch.from_numpy(z) + c.repeat_interleave(particles, 0)[:, :A]

        got = parallel.sample_sharded(keyed_sampler, xT, cond, None, particles=particles, keyed=True)
        want_k = keyed_sampler(xT, cond, (0, n_states * particles))
        assert torch.equal(got, want_k), f"rank {rank}: keyed sharded result differs"
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_states,particles", [(64, 1), (33, 4), (1, 3)])
def test_sharded_sampling_two_ranks_gloo(tmp_path, n_states, particles):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + n_states) % 2000
    mp.spawn(_gloo_worker, args=(2, port, n_states, particles, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _gloo_grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch.distributed as dist
    import flowrl_oracle as orc
    import flowrl_oracle_vjp as ov
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the VJP's N>1 contract: every rank differentiates its own block of states, ONE all-reduce sums the flat buffer
        spec = orc.BackboneSpec(hidden_dims=[32, 32], x_dim=3, activation="mish", time_embedding="positional", time_dim=16)
        params = orc.make_params(spec, 5, seed=0)
        tables = orc.ddpm_tables(3, "vp")
        rng = np.random.default_rng(0)
        n = 10
        xT, cond = rng.standard_normal((n, 3)), rng.standard_normal((n, 5))
        noise, g = rng.standard_normal((3, n, 3)), rng.standard_normal((n, 3))
        kw = dict(steps=3, clip_sampler=True, x_min=-1.0, x_max=1.0)
        _, want, _ = ov.ddpm_sample_vjp(spec, params, tables, xT, cond, noise, g, **kw)
        lo, hi = parallel.shard_bounds(n, rank, world)
        _, mine, _ = ov.ddpm_sample_vjp(spec, params, tables, xT[lo:hi], cond[lo:hi], noise[:, lo:hi], g[lo:hi], **kw)
        to_t = lambda t: {k: to_t(v) if isinstance(v, dict) else torch.from_numpy(np.array(v, copy=True)) for k, v in t.items()}
        got = parallel.all_reduce_grads(to_t(mine))
        for path, v in parallel._leaves(got):
            ref = want
            for k in path:
                ref = ref[k]
            assert np.allclose(v.numpy(), ref, rtol=1e-10, atol=1e-12), path
        # the product's layout: leaves carved out of ONE flat buffer (_sampler.alloc_grad_tree) -> the collective runs on
        # that buffer in place, the leaves (views of it) then hold the sums; no cat / copy-back
        from flowrl_hip import _sampler
        mine_t = to_t(mine)
        paths = [path for path, _ in parallel._leaves(mine_t)]
        flat_tree = _sampler.alloc_grad_tree(mine_t, paths, torch.device("cpu"))
        for path, v in parallel._leaves(mine_t):
            dst = flat_tree
            for k in path:
                dst = dst[k]
            dst.copy_(v.to(torch.float32))
        base = flat_tree.flat.data_ptr()
        out = parallel.all_reduce_grads(flat_tree)
        assert out is flat_tree and flat_tree.flat.data_ptr() == base
        for path, v in parallel._leaves(flat_tree):
            ref = want
            for k in path:
                ref = ref[k]
            assert v.data_ptr() >= base and np.allclose(v.numpy(), ref, rtol=2e-5, atol=1e-6), path
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mp.spawn(_gloo_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


# ---- bench.py launcher: `python bench.py --gpus N` must start N ranks itself ----------------
def _run_bench(*argv, env_extra=None, timeout=180):
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(root / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launcher_starts_n_ranks_over_gloo():
    """The parent starts 2 rank processes (before touching any GPU), they rendezvous on 127.0.0.1 over gloo, run the
    stub step + all-gather with the bench's fence / MAX-over-ranks timing, and rank 0 prints the ONE JSON line."""
    import json
    r = _run_bench("--gpus", "2", "--stub", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["gather_ok"] is True


def test_bench_launcher_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = _run_bench("--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0")
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


def test_bench_refuses_world_size_mismatch():
    r = _run_bench("--gpus", "1", "--stub", env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
