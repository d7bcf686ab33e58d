"""GPU parity of the in-kernel jax.random restatement: csrc/frl_rng.h, csrc/rng.hip and the rng-key
path of the samplers (FrlDesc.rng_impl, FrlFwdBufs.rng_key / rng_out), against oracle/threefry_oracle.py.

 * split / bits: BIT-EXACT (integer work), both counter layouts, odd and even sizes;
 * normal: the uniform is exact; erf_inv goes through log1p and fma-contracted Horner steps, so the draw agrees with
   the numpy evaluation of the same polynomial to a few ulp: |dz| <= 4e-7 max(1, |z|)  (XLA on a GPU is itself not
   bit-identical to XLA on a CPU here);
 * samplers called with a key: x0 / histories within the f32 sampler tolerance of the float64 oracle chain run on the
   ORACLE's noise for that key, and the returned rng bit-exact, for every kind / solver, both f32 builds and bf16.
"""
import dataclasses

import numpy as np
import pytest

import helpers as H
import threefry_oracle as tf

pytestmark = pytest.mark.gpu

IMPLS = ["partitionable", "legacy"]


def _key(seed):
    import torch
    return torch.from_numpy(tf.prng_key(seed)).cuda()


@pytest.mark.parametrize("impl", IMPLS)
def test_split_and_bits_bit_exact(impl):
    import flowrl_hip.random as fr
    for seed in (0, 1, 0xDEADBEEF12345678):
        k = tf.prng_key(seed)
        assert np.array_equal(fr.PRNGKey(seed).cpu().numpy(), k)
        for num in (1, 2, 3, 4, 7, 300):
            assert np.array_equal(fr.split(_key(seed), num, impl).cpu().numpy(), tf.split(k, num, impl)), (seed, num)
        for shape in ((1,), (2,), (7,), (64, 3), (1001, 17)):
            assert np.array_equal(fr.bits(_key(seed), shape, impl).cpu().numpy(), tf.random_bits(k, shape, impl)), (seed, shape)


@pytest.mark.parametrize("impl", IMPLS)
def test_normal(impl):
    import flowrl_hip.random as fr
    k = tf.prng_key(11)
    for shape in ((1,), (5, 3), (200001,), (4000001,)):     # the last: ~40 % of the 2^23 inputs of the float map
        got = fr.normal(_key(11), shape, impl).cpu().numpy()
        ref = tf.normal(k, shape, impl)
        assert got.shape == ref.shape and np.isfinite(got).all()
        err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
        assert err.max() <= 4e-7, (shape, err.max())
    # the documented PRNGKey(0) draws
    z = float(fr.normal(_key(0), (1,), impl).cpu()[0])
    assert abs(z - (1.6226422 if impl == "partitionable" else -0.20584226)) < 5e-7


def test_bad_arguments_are_loud():
    import flowrl_hip as fh
    import flowrl_hip.random as fr
    with pytest.raises(ValueError):
        fr.split(_key(0), 2, "philox")
    with pytest.raises(TypeError):
        import torch
        fr.split(torch.zeros(3, dtype=torch.uint32).cuda(), 2)
    assert fh.random.normal(_key(0), (0,)).numel() == 0


CASES = [
    H.Case("ddpm-dql", "ddpm", A=3, S=11, T=5, B=300),
    H.Case("ddpm-ddim", "ddpm", A=3, S=11, T=5, B=64, solver="ddim"),
    H.Case("ddpm-particles-wide", "ddpm", A=17, S=128, T=6, B=12 * 8, particles=8, schedule="linear", seed=3),
    H.Case("ddpm-512", "ddpm", A=6, S=17, T=4, B=100, hidden=(512, 512), schedule="linear", seed=4),
    H.Case("cddpm-cosine", "cddpm", A=6, S=17, T=12, B=200, temb="learnable_fourier", schedule="cosine", seed=5),
    H.Case("cddpm-dpm2", "cddpm", A=3, S=11, T=6, B=70, temb="learnable_fourier", schedule="cosine", solver="dpm2", seed=6),
    H.Case("cddpm-dpmpp2m", "cddpm", A=3, S=11, T=6, B=70, temb="learnable_fourier", schedule="cosine", solver="dpm++2m", seed=7),
    H.Case("cnf", "cnf", A=8, S=20, T=10, B=64, temb=None, clip=False, act="gelu_tanh", seed=8),
    H.Case("ddpm-bf16", "ddpm", A=3, S=11, T=5, B=256, bf16=True, seed=9),
]


def _run_with_key(case, impl, seed):
    import torch
    params, xT, cond, _ = H.make_inputs(case)
    key = tf.prng_key(seed)
    shape = xT.shape
    noise, rng_ref = tf.sampler_noise(case.kind, case.solver, case.T, key, shape, impl)
    model = H.product_model(case, params)
    x = torch.from_numpy(xT).cuda()
    cd = None if cond is None else torch.from_numpy(cond).cuda()
    kw = dict(history=True, particles=case.particles if case.particles > 1 else None,
              compute_dtype="bf16" if case.bf16 else "f32", tile_policy=case.tile, rng_impl=impl)
    k = torch.from_numpy(key).cuda()
    if case.kind == "cnf":
        rng, x0, (xh, eh) = model.sample(k, x, cd, **kw)
    else:
        rng, x0, (xh, eh) = model.sample(k, x, cd, False, case.solver, **kw)
    torch.cuda.synchronize()
    assert np.array_equal(rng.cpu().numpy(), rng_ref), f"{case.name}: returned rng"
    assert np.array_equal(k.cpu().numpy(), key), "the input key must not be written"
    return params, xT, cond, noise, x0.cpu().numpy(), xh.cpu().numpy(), eh.cpu().numpy()


@pytest.mark.parametrize("tile", ["throughput", "latency"])
@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_sampler_with_key(case, impl, tile):
    if case.bf16 and tile == "latency":
        pytest.skip("tile_policy applies to the f32 builds only")
    case = dataclasses.replace(case, tile=tile)
    params, xT, cond, noise, x0, xh, eh = _run_with_key(case, impl, seed=1234 + case.seed)
    if case.bf16:
        # against the same bf16 kernel fed the oracle's noise for that key: the draws differ by <= 4e-7, which an
        # occasional bf16 operand rounding turns into a bf16-ulp-sized difference of eps (stated tolerance 2e-3)
        import torch
        model = H.product_model(case, params)
        _, y0, (yh, _) = model.sample(None, _t(xT), _t(cond), False, case.solver, noise=_t(noise), history=True, compute_dtype="bf16")
        torch.cuda.synchronize()
        assert H.rel_err(x0, y0.cpu().numpy()) <= 2e-3 and H.rel_err(xh, yh.cpu().numpy()) <= 2e-3
        return
    # (1) against the same kernel fed the DEVICE's draws for the oracle's iteration keys as an explicit noise tensor
    #     (frl_random_normal, checked against the oracle in test_normal): the same chain, so agreement to rounding
    import torch
    import flowrl_hip.random as fr
    keys, _ = tf.sampler_key_chain(case.kind, case.solver, case.T, tf.prng_key(1234 + case.seed), impl)
    model = H.product_model(case, params)
    kw = dict(history=True, particles=case.particles if case.particles > 1 else None, tile_policy=tile)
    if case.kind != "cnf" and case.solver == "ddpm":
        dn = torch.stack([fr.normal(_t(keys[j]), xT.shape, impl) for j in range(case.T)])
        _, y0, (yh, ye) = model.sample(None, _t(xT), None if cond is None else _t(cond), False, case.solver, noise=dn, **kw)
        torch.cuda.synchronize()
        d0, dx = H.rel_err(x0, y0.cpu().numpy()), H.rel_err(xh, yh.cpu().numpy())
        assert d0 <= 2e-6 and dx <= 2e-6, f"{case.name}: in-kernel draws vs frl_random_normal tensor: x0 {d0:.2e}, xt_hist {dx:.2e}"
    # (2) against the float64 oracle chain on the ORACLE's noise for that key: the f32 sampler tolerance plus the
    #     draws' few-ulp difference (4e-7) carried through the chain (1 / sqrt(alpha_hat) amplifies it: 1e-4 bound)
    ref0, refx, refe = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    e0, ex, ee = H.rel_err(x0, ref0), H.rel_err(xh, refx), H.rel_err(eh, refe)
    print(f"{case.name}/{impl}/{tile}: vs oracle x0 {e0:.2e} xt_hist {ex:.2e} eps_hist {ee:.2e}")
    assert e0 <= 1e-4 and ex <= 1e-4 and ee <= 1e-4


def _t(a):
    import torch
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("impl", IMPLS)
def test_sampler_with_key_two_wave_build(impl):
    """Large plain 256-wide batches take the 16-chain two-wave build; with a key (and x_dim <= 4) its EXT 6 variant (one dense
    draw per lane and step as micro-steps, the other wave's MFMAs beside them, values to their owner lanes through LDS).  Same checks as above at 70 001 chains (a ragged last tile): the kernel fed
    frl_random_normal's tensor for the oracle's iteration keys, and the float64 oracle on a slice."""
    import torch
    import flowrl_hip as fh
    import flowrl_hip.random as fr
    case = H.Case("ddpm-dql-two-wave", "ddpm", A=3, S=11, T=5, B=70001, seed=21)
    params, xT, cond, noise, x0, xh, eh = _run_with_key(case, impl, seed=1234 + case.seed)
    assert "kernel_w2<act 0, draws 1>" in fh.last_kernels(), fh.last_kernels()
    keys, _ = tf.sampler_key_chain(case.kind, case.solver, case.T, tf.prng_key(1234 + case.seed), impl)
    model = H.product_model(case, params)
    dn = torch.stack([fr.normal(_t(keys[j]), xT.shape, impl) for j in range(case.T)])
    _, y0, (yh, _) = model.sample(None, _t(xT), _t(cond), False, case.solver, noise=dn, history=True)
    torch.cuda.synchronize()
    assert "kernel_w2<act 0, draws 0>" in fh.last_kernels(), fh.last_kernels()
    d0, dx = H.rel_err(x0, y0.cpu().numpy()), H.rel_err(xh, yh.cpu().numpy())
    assert d0 <= 2e-6 and dx <= 2e-6, f"in-kernel draws vs frl_random_normal tensor: x0 {d0:.2e}, xt_hist {dx:.2e}"
    sl = np.r_[0:2048, case.B - 2049:case.B]
    sub = dataclasses.replace(case, B=len(sl))
    ref0, refx, refe = H.oracle_run(sub, params, xT[sl], cond[sl], noise[:, sl], dtype=np.float64)
    e0, ex, ee = H.rel_err(x0[sl], ref0), H.rel_err(xh[:, sl], refx), H.rel_err(eh[:, sl], refe)
    print(f"two-wave keyed/{impl}: vs oracle x0 {e0:.2e} xt_hist {ex:.2e} eps_hist {ee:.2e}")
    assert e0 <= 1e-4 and ex <= 1e-4 and ee <= 1e-4


def test_keyed_wide_action_stays_on_32_chain_tiles():
    """The keyed two-wave build draws ONE element per lane (16 x x_dim <= 64); a wider action takes a 32-chain build at
    any batch size, and agrees with the tensor-fed call (which does take the two-wave build)."""
    import torch
    import flowrl_hip as fh
    import flowrl_hip.random as fr
    impl = IMPLS[0]
    case = H.Case("ddpm-a6-large", "ddpm", A=6, S=11, T=4, B=66000, seed=22)
    params, xT, cond, noise, x0, xh, eh = _run_with_key(case, impl, seed=77)
    assert "kernel_w2" not in fh.last_kernels(), fh.last_kernels()     # a 32-chain build (EXT 5 or the latency kernel: per-register draws)
    keys, _ = tf.sampler_key_chain(case.kind, case.solver, case.T, tf.prng_key(77), impl)
    model = H.product_model(case, params)
    dn = torch.stack([fr.normal(_t(keys[j]), xT.shape, impl) for j in range(case.T)])
    _, y0, (yh, _) = model.sample(None, _t(xT), _t(cond), False, case.solver, noise=dn, history=True)
    torch.cuda.synchronize()
    assert "kernel_w2" in fh.last_kernels(), fh.last_kernels()
    d0, dx = H.rel_err(x0, y0.cpu().numpy()), H.rel_err(xh, yh.cpu().numpy())
    assert d0 <= 1e-4 and dx <= 1e-4, f"x0 {d0:.2e}, xt_hist {dx:.2e}"    # two different builds: another f32 summation order, not 2e-6


def test_explicit_noise_wins_and_rng_still_advances():
    import torch
    case = CASES[0]
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    key = tf.prng_key(99)
    _, rng_ref = tf.sampler_key_chain("ddpm", "ddpm", case.T, key)
    rng, x0, _ = model.sample(_t(key), _t(xT), _t(cond), False, "ddpm", noise=_t(noise))
    _, y0, _ = model.sample(None, _t(xT), _t(cond), False, "ddpm", noise=_t(noise))
    torch.cuda.synchronize()
    assert np.array_equal(rng.cpu().numpy(), rng_ref)
    assert np.array_equal(x0.cpu().numpy(), y0.cpu().numpy())


def test_empty_batch_advances_rng():
    import torch
    case = dataclasses.replace(CASES[0], B=0)
    params, xT, cond, _ = H.make_inputs(dataclasses.replace(case, B=4))
    model = H.product_model(case, params)
    key = tf.prng_key(5)
    _, rng_ref = tf.sampler_key_chain("ddpm", "ddpm", case.T, key)
    rng, x0, _ = model.sample(_t(key), _t(xT[:0]), _t(cond[:0]), False, "ddpm")
    torch.cuda.synchronize()
    assert x0.shape == (0, case.A) and np.array_equal(rng.cpu().numpy(), rng_ref)


@pytest.mark.parametrize("impl", IMPLS)
def test_gumbel(impl):
    import flowrl_hip.random as fr
    k = tf.prng_key(21)
    for shape in ((1,), (40, 8), (100001,)):
        got = fr.gumbel(_key(21), shape, impl).cpu().numpy()
        ref = tf.gumbel(k, shape, impl)
        assert got.shape == ref.shape and np.isfinite(got).all()
        # -log(-log(u)): two ocml logs against numpy's; near u -> 1 the outer log amplifies the inner one's ulp
        err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
        assert err.max() <= 2e-6, (shape, err.max())
    g = tf.gumbel(k, (400000,), impl)
    assert abs(float(g.mean()) - 0.5772) < 5e-3 and abs(float(g.var()) - np.pi ** 2 / 6) < 2e-2


@pytest.mark.parametrize("temperature", [None, 0.0, 0.7])
def test_sample_and_select_rng_in_rng_out(temperature):
    """jit_sample_and_select (dql.py:130-162) with a threefry key: split -> xT -> sampler (in-kernel noise) -> critic ->
    [split -> categorical], against the oracle run on the oracle's draws for the same key; the returned rng bit-exact."""
    import torch
    import flowrl_hip as fh
    import flowrl_oracle as orc
    Bs, N, A, S = 40, 8, 3, 11
    c = H.Case("sel", "ddpm", A=A, S=S, T=5, B=Bs * N, particles=N, seed=11)
    params, _, cond, _ = H.make_inputs(c)
    cp = orc.make_critic_params(S, A, (256, 256), 2, 12)
    key = tf.prng_key(77)
    # the reference's sequence on the CPU
    rng, xT_rng = tf.split(key, 2)
    xT = tf.normal(xT_rng, (Bs, N, A))
    noise, rng = tf.sampler_noise("ddpm", "ddpm", c.T, rng, (Bs, N, A))
    x0_ref, _, _ = H.oracle_run(c, params, xT.reshape(Bs * N, A), cond, noise.reshape(c.T, Bs * N, A), dtype=np.float64)
    acts = x0_ref.reshape(Bs, N, A)
    if temperature is None:
        want, clear = acts[:, 0], np.ones(Bs, bool)
    else:
        g = None
        if temperature > 0.0:
            rng, sel = tf.split(rng, 2)
            g = tf.gumbel(sel, (Bs, N)).astype(np.float64)
        _, idx_ref, q_ref = orc.select_best(orc.cast_tree(cp, np.float64), cond.astype(np.float64), acts, 2, "mish",
                                            temperature=temperature or 0.0, gumbel=g)
        score = q_ref / (1e-6 + temperature) + g if g is not None else q_ref
        srt = np.sort(score, axis=-1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-3
        want = acts[np.arange(Bs), idx_ref]
        assert clear.sum() >= 30
    model = H.product_model(c, params)
    critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=[256, 256], activation="mish")), ensemble_size=2)
    rng_out, best = fh.sample_and_select(model, critic, fh.tree_from_numpy(cp, "cuda"), _t(key), _t(cond), False, N, "ddpm", temperature)
    torch.cuda.synchronize()
    assert np.array_equal(rng_out.cpu().numpy(), rng)
    np.testing.assert_allclose(best.cpu().numpy()[clear], want[clear], atol=1e-4)


@pytest.mark.parametrize("impl", IMPLS)
def test_keyed_shards_reproduce_the_unsharded_call(impl):
    """State-sharded keyed sampling (flowrl_hip.parallel.sample_sharded(keyed=True) on N ranks): each shard draws the
    elements [first chain, first chain + its chains) of the UNSHARDED normal(key_j, (B, A)) — bufs.rng_chain_offset /
    rng_total_chains — so the shards together are the single call, bit for bit (same build forced on both sides)."""
    import torch
    case = H.Case("shard", "ddpm", A=3, S=11, T=5, B=6 * 50, particles=50, seed=13, tile="throughput")
    params, xT, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    key = _key(4242)
    kw = dict(particles=50, tile_policy="throughput", rng_impl=impl)
    rng_full, full, _ = model.sample(key, _t(xT), _t(cond), False, "ddpm", **kw)
    parts = []
    for lo, hi in ((0, 1), (1, 4), (4, 6)):                       # ragged state shards
        rng_s, x0, _ = model.sample(key, _t(xT[lo * 50:hi * 50]), _t(cond[lo:hi]), False, "ddpm", rng_shard=(lo * 50, case.B), **kw)
        assert torch.equal(rng_s, rng_full)                       # every shard returns the same advanced rng
        parts.append(x0)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts), full)
    # without the offsets a shard is its own batch: different draws
    _, alone, _ = model.sample(key, _t(xT[50:200]), _t(cond[1:4]), False, "ddpm", **kw)
    assert not torch.equal(alone, full[50:200])
    import flowrl_hip as fh
    with pytest.raises(fh.FlowRLHipError):
        model.sample(key, _t(xT[:50]), _t(cond[:1]), False, "ddpm", rng_shard=(case.B - 10, case.B), **kw)
