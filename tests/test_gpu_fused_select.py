"""GPU parity of the ONE-launch sampler + critic scoring + best-of-N selection (frl_sampler_select_fwd through
flowrl_hip.critic.FusedSelect / sample_and_select(fused=True)): jit_sample_and_select, flowrl/agent/offline/dql.py:130-162
(DAC dac.py:151-183, BDPO bdpo.py:60-92 alike), and jit_sample_actions, flowrl/agent/online/qvpo.py:21-44.
PARITY UNPINNED by the reference (no tests / fixtures): the checker is the float64 oracle.

  * the candidates the fused launch samples (want_actions) are BIT-EQUAL to the throughput build of frl_sampler_fwd
    (same arithmetic, same order) — so the sampler's own parity suite covers them;
  * qs = min over the ensemble on those candidates: |hip - oracle64| <= 1e-5 * max(|oracle64|, 1);
  * idx = the oracle's argmax / Gumbel-max wherever its best and second-best logits are further apart than the q
    tolerance; best = candidates[idx] exactly;
  * at a throughput-sized batch the separate frl_critic_select launches run the same kernel code on the same inputs:
    q, idx and best are bit-equal.
"""
import numpy as np
import pytest

import flowrl_oracle as orc
import helpers as H

pytestmark = pytest.mark.gpu


def _critic(fh, hidden, act, E):
    return fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=list(hidden), activation=act)), ensemble_size=E)


def _sample(model, c, xT, cond, noise, **kw):
    import torch
    x = torch.from_numpy(xT.reshape(c.B // c.particles, c.particles, c.A) if c.particles > 1 else xT).cuda()
    cd = torch.from_numpy(cond).cuda()
    parts = c.particles if c.particles > 1 else None
    if c.kind == "cnf":
        return model.sample(None, x, cd, particles=parts, **kw)
    return model.sample(None, x, cd, False, c.solver, noise=torch.from_numpy(noise).cuda(), particles=parts, **kw)


def _fused(c, chid, cact, E, temperature, seed, check_two_call=False):
    import torch
    import flowrl_hip as fh
    params, xT, cond, noise = H.make_inputs(c)
    n_states, N = c.B // c.particles, c.particles
    cp = orc.make_critic_params(c.S, c.A, chid, E, seed)
    rng = np.random.default_rng(seed)
    gumbel = rng.gumbel(size=(n_states, N)).astype(np.float32) if temperature and temperature > 0 else None
    model = H.product_model(c, params)
    critic = _critic(fh, chid, cact, E)
    tcp = fh.tree_from_numpy(cp, "cuda")
    obs = torch.from_numpy(cond).cuda()
    tg = None if gumbel is None else torch.from_numpy(gumbel).cuda()
    sel = fh.FusedSelect(critic, tcp, obs, temperature=temperature, gumbel=tg, want_q=True, want_actions=True)
    _, x0, _ = _sample(model, c, xT, cond, noise, select=sel)
    torch.cuda.synchronize()
    assert "frl_sampler_rs_fused_kernel" in fh.last_kernels()
    x0n = x0.cpu().numpy().reshape(n_states, N, c.A)
    q, idx, best = sel.q.cpu().numpy(), sel.idx.cpu().numpy(), sel.best.cpu().numpy()
    # candidates: the sampler's throughput build, bit for bit (below the two-wave build's 65536 chains)
    if c.B < 65536:
        _, x0s, _ = _sample(model, c, xT, cond, noise, tile_policy="throughput")
        np.testing.assert_array_equal(x0n, x0s.cpu().numpy().reshape(n_states, N, c.A))
    # scoring + selection against the oracle on those candidates
    _, idx_ref, q_ref = orc.select_best(orc.cast_tree(cp, np.float64), cond.astype(np.float64), x0n.astype(np.float64), len(chid), cact,
                                        temperature, None if gumbel is None else gumbel.astype(np.float64))
    err = np.max(np.abs(q - q_ref) / np.maximum(np.abs(q_ref), 1.0))
    assert err <= 1e-5, f"q rel err {err:.2e}"
    scale = 1.0 if gumbel is None else 1.0 / (1e-6 + temperature)
    logits = q_ref * scale + (0.0 if gumbel is None else gumbel.astype(np.float64))
    srt = np.sort(logits, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 4e-5 * scale * np.maximum(np.abs(q_ref).max(), 1.0) if N > 1 else np.ones(n_states, bool)
    assert ((idx >= 0) & (idx < N)).all()
    assert (idx[clear] == idx_ref[clear]).all(), f"{(idx[clear] != idx_ref[clear]).sum()} wrong picks"
    np.testing.assert_array_equal(best, x0n[np.arange(n_states), idx])
    H.record_margin("fused_select", c.name, "f32", q_err=err, q_bar=1e-5, near_ties=int((~clear).sum()))
    if check_two_call:
        b2, i2, q2 = fh.critic_select(critic, tcp, obs, x0, N, temperature=temperature, gumbel=tg)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(q, q2.cpu().numpy())
        np.testing.assert_array_equal(idx, i2.cpu().numpy())
        np.testing.assert_array_equal(best, b2.cpu().numpy())
    return err


CASES = [
    # DQL evaluation: hopper shapes, 8 candidates per state (4 states per tile), ragged last tile
    (H.Case("fsel_dql_hopper_n8", "ddpm", A=3, S=11, T=5, B=301 * 8, particles=8, seed=31), (256, 256, 256), "mish", 2, 0.0),
    # DQL default temperature: categorical draw as Gumbel-max, 32 candidates = one tile per state
    (H.Case("fsel_dql_categorical_n32", "ddpm", A=3, S=11, T=5, B=77 * 32, particles=32, seed=32), (256, 256), "mish", 2, 1.0),
    # QVPO / DACER acting (cfg4 shapes): 376-wide observation behind the cond_embedding, 17-dim actions, 64 candidates
    # = two consecutive tiles per state, relu critic beside the mish actor
    (H.Case("fsel_cfg4_n64", "cddpm", A=17, S=376, T=20, B=41 * 64, particles=64, temb="learnable_fourier", schedule="cosine",
            cond_embed=True, seed=33), (256, 256), "relu", 2, 0.0),
    # three tiles per state, ddim, silu actor / relu critic
    (H.Case("fsel_n96_ddim", "cddpm", A=6, S=40, T=8, B=23 * 96, particles=96, solver="ddim", act="silu", schedule="linear", seed=34), (256, 256, 256), "relu", 3, 0.0),
    # one candidate per state (selection is the identity), Euler flow
    (H.Case("fsel_cnf_n1", "cnf", A=8, S=20, T=10, B=150, temb=None, clip=False, act="gelu_tanh", seed=35), (256, 256), "gelu_tanh", 1, 0.0),
    # four members, four layers, 4 candidates, categorical
    (H.Case("fsel_e4_l4_n4", "ddpm", A=3, S=11, T=3, B=64 * 4, particles=4, seed=36), (256, 256, 256, 256), "mish", 4, 0.5),
]


@pytest.mark.parametrize("c,chid,cact,E,temperature", CASES, ids=[c[0].name for c in CASES])
def test_fused_select_parity(c, chid, cact, E, temperature):
    err = _fused(c, chid, cact, E, temperature, 500 + c.seed)
    print(f"{c.name}: q err {err:.2e}")


def test_fused_select_bit_equal_to_separate_launches_at_throughput_size():
    """65 536 chains: the separate critic launches run the throughput build (same code as the fused tail) on the candidates
    the fused launch wrote: q / idx / best bit-equal.  (The separate SAMPLER launch picks its two-wave 16-chain build at
    this size, whose summation order differs — the candidates are compared at the smaller sizes above.)"""
    c = H.Case("fsel_bit_equal_64k", "ddpm", A=3, S=11, T=2, B=2048 * 32, particles=32, seed=37)
    _fused(c, (256, 256), "mish", 2, 0.0, 537, check_two_call=True)


def test_sample_and_select_fused_end_to_end():
    """sample_and_select(fused=True) against the oracle chain + oracle selection (as test_sample_and_select_end_to_end)."""
    import torch
    import flowrl_hip as fh
    c = H.Case("fsel_e2e", "ddpm", A=3, S=11, T=5, B=40 * 8, particles=8, seed=11)
    params, xT, cond, noise = H.make_inputs(c)
    cp = orc.make_critic_params(11, 3, (256, 256), 2, 12)
    x0_ref, _, _ = H.oracle_run(c, params, xT, cond, noise, dtype=np.float64)
    _, idx_ref, q_ref = orc.select_best(orc.cast_tree(cp, np.float64), cond.astype(np.float64), x0_ref.reshape(40, 8, 3), 2, "mish")
    model = H.product_model(c, params)
    critic = _critic(fh, (256, 256), "mish", 2)
    _, best = fh.sample_and_select(model, critic, fh.tree_from_numpy(cp, "cuda"), None, torch.from_numpy(cond).cuda(), False, 8, "ddpm", 0.0,
                                   xT=torch.from_numpy(xT.reshape(40, 8, 3)).cuda(), noise=torch.from_numpy(noise).cuda(), fused=True)
    torch.cuda.synchronize()
    assert "frl_sampler_rs_fused_kernel" in fh.last_kernels()
    srt = np.sort(q_ref, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    want = x0_ref.reshape(40, 8, 3)[np.arange(40), idx_ref]
    assert clear.sum() >= 30
    np.testing.assert_allclose(best.cpu().numpy()[clear], want[clear], atol=2e-5)


def test_fused_select_refusals_and_fallback():
    """Configurations without a fused build say so (FRL_EUNSUPPORTED); fused="auto" then takes the separate launches."""
    import torch
    import flowrl_hip as fh
    c = H.Case("fsel_refuse", "ddpm", A=3, S=11, T=3, B=10 * 6, particles=6, seed=38)        # 6 candidates: not a power of two
    params, xT, cond, noise = H.make_inputs(c)
    model = H.product_model(c, params)
    critic = _critic(fh, (256, 256), "mish", 2)
    tcp = fh.tree_from_numpy(orc.make_critic_params(11, 3, (256, 256), 2, 1), "cuda")
    obs = torch.from_numpy(cond).cuda()
    args = dict(xT=torch.from_numpy(xT.reshape(10, 6, 3)).cuda(), noise=torch.from_numpy(noise).cuda())
    with pytest.raises(fh.FlowRLHipUnsupported):
        fh.sample_and_select(model, critic, tcp, None, obs, False, 6, "ddpm", 0.0, fused=True, **args)
    _, a_auto = fh.sample_and_select(model, critic, tcp, None, obs, False, 6, "ddpm", 0.0, fused="auto", **args)
    _, a_sep = fh.sample_and_select(model, critic, tcp, None, obs, False, 6, "ddpm", 0.0, fused=False, **args)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a_auto.cpu().numpy(), a_sep.cpu().numpy())
    # a 512-wide critic, and a threefry key (in-kernel draws), have no fused build either
    c8 = H.Case("fsel_refuse8", "ddpm", A=3, S=11, T=3, B=10 * 8, particles=8, seed=39)
    params, xT, cond, noise = H.make_inputs(c8)
    model = H.product_model(c8, params)
    wide = _critic(fh, (512, 512), "mish", 2)
    twide = fh.tree_from_numpy(orc.make_critic_params(11, 3, (512, 512), 2, 1), "cuda")
    obs = torch.from_numpy(cond).cuda()
    args = dict(xT=torch.from_numpy(xT.reshape(10, 8, 3)).cuda(), noise=torch.from_numpy(noise).cuda())
    with pytest.raises(fh.FlowRLHipUnsupported):
        fh.sample_and_select(model, wide, twide, None, obs, False, 8, "ddpm", 0.0, fused=True, **args)
    with pytest.raises(fh.FlowRLHipUnsupported):
        fh.sample_and_select(model, critic, tcp, fh.random.PRNGKey(0), obs, False, 8, "ddpm", 0.0, fused=True)
