"""GPU parity of critic scoring + best-of-N selection (frl_critic_select through flowrl_hip.critic) against the float64
oracle (oracle/flowrl_oracle.py: critic_apply / select_best, restating flowrl/module/critic.py:154-194 and
flowrl/agent/offline/dql.py:151-161).  PARITY UNPINNED by the reference (no tests / fixtures).

  * qs = min over the ensemble:  |hip - oracle64| <= 1e-5 * max(|oracle64|, 1)   (the sampler's fp32 bar)
  * idx: identical to the oracle's argmax wherever the oracle's best and second-best logits are further apart than
    the q tolerance (a float32 tie-break within rounding noise is not an error); chosen actions = actions[idx] exactly.
"""
import numpy as np
import pytest

import flowrl_oracle as orc

pytestmark = pytest.mark.gpu


def _run(B, N, S, A, hidden, act, E, temperature, seed):
    import torch
    import flowrl_hip as fh
    rng = np.random.default_rng(seed)
    params = orc.make_critic_params(S, A, hidden, E, seed)
    obs = rng.standard_normal((B, S)).astype(np.float32) if S > 0 else None
    actions = np.clip(rng.standard_normal((B, N, A)), -1, 1).astype(np.float32)
    gumbel = rng.gumbel(size=(B, N)).astype(np.float32) if temperature and temperature > 0 else None
    p64 = orc.cast_tree(params, np.float64)
    _, idx_ref, q_ref = orc.select_best(p64, None if obs is None else obs.astype(np.float64), actions.astype(np.float64),
                                        len(hidden), act, temperature, None if gumbel is None else gumbel.astype(np.float64))
    critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=list(hidden), activation=act)), ensemble_size=E)
    tp = fh.tree_from_numpy(params, "cuda")
    best, idx, q = fh.critic_select(critic, tp, None if obs is None else torch.from_numpy(obs).cuda(), torch.from_numpy(actions).cuda(),
                                    N, temperature=temperature, gumbel=None if gumbel is None else torch.from_numpy(gumbel).cuda())
    torch.cuda.synchronize()
    q, idx, best = q.cpu().numpy(), idx.cpu().numpy(), best.cpu().numpy()
    err = np.max(np.abs(q - q_ref) / np.maximum(np.abs(q_ref), 1.0))
    assert err <= 1e-5, f"q rel err {err:.2e}"
    scale = 1.0 if gumbel is None else 1.0 / (1e-6 + temperature)
    logits = q_ref * scale + (0.0 if gumbel is None else gumbel.astype(np.float64))
    srt = np.sort(logits, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 4e-5 * scale * np.maximum(np.abs(q_ref).max(), 1.0) if N > 1 else np.ones(B, bool)
    assert (idx[clear] == idx_ref[clear]).all(), f"{(idx[clear] != idx_ref[clear]).sum()} wrong picks"
    assert ((idx >= 0) & (idx < N)).all()
    np.testing.assert_array_equal(best, actions[np.arange(B), idx])
    print(f"B={B} N={N} S={S} A={A} {hidden} {act} E={E} T={temperature}: q err {err:.2e}, {int((~clear).sum())} near-ties")


@pytest.mark.parametrize("cfg", [
    dict(B=100, N=50, S=11, A=3, hidden=(256, 256, 256), act="mish", E=2, temperature=0.0, seed=0),     # DQL eval: 50 candidates
    dict(B=37, N=10, S=17, A=6, hidden=(256, 256), act="relu", E=2, temperature=0.0, seed=1),           # QVPO-style relu critic
    dict(B=64, N=64, S=111, A=8, hidden=(256, 256), act="relu", E=2, temperature=0.0, seed=2),          # ant-sized obs, 64 particles
    dict(B=20, N=7, S=11, A=3, hidden=(256, 256, 256), act="mish", E=2, temperature=1.0, seed=3),       # DQL default: categorical
    dict(B=5, N=1, S=11, A=3, hidden=(256,), act="silu", E=1, temperature=0.0, seed=4),
    dict(B=9, N=130, S=0, A=4, hidden=(256, 256), act="mish", E=3, temperature=0.0, seed=5),            # > 64 candidates, no obs
    dict(B=12, N=16, S=39, A=8, hidden=(512, 512), act="gelu_tanh", E=2, temperature=0.0, seed=6),      # 512-wide geometry
    dict(B=40, N=64, S=376, A=17, hidden=(256, 256), act="relu", E=2, temperature=0.0, seed=7),         # cfg4: QVPO humanoid critic
    dict(B=33, N=1, S=100, A=6, hidden=(256, 256), act="mish", E=2, temperature=0.0, seed=8),           # wide obs, one candidate
    dict(B=33, N=1, S=20, A=6, hidden=(256, 256), act="mish", E=2, temperature=0.0, seed=9),            # direct path (no hoist)
], ids=lambda c: f"B{c['B']}xN{c['N']}-S{c['S']}-{c['act']}-E{c['E']}")
def test_critic_select(cfg):
    _run(**cfg)


def test_critic_select_against_committed_golden():
    """frl_critic_select against tests/golden/golden_critic_select.npz: no oracle run."""
    from pathlib import Path
    import torch
    import flowrl_hip as fh
    z = np.load(Path(__file__).resolve().parent / "golden" / "golden_critic_select.npz")
    cp = orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=[256, 256, 256], activation="mish")), ensemble_size=2)
    best, idx, q = fh.critic_select(critic, fh.tree_from_numpy(cp, "cuda"), torch.from_numpy(z["obs"]).cuda(),
                                    torch.from_numpy(z["actions"]).cuda(), z["actions"].shape[1], temperature=0.0)
    torch.cuda.synchronize()
    q, idx, best = q.cpu().numpy(), idx.cpu().numpy(), best.cpu().numpy()
    assert np.max(np.abs(q - z["q"]) / np.maximum(np.abs(z["q"]), 1.0)) <= 1e-5
    srt = np.sort(z["q"], axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 4e-5 * max(np.abs(z["q"]).max(), 1.0)
    assert (idx[clear] == z["idx"][clear]).all()
    np.testing.assert_array_equal(best, z["actions"][np.arange(len(idx)), idx])


def test_sample_and_select_end_to_end():
    """jit_sample_and_select (dql.py:130-162): sampler (particles broadcast) -> critic -> argmax, against the oracle chain."""
    import torch
    import flowrl_hip as fh
    import helpers as H
    c = H.Case("sel", "ddpm", A=3, S=11, T=5, B=40 * 8, particles=8, seed=11)
    params, xT, cond, noise = H.make_inputs(c)
    cp = orc.make_critic_params(11, 3, (256, 256), 2, 12)
    x0_ref, _, _ = H.oracle_run(c, params, xT, cond, noise, dtype=np.float64)
    _, idx_ref, q_ref = orc.select_best(orc.cast_tree(cp, np.float64), cond.astype(np.float64), x0_ref.reshape(40, 8, 3), 2, "mish")
    model = H.product_model(c, params)
    critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=[256, 256], activation="mish")), ensemble_size=2)
    _, best = fh.sample_and_select(model, critic, fh.tree_from_numpy(cp, "cuda"), None, torch.from_numpy(cond).cuda(), False, 8, "ddpm", 0.0,
                                   xT=torch.from_numpy(xT.reshape(40, 8, 3)).cuda(), noise=torch.from_numpy(noise).cuda())
    torch.cuda.synchronize()
    srt = np.sort(q_ref, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    want = x0_ref.reshape(40, 8, 3)[np.arange(40), idx_ref]
    assert clear.sum() >= 30
    np.testing.assert_allclose(best.cpu().numpy()[clear], want[clear], atol=2e-5)


def test_critic_unsupported_is_loud():
    import torch
    import flowrl_hip as fh
    critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=[384, 384], activation="relu")), ensemble_size=2)
    p = fh.init_critic_params(critic, 11, 3, device="cuda")         # hidden width without a kernel
    with pytest.raises(fh.FlowRLHipUnsupported):
        fh.critic_select(critic, p, torch.zeros((4, 11), device="cuda"), torch.zeros((4, 2, 3), device="cuda"), 2)
