"""flowrl_hip.model — the host mirror of Model.apply_gradient (flowrl/module/model.py:78-86): optax.adam behind an optional
optax.clip_by_global_norm, restated on torch tensors.  optax is not importable here and the reference holds no optimizer
fixtures: PARITY UNPINNED; the check is a float64 numpy restatement of the published update rule (Kingma & Ba with bias
correction, as optax.scale_by_adam writes it) run side by side for several steps, plus known answers of the clip."""
import numpy as np
import pytest
import torch

import flowrl_hip.autograd as fa
import flowrl_hip.model as fm


def _tree(rng):
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    return {"noise_predictor": {"Dense_0": {"kernel": mk(5, 4), "bias": mk(4)}, "Dense_1": {"kernel": mk(4, 2), "bias": mk(2)}}}


def _np_adam(p, g, mu, nu, k, lr, b1=0.9, b2=0.999, eps=1e-8):
    mu = b1 * mu + (1 - b1) * g
    nu = b2 * nu + (1 - b2) * g * g
    return p - lr * (mu / (1 - b1 ** k)) / (np.sqrt(nu / (1 - b2 ** k)) + eps), mu, nu


def test_adam_and_clip_against_numpy_restatement():
    rng = np.random.default_rng(0)
    params = _tree(rng)
    target = _tree(rng)
    state = fm.TrainState.create(params, fm.adam(3e-2), clip_grad_norm=0.7)
    ref = {"/".join(p): v.double().numpy() for p, v in fa.tree_leaves(params)}
    mu = {k: np.zeros_like(v) for k, v in ref.items()}
    nu = {k: np.zeros_like(v) for k, v in ref.items()}
    tgt = {"/".join(p): v.double().numpy() for p, v in fa.tree_leaves(target)}

    class M:      # the smallest "model": params + replace
        def __init__(self, params): self.params = params
        def replace(self, params): return M(params)

    def loss_fn(p, _rng):
        l = sum(((a - b) ** 2).sum() for (_, a), (_, b) in zip(fa.tree_leaves(p), fa.tree_leaves(target)))
        return l, {"loss": l.detach()}

    m = M(params)
    losses = []
    for k in range(1, 9):
        m_new, state, info = fm.apply_gradient(m, state, loss_fn)
        losses.append(float(info["loss"]))
        g = {n: 2.0 * (ref[n] - tgt[n]) for n in ref}
        norm = np.sqrt(sum((v ** 2).sum() for v in g.values()))
        if not norm < 0.7:
            g = {n: v / norm * 0.7 for n, v in g.items()}
        for n in ref:
            ref[n], mu[n], nu[n] = _np_adam(ref[n], g[n], mu[n], nu[n], k, 3e-2)
        for path, v in fa.tree_leaves(m_new.params):
            assert np.abs(v.double().numpy() - ref["/".join(path)]).max() <= 2e-6, (k, path)
        assert all(a is not b for (_, a), (_, b) in zip(fa.tree_leaves(m.params), fa.tree_leaves(m_new.params)))   # functional update
        m = m_new
    assert state.step == 8 and state.opt_state["count"] == 8 and losses[-1] < losses[0]


def test_clip_known_answers_and_empty_optimizer():
    g = {"a": {"kernel": torch.tensor([3.0, 0.0]), "bias": torch.tensor([4.0])}}
    assert float(fm.global_norm(g)) == 5.0
    c = fm.clip_by_global_norm(g, 10.0)
    assert torch.equal(c["a"]["kernel"], g["a"]["kernel"]) and torch.equal(c["a"]["bias"], g["a"]["bias"])     # below: untouched
    c = fm.clip_by_global_norm(g, 2.5)
    assert torch.allclose(c["a"]["kernel"], torch.tensor([1.5, 0.0])) and torch.allclose(c["a"]["bias"], torch.tensor([2.0]))
    st = fm.TrainState.create(g)                      # Model.create(optimizer=None): the empty optimizer
    with pytest.raises(ValueError, match="Empty optimizer"):
        st.apply_gradients(g, g)
    st = fm.TrainState.create(g, fm.adam(1e-3))
    with pytest.raises(KeyError):
        st.apply_gradients(g, {"a": {"kernel": g["a"]["kernel"]}})


def test_continuous_tables_cache_survives_replace():
    """`model.apply_gradient` replaces the model every optimizer step; the device-table cache of
    ContinuousDDPM must travel with `replace()` (same schedule function / t_diffusion), and miss when they change."""
    import numpy as np
    import flowrl_hip as fh
    from flowrl_hip.flow import ContinuousDDPM
    net = fh.DDPMBackbone(noise_predictor=fh.MLP(hidden_dims=[256, 256], output_dim=3, activation="mish"),
                          time_embedding=fh.LearnableFourierEmbedding(output_dim=64), cond_embedding=None)
    inputs = (np.ones((1, 3)), np.zeros((1, 1)), np.ones((1, 5)))
    m = ContinuousDDPM.create(net, 0, inputs, x_dim=3, steps=6, noise_schedule="cosine", device="cpu")
    t1 = m._device_tables(6, m.t_schedule_n, False, "cpu")
    m2 = m.replace(params=m.params)
    assert m2 is not m
    t2 = m2._device_tables(6, m.t_schedule_n, False, "cpu")
    assert all(t1[k] is t2[k] for k in t1), "replace() dropped the table cache: every step would rebuild and re-upload it"
    m3 = m2.replace(t_diffusion=[1e-3, 0.9])
    t3 = m3._device_tables(6, m.t_schedule_n, False, "cpu")
    assert t3["ts"] is not t1["ts"] and float(t3["ts"][-1]) != float(t1["ts"][-1])
