"""Host logic of flowrl_hip.autograd that needs no GPU: the parameter-tree flattening (sorted keys, as jax.tree_util
flattens dicts) and value_and_grad's jax.grad conventions (grads in the layout of params, zeros for unreached leaves,
has_aux)."""
import pytest
import torch

import flowrl_hip.autograd as fa


def _tree():
    return {"noise_predictor": {"Dense_1": {"kernel": torch.arange(6.0).reshape(2, 3), "bias": torch.ones(3)},
                                "Dense_0": {"kernel": torch.full((2, 2), 2.0), "bias": torch.zeros(2)}},
            "MLP_0": {"Dense_0": {"kernel": torch.eye(2), "bias": torch.zeros(2)}}}


def test_tree_roundtrip_and_order():
    t = _tree()
    flat = fa.tree_leaves(t)
    paths = [p for p, _ in flat]
    assert paths == sorted(paths) and paths[0] == ("MLP_0", "Dense_0", "bias") and len(paths) == 6
    back = fa.tree_build(paths, [v for _, v in flat])
    assert fa.tree_leaves(back)[3][1] is flat[3][1] and set(back) == set(t)


def test_value_and_grad_conventions():
    t = _tree()

    def loss_fn(p):
        w = p["noise_predictor"]["Dense_1"]["kernel"]
        assert w.requires_grad and w.data_ptr() == t["noise_predictor"]["Dense_1"]["kernel"].data_ptr()   # aliases, not copies
        return (w ** 2).sum() + p["MLP_0"]["Dense_0"]["bias"].sum(), {"n": 3}

    (loss, aux), g = fa.value_and_grad(loss_fn, t, has_aux=True)
    assert aux == {"n": 3} and float(loss) == 55.0
    assert torch.equal(g["noise_predictor"]["Dense_1"]["kernel"], 2 * t["noise_predictor"]["Dense_1"]["kernel"])
    assert torch.equal(g["MLP_0"]["Dense_0"]["bias"], torch.ones(2))
    assert float(g["noise_predictor"]["Dense_0"]["kernel"].abs().max()) == 0.0      # unreached leaf: zeros, as jax.grad gives
    assert not t["noise_predictor"]["Dense_1"]["kernel"].requires_grad             # the caller's tree is untouched
    g2 = fa.grad(lambda p: (p["MLP_0"]["Dense_0"]["kernel"] * 3).sum(), t)
    assert torch.equal(g2["MLP_0"]["Dense_0"]["kernel"], torch.full((2, 2), 3.0))
    with pytest.raises(TypeError):
        fa.value_and_grad(lambda p: p["MLP_0"]["Dense_0"]["kernel"], t)


def test_condition_gradient_is_refused():
    with pytest.raises(NotImplementedError):
        fa._no_cond_grad(torch.zeros(2, 3, requires_grad=True))
    fa._no_cond_grad(None)
    fa._no_cond_grad(torch.zeros(2, 3))
