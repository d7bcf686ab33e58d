"""GPU: the host wrapper's OPT-IN prepare cache (cache_prepare=True; flowrl_hip/_sampler.py: prepared tables / packed edge layers keyed by the
descriptor and (data_ptr, _version) of every parameter and schedule tensor).  A hit must give the same bits as a
cold call; an in-place parameter update must miss; entries are bounded."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [
    H.Case("ddpm", "ddpm", A=3, S=11, T=5, B=200),
    H.Case("cddpm-particles", "cddpm", A=6, S=64, T=8, B=16 * 8, particles=8, temb="learnable_fourier", schedule="cosine", seed=2),
    H.Case("cnf", "cnf", A=8, S=20, T=6, B=64, temb=None, clip=False, seed=3),
], ids=lambda c: c.name)
def test_cache_hit_is_bit_identical_and_update_misses(case):
    import torch
    from flowrl_hip import _sampler
    _sampler.clear_prepare_cache()
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    cold = H.product_run(case, model, xT, cond, noise, cache_prepare=True)                # miss: prepare + run, entry stored
    n1 = len(_sampler._prep_cache)
    assert n1 == 1
    hot = H.product_run(case, model, xT, cond, noise, cache_prepare=True)                # hit: run only
    assert len(_sampler._prep_cache) == n1
    for a, b in zip(cold, hot):
        assert np.array_equal(a, b)
    ref = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    assert H.rel_err(hot[0], ref[0]) <= 5e-5

    # in-place update of one leaf (what an optimizer step does): the version counter changes, the call must re-prepare
    name = "vel_predictor" if case.kind == "cnf" else "noise_predictor"
    with torch.no_grad():
        model.params[name]["Dense_0"]["bias"].add_(0.25)
        if "MLP_0" in model.params:
            model.params["MLP_0"]["Dense_1"]["bias"].mul_(0.5)
    params[name]["Dense_0"]["bias"] = params[name]["Dense_0"]["bias"] + np.float32(0.25)
    if "MLP_0" in params:
        params["MLP_0"]["Dense_1"]["bias"] = params["MLP_0"]["Dense_1"]["bias"] * np.float32(0.5)
    upd = H.product_run(case, model, xT, cond, noise, cache_prepare=True)
    assert len(_sampler._prep_cache) == n1 + 1
    assert not np.array_equal(upd[0], hot[0])
    ref2 = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    assert H.rel_err(upd[0], ref2[0]) <= 5e-5, "stale prepared tables after an in-place parameter update"


def test_cache_is_bounded_and_optional():
    import torch
    from flowrl_hip import _sampler
    _sampler.clear_prepare_cache()
    case = H.Case("ddpm", "ddpm", A=3, S=11, T=5, B=64)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    x, c, nz = (torch.from_numpy(v).cuda() for v in (xT, cond, noise))
    ref = model.sample(None, x, c, False, "ddpm", noise=nz, cache_prepare=False)[1]
    assert len(_sampler._prep_cache) == 0
    for b in range(1, _sampler.PREP_CACHE_ENTRIES + 5):                     # a new batch size = a new descriptor
        model.sample(None, x[:b], c[:b], False, "ddpm", noise=nz[:, :b].contiguous(), cache_prepare=True)
    assert len(_sampler._prep_cache) == _sampler.PREP_CACHE_ENTRIES
    got = model.sample(None, x, c, False, "ddpm", noise=nz, cache_prepare=True)[1]
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_update_through_dot_data_is_seen_by_default():
    """`p.data.copy_/mul_` (the EMA-target pattern) does not bump torch's version counter, so a cache keyed on
    (data_ptr, _version) would silently serve stale tables.  The cache is therefore opt-in: a default call re-prepares
    and must see the update; an opted-in caller invalidates explicitly with clear_prepare_cache()."""
    import torch
    from flowrl_hip import _sampler
    _sampler.clear_prepare_cache()
    case = H.Case("ddpm", "ddpm", A=3, S=11, T=5, B=96)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    before = H.product_run(case, model, xT, cond, noise)
    assert len(_sampler._prep_cache) == 0, "the prepare cache must be off unless asked for"
    leaf = model.params["noise_predictor"]["Dense_0"]["bias"]
    v0 = leaf._version
    leaf.data.mul_(0.5)                      # bypasses the version counter
    model.params["MLP_0"]["Dense_1"]["bias"].data.add_(0.125)
    assert leaf._version == v0
    params["noise_predictor"]["Dense_0"]["bias"] = params["noise_predictor"]["Dense_0"]["bias"] * np.float32(0.5)
    params["MLP_0"]["Dense_1"]["bias"] = params["MLP_0"]["Dense_1"]["bias"] + np.float32(0.125)
    after = H.product_run(case, model, xT, cond, noise)
    ref = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    assert not np.array_equal(before[0], after[0])
    assert H.rel_err(after[0], ref[0]) <= 5e-5, "stale tables after an update through .data"
    # opted in: explicit invalidation is the contract
    hot = H.product_run(case, model, xT, cond, noise, cache_prepare=True)
    leaf.data.mul_(2.0)
    _sampler.clear_prepare_cache()
    params["noise_predictor"]["Dense_0"]["bias"] = params["noise_predictor"]["Dense_0"]["bias"] * np.float32(2.0)
    again = H.product_run(case, model, xT, cond, noise, cache_prepare=True)
    ref2 = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    assert H.rel_err(again[0], ref2[0]) <= 5e-5 and not np.array_equal(hot[0], again[0])
