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


@pytest.mark.parametrize("cond_embed", [False, True], ids=["plain-w2-shape", "cond-embed-hoist"])
def test_prepared_workspace_serves_any_batch_size(cond_embed):
    """frl_sampler_run used to need the n_states frl_sampler_prepare saw (the packed edge-layer images
    switch geometry at 65 536 chains of a plain 256-wide sampler; per-state rows sat in front of prepared regions).  Now
    everything prepare writes is independent of n_states: ONE prepare at a small batch, then runs at batches on both
    sides of the threshold over the same workspace, each bit-identical to its own frl_sampler_fwd."""
    import ctypes as C
    import torch
    from flowrl_hip import _lib, _sampler
    lib = _lib.load()
    if cond_embed:
        case = H.Case("dacer-like", "cddpm", A=6, S=40, T=4, B=70016, particles=8, temb="learnable_fourier", schedule="cosine", cond_embed=True)
    else:
        case = H.Case("dql", "ddpm", A=3, S=11, T=3, B=70016)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    dev = torch.device("cuda", torch.cuda.current_device())
    x, c, nz = (torch.from_numpy(v).to(dev) for v in (xT, cond, noise))
    keep = []
    kind = _lib.FRL_KIND_DDPM if case.kind == "ddpm" else _lib.FRL_KIND_CDDPM

    def desc_for(n_states):
        return _sampler.build_desc(model.network, kind=kind, solver=_lib.FRL_SOLVER_DDPM, steps=case.T, n_states=n_states,
                                   particles=case.particles, x_dim=case.A, cond_dim=case.S, clip_sampler=True, x_min=-1.0, x_max=1.0,
                                   want_history=False)

    fp = _sampler.build_params(model.network, model.params, dev, keep)
    fs = _lib.FrlSchedule()
    if case.kind == "ddpm":
        tabs = dict(betas=model.betas, alphas=model.alphas, alpha_hats=model.alpha_hats, postvars=model.postvars)
    else:
        from flowrl_hip import schedules as sch
        ts, ah, al, be = sch.continuous_tables(case.T, model.noise_schedule_func, model.t_diffusion, model.t_schedule_n)
        tabs = {k: torch.from_numpy(v).to(dev) for k, v in dict(ts=ts, alpha_hats=ah, alphas=al, betas=be).items()}
    for k, v in tabs.items():
        keep.append(v)
        setattr(fs, k, v.data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    n_big = case.B // case.particles
    sizes = [n_big, 1000 // case.particles, 37]
    need = max(lib.frl_workspace_bytes(C.byref(desc_for(n))) for n in sizes)
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    d_small = desc_for(sizes[1])
    _lib.check(lib.frl_sampler_prepare(C.byref(d_small), C.byref(fp), C.byref(fs), ws.data_ptr(), need, stream))     # ONCE, at the small batch
    for n in sizes:
        B = n * case.particles
        d = desc_for(n)
        xb, cb, nb = x[:B].contiguous(), c[:n].contiguous(), nz[:, :B].contiguous()
        got, ref = torch.empty_like(xb), torch.empty_like(xb)
        bufs = _lib.FrlFwdBufs()
        bufs.xT, bufs.cond, bufs.noise, bufs.x0 = xb.data_ptr(), cb.data_ptr(), nb.data_ptr(), got.data_ptr()
        _lib.check(lib.frl_sampler_run(C.byref(d), C.byref(fp), C.byref(bufs), ws.data_ptr(), need, stream))
        ws2 = torch.zeros(need, dtype=torch.uint8, device=dev)
        bufs.x0 = ref.data_ptr()
        _lib.check(lib.frl_sampler_fwd(C.byref(d), C.byref(fp), C.byref(fs), C.byref(bufs), ws2.data_ptr(), need, stream))
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        assert torch.equal(got, ref), f"run at n_states = {n} over a workspace prepared at n_states = {sizes[1]} differs from its own frl_sampler_fwd"
    o = H.oracle_run(H.Case(**{**case.__dict__, "B": 37 * case.particles}), params, xT[:37 * case.particles], cond[:37], noise[:, :37 * case.particles], dtype=np.float64)
    assert H.rel_err(got.cpu().numpy(), o[0]) <= 1e-4
