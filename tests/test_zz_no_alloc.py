"""GPU: frl_sampler_fwd_static — the one entry point that allocates nothing (include/flowrl_hip.h).

It is frl_sampler_fwd with the workspace in a file-scope __device__ array inside the library.  Checked here, through
the C ABI, against frl_sampler_fwd itself (bit-identical: same kernels, same tables) and against the float64 oracle,
on two calls in a row — the second with a smaller input, so stale contents of the shared array would show — plus a
cond_embedding + per-state-hoist call (per-state rows in the static array), a keyed call (iteration keys there), and
the refusal of a call whose workspace exceeds the static limit.  Collected last (file name) so that a failure here does
not hide the other tests under `pytest -x`."""
import ctypes as C

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _both(case, params, xT, cond, noise):
    model = H.product_model(case, params)
    dyn = H.product_run(case, model, xT, cond, noise)                        # frl_sampler_fwd, caller workspace
    sta = H.product_run(case, model, xT, cond, noise, static_scratch=True)   # frl_sampler_fwd_static
    return dyn, sta


def test_static_matches_fwd_two_calls_second_smaller():
    big = H.Case("static-big", "ddpm", A=3, S=11, T=5, B=3000, seed=5)
    small = H.Case("static-small", "ddpm", A=3, S=11, T=5, B=70, seed=6)
    for case in (big, small):
        params, xT, cond, noise = H.make_inputs(case)
        dyn, sta = _both(case, params, xT, cond, noise)
        for a, b in zip(dyn, sta):
            assert np.array_equal(a, b), f"{case.name}: static-scratch entry point differs from frl_sampler_fwd"
        ref = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
        r32 = H.oracle_run(case, params, xT, cond, noise, dtype=np.float32)
        for k in (0, 2):   # the suite's bar (tests/test_gpu_parity.py): 1e-5, or 4x numpy float32's own deviation
            assert H.rel_err(sta[k], ref[k]) <= max(1e-5, 4.0 * H.rel_err(r32[k], ref[k])), case.name


def test_static_other_shapes_back_to_back():
    """Different descriptors share the one array: a wide cond_embedding + hoist call (cfeat / zs rows per state), then a
    512-wide CNF call, then a tiny call again."""
    cases = [
        H.Case("static-cddpm-embed", "cddpm", A=17, S=376, T=6, B=64 * 16, particles=16, temb="learnable_fourier",
               schedule="cosine", cond_embed=True, seed=7),
        H.Case("static-cnf-512", "cnf", A=8, S=39, T=4, B=500, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None,
               clip=False, seed=8),
        H.Case("static-tiny", "ddpm", A=2, S=1, T=20, B=33, schedule="linear", clip=False, seed=9),
    ]
    for case in cases:
        params, xT, cond, noise = H.make_inputs(case)
        dyn, sta = _both(case, params, xT, cond, noise)
        for a, b in zip(dyn, sta):
            assert np.array_equal(a, b), case.name
        ref = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
        r32 = H.oracle_run(case, params, xT, cond, noise, dtype=np.float32)
        for k in (0, 2):   # the suite's bar (tests/test_gpu_parity.py): 1e-5, or 4x numpy float32's own deviation
            assert H.rel_err(sta[k], ref[k]) <= max(1e-5, 4.0 * H.rel_err(r32[k], ref[k])), case.name


def test_static_keyed_call_returns_same_rng():
    import torch
    import flowrl_hip as fh
    case = H.Case("static-keyed", "ddpm", A=3, S=11, T=5, B=400, seed=10)
    params, xT, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    key = fh.random.PRNGKey(1234, device="cuda")
    x, c = torch.from_numpy(xT).cuda(), torch.from_numpy(cond).cuda()
    r1, a1, _ = model.sample(key, x, c, False, "ddpm")
    r2, a2, _ = model.sample(key, x, c, False, "ddpm", static_scratch=True)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(r1, r2)


def test_static_refuses_beyond_the_limit(lib):
    import torch
    from flowrl_hip import _lib
    assert lib.frl_static_workspace_limit() == 12 << 20
    # per-state rows (128 cfeat + 256 hoisted floats) x 20000 states = 30 MB > 12 MiB
    case = H.Case("static-too-big", "cddpm", A=17, S=376, T=4, B=20000 * 4, particles=4, temb="learnable_fourier",
                  schedule="cosine", cond_embed=True, seed=11)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    with pytest.raises(_lib.FlowRLHipError) as ei:
        H.product_run(case, model, xT, cond, noise, static_scratch=True)
    assert ei.value.code == _lib.FRL_EWORKSPACE and "static" in str(ei.value)
    x0 = H.product_run(case, model, xT, cond, noise)[0]        # the ordinary entry point takes it
    assert np.isfinite(x0).all()
    torch.cuda.synchronize()
