"""GPU parity at BASELINE.json's FULL sizes (cfg2 262144 chains, cfg3 1048576 chains, cfg4 8192 states x 64 particles),
where the float64 oracle cannot run the whole batch in seconds.  Size-independent properties tie the big run to it:

  (1) chains are independent, so a scattered SUBSET of the big batch's rows, run on its own (a small batch: another
      kernel build, other tiles), must reproduce the big run's rows to rounding — and that subset IS small enough for
      the float64 oracle, under the standard f32 bar of tests/test_gpu_parity.py;
  (2) rows with identical inputs give BIT-identical outputs wherever they sit in the batch (tile / wave / workgroup
      independence of the throughput build);
  (3) with a threefry key instead of a noise tensor (cfg2), the in-kernel draws of the subset's elements are the
      oracle's draws for those flat indices of the FULL (B, A) shape.
"""
import dataclasses

import numpy as np
import pytest

import helpers as H
import threefry_oracle as tf

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _subset_check(case, n_sub_states, dup_states):
    import torch
    params, xT, cond, noise = H.make_inputs(case)
    P = case.particles
    n_states = case.B // P
    rng = np.random.default_rng(5)
    # (2) duplicate the first `dup_states` states into the last ones
    xs = xT.reshape(n_states, P, case.A)
    nz = noise.reshape(case.T, n_states, P, case.A)
    xs[-dup_states:] = xs[:dup_states]
    nz[:, -dup_states:] = nz[:, :dup_states]
    if cond is not None:
        cond[-dup_states:] = cond[:dup_states]
    model = H.product_model(case, params)
    x0, xh, eh = H.product_run(case, model, xT, cond, noise, history=False)
    assert x0.shape == (case.B, case.A) and np.isfinite(x0).all()
    big = x0.reshape(n_states, P, case.A)
    assert np.array_equal(big[-dup_states:], big[:dup_states]), "identical rows, different tiles: outputs differ"
    # (1) scattered subset of states -> its own small run -> the oracle
    sel = np.sort(rng.choice(n_states - dup_states, n_sub_states, replace=False))
    sub = dataclasses.replace(case, B=n_sub_states * P)
    sx = np.ascontiguousarray(xs[sel].reshape(-1, case.A))
    sn = np.ascontiguousarray(nz[:, sel].reshape(case.T, -1, case.A))
    sc = None if cond is None else np.ascontiguousarray(cond[sel])
    y0, _, _ = H.product_run(sub, H.product_model(sub, params), sx, sc, sn, history=False)
    want = big[sel].reshape(-1, case.A)
    e_self = H.rel_err(y0, want)
    ref0, refx, _ = H.oracle_run(sub, params, sx, sc, sn, dtype=np.float64)
    o32 = H.oracle_run(sub, params, sx, sc, sn, dtype=np.float32)
    tol = max(1e-5, 4.0 * H.rel_err(o32[0], ref0))
    e_or = H.rel_err(want, ref0)
    print(f"{case.name}: full-size rows vs their own small run {e_self:.2e}; vs float64 oracle {e_or:.2e} (tol {tol:.1e})")
    assert e_self <= tol and e_or <= tol
    return params, xT, cond, model, sel


def test_cfg2_full_size():
    case = H.Case("cfg2-full", "ddpm", A=3, S=11, T=5, B=262144)
    params, xT, cond, model, sel = _subset_check(case, 2048, 1024)
    # (3) the same batch with a key: rows `sel` against the oracle chain on the oracle's draws for those elements
    import torch
    key = tf.prng_key(2024)
    keys, rng_ref = tf.sampler_key_chain("ddpm", "ddpm", case.T, key)
    rng, x0, _ = model.sample(_t(key), _t(xT), _t(cond), False, "ddpm")
    torch.cuda.synchronize()
    assert np.array_equal(rng.cpu().numpy(), rng_ref)
    z = np.stack([tf.normal(keys[j], (case.B, case.A))[sel] for j in range(case.T)])
    sub = dataclasses.replace(case, B=len(sel))
    ref0, _, _ = H.oracle_run(sub, params, xT[sel], cond[sel], z, dtype=np.float64)
    e = H.rel_err(x0.cpu().numpy()[sel], ref0)
    print(f"cfg2-full with a threefry key: rows vs oracle on the oracle's draws {e:.2e}")
    assert e <= 1e-4


def test_cfg3_full_size():
    case = H.Case("cfg3-full", "cnf", A=8, S=39, T=10, B=1048576, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None, clip=False, seed=3)
    _subset_check(case, 1024, 512)


def test_cfg4_full_size():
    case = H.Case("cfg4-full", "cddpm", A=17, S=376, T=20, B=8192 * 64, particles=64, temb="learnable_fourier", schedule="cosine",
                  cond_embed=True, seed=4)
    _subset_check(case, 16, 8)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_cfg5_full_size_vjp(dtype):
    """cfg5 (the DQL chain VJP at cfg2's full batch).  dL/dtheta is a sum over chains and linear in the cotangent:
      * additivity: grads(full batch) = grads(first half) + grads(second half);
      * linearity: grads(g1 + 2 g2) = grads(g1) + 2 grads(g2);
      * dL/dxT is per chain: a scattered subset's rows, run on their own, reproduce the big run's rows —
    to float32 summation noise (f32: different split-K ranges; bf16: additivity sees the same operand
    roundings in every run, so the same bound, while linearity only holds to the variant's stated tolerance — the
    cotangent-dependent operands dL/dz are rounded to bf16, which is not linear: 1.5e-2 as in test_gpu_vjp_bf16.py).  The per-chain arithmetic itself is pinned against the float64 oracle at small sizes in
    tests/test_gpu_vjp.py / test_gpu_vjp_bf16.py."""
    import torch
    from test_gpu_vjp import product_vjp, _flatten
    case = H.Case("cfg5-full", "ddpm", A=3, S=11, T=5, B=262144, clip=False, seed=5, bf16=(dtype == "bf16"))
    params, xT, cond, noise = H.make_inputs(case)
    rng = np.random.default_rng(9)
    g1 = rng.standard_normal((case.B, case.A)).astype(np.float32)
    g2 = rng.standard_normal((case.B, case.A)).astype(np.float32)
    model = H.product_model(case, params)

    def run(c, sl, g):
        grads, gx, _ = product_vjp(c, model, xT[sl], cond[sl], np.ascontiguousarray(noise[:, sl]), np.ascontiguousarray(g[sl]), compute_dtype=dtype)
        torch.cuda.synchronize()
        return {k: v.double().cpu().numpy() for k, v in _flatten(grads).items()}, gx.cpu().numpy()

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    full, gx_full = run(case, slice(None), g1)
    # run-to-run determinism: the weight gradients are per-workgroup partial tiles added in slot order by
    # frl_bwd_reduce_kernel (no float atomics) -> a second call gives the same bits, every leaf
    again, gx_again = run(case, slice(None), g1)
    for k in full:
        assert np.array_equal(full[k], again[k]), f"{k}: VJP not bitwise reproducible"
    assert np.array_equal(gx_full, gx_again)
    half = dataclasses.replace(case, B=case.B // 2)
    ga, _ = run(half, slice(0, case.B // 2), g1)
    gb, _ = run(half, slice(case.B // 2, None), g1)
    worst_add = max(rel(ga[k] + gb[k], full[k]) for k in full)
    f2, _ = run(case, slice(None), g2)
    f12, _ = run(case, slice(None), g1 + 2.0 * g2)
    worst_lin = max(rel(full[k] + 2.0 * f2[k], f12[k]) for k in full)
    sel = np.sort(rng.choice(case.B, 2048, replace=False))
    sub = dataclasses.replace(case, B=len(sel))
    _, gx_sub = run(sub, sel, g1)
    # the small run's forward takes the latency build: its chain states differ from the big run's by float32 rounding,
    # which the backward carries (times 1 / sqrt(alpha_hat)) into dL/dxT — bar: 3e-5 of the largest entry
    e_gx = float(np.abs(gx_sub - gx_full[sel]).max() / np.abs(gx_full[sel]).max())
    print(f"cfg5-full/{dtype}: additivity {worst_add:.2e}, linearity {worst_lin:.2e}, dL/dxT rows vs their own small run {e_gx:.2e}")
    assert np.all([np.isfinite(v).all() for v in full.values()])
    assert worst_add <= 2e-5 and worst_lin <= (1.5e-2 if dtype == "bf16" else 2e-5) and e_gx <= 3e-5


W2_CASES = [
    H.Case("w2-ddpm-ragged", "ddpm", A=3, S=11, T=5, B=65536 + 37),
    H.Case("w2-ddim-relu-wide", "ddpm", A=16, S=48, T=4, B=65536 + 16, solver="ddim", act="relu", clip=False, seed=21),
    H.Case("w2-cnf-nocond", "cnf", A=6, S=0, T=3, B=65536, act="silu", temb=None, clip=False, seed=22, tile="throughput"),
    H.Case("w2-cddpm-elu", "cddpm", A=8, S=17, T=4, B=65536 + 1, act="elu", temb="learnable_fourier", schedule="cosine", seed=23),
    H.Case("w2-two-layers-gelu", "ddpm", A=5, S=20, T=3, B=70000, hidden=(256, 256), act="gelu_tanh", seed=24),
]


@pytest.mark.parametrize("case", W2_CASES, ids=lambda c: c.name)
def test_two_wave_build(case):
    """Batches of >= 65536 chains of a plain 256-wide sampler run on 16-chain tiles at two waves per SIMD
    (frl_sampler_rs_kernel_w2, csrc/sampler_rs.hip; plan_two_wave in csrc/frl_internal.h picks it).  A scattered subset of
    the batch, run on its own (below the threshold: the 32-chain or latency build), must give the same final actions AND
    the same history rows to rounding, and matches the float64 oracle under the standard f32 bar."""
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    x0, xh, eh = H.product_run(case, model, xT, cond, noise, history=True)
    assert x0.shape == (case.B, case.A) and np.isfinite(x0).all()
    rng = np.random.default_rng(11)
    sel = np.sort(np.concatenate([rng.choice(case.B - 64, 1000, replace=False), np.arange(case.B - 64, case.B)]))   # ragged tail included
    sub = dataclasses.replace(case, B=len(sel), tile="auto")
    sn = np.ascontiguousarray(noise[:, sel])
    sc = None if cond is None else np.ascontiguousarray(cond[sel])
    y0, yh, zh = H.product_run(sub, H.product_model(sub, params), np.ascontiguousarray(xT[sel]), sc, sn, history=True)
    ref0, refx, refe = H.oracle_run(sub, params, xT[sel], sc, sn, dtype=np.float64)
    o32 = H.oracle_run(sub, params, xT[sel], sc, sn, dtype=np.float32)
    tol = max(1e-5, 4.0 * H.rel_err(o32[0], ref0))
    e_self = max(H.rel_err(x0[sel], y0), H.rel_err(xh[:, sel], yh), H.rel_err(eh[:, sel], zh))
    e_or = max(H.rel_err(x0[sel], ref0), H.rel_err(xh[:, sel], refx), H.rel_err(eh[:, sel], refe))
    e_first = H.rel_err(eh[0][sel], refe[0])
    print(f"{case.name}: two-wave rows vs their own small run {e_self:.2e}; vs float64 oracle {e_or:.2e} (tol {tol:.1e}); first evaluation {e_first:.2e}")
    assert e_self <= tol and e_or <= tol and e_first <= 1e-5
    again, _, _ = H.product_run(case, model, xT, cond, noise, history=False)
    assert np.array_equal(again, x0), "two-wave build: second run differs (and the history-less build from the history one)"
