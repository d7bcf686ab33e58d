"""CPU tests of the oracle (oracle/flowrl_oracle.py): known answers, table values, float32-vs-float64,
an independent torch restatement, and the committed golden vectors.  PARITY UNPINNED by the reference
(no tests / fixtures exist upstream) — these are what pins the oracle instead."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

import flowrl_oracle as orc
import helpers as H

GOLD = Path(__file__).resolve().parent / "golden"


# ---- table values hand-computed in SURVEY.md appendix A.6 -----------------------------------------
def test_vp_tables_T5():
    b, a, ah, pv = orc.ddpm_tables(5, "vp", dtype=np.float64)
    np.testing.assert_allclose(b, [0, 0.195874558, 0.458818193, 0.635781020, 0.754878188, 0.835031379], atol=2e-9)
    np.testing.assert_allclose(ah, [1, 0.804125442, 0.435178059, 0.158500109, 0.038851834, 0.006409333], atol=2e-9)
    np.testing.assert_allclose(pv, [0, 0, 0.159113527, 0.426741671, 0.660907377, 0.807766121], atol=2e-9)
    b32, _, ah32, pv32 = orc.ddpm_tables(5, "vp", dtype=np.float32)
    assert b32.dtype == np.float32
    np.testing.assert_allclose(ah32, ah, rtol=5e-7)
    np.testing.assert_allclose(pv32, pv, rtol=2e-6)
    # DAC / BDPO: approx_postvar -> postvars = betas (dac.py:232)
    assert np.array_equal(orc.ddpm_tables(5, "vp", approx_postvar=True)[3], orc.ddpm_tables(5, "vp")[0])


def test_linear_tables_T20():
    _, _, ah, _ = orc.ddpm_tables(20, "linear", dtype=np.float64)
    np.testing.assert_allclose([ah[1], ah[10], ah[20]], [0.9999, 0.952854108, 0.816777103], atol=2e-9)


def test_cosine_discrete_tables():
    b, a, ah, pv = orc.ddpm_tables(10, "cosine", dtype=np.float64)
    assert b[0] == 0 and (b[1:] > 0).all() and (b <= 0.999).all()
    # alphas_cumprod ratio definition (ddpm.py:30-32)
    t = np.linspace(0, 10, 11) / 10
    ac = np.cos((t + 0.008) / 1.008 * np.pi / 2) ** 2
    np.testing.assert_allclose(ah[1:], np.cumprod(np.clip(ac[1:] / ac[:-1], 1 - 0.999, 1)), rtol=1e-12)


def test_continuous_cosine_tables_T20():
    ts, ah, al, be = orc.continuous_tables(20, "cosine", dtype=np.float64)
    np.testing.assert_allclose([ts[0], ts[1], ts[20]], [0.001, 0.05068, 0.9946], atol=1e-12)
    np.testing.assert_allclose([ah[0], ah[1], ah[10], ah[20]],
                               [0.9999587158, 0.9918156344, 0.4972723019, 7.082122865e-05], rtol=2e-9)
    np.testing.assert_allclose([be[1], be[10], be[20]], [0.008143418, 0.134313654, 0.990364901], atol=2e-9)
    assert al[0] == 1.0 and be[0] == 0.0


def test_continuous_linear_tables():
    ts, ah, al, be = orc.continuous_tables(10, "linear", dtype=np.float64)
    assert ts[0] == 1e-3 and ts[-1] == 1.0
    np.testing.assert_allclose(ah, np.exp(-(19.9 / 4) * ts ** 2 - 0.05 * ts) ** 2, rtol=1e-12)


def test_positional_embedding():
    e = orc.positional_embedding(np.zeros((3, 1)), 64)
    assert np.array_equal(e[:, :32], np.ones((3, 32))) and np.array_equal(e[:, 32:], np.zeros((3, 32)))   # cos first
    t = np.full((1, 1), 5.0)
    e = orc.positional_embedding(t, 64)[0]
    f = np.exp(np.arange(32) * -(np.log(10000.0) / 31))
    np.testing.assert_allclose([f[0], f[1], f[31]], [1.0, 0.7429639508, 1.0e-4], rtol=1e-9)
    np.testing.assert_allclose(e, np.concatenate([np.cos(5 * f), np.sin(5 * f)]), rtol=1e-12)


def test_fourier_embedding():
    w = np.linspace(-0.3, 0.3, 32).reshape(32, 1)
    e = orc.learnable_fourier_embedding(np.full((2, 1), 0.25), w)
    np.testing.assert_allclose(e[0], np.concatenate([np.cos(2 * np.pi * 0.25 * w[:, 0]), np.sin(2 * np.pi * 0.25 * w[:, 0])]), rtol=1e-12)


def test_activations_closed_form():
    x = np.linspace(-30, 30, 2001)
    np.testing.assert_allclose(orc.mish(x), x * np.tanh(np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)), rtol=1e-12)
    en = np.exp(x) * (np.exp(x) + 2)                      # rational form used by the kernel
    np.testing.assert_allclose(orc.mish(x), x * en / (en + 2), rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(orc.silu(x), x / (1 + np.exp(-x)), rtol=1e-12)
    u = math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)
    np.testing.assert_allclose(orc.gelu_tanh(x), x / (1 + np.exp(-2 * u)), rtol=1e-9, atol=1e-12)   # sigmoid form used by the kernel
    assert orc.relu(np.array([-1.0, 2.0])).tolist() == [0.0, 2.0]
    np.testing.assert_allclose(orc.elu(np.array([-1.0, 2.0])), [math.expm1(-1.0), 2.0])


# ---- closed-form sampler answers ----------------------------------------------------------------------
def _zero_net(spec, cond_dim, bias_out):
    p = orc.cast_tree(orc.make_params(spec, cond_dim, seed=0), np.float64)
    for name, sub in p.items():
        if name == "time_embedding":
            continue
        for d in sub.values():
            d["kernel"] = np.zeros_like(d["kernel"])
            d["bias"] = np.zeros_like(d["bias"])
    last = p[spec.predictor_name][f"Dense_{len(spec.hidden_dims)}"]
    last["bias"] = np.asarray(bias_out, dtype=np.float64)
    return p


def test_ddpm_zero_weight_net_closed_form():
    """All-zero weights => eps == output bias e0 at every step: the chain is an affine recursion."""
    T, A = 5, 3
    spec = orc.BackboneSpec([256, 256, 256], A)
    e0 = np.array([0.3, -0.2, 0.1])
    p = _zero_net(spec, 11, e0)
    tabs = orc.ddpm_tables(T, "vp", dtype=np.float64)
    b, al, ah, pv = tabs
    rng = np.random.default_rng(0)
    xT, cond, nz = rng.standard_normal((4, A)), rng.standard_normal((4, 11)), rng.standard_normal((T, 4, A))
    x0, (xh, eh) = orc.ddpm_sample(spec, p, tabs, xT, cond, nz, steps=T, clip_sampler=True, x_min=-1, x_max=1)
    x = xT
    for j, t in enumerate(range(T, 0, -1)):
        assert np.allclose(eh[j], e0) and np.allclose(xh[j], x)
        x0h = np.clip((x - np.sqrt(1 - ah[t]) * e0) / np.sqrt(ah[t]), -1, 1)
        x = (np.sqrt(ah[t - 1]) * (1 - al[t]) * x0h + np.sqrt(al[t]) * (1 - ah[t - 1]) * x) / (1 - ah[t])
        x = x + (t > 1) * np.sqrt(pv[t]) * nz[j]
    np.testing.assert_allclose(x0, x, rtol=1e-12)


def test_ddpm_T1_no_noise():
    """T = 1: x0 = clip(x0_hat) exactly (coefficient of xt is 0, noise is multiplied by (t>1) = 0)."""
    spec = orc.BackboneSpec([256], 2)
    p = orc.cast_tree(orc.make_params(spec, 3, seed=1), np.float64)
    tabs = orc.ddpm_tables(1, "linear", dtype=np.float64)
    rng = np.random.default_rng(1)
    xT, cond = rng.standard_normal((5, 2)), rng.standard_normal((5, 3))
    nz = 1e6 * np.ones((1, 5, 2))
    x0, (xh, eh) = orc.ddpm_sample(spec, p, tabs, xT, cond, nz, steps=1, clip_sampler=True, x_min=-0.5, x_max=0.5)
    ah = tabs[2]
    want = np.clip((xT - np.sqrt(1 - ah[1]) * eh[0]) / np.sqrt(ah[1]), -0.5, 0.5)
    np.testing.assert_allclose(x0, want, rtol=1e-9, atol=1e-12)
    assert (np.abs(x0) <= 0.5 + 1e-12).all()


def test_ddim_uses_alpha_not_alpha_hat_and_clips_xt():
    spec = orc.BackboneSpec([256], 2)
    e0 = np.array([5.0, -5.0])
    p = _zero_net(spec, 0, e0)
    tabs = orc.ddpm_tables(3, "vp", dtype=np.float64)
    _, al, ah, _ = tabs
    xT = np.array([[0.2, -0.1]])
    x0, _ = orc.ddpm_sample(spec, p, tabs, xT, None, np.zeros((3, 1, 2)), steps=3, solver="ddim", clip_sampler=True, x_min=-1, x_max=1)
    x = xT
    for t in (3, 2, 1):
        x = np.clip((x - np.sqrt(1 - ah[t]) * e0) / np.sqrt(al[t]) + np.sqrt(1 - ah[t - 1]) * e0, -1, 1)
    np.testing.assert_allclose(x0, x, rtol=1e-12)


def test_dpm_solvers_known_answers():
    """continuous_ddpm.py:179-212.  (1) A network whose output does not depend on (x, t) makes dpm2's second evaluation
    equal the first, so the midpoint step collapses to the ddim step.  (2) DPM-Solver++(2M)'s first step has no
    history (prev_h = 0) and is the first-order update, which is algebraically the ddim update when nothing clips:
    (sigma_next / sigma_curr) xt - alpha_next expm1(-h) x0_hat  ==  alpha_next x0_hat + sigma_next eps.
    (3) Its second step uses r = prev_h / h and the previous x0_hat, checked against a hand-written recursion."""
    T, A = 6, 3
    spec = orc.BackboneSpec([256, 256], A, time_embedding="learnable_fourier")
    e0 = np.array([0.3, -0.2, 0.1])
    p = _zero_net(spec, 5, e0)
    p["time_embedding"] = {"kernel": np.zeros((32, 1))}
    tabs = orc.continuous_tables(T, "cosine", dtype=np.float64)
    ts, ah, al, be = tabs
    rng = np.random.default_rng(5)
    xT, cond = 0.1 * rng.standard_normal((4, A)), rng.standard_normal((4, 5))
    nz = np.zeros((T, 4, A))
    kw = dict(steps=T, clip_sampler=False, noise_schedule="cosine")
    ddim, _ = orc.continuous_ddpm_sample(spec, p, tabs, xT, cond, nz, solver="ddim", **kw)
    dpm2, (xh2, eh2) = orc.continuous_ddpm_sample(spec, p, tabs, xT, cond, nz, solver="dpm2", **kw)
    np.testing.assert_allclose(dpm2, ddim, rtol=1e-12, atol=1e-14)
    assert xh2.shape == (T, 4, A) and np.allclose(eh2, e0)
    # (2): one step of each from the same state
    tabs1 = orc.continuous_tables(1, "cosine", dtype=np.float64)
    a1, _ = orc.continuous_ddpm_sample(spec, p, tabs1, xT, cond, nz[:1], solver="ddim", steps=1)
    b1, _ = orc.continuous_ddpm_sample(spec, p, tabs1, xT, cond, nz[:1], solver="dpm++2m", steps=1)
    np.testing.assert_allclose(b1, a1, rtol=1e-9, atol=1e-12)
    # (3): full multistep recursion with a constant eps
    lam = np.log(np.sqrt(ah) / np.sqrt(1 - ah))
    x, prev, prev_h = xT, np.zeros_like(xT), 0.0
    for i in range(T, 0, -1):
        x0h = (x - np.sqrt(1 - ah[i]) * e0) / np.sqrt(ah[i])
        h = lam[i - 1] - lam[i]
        d = x0h if prev_h <= 1e-8 else (1 + 0.5 * h / prev_h) * x0h - (0.5 * h / prev_h) * prev
        x = np.sqrt(1 - ah[i - 1]) / np.sqrt(1 - ah[i]) * x - np.sqrt(ah[i - 1]) * np.expm1(-h) * d
        prev, prev_h = x0h, h
    got, _ = orc.continuous_ddpm_sample(spec, p, tabs, xT, cond, nz, solver="dpm++2m", **kw)
    np.testing.assert_allclose(got, x, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(got, ddim, rtol=1e-9, atol=1e-11)   # a constant eps is integrated exactly by both
    # ... and with a real network the second-order correction is really applied
    pr = orc.cast_tree(orc.make_params(spec, 5, seed=4), np.float64)
    r1, _ = orc.continuous_ddpm_sample(spec, pr, tabs, xT, cond, nz, solver="ddim", **kw)
    r2, _ = orc.continuous_ddpm_sample(spec, pr, tabs, xT, cond, nz, solver="dpm++2m", **kw)
    r3, _ = orc.continuous_ddpm_sample(spec, pr, tabs, xT, cond, nz, solver="dpm2", **kw)
    assert np.abs(r2 - r1).max() > 1e-3 and np.abs(r3 - r1).max() > 1e-3


def test_cnf_constant_velocity():
    """vel == const => x1 = x0 + const (Euler is exact), histories ordered by loop iteration."""
    spec = orc.BackboneSpec([256, 256], 4, activation="gelu_tanh", time_embedding=None, predictor_name="vel_predictor")
    v = np.array([1.0, -2.0, 0.5, 0.0])
    p = _zero_net(spec, 7, v)
    rng = np.random.default_rng(2)
    x0, cond = rng.standard_normal((6, 4)), rng.standard_normal((6, 7))
    x1, (xh, vh) = orc.cnf_sample(spec, p, x0, cond, steps=10)
    np.testing.assert_allclose(x1, x0 + v, rtol=1e-12)
    np.testing.assert_allclose(xh[3], x0 + 0.3 * v, rtol=1e-12)
    assert np.allclose(vh, v) and xh.shape == (10, 6, 4)


def test_cnf_raw_time_is_input_column_A():
    """time_embedding=None: the scalar t is concatenated between x and cond (cnf.py:38-41)."""
    spec = orc.BackboneSpec([256], 2, activation="relu", time_embedding=None, predictor_name="vel_predictor")
    p = orc.cast_tree(orc.make_params(spec, 3, seed=3), np.float64)
    x, c = np.ones((1, 2)), np.ones((1, 3))
    a = orc.backbone_apply(spec, p, x, np.full((1, 1), 0.0), c)
    b = orc.backbone_apply(spec, p, x, np.full((1, 1), 0.7), c)
    W0, b0 = p["vel_predictor"]["Dense_0"]["kernel"], p["vel_predictor"]["Dense_0"]["bias"]
    assert W0.shape == (2 + 1 + 3, 256)
    h = np.maximum(np.concatenate([x, [[0.7]], c], -1) @ W0 + b0, 0)
    np.testing.assert_allclose(b, h @ p["vel_predictor"]["Dense_1"]["kernel"] + p["vel_predictor"]["Dense_1"]["bias"], rtol=1e-12)
    assert not np.allclose(a, b)


def test_clip_saturation():
    c = H.Case("sat", "ddpm", A=3, S=11, T=5, B=16, x_min=-0.25, x_max=0.25)
    params, xT, cond, noise = H.make_inputs(c)
    x0, xh, eh = H.oracle_run(c, params, xT, cond, np.zeros_like(noise))
    assert (np.abs(x0) <= 0.25 * 5.2).all()   # last step: x0 = 5.1 * (0.196 * clip(x0h) + ...) stays bounded by the clip


# ---- float32 vs float64, independent torch restatement ----------------------------------------------------
@pytest.mark.parametrize("case", [
    H.Case("cfg2", "ddpm", A=3, S=11, T=5, B=64),
    H.Case("cfg1", "ddpm", A=2, S=1, T=20, B=64, schedule="linear", clip=False),
    H.Case("cddpm", "cddpm", A=6, S=17, T=20, B=64, temb="learnable_fourier", schedule="cosine"),
    H.Case("cddpm-dpm2", "cddpm", A=6, S=17, T=20, B=64, temb="learnable_fourier", schedule="cosine", solver="dpm2"),
    H.Case("cddpm-dpmpp2m", "cddpm", A=6, S=17, T=20, B=64, temb="learnable_fourier", schedule="cosine", solver="dpm++2m"),
    H.Case("cnf", "cnf", A=8, S=39, T=10, B=64, temb=None, clip=False, act="gelu_tanh"),
], ids=lambda c: c.name)
def test_float32_oracle_tracks_float64(case):
    params, xT, cond, noise = H.make_inputs(case)
    r64 = H.oracle_run(case, params, xT, cond, noise, np.float64)
    r32 = H.oracle_run(case, params, xT, cond, noise, np.float32)
    assert r32[0].dtype == np.float32
    tol = 1e-4 if case.solver.startswith("dpm") else 2e-5   # dpm2 chains 2 T = 40 float32 evaluations; 2M differences log-SNRs in float32
    for a, b in zip(r32, r64):
        assert H.rel_err(a, b) < tol


def _torch_backbone(case, params, x, t, cond):
    """Independent restatement of DDPMBackbone with torch ops (different BLAS / transcendental library)."""
    P = lambda *k: torch.from_numpy(np.asarray(_get(params, k), dtype=np.float64))
    def _get(tree, keys):
        for k in keys:
            tree = tree[k]
        return tree
    act = {"mish": torch.nn.functional.mish, "silu": torch.nn.functional.silu, "relu": torch.relu,
           "gelu_tanh": lambda v: torch.nn.functional.gelu(v, approximate="tanh"), "tanh": torch.tanh,
           "elu": torch.nn.functional.elu}[case.act]
    half = 32
    if case.temb == "positional":
        f = torch.exp(torch.arange(half, dtype=torch.float64) * -(math.log(10000.0) / (half - 1)))
        te = torch.cat([torch.cos(t * f), torch.sin(t * f)], -1)
    else:
        f = 2 * math.pi * t @ P("time_embedding", "kernel").T
        te = torch.cat([torch.cos(f), torch.sin(f)], -1)
    for i in range(2):
        te = torch.nn.functional.mish(te @ P("MLP_0", f"Dense_{i}", "kernel") + P("MLP_0", f"Dense_{i}", "bias"))
    h = torch.cat([x, te, cond], -1)
    n = len(case.hidden)
    for i in range(n):
        h = act(h @ P("noise_predictor", f"Dense_{i}", "kernel") + P("noise_predictor", f"Dense_{i}", "bias"))
    return h @ P("noise_predictor", f"Dense_{n}", "kernel") + P("noise_predictor", f"Dense_{n}", "bias")


@pytest.mark.parametrize("act,temb", [("mish", "positional"), ("silu", "learnable_fourier"), ("gelu_tanh", "positional"),
                                      ("elu", "positional"), ("tanh", "positional"), ("relu", "positional")])
def test_network_against_torch_restatement(act, temb):
    case = H.Case("net", "ddpm", A=3, S=11, T=5, B=32, act=act, temb=temb, seed=7)
    params, xT, cond, _ = H.make_inputs(case)
    p64 = orc.cast_tree(params, np.float64)
    t = np.full((case.B, 1), 3.0)
    got = orc.backbone_apply(case.spec, p64, xT.astype(np.float64), t, cond.astype(np.float64))
    want = _torch_backbone(case, params, torch.from_numpy(xT).double(), torch.from_numpy(t), torch.from_numpy(cond).double()).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)


# ---- golden vectors ---------------------------------------------------------------------------------------------
def _load_golden(name):
    z = np.load(GOLD / f"{name}.npz")
    params = orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    return z, params


@pytest.mark.parametrize("case", [
    H.Case("golden_cfg1_toy2d", "ddpm", A=2, S=1, T=20, B=48, schedule="linear", clip=False, x_min=-4.0, x_max=4.0, seed=101),
    H.Case("golden_cfg2_hopper", "ddpm", A=3, S=11, T=5, B=48, seed=102),
    H.Case("golden_cddpm_cosine", "cddpm", A=6, S=17, T=20, B=48, temb="learnable_fourier", schedule="cosine", seed=103),
    H.Case("golden_cnf_raw_t", "cnf", A=8, S=20, T=10, B=48, temb=None, clip=False, seed=104),
], ids=lambda c: c.name)
def test_oracle_reproduces_golden(case):
    z, params = _load_golden(case.name)
    x0, xh, eh = H.oracle_run(case, params, z["xT"], z["cond"], z["noise"], np.float64)
    np.testing.assert_allclose(x0, z["x0"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(xh, z["xt_hist"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(eh, z["eps_hist"], rtol=1e-11, atol=1e-13)
    # float32 oracle stays within the parity tolerance of the stored float64 answer
    r32 = H.oracle_run(case, params, z["xT"], z["cond"], z["noise"], np.float32)
    assert H.rel_err(r32[0], z["x0"]) < 2e-5


def _golden_params(z, case):
    """Parameters of a golden file: stored verbatim ("p:" entries), or — cfg3's 3.3 MB of 512-wide kernels — regenerated
    from the stored seed and checked against the stored SHA-256 of their bytes."""
    import hashlib
    if "params_sha256" not in z.files:
        return orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    assert int(z["params_seed"]) == case.seed
    params = orc.make_params(case.spec, case.S, seed=case.seed, cond_in_dim=case.S)
    h = hashlib.sha256()
    for k, v in sorted(orc.flatten_tree(params).items()):
        h.update(k.encode())
        h.update(np.ascontiguousarray(v, dtype=np.float32).tobytes())
    assert h.hexdigest() == str(z["params_sha256"]), "the parameter generator changed: regenerate tests/golden"
    return params


GOLDEN_R2 = [
    H.Case("golden_cfg3_fql_512x4", "cnf", A=8, S=39, T=10, B=24, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None, clip=False, seed=105),
    H.Case("golden_cfg4_dacer_embed", "cddpm", A=17, S=376, T=20, B=4 * 8, particles=8, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, seed=106),
]


GOLDEN_RES = [
    H.Case("golden_res_dac_resnet", "ddpm", A=3, S=11, T=5, B=24, hidden=(256, 256), backbone="residual", ln=True, seed=109),
    H.Case("golden_simba_cnf", "cnf", A=6, S=17, T=10, B=24, hidden=(256,), backbone="simba", act="relu", seed=110),
]


@pytest.mark.parametrize("case", GOLDEN_R2 + GOLDEN_RES, ids=lambda c: c.name)
def test_oracle_reproduces_golden_r2(case):
    z = np.load(GOLD / f"{case.name}.npz")
    params = _golden_params(z, case)
    x0, xh, eh = H.oracle_run(case, params, z["xT"], z["cond"], z["noise"], np.float64)
    np.testing.assert_allclose(x0, z["x0"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(xh, z["xt_hist"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(eh, z["eps_hist"], rtol=1e-11, atol=1e-13)


def test_oracle_reproduces_golden_eval_vjp():
    """tests/golden/golden_eval_vjp.npz: the float64 backward of ONE per-sample-time evaluation (every leaf + dL/dx)."""
    import flowrl_oracle_vjp as ov
    z = np.load(GOLD / "golden_eval_vjp.npz")
    case = H.Case("golden_eval_vjp", "cddpm", A=6, S=17, T=20, B=40, temb="learnable_fourier", schedule="cosine", act="silu", seed=111)
    p64 = orc.cast_tree(orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")}), np.float64)
    out, cache = ov.backbone_fwd(case.spec, p64, z["x"].astype(np.float64), z["t"].astype(np.float64), z["cond"].astype(np.float64))
    gref: dict = {}
    gx = ov.backbone_bwd(case.spec, p64, cache, z["g"].astype(np.float64), gref)
    np.testing.assert_allclose(out, z["out"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gx, z["g_x"], rtol=1e-10, atol=1e-12)
    flat = orc.flatten_tree(gref)
    assert {"g:" + k for k in flat} == {k for k in z.files if k.startswith("g:")}
    for k, v in flat.items():
        np.testing.assert_allclose(v, z["g:" + k], rtol=1e-10, atol=1e-12)


def test_oracle_reproduces_golden_critic():
    z = np.load(GOLD / "golden_critic_select.npz")
    cp = orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    best, idx, q = orc.select_best(orc.cast_tree(cp, np.float64), z["obs"].astype(np.float64), z["actions"].astype(np.float64), 3, "mish")
    np.testing.assert_allclose(q, z["q"], rtol=1e-11, atol=1e-13)
    assert np.array_equal(idx, z["idx"]) and np.array_equal(best, z["best"])


def test_bf16_round_matches_torch():
    import torch
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.integers(-6, 6, 4096), [0.0, 1.0, 1.00390625, -1.01171875]])
    want = torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    assert np.array_equal(orc.bf16_round(a), want)
    assert orc.bf16_round(a.astype(np.float32)).dtype == np.float32


def test_critic_select_known_answers():
    """critic.py:154-194 + dql.py:151-161.  A critic whose members are q_e(obs, a) = w_e . a + b_e (zero hidden
    kernels would kill the input, so one relu layer passes |a| through): the min over the ensemble and the argmax
    over the candidates are computable by hand; the categorical draw is argmax(logits + Gumbel)."""
    S, A, H_, E, B, N = 2, 2, 256, 2, 3, 5
    p = orc.make_critic_params(S, A, (H_,), E, 0, dtype=np.float64, bias_std=0.0)
    k0 = np.zeros((E, S + A, H_)); k1 = np.zeros((E, H_, 1)); b1 = np.zeros((E, 1))
    # hidden unit 0 = relu(a_0), unit 1 = relu(-a_0): q_0 = |a_0| ... member 1 scores a_1 + 10
    k0[:, S + 0, 0], k0[:, S + 0, 1] = 1.0, -1.0
    k0[1, S + 1, 2], k0[1, S + 1, 3] = 1.0, -1.0
    k1[0, 0, 0] = k1[0, 1, 0] = 1.0                       # member 0: |a_0|
    k1[1, 2, 0], k1[1, 3, 0] = 1.0, -1.0; b1[1, 0] = 10.0  # member 1: a_1 + 10
    p["base"]["backbone"]["Dense_0"]["kernel"] = k0
    p["base"]["Dense_0"] = {"kernel": k1, "bias": b1}
    rng = np.random.default_rng(3)
    obs, act = rng.standard_normal((B, S)), rng.uniform(-1, 1, (B, N, A))
    q = orc.critic_apply(p, np.repeat(obs[:, None], N, 1), act, 1, "relu")
    assert q.shape == (E, B, N, 1)
    np.testing.assert_allclose(q[0, ..., 0], np.abs(act[..., 0]), atol=1e-12)
    np.testing.assert_allclose(q[1, ..., 0], act[..., 1] + 10.0, atol=1e-12)
    chosen, idx, qs = orc.select_best(p, obs, act, 1, "relu", temperature=0.0)
    np.testing.assert_allclose(qs, np.abs(act[..., 0]), atol=1e-12)           # |a_0| <= 1 < a_1 + 10: member 0 is the min
    assert (idx == np.abs(act[..., 0]).argmax(-1)).all()
    np.testing.assert_array_equal(chosen, act[np.arange(B), idx])
    g = np.zeros((B, N)); g[:, 3] = 1e6                                       # overwhelming Gumbel noise picks candidate 3
    assert (orc.select_best(p, obs, act, 1, "relu", temperature=1.0, gumbel=g)[1] == 3).all()


def test_layer_norm_known_answers():
    """flax.linen.LayerNorm restated (oracle layer_norm / mlp_apply(ln=True), mlp.py:26-29): closed forms and an
    independent torch evaluation (torch's layer_norm uses the two-pass variance; float64 makes the forms agree)."""
    import torch
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 256))
    p = {"scale": 1.0 + 0.2 * rng.standard_normal(256), "bias": 0.1 * rng.standard_normal(256)}
    y = orc.layer_norm(p, x)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (256,), torch.from_numpy(p["scale"]), torch.from_numpy(p["bias"]), eps=1e-6).numpy()
    assert np.abs(y - ref).max() < 1e-12
    # identity parameters: zero mean, unit variance (up to eps); constant rows map to the bias exactly
    one = {"scale": np.ones(256), "bias": np.zeros(256)}
    z = orc.layer_norm(one, x)
    assert np.abs(z.mean(-1)).max() < 1e-12 and np.abs(z.var(-1) - 1).max() < 1e-5
    c = orc.layer_norm(p, np.full((2, 256), 3.25))
    assert np.array_equal(c, np.broadcast_to(p["bias"], (2, 256)))
    # a LayerNorm backbone differs from the plain one and uses the LayerNorm_i leaves
    spec = orc.BackboneSpec(hidden_dims=[256, 256], x_dim=3, layer_norm=True)
    prm = orc.make_params(spec, 11, seed=1, dtype=np.float64)
    assert set(prm["noise_predictor"]) == {"Dense_0", "Dense_1", "Dense_2", "LayerNorm_0", "LayerNorm_1"}
    xx, t, cd = rng.standard_normal((4, 3)), np.full((4, 1), 2.0), rng.standard_normal((4, 11))
    a = orc.backbone_apply(spec, prm, xx, t, cd)
    b = orc.backbone_apply(orc.BackboneSpec(hidden_dims=[256, 256], x_dim=3), prm, xx, t, cd)
    assert a.shape == (4, 3) and np.abs(a - b).max() > 1e-3


def test_residual_backbones_against_torch_restatement():
    """residual_mlp_apply / simba_apply (mlp.py:37-98, simba.py:10-63) against an independent torch float64 restatement
    built from torch.nn.functional primitives (layer_norm, linear): structure, LayerNorm placement, head."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    for backbone, ln in (("residual", True), ("residual", False), ("simba", True)):
        spec = orc.BackboneSpec(hidden_dims=[256, 256], x_dim=3, activation="mish", time_embedding=None, backbone=backbone, layer_norm=ln)
        p = orc.cast_tree(orc.make_params(spec, 5, seed=3), np.float64)
        x, t, c = rng.standard_normal((7, 3)), rng.uniform(size=(7, 1)), rng.standard_normal((7, 5))
        got = orc.backbone_apply(spec, p, x, t, c)
        q = p["noise_predictor"]
        T = lambda a: torch.from_numpy(np.asarray(a))
        lin = lambda d, v: F.linear(v, T(d["kernel"]).T, T(d["bias"]))
        lnf = lambda d, v: F.layer_norm(v, (256,), T(d["scale"]), T(d["bias"]), eps=1e-6)
        h = lin(q["Dense_0"], T(np.concatenate([x, t, c], -1)))
        blk = "SimbaBlock" if backbone == "simba" else "ResidualLinear"
        for i in range(2):
            b = q[f"{blk}_{i}"]
            y = lnf(b["LayerNorm_0"], h) if (ln or backbone == "simba") else h
            h = h + lin(b["Dense_1"], F.mish(lin(b["Dense_0"], y)))
        want = lin(q["Dense_1"], lnf(q["LayerNorm_0"], h) if backbone == "simba" else F.mish(h))
        assert np.abs(got - want.numpy()).max() <= 1e-12 * max(1.0, np.abs(got).max()), backbone
