"""GPU parity of the bf16-MFMA chain VJP (BASELINE config 5: "DQL backprop-through-chain VJP ... bf16 MFMA variant"):
DDPM.sample(compute_dtype="bf16", history=True) followed by DDPM.sample_vjp(compute_dtype="bf16") ->
frl_sampler_bwd with FrlDesc.dtype = FRL_DTYPE_BF16 (csrc/sampler_bwd_bf16.hip).  Loss = sum(g * x0).

A STATED-TOLERANCE variant, checked two ways per parameter leaf (and for dL/dxT), error = ||got - ref||_2 / ||ref||_2:

  (1) against the oracle's bf16-OPERAND EMULATION (oracle/flowrl_oracle_vjp.py, BackboneSpec.bf16_operands: the same
      operand roundings in forward and backward, exact sums) evaluated on the SAME chain states (the kernel's own
      history[0]): what remains is f32 accumulation order, the kernel's cheaper activation forms, roundings that flip
      because of those, and the rare clip mask that flips on a chain whose x0-estimate sits within that noise of the
      clip boundary.                                                                          <= TOL_EMU
  (2) against the EXACT float64 oracle (its own chain): the error of the variant itself, no worse than 1.5 x the
      emulation's own error (+ 2e-3) — i.e. it is the error of bf16 operands, not of the kernel — and, where the
      loss is smooth in the chain state (no clipping, or a schedule that does not amplify: linear),  <= TOL_EXACT
      (SURVEY 8d cfg5 states 3e-2 on the gradients; measured 5e-3 .. 1e-2).
      With clip_sampler on the vp / T = 5 schedule of the DQL config and random orthogonal(sqrt 2) weights the loss is
      NOT smooth: 1/sqrt(alpha_hat_T) = 12.5 multiplies the bf16 error of eps into x0-estimates of which most sit at
      the clip boundary, and every chain whose mask (ddpm.py:200-201) flips changes the gradient by O(1/B).  There
      the float64 emulation itself is 0.1 .. 0.4 away from the exact gradient, and so is the kernel; the assertion
      is only the 1.5 x bound (+ 2e-2).  The same holds for relu (its derivative is a step: bf16 noise flips it).

PARITY UNPINNED by the reference (no bf16 path, no VJP fixtures upstream).
"""
import dataclasses

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL_EMU = 1.5e-2
TOL_EXACT = 3e-2


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def _rel_l2(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(np.asarray(got, np.float64) - ref) / max(np.linalg.norm(ref), 1e-30))


def _oracle(case, params, xT, cond, noise, g, xt_hist=None):
    from test_gpu_vjp import oracle_vjp
    _, gref, gx = oracle_vjp(case, params, xT, cond, noise, g, xt_hist=xt_hist)
    out = _flatten(gref)
    out["dL/dxT"] = gx
    return out


def _run(case, stash=False):
    import torch
    from test_gpu_vjp import product_vjp
    assert case.bf16
    params, xT, cond, noise = H.make_inputs(case)
    g = np.random.default_rng(77 + case.seed).standard_normal((case.B, case.A)).astype(np.float32)
    model = H.product_model(case, params)
    grads, gxT, xh = product_vjp(case, model, xT, cond, noise, g, compute_dtype="bf16", stash=stash)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in _flatten(grads).items()}
    got["dL/dxT"] = gxT.cpu().numpy()
    exact = _oracle(dataclasses.replace(case, bf16=False), params, xT, cond, noise, g)
    emu_own = _oracle(case, params, xT, cond, noise, g)
    emu = _oracle(case, params, xT, cond, noise, g, xt_hist=xh.cpu().numpy())
    assert set(got) == set(exact), (sorted(got), sorted(exact))
    smooth = not (case.clip and case.schedule == "vp") and case.act != "relu"
    rows = []
    for k in sorted(exact):
        assert got[k].shape == np.asarray(exact[k]).shape, k
        assert np.isfinite(got[k]).all(), k
        rows.append((k, _rel_l2(got[k], emu[k]), _rel_l2(got[k], exact[k]), _rel_l2(emu_own[k], exact[k])))
    w_emu = max(rows, key=lambda r: r[1])
    w_ex = max(rows, key=lambda r: r[2])
    print(f"{case.name}: worst leaf vs emulation (same states) {w_emu[1]:.2e} ({w_emu[0]}); vs exact {w_ex[2]:.2e} ({w_ex[0]}), "
          f"emulation's own {w_ex[3]:.2e}{'' if smooth else '  [non-smooth loss: clip / relu masks flip]'}")
    H.record_margin("vjp_bf16", case.name, "stash" if stash else "recompute", leaf_l2_vs_emulation=w_emu[1], bar_vs_emulation=TOL_EMU,
                    emulation_frac_of_bar=w_emu[1] / TOL_EMU, leaf_l2_vs_exact=w_ex[2], emulation_own_vs_exact=w_ex[3])
    for k, e_emu, e_ex, e_own in rows:
        assert e_emu <= TOL_EMU, f"{case.name}: {k} vs emulation {e_emu:.2e}"
        assert e_ex <= 1.5 * e_own + (2e-3 if smooth else 2e-2), f"{case.name}: {k} vs exact {e_ex:.2e} (emulation {e_own:.2e})"
        if smooth:
            assert e_ex <= TOL_EXACT, f"{case.name}: {k} vs exact {e_ex:.2e}"


CASES = [
    H.Case("cfg5_dql_b256", "ddpm", A=3, S=11, T=5, B=256, bf16=True),
    H.Case("cfg5_noclip_b512", "ddpm", A=3, S=11, T=5, B=512, clip=False, seed=9, bf16=True),
    H.Case("ragged_b70_linear", "ddpm", A=3, S=11, T=5, B=70, schedule="linear", seed=1, bf16=True),
    H.Case("particles", "ddpm", A=3, S=11, T=5, B=8 * 50, particles=50, schedule="linear", seed=2, bf16=True),
    H.Case("silu_noclip", "ddpm", A=6, S=17, T=4, B=128, act="silu", clip=False, seed=3, bf16=True),
    H.Case("relu_fourier_a16", "ddpm", A=16, S=16, T=3, B=96, act="relu", temb="learnable_fourier", schedule="linear", seed=4, bf16=True),
    H.Case("two_hidden", "ddpm", A=3, S=11, T=5, B=64, hidden=(256, 256), seed=5, bf16=True),
    H.Case("one_hidden_nocond", "ddpm", A=2, S=0, T=6, B=64, hidden=(256,), clip=False, seed=6, bf16=True),
    H.Case("raw_time", "ddpm", A=3, S=11, T=5, B=64, temb=None, schedule="linear", seed=7, bf16=True),
    H.Case("multi_tile", "ddpm", A=3, S=11, T=5, B=20000, clip=False, seed=8, bf16=True),
    H.Case("cddpm_cosine", "cddpm", A=6, S=17, T=20, B=64, temb="learnable_fourier", schedule="cosine", clip=False, seed=10, bf16=True),
    H.Case("cnf_raw_t", "cnf", A=8, S=20, T=10, B=64, temb=None, clip=False, seed=12, bf16=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_vjp_bf16(case):
    _run(case)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_vjp_bf16_training_forward_stash(case):
    """The same bars with the bf16 TRAINING forward (frl_sampler_fwd_train, dtype bf16: the sampler + the stash of
    act(z) / act'(z)) and the backward from that stash (have_stash: no forward MFMAs, no activation VALU in the sweep)."""
    _run(case, stash=True)


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[10], CASES[11]], ids=lambda c: c.name)
def test_bf16_training_forward_is_the_sampler(case):
    """frl_sampler_fwd_train(bf16) returns bit for bit what the plain bf16 sampler returns (x0 and both histories), and the
    stash backward returns bit for bit the gradients of the recompute backward (same dL/dz, same fixed-order reduce)."""
    import torch
    from test_gpu_vjp import product_vjp
    params, xT, cond, noise = H.make_inputs(case)
    g = np.random.default_rng(5).standard_normal((case.B, case.A)).astype(np.float32)
    model = H.product_model(case, params)
    a, b = {}, {}
    g0, gx0, _ = product_vjp(case, model, xT, cond, noise, g, compute_dtype="bf16", stash=False, out=a)
    g1, gx1, _ = product_vjp(case, model, xT, cond, noise, g, compute_dtype="bf16", stash=True, out=b)
    torch.cuda.synchronize()
    for k in ("x0", "xt_hist", "eps_hist"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(gx0, gx1)
    f0, f1 = _flatten(g0), _flatten(g1)
    for k in f0:
        assert torch.equal(f0[k], f1[k]), k


def test_vjp_bf16_unsupported_is_loud():
    import torch
    import flowrl_hip as fh
    case = H.Case("wide", "ddpm", A=17, S=24, T=3, B=32)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    xh = torch.zeros((3, 32, 17), device="cuda")
    with pytest.raises(fh.FlowRLHipUnsupported):
        model.sample_vjp(xh, torch.zeros((32, 17), device="cuda"), torch.from_numpy(cond).cuda(), compute_dtype="bf16")
