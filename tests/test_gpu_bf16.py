"""GPU parity of the bf16-MFMA variant (FrlDesc.dtype = FRL_DTYPE_BF16, `compute_dtype="bf16"`): bf16 operands,
f32 accumulation, f32 biases / time branch / activations / chain state.  It is a STATED-TOLERANCE variant
(BASELINE.json north_star), checked two ways against the float64 oracle on identical inputs:

  (1) against the oracle's bf16-OPERAND EMULATION (BackboneSpec.bf16_operands: the same roundings, exact sums):
      what remains is f32 accumulation order, the kernel's cheaper activation forms, and roundings that flip
      because of those (one bf16 ulp = 2^-8 relative on single layer inputs).
      eps_hist:  rms |hip - emu| / max(|emu|, 1)  <=  TOL_EMU_RMS
  (2) against the EXACT float64 oracle: the error of the variant itself.  x0 (clipped to +-1 where the config clips):
      rms |hip - exact| <= TOL_X0_RMS, and the HIP max error is no worse than 1.5 x the emulation's own (+ 1e-3) —
      i.e. the error is that of bf16 operands, not of the kernel.
      Measured on MI355X (random orthogonal(sqrt 2) weights, the worst case for error growth): hip-vs-emulation eps rms
      7e-7 .. 2e-4; x0 rms error 2e-4 .. 9e-3, worst element 6e-4 .. 3e-1 (isolated chains next to a clip boundary), identical to the
      emulation's to 3 digits.

PARITY UNPINNED by the reference (no bf16 path, no fixtures exist upstream).
"""
import dataclasses

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL_EMU_RMS = 3e-3
TOL_X0_RMS = 2e-2


def _check(case):
    assert case.bf16
    params, xT, cond, noise = H.make_inputs(case)
    exact = H.oracle_run(dataclasses.replace(case, bf16=False), params, xT, cond, noise, dtype=np.float64)
    emu = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    model = H.product_model(case, params)
    x0, xh, eh = H.product_run(case, model, xT, cond, noise, history=True)
    assert np.isfinite(x0).all()
    # (1) first step's network output: identical inputs, so only kernel-vs-emulation arithmetic differs
    e_first = H.rms_err(eh[0], emu[2][0])
    e_hist = H.rms_err(eh, emu[2])
    # (2) end-to-end error of the variant
    err_hip = float(np.max(np.abs(x0 - exact[0])))
    err_emu = float(np.max(np.abs(emu[0] - exact[0])))
    rms_hip = float(np.sqrt(np.mean((x0 - exact[0]) ** 2)))
    print(f"{case.name}: eps rms vs emulation first-step {e_first:.2e} all-steps {e_hist:.2e}; "
          f"x0 vs exact: hip max {err_hip:.2e} rms {rms_hip:.2e}, emulation max {err_emu:.2e}")
    assert e_first <= TOL_EMU_RMS and e_hist <= 4 * TOL_EMU_RMS, case.name
    assert rms_hip <= TOL_X0_RMS and err_hip <= 1.5 * err_emu + 1e-3, case.name


CASES = [
    H.Case("cfg2_dql", "ddpm", A=3, S=11, T=5, B=4096, bf16=True),
    H.Case("ragged_b70", "ddpm", A=3, S=11, T=5, B=70, seed=1, bf16=True),
    H.Case("one_hidden", "ddpm", A=2, S=1, T=6, B=200, hidden=(256,), clip=False, schedule="linear", seed=2, bf16=True),
    H.Case("two_hidden_silu", "ddpm", A=6, S=17, T=4, B=300, hidden=(256, 256), act="silu", seed=3, bf16=True),
    H.Case("four_hidden_relu_ddim", "ddpm", A=8, S=24, T=3, B=260, hidden=(256,) * 4, act="relu", solver="ddim", seed=4, bf16=True),
    H.Case("cnf_gelu_raw_t", "cnf", A=8, S=39, T=10, B=500, act="gelu_tanh", temb=None, clip=False, seed=5, bf16=True),
    H.Case("wide_cond_particles", "cddpm", A=17, S=100, T=6, B=16 * 64, particles=64, temb="learnable_fourier",
           schedule="cosine", seed=6, bf16=True),
    H.Case("cfg4_cond_embed", "cddpm", A=17, S=376, T=20, B=8 * 64, particles=64, temb="learnable_fourier",
           schedule="cosine", cond_embed=True, seed=7, bf16=True),
    # the 512-wide build (one wave per SIMD, 128-chain workgroup tiles): cfg3's FQL net, ragged batches, a DDPM net with a time branch
    H.Case("cfg3_fql_h512", "cnf", A=8, S=39, T=10, B=700, hidden=(512,) * 4, act="gelu_tanh", temb=None, clip=False, seed=8, bf16=True),
    H.Case("h512_ddpm_mish", "ddpm", A=6, S=17, T=5, B=130, hidden=(512, 512), seed=9, bf16=True),
    H.Case("h512_one_hidden_relu", "ddpm", A=16, S=64, T=4, B=64, hidden=(512,), act="relu", schedule="linear", seed=10, bf16=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_bf16_variant(case):
    _check(case)


def test_bf16_unsupported_is_loud():
    import flowrl_hip as fh
    for case in (H.Case("h128", "cnf", A=8, S=39, T=10, B=64, hidden=(128,) * 2, act="gelu_tanh", temb=None, clip=False, bf16=True),
                 H.Case("h512_wide_cond", "cnf", A=8, S=100, T=10, B=64, hidden=(512,) * 2, act="gelu_tanh", temb=None, clip=False, bf16=True)):
        params, xT, cond, noise = H.make_inputs(case)
        model = H.product_model(case, params)
        with pytest.raises(fh.FlowRLHipUnsupported):
            H.product_run(case, model, xT, cond, noise)
