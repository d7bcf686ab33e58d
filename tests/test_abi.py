"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/flowrl_hip.h
declares, and rejects bad descriptors before any launch (no compute calls: there is no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from flowrl_hip import _lib

ROOT = Path(__file__).resolve().parent.parent


def _declared_functions():
    text = (ROOT / "include" / "flowrl_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = _declared_functions()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/flowrl_hip.h but not exported"
        assert n in _lib.EXPORTS, f"{n} has no ctypes signature in flowrl_hip/_lib.py"
    assert sorted(_lib.EXPORTS) == names


def test_abi_version(lib):
    assert lib.frl_abi_version() == _lib.FRL_ABI_VERSION


def _desc(**kw):
    d = _lib.FrlDesc()
    d.abi_version = _lib.FRL_ABI_VERSION
    d.kind, d.solver, d.steps = _lib.FRL_KIND_DDPM, _lib.FRL_SOLVER_DDPM, 5
    d.n_states, d.particles, d.x_dim, d.cond_dim = 262144, 1, 3, 11
    d.time_embed, d.time_dim, d.n_hidden = _lib.FRL_TEMB_POSITIONAL, 64, 3
    for i in range(3):
        d.hidden[i] = 256
    d.act, d.clip, d.x_min, d.x_max, d.dtype = 0, 1, -1.0, 1.0, _lib.FRL_DTYPE_F32
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_validate_and_workspace(lib):
    d = _desc()
    assert lib.frl_validate(C.byref(d)) == _lib.FRL_OK
    ws = lib.frl_workspace_bytes(C.byref(d))
    assert ws > 0 and ws % 256 == 0 and ws < (1 << 20)


def test_algorithmic_flops_match_survey(lib):
    # cfg2 M_step = 78*256 + 2*256^2 + 256*3 = 151,808 -> 3.98e11 FLOP per call
    d = _desc()
    ex = C.c_double(0)
    alg = lib.frl_algorithmic_flops(C.byref(d), C.byref(ex))
    assert alg == 2.0 * 262144 * 5 * 151808
    assert 0.9 * alg < ex.value < 1.0 * alg      # time slice hoisted (-64 rows), edge layers padded
    d = _desc(x_dim=2, cond_dim=1, steps=20, n_states=4096)   # cfg1: 148,736 MACs
    assert lib.frl_algorithmic_flops(C.byref(d), None) == 2.0 * 4096 * 20 * 148736
    # dpm2 evaluates the network twice per step (continuous_ddpm.py:163,191)
    d = _desc(kind=_lib.FRL_KIND_CDDPM, time_embed=_lib.FRL_TEMB_FOURIER, solver=_lib.FRL_SOLVER_DPM2)
    assert lib.frl_validate(C.byref(d)) == _lib.FRL_OK
    assert lib.frl_algorithmic_flops(C.byref(d), None) == 2.0 * 262144 * 10 * 151808


@pytest.mark.parametrize("kw,code", [
    (dict(abi_version=99), _lib.FRL_EINVAL),
    (dict(kind=7), _lib.FRL_EINVAL),
    (dict(steps=0), _lib.FRL_EINVAL),
    (dict(x_dim=0), _lib.FRL_EINVAL),
    (dict(particles=0), _lib.FRL_EINVAL),
    (dict(n_hidden=0), _lib.FRL_EINVAL),
    (dict(act=17), _lib.FRL_EINVAL),
    (dict(time_dim=7), _lib.FRL_EINVAL),
    (dict(x_min=2.0, x_max=1.0), _lib.FRL_EINVAL),
    (dict(kind=_lib.FRL_KIND_CNF, solver=_lib.FRL_SOLVER_DDPM), _lib.FRL_EINVAL),
    (dict(solver=5), _lib.FRL_EINVAL),
    (dict(solver=_lib.FRL_SOLVER_DPM2), _lib.FRL_EINVAL),            # the discrete DDPM knows ddpm / ddim only (ddpm.py:16-18)
    (dict(kind=_lib.FRL_KIND_CDDPM, time_embed=_lib.FRL_TEMB_FOURIER, solver=_lib.FRL_SOLVER_DPMPP2M, dtype=_lib.FRL_DTYPE_BF16),
     _lib.FRL_EUNSUPPORTED),
    (dict(kind=_lib.FRL_KIND_CDDPM, time_embed=_lib.FRL_TEMB_FOURIER, solver=_lib.FRL_SOLVER_DPM2, steps=200), _lib.FRL_EUNSUPPORTED),
    (dict(dtype=_lib.FRL_DTYPE_BF16, x_dim=33), _lib.FRL_EUNSUPPORTED),
    (dict(dtype=_lib.FRL_DTYPE_BF16, act=_lib.FRL_ACT["tanh"]), _lib.FRL_EUNSUPPORTED),
    (dict(dtype=7), _lib.FRL_EINVAL),
    (dict(x_dim=40), _lib.FRL_EUNSUPPORTED),
])
def test_bad_descriptors_rejected(lib, kw, code):
    d = _desc(**kw)
    assert lib.frl_validate(C.byref(d)) == code
    assert lib.frl_last_error() != b""
    assert lib.frl_workspace_bytes(C.byref(d)) == 0


def test_null_and_workspace_errors_before_launch(lib):
    d = _desc()
    p, s, b = _lib.FrlParams(), _lib.FrlSchedule(), _lib.FrlFwdBufs()
    need = lib.frl_workspace_bytes(C.byref(d))
    assert lib.frl_sampler_fwd(C.byref(d), C.byref(p), C.byref(s), C.byref(b), None, need, None) == _lib.FRL_EWORKSPACE
    assert lib.frl_sampler_fwd(C.byref(d), C.byref(p), C.byref(s), C.byref(b), 256, 16, None) == _lib.FRL_EWORKSPACE
    assert lib.frl_sampler_fwd(C.byref(d), C.byref(p), C.byref(s), C.byref(b), 260, need, None) == _lib.FRL_EWORKSPACE
    # aligned fake workspace pointer, NULL parameter leaves -> rejected before any kernel launch
    assert lib.frl_sampler_fwd(C.byref(d), C.byref(p), C.byref(s), C.byref(b), 4096, need, None) == _lib.FRL_EINVAL
    assert b"NULL" in lib.frl_last_error()
    with pytest.raises(_lib.FlowRLHipError):
        _lib.check(_lib.FRL_EINVAL)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.FlowRLHipError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensor_is_refused():
    """The product path never computes on the CPU: a host tensor raises instead of falling back."""
    import numpy as np
    import torch
    import helpers as H
    c = H.Case("cpu", "ddpm", A=3, S=11, T=5, B=4)
    params, xT, cond, noise = H.make_inputs(c)
    model = H.product_model(c, params, device="cpu")
    with pytest.raises(RuntimeError, match="GPU only"):
        model.sample(None, torch.from_numpy(xT), torch.from_numpy(cond), noise=torch.from_numpy(noise))


def test_ffi_shim_compile_smoke():
    """COMPILE SMOKE ONLY: `g++ -fsyntax-only` of flow-rl_amd/csrc/xla_ffi_shim.cc against tests/ffi_stub (a minimal
    in-tree declaration of the xla::ffi names the shim uses — NOT jaxlib's header, which this image lacks).  Catches
    typos, missing includes (hipStream_t) and handler-signature / binding mismatches; says nothing about jaxlib."""
    import shutil
    import subprocess
    root = Path(__file__).resolve().parent.parent
    gxx = shutil.which("g++")
    if gxx is None or not Path("/opt/rocm/include/hip/hip_runtime_api.h").exists():
        pytest.skip("g++ or the HIP runtime headers are not available")
    r = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-D__HIP_PLATFORM_AMD__", f"-I{root / 'tests' / 'ffi_stub'}",
                        f"-I{root / 'include'}", "-I/opt/rocm/include", str(root / "flow-rl_amd" / "csrc" / "xla_ffi_shim.cc")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    src = (root / "flow-rl_amd" / "csrc" / "xla_ffi_shim.cc").read_text()
    for sym in ("FrlSamplerFwd", "FrlSamplerFwdTrain", "FrlSamplerBwd", "frl_sampler_fwd(", "frl_sampler_fwd_train(", "frl_sampler_bwd(", "rng_impl", "cond_embed", "layer_norm", "dtype", "backbone", "have_stash"):
        assert sym in src, sym
