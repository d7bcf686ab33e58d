"""ctypes binding of libflowrl_hip.so (the C ABI declared in include/flowrl_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, this module
raises.  Build the library with ``python flow-rl_amd/build.py`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

FRL_ABI_VERSION = 6
FRL_MAX_HIDDEN = 8

FRL_KIND_DDPM, FRL_KIND_CDDPM, FRL_KIND_CNF = 0, 1, 2
FRL_SOLVER_DDPM, FRL_SOLVER_DDIM, FRL_SOLVER_EULER, FRL_SOLVER_DPM2, FRL_SOLVER_DPMPP2M = 0, 1, 2, 3, 4
FRL_TEMB_NONE, FRL_TEMB_POSITIONAL, FRL_TEMB_FOURIER = 0, 1, 2
FRL_ACT = {"mish": 0, "silu": 1, "relu": 2, "gelu_tanh": 3, "gelu": 3, "tanh": 4, "elu": 5}
FRL_DTYPE_F32, FRL_DTYPE_BF16 = 0, 1
FRL_BACKBONE_MLP, FRL_BACKBONE_RESIDUAL, FRL_BACKBONE_SIMBA = 0, 1, 2
FRL_TILE = {"auto": 0, "throughput": 1, "latency": 2, "latency32": 3, "latency16": 4}
FRL_RNG = {"none": 0, "partitionable": 1, "legacy": 2}   # jax.random threefry layouts (include/flowrl_hip.h FRL_RNG_*)

FRL_OK, FRL_EINVAL, FRL_EUNSUPPORTED, FRL_EWORKSPACE, FRL_EHIP = 0, -1, -2, -3, -4

_fp = C.c_void_p  # device pointers travel as integers


class FrlDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("kind", C.c_int32), ("solver", C.c_int32), ("steps", C.c_int32),
        ("n_states", C.c_int64), ("particles", C.c_int32), ("x_dim", C.c_int32), ("cond_dim", C.c_int32),
        ("time_embed", C.c_int32), ("time_dim", C.c_int32), ("n_hidden", C.c_int32),
        ("hidden", C.c_int32 * FRL_MAX_HIDDEN), ("act", C.c_int32), ("clip", C.c_int32),
        ("x_min", C.c_float), ("x_max", C.c_float), ("dtype", C.c_int32), ("want_history", C.c_int32),
        ("cond_embed", C.c_int32), ("cond_in_dim", C.c_int32), ("tile_policy", C.c_int32), ("rng_impl", C.c_int32), ("layer_norm", C.c_int32),
        ("backbone", C.c_int32), ("res_multiplier", C.c_int32),
    ]


class FrlParams(C.Structure):
    _fields_ = [
        ("time_kernel", _fp), ("tmlp_kernel", _fp * 2), ("tmlp_bias", _fp * 2),
        ("kernel", _fp * (FRL_MAX_HIDDEN + 1)), ("bias", _fp * (FRL_MAX_HIDDEN + 1)),
        ("cond_kernel", _fp * 2), ("cond_bias", _fp * 2),
        ("ln_scale", _fp * FRL_MAX_HIDDEN), ("ln_bias", _fp * FRL_MAX_HIDDEN),
        ("res_kernel", (_fp * 2) * FRL_MAX_HIDDEN), ("res_bias", (_fp * 2) * FRL_MAX_HIDDEN),
        ("res_ln_scale", _fp * FRL_MAX_HIDDEN), ("res_ln_bias", _fp * FRL_MAX_HIDDEN),
        ("final_ln_scale", _fp), ("final_ln_bias", _fp),
    ]


class FrlSchedule(C.Structure):
    _fields_ = [("betas", _fp), ("alphas", _fp), ("alpha_hats", _fp), ("postvars", _fp), ("ts", _fp),
                ("mid_ts", _fp), ("mid_alphas", _fp), ("mid_sigmas", _fp)]


class FrlFwdBufs(C.Structure):
    _fields_ = [("xT", _fp), ("cond", _fp), ("noise", _fp), ("x0", _fp), ("xt_hist", _fp), ("eps_hist", _fp),
                ("rng_key", _fp), ("rng_out", _fp), ("rng_chain_offset", C.c_int64), ("rng_total_chains", C.c_int64)]


class FrlBwdBufs(C.Structure):
    _fields_ = [("xt_hist", _fp), ("cond", _fp), ("g_x0", _fp), ("g_xT", _fp),
                ("grad_kernel", _fp * (FRL_MAX_HIDDEN + 1)), ("grad_bias", _fp * (FRL_MAX_HIDDEN + 1)), ("grad_tb", _fp),
                ("grad_tmlp_kernel", _fp * 2), ("grad_tmlp_bias", _fp * 2), ("grad_time_kernel", _fp),
                ("eps_hist", _fp), ("have_stash", C.c_int32), ("grad_cond_kernel", _fp * 2), ("grad_cond_bias", _fp * 2)]


class FrlCriticDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("particles", C.c_int32), ("n_states", C.c_int64), ("obs_dim", C.c_int32),
                ("act_dim", C.c_int32), ("n_hidden", C.c_int32), ("hidden", C.c_int32 * FRL_MAX_HIDDEN), ("act", C.c_int32),
                ("ensemble", C.c_int32)]


class FrlCriticParams(C.Structure):
    _fields_ = [("kernel", _fp * (FRL_MAX_HIDDEN + 1)), ("bias", _fp * (FRL_MAX_HIDDEN + 1))]


class FrlSelectBufs(C.Structure):
    _fields_ = [("obs", _fp), ("actions", _fp), ("gumbel", _fp), ("inv_temperature", C.c_float), ("q", _fp),
                ("best_idx", _fp), ("best_action", _fp)]


class FrlEvalBufs(C.Structure):
    _fields_ = [("x", _fp), ("cond", _fp), ("t", _fp), ("out", _fp)]


class FrlEvalBwdBufs(C.Structure):
    _fields_ = [("x", _fp), ("cond", _fp), ("t", _fp), ("g_out", _fp), ("g_x", _fp),
                ("grad_kernel", _fp * (FRL_MAX_HIDDEN + 1)), ("grad_bias", _fp * (FRL_MAX_HIDDEN + 1)),
                ("grad_tmlp_kernel", _fp * 2), ("grad_tmlp_bias", _fp * 2), ("grad_time_kernel", _fp),
                ("grad_cond_kernel", _fp * 2), ("grad_cond_bias", _fp * 2),
                ("grad_ln_scale", _fp * FRL_MAX_HIDDEN), ("grad_ln_bias", _fp * FRL_MAX_HIDDEN),
                ("grad_res_kernel", (_fp * 2) * FRL_MAX_HIDDEN), ("grad_res_bias", (_fp * 2) * FRL_MAX_HIDDEN),
                ("grad_res_ln_scale", _fp * FRL_MAX_HIDDEN), ("grad_res_ln_bias", _fp * FRL_MAX_HIDDEN),
                ("grad_final_ln_scale", _fp), ("grad_final_ln_bias", _fp)]


FRL_NOISE_DDPM, FRL_NOISE_CDDPM_LINEAR, FRL_NOISE_CDDPM_COSINE, FRL_NOISE_FLOW = 0, 1, 2, 3
FRL_MSE_WORKSPACE_BYTES = 2048


class FlowRLHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libflowrl_hip error {code}: {msg}")
        self.code = code


class FlowRLHipUnsupported(FlowRLHipError):
    """The configuration is valid for the reference but this build has no kernel for it."""


LIB_PATH = Path(os.environ.get("FLOWRL_HIP_LIB", Path(__file__).resolve().parent / "libflowrl_hip.so"))

# every symbol include/flowrl_hip.h declares: name -> (restype, argtypes)
EXPORTS = {
    "frl_abi_version": (C.c_int, []),
    "frl_last_error": (C.c_char_p, []),
    "frl_last_kernels": (C.c_char_p, []),
    "frl_validate": (C.c_int, [C.POINTER(FrlDesc)]),
    "frl_workspace_bytes": (C.c_size_t, [C.POINTER(FrlDesc)]),
    "frl_algorithmic_flops": (C.c_double, [C.POINTER(FrlDesc), C.POINTER(C.c_double)]),
    "frl_hoisted_flops": (C.c_double, [C.POINTER(FrlDesc)]),
    "frl_sampler_prepare": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlSchedule),
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_sampler_run": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlFwdBufs),
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_sampler_fwd": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlSchedule),
                                  C.POINTER(FrlFwdBufs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_static_workspace_limit": (C.c_size_t, []),
    "frl_sampler_fwd_static": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlSchedule),
                                         C.POINTER(FrlFwdBufs), C.c_void_p]),
    "frl_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(FrlDesc)]),
    "frl_sampler_bwd": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlSchedule), C.POINTER(FrlBwdBufs),
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_sampler_fwd_train": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlSchedule), C.POINTER(FrlFwdBufs),
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_critic_workspace_bytes": (C.c_size_t, [C.POINTER(FrlCriticDesc)]),
    "frl_critic_select": (C.c_int, [C.POINTER(FrlCriticDesc), C.POINTER(FrlCriticParams), C.POINTER(FrlSelectBufs),
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_sampler_select_workspace_bytes": (C.c_size_t, [C.POINTER(FrlDesc), C.POINTER(FrlCriticDesc)]),
    "frl_sampler_select_fwd": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlSchedule), C.POINTER(FrlFwdBufs),
                                         C.POINTER(FrlCriticDesc), C.POINTER(FrlCriticParams), C.POINTER(FrlSelectBufs),
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_random_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "frl_random_normal": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "frl_random_bits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "frl_random_gumbel": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "frl_random_uniform": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "frl_random_randint": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "frl_net_eval_workspace_bytes": (C.c_size_t, [C.POINTER(FrlDesc)]),
    "frl_net_eval": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlEvalBufs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_net_eval_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(FrlDesc)]),
    "frl_net_eval_train": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlEvalBufs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_net_eval_bwd": (C.c_int, [C.POINTER(FrlDesc), C.POINTER(FrlParams), C.POINTER(FrlEvalBwdBufs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_add_noise": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float,
                                C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frl_ddpm_onestep_update": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frl_mse_loss": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "frl_selftest_mfma32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once) and type every export.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FlowRLHipError(FRL_EHIP, f"{LIB_PATH} not found: build it with `python flow-rl_amd/build.py` "
                                       "(there is no CPU fallback for the sampler)")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)   # AttributeError if an export is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.frl_abi_version()
    if got != FRL_ABI_VERSION:
        raise FlowRLHipError(FRL_EINVAL, f"ABI mismatch: library {got}, binding {FRL_ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == FRL_OK:
        return
    msg = load().frl_last_error().decode("utf-8", "replace")
    if rc == FRL_EUNSUPPORTED:
        raise FlowRLHipUnsupported(rc, msg)
    raise FlowRLHipError(rc, msg)
