"""jax.random's threefry2x32 key handling on the GPU (csrc/rng.hip, csrc/frl_rng.h): the pieces the callers either side
of the samplers use (dql.py:142-146 splits the agent's rng and draws xT before calling ``sample``), and the key type the
samplers accept as ``rng`` to draw their per-step noise in-kernel (ddpm.py:188,204-206).

A key is a (2,) ``torch.uint32`` tensor on the GPU holding ``jax.random.key_data(key)``.  ``impl`` selects the counter
layout: "partitionable" (jax_threefry_partitionable=True, the default of the reference's jax 0.5.3) or "legacy".
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Sequence, Union

import numpy as np
import torch

from . import _lib

DEFAULT_IMPL = "partitionable"


def PRNGKey(seed: int, device: Union[str, torch.device] = "cuda") -> torch.Tensor:
    """jax.random.PRNGKey(seed) as raw key data: [seed >> 32, seed & 0xffffffff]."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return torch.tensor([seed >> 32, seed & 0xFFFFFFFF], dtype=torch.uint32).to(device)


def is_key(rng) -> bool:
    if isinstance(rng, torch.Tensor):
        return rng.dtype == torch.uint32 and tuple(rng.shape) == (2,)
    return isinstance(rng, np.ndarray) and rng.dtype == np.uint32 and rng.shape == (2,)


def as_key(rng, device) -> torch.Tensor:
    if isinstance(rng, np.ndarray):
        rng = torch.from_numpy(np.ascontiguousarray(rng))
    if not is_key(rng):
        raise TypeError("rng key: a (2,) uint32 tensor (jax.random.key_data) expected")
    return rng.to(device).contiguous()


def impl_id(impl) -> int:
    impl = DEFAULT_IMPL if impl is None else impl
    if impl not in ("partitionable", "legacy"):
        raise ValueError(f"rng impl {impl!r}: 'partitionable' or 'legacy'")
    return _lib.FRL_RNG[impl]


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def split(key: torch.Tensor, num: int = 2, impl: str = None) -> torch.Tensor:
    """jax.random.split(key, num) -> (num, 2) uint32."""
    lib = _lib.load()
    key = as_key(key, key.device if isinstance(key, torch.Tensor) and key.is_cuda else "cuda")
    out = torch.empty((int(num), 2), dtype=torch.uint32, device=key.device)
    with torch.cuda.device(key.device):
        _lib.check(lib.frl_random_split(key.data_ptr(), impl_id(impl), int(num), out.data_ptr(), _stream(key.device)))
    return out


def _fill(fn_name, key, shape, dtype, impl):
    lib = _lib.load()
    key = as_key(key, key.device if isinstance(key, torch.Tensor) and key.is_cuda else "cuda")
    shape = tuple(int(v) for v in (shape if isinstance(shape, Sequence) else (shape,)))
    out = torch.empty(shape, dtype=dtype, device=key.device)
    with torch.cuda.device(key.device):
        _lib.check(getattr(lib, fn_name)(key.data_ptr(), impl_id(impl), math.prod(shape), out.data_ptr(), _stream(key.device)))
    return out


def normal(key: torch.Tensor, shape, impl: str = None) -> torch.Tensor:
    """jax.random.normal(key, shape) (float32)."""
    return _fill("frl_random_normal", key, shape, torch.float32, impl)


def bits(key: torch.Tensor, shape, impl: str = None) -> torch.Tensor:
    """jax.random.bits(key, shape) (uint32)."""
    return _fill("frl_random_bits", key, shape, torch.uint32, impl)


def gumbel(key: torch.Tensor, shape, impl: str = None) -> torch.Tensor:
    """jax.random.gumbel(key, shape) (float32): the noise jax.random.categorical adds to its logits (dql.py:159-160)."""
    return _fill("frl_random_gumbel", key, shape, torch.float32, impl)


def uniform(key: torch.Tensor, shape, minval: float = 0.0, maxval: float = 1.0, impl: str = None) -> torch.Tensor:
    """jax.random.uniform(key, shape, minval=, maxval=) (float32): the t draw of ContinuousDDPM.add_noise
    (continuous_ddpm.py:110) and CNF.linear_interpolation (cnf.py:87)."""
    lib = _lib.load()
    key = as_key(key, key.device if isinstance(key, torch.Tensor) and key.is_cuda else "cuda")
    shape = tuple(int(v) for v in (shape if isinstance(shape, Sequence) else (shape,)))
    out = torch.empty(shape, dtype=torch.float32, device=key.device)
    with torch.cuda.device(key.device):
        _lib.check(lib.frl_random_uniform(key.data_ptr(), impl_id(impl), math.prod(shape), float(minval), float(maxval), out.data_ptr(), _stream(key.device)))
    return out


def randint(key: torch.Tensor, shape, minval: int, maxval: int, impl: str = None) -> torch.Tensor:
    """jax.random.randint(key, shape, minval, maxval) (int32): the t draw of DDPM.add_noise (ddpm.py:167)."""
    lib = _lib.load()
    key = as_key(key, key.device if isinstance(key, torch.Tensor) and key.is_cuda else "cuda")
    shape = tuple(int(v) for v in (shape if isinstance(shape, Sequence) else (shape,)))
    out = torch.empty(shape, dtype=torch.int32, device=key.device)
    with torch.cuda.device(key.device):
        _lib.check(lib.frl_random_randint(key.data_ptr(), impl_id(impl), math.prod(shape), int(minval), int(maxval), out.data_ptr(), _stream(key.device)))
    return out
