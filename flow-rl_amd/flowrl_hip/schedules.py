"""Noise-schedule tables, built on the host in float32 with the reference's op order.

References (paths relative to the flow-rl checkout):
  * discrete DDPM      flowrl/flow/ddpm.py:23-42 (schedules), :132-150 (tables)
  * continuous VP      flowrl/flow/continuous_ddpm.py:19-40 (schedules), :84-87, :134-138 (per-call tables)
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

F32 = np.float32


def linear_beta_schedule(T: int = 1000, beta_min: float = 1e-4, beta_max: float = 2e-2) -> np.ndarray:
    return np.linspace(beta_min, beta_max, T, dtype=F32)


def cosine_beta_schedule(T: int = 1000, s: float = 0.008) -> np.ndarray:
    steps = T + 1
    t = np.linspace(0, T, steps, dtype=F32) / F32(T)
    a = np.cos((t + F32(s)) / F32(1 + s) * F32(np.pi) * F32(0.5)) ** 2
    a = a / a[0]
    betas = 1 - (a[1:] / a[:-1])
    return np.clip(betas, 0, 0.999).astype(F32)


def vp_beta_schedule(T: int = 1000) -> np.ndarray:
    t = np.arange(1, T + 1)
    b_max, b_min = 10.0, 0.1
    alpha = np.exp((-b_min / T - 0.5 * (b_max - b_min) * (2 * t - 1).astype(F32) / F32(T ** 2)).astype(F32))
    return (1 - alpha).astype(F32)


def ddpm_tables(steps: int, noise_schedule: str, noise_schedule_params: Optional[Dict] = None,
                approx_postvar: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(betas, alphas, alpha_hats, postvars), each (steps+1,) float32 — DDPM.create, ddpm.py:132-150."""
    kw = dict(noise_schedule_params or {})
    kw["T"] = steps
    if noise_schedule == "linear":
        betas = linear_beta_schedule(**kw)
    elif noise_schedule == "cosine":
        betas = cosine_beta_schedule(**kw)
    elif noise_schedule == "vp":
        betas = vp_beta_schedule(**kw)
    else:
        raise NotImplementedError(f"Unsupported noise schedule: {noise_schedule}")
    betas = np.concatenate([np.zeros((1,), F32), betas]).astype(F32)
    alphas = (1 - betas).astype(F32)
    alpha_hats = np.cumprod(alphas, dtype=F32)
    if approx_postvar:
        postvars = betas.copy()
    else:
        postvars = betas[1:] * (1 - alpha_hats[:-1]) / (1 - alpha_hats[1:])
        postvars = np.concatenate([np.zeros((1,), F32), postvars]).astype(F32)
    return betas, alphas, alpha_hats, postvars


def linear_noise_schedule(t: np.ndarray, beta0: float = 0.1, beta1: float = 20.0):
    log_alpha = -F32((beta1 - beta0) / 4.0) * (t ** 2) - F32(beta0 / 2.0) * t
    alpha = np.exp(log_alpha).astype(F32)
    return alpha, np.sqrt(1.0 - alpha ** 2).astype(F32)


def cosine_noise_schedule(t: np.ndarray, s: float = 0.008):
    alpha = np.cos(F32(np.pi / 2.0) * (np.clip(t, 0.0, 0.9946).astype(F32) + F32(s)) / F32(1 + s)) / np.cos(
        F32(np.pi / 2.0) * F32(s) / F32(1 + s))
    alpha = alpha.astype(F32)
    return alpha, np.sqrt(1.0 - alpha ** 2).astype(F32)


def quad_t_schedule(T: int, n: float = 1.0, tmin: float = 1e-3, tmax: float = 1.0) -> np.ndarray:
    return (F32(tmax - tmin) * (np.linspace(0, 1, T + 1, dtype=F32) ** F32(n)) + F32(tmin)).astype(F32)


def continuous_tables(steps: int, noise_schedule_func: Callable, t_diffusion, t_schedule_n: float):
    """(ts, alpha_hats, alphas, betas) — ContinuousDDPM.sample, continuous_ddpm.py:134-138."""
    ts = quad_t_schedule(steps, n=t_schedule_n, tmin=t_diffusion[0], tmax=t_diffusion[1])
    alpha_hats = (noise_schedule_func(ts)[0] ** 2).astype(F32)
    alphas = np.concatenate([np.ones((1,), F32), alpha_hats[1:] / alpha_hats[:-1]]).astype(F32)
    betas = (1 - alphas).astype(F32)
    return ts, alpha_hats, alphas, betas


def continuous_midpoints(ts: np.ndarray, noise_schedule_func: Callable):
    """(midpoints, mid_alphas, mid_sigmas), each (steps,) — the dpm2 tables of continuous_ddpm.py:145-147."""
    mid = ((ts[:-1] + ts[1:]) / F32(2)).astype(F32)
    al, sg = noise_schedule_func(mid)
    return mid, np.asarray(al, F32), np.asarray(sg, F32)
