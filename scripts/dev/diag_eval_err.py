"""Development aid: where does the f32 kernels' deviation from the float64 oracle come from?  Prints, per build and
activation, the error of the FIRST network evaluation (eps_hist[0]: no chain feedback) and of the final state, next to
numpy float32's own.  Usage (GPU box): python scripts/dev/diag_eval_err.py"""
import dataclasses
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "flow-rl_amd"), str(ROOT / "oracle"), str(ROOT / "tests")]
import helpers as H  # noqa: E402


def stats(got, ref):
    d = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
    return f"max {d.max():.2e} rms {np.sqrt((d * d).mean()):.2e}"


base = H.Case("ln-cddpm-particles-wide", "cddpm", A=17, S=128, T=8, B=12 * 8, particles=8, temb="learnable_fourier",
              schedule="cosine", ln=True, seed=42)
for ln in (True, False):
    for act in ("mish", "relu", "silu"):
        c0 = dataclasses.replace(base, act=act, ln=ln)
        params, xT, cond, noise = H.make_inputs(c0)
        r64 = H.oracle_run(c0, params, xT, cond, noise, dtype=np.float64)
        r32 = H.oracle_run(c0, params, xT, cond, noise, dtype=np.float32)
        print(f"--- ln={ln} act={act}: numpy f32  eps0 {stats(r32[2][0], r64[2][0])} | x0 {stats(r32[0], r64[0])}")
        for tile in ("throughput", "latency32", "latency16"):
            c = dataclasses.replace(c0, tile=tile)
            try:
                model = H.product_model(c, params)
                x0, xh, eh = H.product_run(c, model, xT, cond, noise)
            except Exception as e:  # noqa: BLE001
                print(f"    {tile:11s} unsupported: {str(e)[:60]}")
                continue
            print(f"    {tile:11s} eps0 {stats(eh[0], r64[2][0])} | x0 {stats(x0, r64[0])}")
