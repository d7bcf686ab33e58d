"""Diagnostic (GPU box): error statistics of the HIP sampler vs the float64 oracle, next to the
float32 oracle's own noise, per step."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "flow-rl_amd"), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import helpers as H

cases = [H.Case("nocond", "ddpm", A=6, S=0, T=5, B=130, seed=6)] + \
        [H.Case(f"cfg2-s{s}", "ddpm", A=3, S=11, T=5, B=1024, seed=s) for s in range(4)] + \
        [H.Case("toy", "ddpm", A=2, S=1, T=20, B=1024, schedule="linear", clip=False, seed=1)]
for c in cases:
    p, x, cd, nz = H.make_inputs(c)
    r64 = H.oracle_run(c, p, x, cd, nz, np.float64)
    r32 = H.oracle_run(c, p, x, cd, nz, np.float32)
    m = H.product_model(c, p)
    g = H.product_run(c, m, x, cd, nz)
    def st(a, b):
        d = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1)
        return f"max {d.max():.2e} rms {np.sqrt((d**2).mean()):.2e} p99.9 {np.quantile(d, 0.999):.2e}"
    print(c.name)
    for nm, r in (("hip", g), ("np32", r32)):
        print(f"  {nm:5s} x0[{st(r[0], r64[0])}] eps[{st(r[2], r64[2])}] xt[{st(r[1], r64[1])}]")
        print("        eps max per step", " ".join(f"{np.abs(r[2][j]-r64[2][j]).max():.1e}" for j in range(min(c.T, 8))))
