"""Development aid: per-leaf errors of one VJP case (stash path) against the float64 oracle."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "flow-rl_amd"), str(ROOT / "oracle"), str(ROOT / "tests")]
import helpers as H  # noqa: E402
import test_gpu_vjp as T  # noqa: E402
import torch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dacer_humanoid"
case = {c.name: c for c in T.WIDE_CASES + T.CASES}[name]
params, xT, cond, noise = H.make_inputs(case)
g = np.random.default_rng(77 + case.seed).standard_normal((case.B, case.A)).astype(np.float32)
x0_ref, gref, gx_ref = T.oracle_vjp(case, params, xT, cond, noise, g)
model = H.product_model(case, params)
grads, gxT, _ = T.product_vjp(case, model, xT, cond, noise, g, stash=True)
torch.cuda.synchronize()
got = {k: v.cpu().numpy() for k, v in T._flatten(grads).items()}
ref = T._flatten(gref)
got["dL/dxT"], ref["dL/dxT"] = gxT.cpu().numpy(), gx_ref
for k in sorted(ref):
    r = np.asarray(ref[k], np.float64)
    d = got[k].astype(np.float64) - r
    print(f"{k:45s} max {np.abs(d).max() / max(np.abs(r).max(), 1e-30):.2e}  |ref| {np.abs(r).max():.3e}  |got| {np.abs(got[k]).max():.3e}")
