"""Developer diagnostic: per-leaf error of the bf16 chain VJP against the bf16-operand emulation on the kernel's own states."""
import sys, os, dataclasses
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import conftest  # noqa
import helpers as H
import torch
import test_gpu_vjp_bf16 as T
from test_gpu_vjp import product_vjp

name = sys.argv[1] if len(sys.argv) > 1 else "relu_fourier_a16"
stash = len(sys.argv) > 2 and sys.argv[2] == "stash"
case = [c for c in T.CASES if c.name == name][0]
for kv in os.environ.get("CASE_OVERRIDE", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        cur = getattr(case, k)
        case = dataclasses.replace(case, **{k: (type(cur)(v) if not isinstance(cur, bool) else v == "1") if cur is not None else v})
print(case)
params, xT, cond, noise = H.make_inputs(case)
g = np.random.default_rng(77 + case.seed).standard_normal((case.B, case.A)).astype(np.float32)
model = H.product_model(case, params)
grads, gxT, xh = product_vjp(case, model, xT, cond, noise, g, compute_dtype="bf16", stash=stash)
torch.cuda.synchronize()
got = {k: v.cpu().numpy() for k, v in T._flatten(grads).items()}
got["dL/dxT"] = gxT.cpu().numpy()
emu = T._oracle(case, params, xT, cond, noise, g, xt_hist=xh.cpu().numpy())
worst = max(T._rel_l2(got[k], emu[k]) for k in emu)
print("WORST", worst)
for k in sorted(emu) if worst > 1e-2 else []:
    print(k, f"{T._rel_l2(got[k], emu[k]):.2e}", f"|ref| {np.linalg.norm(emu[k]):.3e}")
