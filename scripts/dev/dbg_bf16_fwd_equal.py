"""Debug aid: where does the bf16 training forward differ from the plain bf16 sampler (they must agree bit for bit)?"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "flow-rl_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import helpers as H
from test_gpu_vjp import product_vjp
from test_gpu_vjp_bf16 import CASES

for ci in (0, 3):
    case = CASES[ci]
    params, xT, cond, noise = H.make_inputs(case)
    g = np.random.default_rng(5).standard_normal((case.B, case.A)).astype(np.float32)
    model = H.product_model(case, params)
    outs = []
    for stash in (False, False, True, True):
        o = {}
        product_vjp(case, model, xT, cond, noise, g, compute_dtype="bf16", stash=stash, out=o)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items() if k in ("x0", "xt_hist", "eps_hist")})
    for name, (i, j) in {"plain vs plain": (0, 1), "train vs train": (2, 3), "plain vs train": (0, 2)}.items():
        for k in ("eps_hist", "xt_hist", "x0"):
            a, b = outs[i][k], outs[j][k]
            d = (a != b)
            print(case.name, name, k, "differing", int(d.sum()), "of", d.numel(), "max abs", float((a - b).abs().max()),
                  "first idx", d.nonzero()[:3].tolist() if d.any() else None)
