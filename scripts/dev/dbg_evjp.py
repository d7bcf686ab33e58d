"""Developer diagnostic: per-leaf error of the single-evaluation VJP (two backward calls from one stash) against the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import conftest  # noqa
import helpers as H
import flowrl_oracle as orc
import flowrl_oracle_vjp as ov
import torch
import test_gpu_train as T

name = sys.argv[1] if len(sys.argv) > 1 else "evjp-ragged-nocond"
case = [c for c in T.EVAL_VJP_CASES if c.name == name][0]
params, x, cond, _ = H.make_inputs(case)
rng = np.random.default_rng(150 + case.seed)
t = T._t_for(case, rng, case.B)
g = rng.standard_normal((case.B, case.A)).astype(np.float32)
p64 = orc.cast_tree(params, np.float64)
cd = None if cond is None else np.repeat(cond, case.particles, axis=0).astype(np.float64)
out_ref, cache = ov.backbone_fwd(case.spec, p64, x.astype(np.float64), t.astype(np.float64), cd)
gref = {}
gx_ref = ov.backbone_bwd(case.spec, p64, cache, g.astype(np.float64), gref)
model = H.product_model(case, params)
parts = case.particles if case.particles > 1 else None
c = None if cond is None else torch.from_numpy(cond).cuda()
out, ctx = model(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), c, True, model.params, particles=parts, keep_for_vjp=True)
print("fwd err", H.rel_err(out.cpu().numpy(), out_ref))
ref = T._flatten(gref)
for call, wg in enumerate((True, False, True)):
    grads, gx = model.eval_vjp(ctx, torch.from_numpy(g).cuda(), want_gx=wg)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in T._flatten(grads).items()}
    for k in sorted(ref):
        r = np.asarray(ref[k], np.float64)
        d = got[k] - r
        rows = ""
        if d.ndim == 2:
            bad = np.where(np.abs(d).max(axis=1) > 1e-4 * np.abs(r).max())[0]
            rows = f" bad rows {bad[:8]}..{bad[-1] if len(bad) else ''} ({len(bad)})"
        print(call, k, f"{np.abs(d).max() / max(np.abs(r).max(), 1e-30):.2e}", rows)
    if gx is not None:
        print(call, "gx", H.rel_err(gx.cpu().numpy(), gx_ref))
