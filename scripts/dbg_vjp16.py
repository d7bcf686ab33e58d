"""Development aid: bf16 VJP vs the emulation, leaf by leaf (run on the GPU box)."""
import sys, dataclasses
sys.path[:0] = ['tests', 'oracle', 'flow-rl_amd']
import numpy as np, torch
import helpers as H
import test_gpu_vjp_bf16 as t

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_dql_b256"
case = [c for c in t.CASES if c.name == name][0]
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    v = {"B": int, "T": int, "act": str, "L": lambda x: (256,) * int(x), "clip": lambda x: x == "1", "schedule": str}[k](v)
    case = dataclasses.replace(case, **{("hidden" if k == "L" else k): v})
print(case)
params, xT, cond, noise = H.make_inputs(case)
g = np.random.default_rng(77 + case.seed).standard_normal((case.B, case.A)).astype(np.float32)
model = H.product_model(case, params)
x = torch.from_numpy(xT).cuda()
c = None if cond is None else torch.from_numpy(cond).cuda()
parts = case.particles if case.particles > 1 else None
_, x0, (xh, _) = model.sample(None, x, c, True, case.solver, noise=torch.from_numpy(noise).cuda(), history=True,
                              particles=parts, params=model.params, compute_dtype="bf16")
from flowrl_hip import _sampler as _S
_w = _S._workspace(256 << 20, x.device); _w.fill_(0xFF)          # sentinel: every never-written bf16 reads as NaN (0xFFFF)
torch.cuda.synchronize()
grads, gxT = model.sample_vjp(xh, torch.from_numpy(g).cuda(), c, case.solver, particles=parts, compute_dtype="bf16")
torch.cuda.synchronize()
got = {k: v.cpu().numpy() for k, v in t._flatten(grads).items()}
got["dL/dxT"] = gxT.cpu().numpy()
emu = t._oracle(case, params, xT, cond, noise, g, xt_hist=xh.cpu().numpy())
for k in sorted(got):
    a, r = got[k], np.asarray(emu[k])
    nn = int(np.isnan(a).sum())
    fin = np.isfinite(a)
    err = np.linalg.norm((a - r)[fin]) / max(np.linalg.norm(r[fin]), 1e-30) if fin.any() else float('nan')
    print(f"{k:40s} shape {a.shape} nan {nn}/{a.size} rel-l2(finite) {err:.3e} |ref| {np.abs(r).max():.3e} |got| {np.abs(a[fin]).max() if fin.any() else 0:.3e}")
    if k == "dL/dxT":
        for w0 in range(0, case.B, 32):
            aa, rr = a[w0:w0 + 32], r[w0:w0 + 32]
            f2 = np.isfinite(aa)
            print(f"   chains {w0}..{w0+31}: nan {int((~f2).sum())} rel-l2 {np.linalg.norm((aa-rr)[f2])/max(np.linalg.norm(rr[f2]),1e-30):.2e}")

# ---- stash forensics: where do non-finite / absurd values first appear? ----
import ctypes as C
from flowrl_hip import _lib, _sampler
ws = list(_sampler._ws_cache.values())[0]
L = len(case.hidden); T = case.T
desc = _sampler.build_desc(model.network, kind=_lib.FRL_KIND_DDPM, solver=_lib.FRL_SOLVER_DDPM, steps=T, n_states=case.B // case.particles,
                           particles=case.particles, x_dim=case.A, cond_dim=case.S, clip_sampler=case.clip, x_min=case.x_min,
                           x_max=case.x_max, want_history=True, compute_dtype="bf16")
fwd_floats = _lib.load().frl_workspace_bytes(C.byref(desc)) // 4
bf_ns = 1 + 4 * (L - 1) + 1
n_wt = 8 * ((case.B + 255) // 256); n_ts = n_wt * T
off = fwd_floats + bf_ns * 8192
wsf = ws.view(torch.float32)
for nm in ("hd", "ad", "gd"):
    arr = wsf[off:off + n_ts * L * 4096].view(torch.bfloat16).float().reshape(n_wt, T, L, 16, 64, 8).cpu().numpy()
    off += n_ts * L * 4096
    raw = wsf[off - n_ts * L * 4096:off].view(torch.int16).reshape(n_wt, T, L, 16, 64, 8).cpu().numpy()
    print(nm, "sentinel (never written) elements:", int((raw == -1).sum()))
    if nm == "hd":
        hd_arr = arr
    bad = ~np.isfinite(arr) | (np.abs(arr) > 1e6)
    print(nm, "bad elements:", int(bad.sum()), "of", arr.size)
    if bad.any():
        idx = np.argwhere(bad.any(axis=-1))
        seen = {}
        for wt, j, l, pc, ln in idx:
            seen.setdefault((int(wt), int(j), int(l), int(pc)), []).append(int(ln))
        for k in sorted(seen, key=lambda k: (k[0], -k[1], k[2]))[:24]:
            print("   wt %d step j=%d layer %d piece %d lanes %s" % (*k, seen[k]))
            if nm == "ad":
                for ln in seen[k][:3]:
                    print("        lane", ln, "d =", arr[k[0], k[1], k[2], k[3], ln].tolist(), " h =", hd_arr[k[0], k[1], k[2], k[3], ln].tolist())
