"""Wall time of one training-size chain VJP (training forward + backward from the stash), HIP events over 50 calls.
usage: python scripts/dev/time_train_step.py [dql|dacer|bc] [batch]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "tests"), str(ROOT / "oracle"), str(ROOT / "flow-rl_amd")]
import helpers as H  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "dql"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "dacer":
    case = H.Case("dacer", "cddpm", A=17, S=376, T=20, B=B, temb="learnable_fourier", schedule="cosine", cond_embed=True, seed=1)
else:
    case = H.Case("dql", "ddpm", A=3, S=11, T=5, B=B, seed=1)
params, xT, cond, noise = H.make_inputs(case)
model = H.product_model(case, params)
x, c, nz = torch.from_numpy(xT).cuda(), torch.from_numpy(cond).cuda(), torch.from_numpy(noise).cuda()
g = torch.randn(case.B, case.A, device="cuda")
from flowrl_hip.flow.ddpm import ddpm_loss  # noqa: E402
import flowrl_hip.random as fr  # noqa: E402
key = fr.PRNGKey(0)


def step():
    if which == "bc":      # the eps-MSE loss of jit_update_ddpm with its gradients: one evaluation per sample + its VJP
        return ddpm_loss(key, model, x, c, want_grads=True)
    _, x0, hist = model.sample(None, x, c, True, case.solver, noise=nz, params=model.params, keep_for_vjp=True)
    return model.sample_vjp(hist[0], g, c, case.solver, params=model.params, ctx=hist[2])


for _ in range(40):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
e0.record()
t0 = time.perf_counter()
for _ in range(50):
    step()
t_host = (time.perf_counter() - t0) / 50 * 1e3      # time to SUBMIT a step (no synchronisation in the loop)
e1.record()
torch.cuda.synchronize()
print(f"{which} B={B} T={case.T}: {e0.elapsed_time(e1) / 50:.3f} ms per forward+backward on the stream; host submits a step in {t_host:.3f} ms")
