"""Timing of the chain VJP at a given batch (development aid): forward-with-history, backward, per-kernel via rocprofv3."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "flow-rl_amd"), str(ROOT)]
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
w = dict(bench.WORKLOADS["cfg2"]); w["B"] = B
dev = torch.device("cuda", 0)
model = bench.build_model(w, dev)
xT, cond, noise = bench.make_inputs(w, dev, 1)
g = torch.randn_like(xT)
def fwd():
    return model.sample(None, xT, cond, True, "ddpm", noise=noise, history=True, params=model.params)
def ev():
    return torch.cuda.Event(enable_timing=True)
for it in range(3):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    _, x0, (xh, eh) = fwd()
    e1.record()
    grads, gxT = model.sample_vjp(xh, g, cond, "ddpm")
    e2.record()
    torch.cuda.synchronize()
    print(f"B={B} fwd+hist {e0.elapsed_time(e1):.3f} ms  bwd {e1.elapsed_time(e2):.3f} ms", flush=True)
print("finite", all(torch.isfinite(v).all().item() for v in [gxT, grads["noise_predictor"]["Dense_1"]["kernel"]]))
