"""What the noise-TENSOR form of a sampler call costs before the sampler starts: drawing noise (T, B, A) with jax.random.normal's
algorithm on the GPU (frl_random_normal, one key per step as the scan does) and, for comparison, torch.randn of the same shape.
Prints one JSON line per workload; compare with the keyed (in-kernel draws) bench lines."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "flow-rl_amd")]
import torch
import flowrl_hip.random as fr

for name, (T, B, A) in {"cfg2": (5, 262144, 3), "cfg4": (20, 8192 * 64, 17)}.items():
    key = fr.PRNGKey(0)
    keys = fr.split(key, T)

    def draw():
        return [fr.normal(keys[j].contiguous(), (B, A)) for j in range(T)]

    def draw_torch():
        return torch.randn((T, B, A), device="cuda")

    out = {}
    for tag, fn in (("frl_random_normal", draw), ("torch_randn", draw_torch)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        out[tag + "_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    out.update(workload=name, elements=T * B * A, bytes=4 * T * B * A)
    print(json.dumps(out), flush=True)
