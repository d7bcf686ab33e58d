"""Generate the committed golden fixtures (tests/golden/*.npz) from the float64 oracle.

    python tests/golden/make_golden.py

The reference cannot be imported here (jax/flax absent) and ships no fixtures of its own, so these
vectors pin the ORACLE against accidental change and give the GPU tests reference outputs that do not
depend on re-running it: inputs (params, xT, cond, noise), float64 outputs (x0, xt_hist, eps_hist).
Parameters are stored verbatim (float32, Flax tree paths prefixed "p:"); the hidden width stays 256
(the kernel's geometry) and the batch is 48 chains to keep the files small.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "oracle"), str(ROOT / "tests")]
import flowrl_oracle as orc  # noqa: E402
import helpers as H  # noqa: E402

CASES = [
    H.Case("golden_cfg1_toy2d", "ddpm", A=2, S=1, T=20, B=48, schedule="linear", clip=False, x_min=-4.0, x_max=4.0, seed=101),
    H.Case("golden_cfg2_hopper", "ddpm", A=3, S=11, T=5, B=48, seed=102),
    H.Case("golden_cddpm_cosine", "cddpm", A=6, S=17, T=20, B=48, temb="learnable_fourier", schedule="cosine", seed=103),
    H.Case("golden_cnf_raw_t", "cnf", A=8, S=20, T=10, B=48, temb=None, clip=False, seed=104),
]


def main():
    out = Path(__file__).resolve().parent
    for c in CASES:
        params, xT, cond, noise = H.make_inputs(c)
        if c.name == "golden_cfg1_toy2d":
            cond = np.zeros_like(cond)            # toy2d feeds obs = zeros (examples/toy2d/main_toy2d.py:57-58)
        x0, xh, eh = H.oracle_run(c, params, xT, cond, noise, dtype=np.float64)
        flat = {"p:" + k: v for k, v in orc.flatten_tree(params).items()}
        np.savez_compressed(out / f"{c.name}.npz", xT=xT, cond=cond, noise=noise, x0=x0, xt_hist=xh, eps_hist=eh, **flat)
        print(c.name, x0.shape, float(np.abs(x0).max()))


if __name__ == "__main__":
    main()
