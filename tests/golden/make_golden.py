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

# Round 2: one fixture per remaining BASELINE geometry.  cfg3's 512x4 parameters are 3.3 MB of random floats, so that
# file stores the generator seed and a SHA-256 of the parameter bytes instead of the values (a changed generator fails
# loudly in the test); everything else stays verbatim.
BIG_CASES = [
    H.Case("golden_cfg3_fql_512x4", "cnf", A=8, S=39, T=10, B=24, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None, clip=False, seed=105),
    H.Case("golden_cfg4_dacer_embed", "cddpm", A=17, S=376, T=20, B=4 * 8, particles=8, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, seed=106),
]
# residual predictors: 1 .. 4 MB of block kernels each -> seed + SHA-256, as cfg3
RES_CASES = [
    H.Case("golden_res_dac_resnet", "ddpm", A=3, S=11, T=5, B=24, hidden=(256, 256), backbone="residual", ln=True, seed=109),
    H.Case("golden_simba_cnf", "cnf", A=6, S=17, T=10, B=24, hidden=(256,), backbone="simba", act="relu", seed=110),
]
# the evaluation under jax.grad (frl_net_eval_train / frl_net_eval_bwd): per-sample time, every gradient leaf
EVAL_VJP_CASE = H.Case("golden_eval_vjp", "cddpm", A=6, S=17, T=20, B=40, temb="learnable_fourier", schedule="cosine", act="silu", seed=111)
# cfg3's FQL actor_bc under the flow-matching loss (layer-by-layer VJP, csrc/net_vjp_layered.hip): 817 k parameters -> the
# parameters by seed + SHA-256, every gradient leaf by 4096 seeded entries + its L2 norm
EVAL_VJP_512 = H.Case("golden_eval_vjp_fql512x4", "cnf", A=8, S=39, T=10, B=64, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None,
                      clip=False, ln=True, ln_post=True, seed=112)
VJP_CASE = H.Case("golden_cfg5_vjp", "ddpm", A=3, S=11, T=5, B=40, seed=107)
CRITIC = dict(name="golden_critic_select", B=12, N=10, S=11, A=3, hidden=(256, 256, 256), act="mish", E=2, seed=108)


def params_digest(params):
    import hashlib
    h = hashlib.sha256()
    for k, v in sorted(orc.flatten_tree(params).items()):
        h.update(k.encode())
        h.update(np.ascontiguousarray(v, dtype=np.float32).tobytes())
    return h.hexdigest()


CASES = [
    H.Case("golden_cfg1_toy2d", "ddpm", A=2, S=1, T=20, B=48, schedule="linear", clip=False, x_min=-4.0, x_max=4.0, seed=101),
    H.Case("golden_cfg2_hopper", "ddpm", A=3, S=11, T=5, B=48, seed=102),
    H.Case("golden_cddpm_cosine", "cddpm", A=6, S=17, T=20, B=48, temb="learnable_fourier", schedule="cosine", seed=103),
    H.Case("golden_cnf_raw_t", "cnf", A=8, S=20, T=10, B=48, temb=None, clip=False, seed=104),
]


def leaf_sample_indices(name: str, size: int, n: int = 4096):
    import zlib
    return np.random.default_rng(zlib.crc32(name.encode())).integers(0, size, size=min(n, size))


def make_eval_vjp_512():
    import flowrl_oracle_vjp as ov
    sys.path.insert(0, str(ROOT / "tests"))
    from test_gpu_vjp import _flatten
    c = EVAL_VJP_512
    params, x, cond, _ = H.make_inputs(c)
    rng = np.random.default_rng(150 + c.seed)
    t = rng.uniform(0.0, 1.0, size=(c.B, 1)).astype(np.float32)
    g = rng.standard_normal((c.B, c.A)).astype(np.float32)
    p64 = orc.cast_tree(params, np.float64)
    ev_out, cache = ov.backbone_fwd(c.spec, p64, x.astype(np.float64), t.astype(np.float64), cond.astype(np.float64))
    gref: dict = {}
    gx = ov.backbone_bwd(c.spec, p64, cache, g.astype(np.float64), gref)
    keep = dict(x=x, cond=cond, t=t, g=g, out=ev_out, g_x=gx, params_sha256=np.array(params_digest(params)), params_seed=np.array(c.seed))
    for k, v in _flatten(gref).items():
        v = np.asarray(v, np.float64).reshape(-1)
        keep["gs:" + k] = v[leaf_sample_indices(k, v.size)]
        keep["gn:" + k] = np.array([np.linalg.norm(v), np.abs(v).max()])
    np.savez_compressed(Path(__file__).resolve().parent / f"{c.name}.npz", **keep)
    print(c.name, len(_flatten(gref)), "leaves")


def main():
    out = Path(__file__).resolve().parent
    if len(sys.argv) > 1 and sys.argv[1] == "eval512":
        make_eval_vjp_512()
        return
    make_eval_vjp_512()
    for c in CASES:
        params, xT, cond, noise = H.make_inputs(c)
        if c.name == "golden_cfg1_toy2d":
            cond = np.zeros_like(cond)            # toy2d feeds obs = zeros (examples/toy2d/main_toy2d.py:57-58)
        x0, xh, eh = H.oracle_run(c, params, xT, cond, noise, dtype=np.float64)
        flat = {"p:" + k: v for k, v in orc.flatten_tree(params).items()}
        np.savez_compressed(out / f"{c.name}.npz", xT=xT, cond=cond, noise=noise, x0=x0, xt_hist=xh, eps_hist=eh, **flat)
        print(c.name, x0.shape, float(np.abs(x0).max()))
    for c in BIG_CASES + RES_CASES:
        params, xT, cond, noise = H.make_inputs(c)
        x0, xh, eh = H.oracle_run(c, params, xT, cond, noise, dtype=np.float64)
        keep = dict(xT=xT, cond=cond, noise=noise, x0=x0, xt_hist=xh, eps_hist=eh)
        if c.hidden[0] == 512 or c.backbone != "mlp":
            keep.update(params_sha256=np.array(params_digest(params)), params_seed=np.array(c.seed))
        else:
            keep.update({"p:" + k: v for k, v in orc.flatten_tree(params).items()})
        np.savez_compressed(out / f"{c.name}.npz", **keep)
        print(c.name, x0.shape, float(np.abs(x0).max()))
    # chain VJP (cfg5 shapes): cotangent in, every gradient leaf + dL/dxT out (float64 manual-backprop oracle)
    sys.path.insert(0, str(ROOT / "tests"))
    from test_gpu_vjp import oracle_vjp, _flatten
    c = VJP_CASE
    params, xT, cond, noise = H.make_inputs(c)
    g = np.random.default_rng(77 + c.seed).standard_normal((c.B, c.A)).astype(np.float32)
    x0, gref, gx = oracle_vjp(c, params, xT, cond, noise, g)
    np.savez_compressed(out / f"{c.name}.npz", xT=xT, cond=cond, noise=noise, g=g, x0=x0, g_xT=gx,
                        **{"p:" + k: v for k, v in orc.flatten_tree(params).items()},
                        **{"g:" + k: v for k, v in _flatten(gref).items()})
    print(c.name, len(_flatten(gref)), "leaves")
    # single-evaluation VJP: per-sample t, cotangent in, every leaf + dL/dx out
    import flowrl_oracle_vjp as ov
    c = EVAL_VJP_CASE
    params, x, cond, _ = H.make_inputs(c)
    rng = np.random.default_rng(150 + c.seed)
    t = rng.uniform(1e-3, 0.9946, size=(c.B, 1)).astype(np.float32)
    g = rng.standard_normal((c.B, c.A)).astype(np.float32)
    p64 = orc.cast_tree(params, np.float64)
    ev_out, cache = ov.backbone_fwd(c.spec, p64, x.astype(np.float64), t.astype(np.float64), cond.astype(np.float64))
    gref: dict = {}
    gx = ov.backbone_bwd(c.spec, p64, cache, g.astype(np.float64), gref)
    np.savez_compressed(Path(__file__).resolve().parent / f"{c.name}.npz", x=x, cond=cond, t=t, g=g, out=ev_out, g_x=gx,
                        **{"p:" + k: v for k, v in orc.flatten_tree(params).items()},
                        **{"g:" + k: v for k, v in _flatten(gref).items()})
    print(c.name, len(_flatten(gref)), "leaves")
    # critic scoring + best-of-N (dql.py:151-161): q = min over the ensemble, argmax index, chosen action
    k = CRITIC
    rng = np.random.default_rng(k["seed"])
    cp = orc.make_critic_params(k["S"], k["A"], k["hidden"], k["E"], k["seed"])
    obs = rng.standard_normal((k["B"], k["S"])).astype(np.float32)
    actions = np.clip(rng.standard_normal((k["B"], k["N"], k["A"])), -1, 1).astype(np.float32)
    best, idx, q = orc.select_best(orc.cast_tree(cp, np.float64), obs.astype(np.float64), actions.astype(np.float64), len(k["hidden"]), k["act"])
    np.savez_compressed(out / f"{k['name']}.npz", obs=obs, actions=actions, q=q, idx=idx, best=best,
                        **{"p:" + n: v for n, v in orc.flatten_tree(cp).items()})
    print(k["name"], q.shape)


if __name__ == "__main__":
    main()
