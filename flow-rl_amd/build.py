"""Build libflowrl_hip.so (gfx950) in-tree with hipcc.

    python flow-rl_amd/build.py [--force] [--jobs N]

Cross-compiles without a GPU.  Output: flow-rl_amd/flowrl_hip/libflowrl_hip.so (git-ignored, but it
travels to the GPU box with the tree).  Objects are cached under flow-rl_amd/csrc/_obj/ and rebuilt
when a source or header is newer.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
CSRC = HERE / "csrc"
OBJ = CSRC / "_obj"
OUT = HERE / "flowrl_hip" / "libflowrl_hip.so"
ARCH = "gfx950"
# (source, object stem, extra defines); the register-streamed sampler is one translation unit per activation
SOURCES = [("capi.hip", "capi", []), ("prepare.hip", "prepare", []),
           ("cond_embed.hip", "cond_embed", []), ("critic_select.hip", "critic_select", []), ("time_bwd.hip", "time_bwd", []), ("rng.hip", "rng", []), ("net_eval.hip", "net_eval", []), ("net_vjp_layered.hip", "net_vjp_layered", []),
           ("sampler_bf16.hip", "sampler_bf16", ["-fno-slp-vectorize"]),
           ] + [("sampler_bwd_bf16.hip", f"sampler_bwd_bf16_p{p}", ["-fno-slp-vectorize", f"-DFRL_BWD16_PHASE={p}"]) for p in range(3)] + \
          [("sampler_bwd.hip", f"sampler_bwd_p{p}", [f"-DFRL_BWD_PHASE={p}"]) for p in range(3)] + [("sampler_bwd_lat.hip", "sampler_bwd_lat", [])] + \
          [("sampler_rs.hip", f"sampler_rs_act{a}", [f"-DFRL_RS_ACT={a}"]) for a in range(7)] + \
          [("sampler_res.hip", f"sampler_res_act{a}", [f"-DFRL_RES_ACT={a}"]) for a in range(6)] + \
          [("sampler_lat.hip", f"sampler_lat_act{a}", [f"-DFRL_RS_ACT={a}"]) for a in range(7)]   # 6 = identity: the post-activation-LayerNorm units
HEADERS = [CSRC / "frl_internal.h", CSRC / "frl_device.h", CSRC / "frl_bf16.h", CSRC / "frl_rng.h", CSRC / "frl_actgrad.h", ROOT / "include" / "flowrl_hip.h"]
FLAGS = ["-O3", "-fPIC", f"--offload-arch={ARCH}", "-std=c++17", "-Wno-unused-value",
         f"-I{ROOT / 'include'}", f"-I{CSRC}"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain required to build libflowrl_hip.so)")
    return exe


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    cc = hipcc()
    todo = []
    objs = []
    for src, stem, defs in SOURCES:
        s = CSRC / src
        o = OBJ / (stem + ".o")
        objs.append(o)
        if force or _stale(o, [s, *HEADERS, Path(__file__)]):
            todo.append((s, o, defs))

    def compile_one(so):
        s, o, defs = so
        cmd = [cc, *FLAGS, *defs, "-c", str(s), "-o", str(o)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True)
        if verbose:
            print(f"[build] {o.name}: {time.perf_counter() - t0:.0f} s", flush=True)

    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(compile_one, todo))
    if force or todo or _stale(OUT, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(OUT)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # the torch-free driver (bench/native_bench.cpp): host code against include/flowrl_hip.h + RCCL, finds the library
    # next to the package through its rpath
    nb_src, nb_out = ROOT / "bench" / "native_bench.cpp", ROOT / "bench" / "native_bench"
    if nb_src.exists() and (force or _stale(nb_out, [nb_src, ROOT / "include" / "flowrl_hip.h", OUT])):
        cmd = [cc, "-O2", "-std=c++17", str(nb_src), f"-I{ROOT / 'include'}", f"-L{OUT.parent}", "-lflowrl_hip", "-lrccl",
               "-Wl,-rpath,$ORIGIN/../flow-rl_amd/flowrl_hip", "-o", str(nb_out)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    args = ap.parse_args()
    print(build(force=args.force, jobs=args.jobs))
    sys.exit(0)
