"""Developer diagnostic: host-side cost of one sample() call (Python marshalling + ctypes) against the kernel time, small batch."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "flow-rl_amd"))
import torch
import bench

w = dict(bench.WORKLOADS["cfg2"]); w["B"] = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w.update(dtype="f32", tile_policy="auto", rng="none", rng_shard=None, cache_prepare=False, recompute=False)
dev = torch.device("cuda", 0)
model = bench.build_model(w, dev)
xT, cond, noise = bench.make_inputs(w, dev, 0)
for _ in range(20):
    bench.sample(model, w, xT, cond, noise)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    bench.sample(model, w, xT, cond, noise)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={w['B']}: host issue {1e6 * (t1 - t0) / n:.1f} us/call, with drain {1e6 * (t2 - t0) / n:.1f} us/call")
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    bench.sample(model, w, xT, cond, noise)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
