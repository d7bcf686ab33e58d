"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the jax.random pieces the samplers use.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this file; the product path never does.

The reference draws its per-step noise with jax.random (jax==0.5.3, /root/reference/setup.py:18-19), which is NOT
under /root/reference and not installed here; this file restates the published algorithm:

  threefry2x32          Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), 20 rounds, the
                        Threefry-2x32 rotation table; jax/_src/prng.py::_threefry2x32_lowering applies it to uint32 pairs.
  PRNGKey(seed)         jax/_src/prng.py::threefry_seed: key = [seed >> 32, seed & 0xffffffff].
  split / random_bits   two key-derivation layouts:
      "partitionable"   (jax_threefry_partitionable=True, the default from jax 0.5.0 on, i.e. the reference's):
                        element with flat row-major index i of the requested shape gets (o0, o1) =
                        threefry(key, (i >> 32, i & 0xffffffff)); split(key, n)[i] = (o0, o1); 32 random bits = o0 ^ o1.
      "legacy"          (the default before 0.5.0): counts = iota(n) zero-padded to even length and cut into halves
                        (x0, x1); out = concat(threefry(key, (x0, x1)))[:n]; split(key, n) = 2n such words reshaped (n, 2).
  uniform(lo, hi)       mantissa trick: f = bitcast((bits >> 9) | 0x3f800000) - 1;  u = max(lo, f * (hi - lo) + lo)  (f32).
  normal                sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1)),  f32;  erf_inv = XLA's single-precision
                        polynomial (M. Giles, "Approximating the erfinv function"): w = -log1p(-x*x), two branches at w = 5.

Call sites that fix the protocol: flowrl/flow/ddpm.py:188,204-206; continuous_ddpm.py:161,174,190; cnf.py:126.

PINNING: Random123's published known-answer vectors for threefry2x32-20 (the same three jax's own test-suite uses), and
the values the JAX documentation printed for the legacy layout: split(PRNGKey(0)) and normal(PRNGKey(0), (1,)).  The
partitionable layout has no published vector we can cite from memory: PARITY UNPINNED beyond the block function.
"""
import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M32 = np.uint64(0xFFFFFFFF)


def threefry2x32(k0, k1, c0, c1):
    """Elementwise Threefry-2x32-20 on uint32 arrays (broadcast); returns (o0, o1) uint32."""
    k0 = np.asarray(k0, np.uint64); k1 = np.asarray(k1, np.uint64)
    x0 = np.asarray(c0, np.uint64); x1 = np.asarray(c1, np.uint64)
    ks = (k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & _M32)
    x0 = (x0 + ks[0]) & _M32
    x1 = (x1 + ks[1]) & _M32
    for blk in range(5):
        for r in _ROT[blk & 1]:
            x0 = (x0 + x1) & _M32
            x1 = ((x1 << np.uint64(r)) | (x1 >> np.uint64(32 - r))) & _M32
            x1 = x1 ^ x0
        x0 = (x0 + ks[(blk + 1) % 3]) & _M32
        x1 = (x1 + ks[(blk + 2) % 3] + np.uint64(blk + 1)) & _M32
    return x0.astype(np.uint32), x1.astype(np.uint32)


def prng_key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([seed >> 32, seed & 0xFFFFFFFF], np.uint32)


def _legacy_words(key, n):
    cnt = np.arange(n + (n & 1), dtype=np.uint64)
    cnt[n:] = 0
    half = cnt.size // 2
    o0, o1 = threefry2x32(key[0], key[1], cnt[:half], cnt[half:])
    return np.concatenate([o0, o1])[:n]


def split(key, num=2, impl="partitionable"):
    key = np.asarray(key, np.uint32)
    if impl == "partitionable":
        i = np.arange(num, dtype=np.uint64)
        o0, o1 = threefry2x32(key[0], key[1], i >> np.uint64(32), i & _M32)
        return np.stack([o0, o1], axis=1)
    return _legacy_words(key, 2 * num).reshape(num, 2)


def random_bits(key, shape, impl="partitionable"):
    key = np.asarray(key, np.uint32)
    n = int(np.prod(shape, dtype=np.int64))
    if impl == "partitionable":
        i = np.arange(n, dtype=np.uint64)
        o0, o1 = threefry2x32(key[0], key[1], i >> np.uint64(32), i & _M32)
        return (o0 ^ o1).reshape(shape)
    return _legacy_words(key, n).reshape(shape)


def uniform_from_bits(bits, lo, hi):
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo = np.float32(lo); hi = np.float32(hi)
    return np.maximum(lo, f * np.float32(hi - lo) + lo)


def erf_inv_f32(x):
    x = np.asarray(x, np.float32)
    w = -np.log1p(-(x * x)).astype(np.float32)
    small = w < np.float32(5.0)
    ws = (w - np.float32(2.5)).astype(np.float32)
    wl = (np.sqrt(np.maximum(w, np.float32(0))) - np.float32(3.0)).astype(np.float32)
    cs = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503,
          -0.00417768164, 0.246640727, 1.50140941)
    cl = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613,
          0.00943887047, 1.00167406, 2.83297682)
    ps = np.full(x.shape, cs[0], np.float32)
    pl = np.full(x.shape, cl[0], np.float32)
    for a, b in zip(cs[1:], cl[1:]):
        ps = (np.float32(a) + ps * ws).astype(np.float32)
        pl = (np.float32(b) + pl * wl).astype(np.float32)
    out = np.where(small, ps, pl).astype(np.float32) * x
    return np.where(np.abs(x) == 1, np.float32(np.inf) * x, out).astype(np.float32)


def normal_from_bits(bits):
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0), dtype=np.float32)
    return (np.float32(np.sqrt(2)) * erf_inv_f32(uniform_from_bits(bits, lo, np.float32(1.0)))).astype(np.float32)


def normal(key, shape, impl="partitionable"):
    return normal_from_bits(random_bits(key, shape, impl))


def gumbel(key, shape, impl="partitionable"):
    """jax.random.gumbel: -log(-log(uniform(key, shape, minval=finfo.tiny, maxval=1))) in float32."""
    tiny = np.finfo(np.float32).tiny
    u = uniform_from_bits(random_bits(key, shape, impl), tiny, np.float32(1.0))
    return (-np.log(-np.log(u))).astype(np.float32)


def uniform(key, shape, minval=0.0, maxval=1.0, impl="partitionable"):
    """jax.random.uniform(key, shape, float32, minval, maxval) (jax/_src/random.py::_uniform): the mantissa trick on
    random_bits, then * (maxval - minval) + minval, floored at minval.  PARITY UNPINNED above the bits (no published
    vector quoted); call sites continuous_ddpm.py:110, cnf.py:87."""
    return uniform_from_bits(random_bits(key, shape, impl), minval, maxval)


def randint(key, shape, minval, maxval, impl="partitionable"):
    """jax.random.randint(key, shape, minval, maxval) for int32 (jax/_src/random.py::_randint, jax 0.5.3):
        k1, k2 = split(key); higher, lower = random_bits(k1), random_bits(k2)            (uint32)
        span = uint32(maxval - minval) (1 when maxval <= minval)
        multiplier = ((2**16 mod span) ** 2) mod span                                    (= 2**32 mod span)
        offset = ((higher mod span) * multiplier + (lower mod span)) mod span            (wrapping uint32 arithmetic)
        result = minval + int32(offset)
    PARITY UNPINNED above the bits (restated from the published source, no vector quoted); call site ddpm.py:167."""
    k1, k2 = split(key, 2, impl)
    hi = random_bits(k1, shape, impl).astype(np.uint32)
    lo = random_bits(k2, shape, impl).astype(np.uint32)
    span = np.uint32(1) if maxval <= minval else np.uint32(maxval - minval)
    mult = np.uint32(65536) % span
    mult = np.uint32((np.uint64(mult) * np.uint64(mult)) & _M32) % span          # the product wraps in uint32 (span < 2**16 never does)
    with np.errstate(over="ignore"):
        off = ((hi % span) * mult + (lo % span)).astype(np.uint32) % span
    return (np.int32(minval) + off.astype(np.int32)).astype(np.int32)


def sampler_key_chain(kind, solver, steps, key, impl="partitionable"):
    """The split sequence inside the reference's scans.  Returns (noise_keys (n_steps, 2) uint32 — the key_ of every
    loop iteration, zeros where the solver draws none — and the rng the sampler returns)."""
    rng = np.asarray(key, np.uint32).copy()
    keys = np.zeros((steps, 2), np.uint32)
    for j in range(steps):
        if kind == "ddpm":                                   # ddpm.py:188,204
            rng = split(rng, 2, impl)[0]
            if solver == "ddpm":
                rng, keys[j] = split(rng, 2, impl)
        elif kind == "cddpm":                                # continuous_ddpm.py:161,190
            rng, _, k = split(rng, 3, impl)
            if solver == "ddpm":
                keys[j] = k
            if solver == "dpm2":
                rng = split(rng, 2, impl)[0]
        else:                                                # cnf.py:126
            rng = split(rng, 2, impl)[0]
    return keys, rng


def sampler_noise(kind, solver, steps, key, shape, impl="partitionable"):
    """noise (steps, *shape) as the reference's loop draws it (zeros for solvers without noise) and the returned rng."""
    keys, rng = sampler_key_chain(kind, solver, steps, key, impl)
    z = np.zeros((steps, *shape), np.float32)
    if solver == "ddpm" and kind != "cnf":
        for j in range(steps):
            z[j] = normal(keys[j], shape, impl)
    return z, rng
