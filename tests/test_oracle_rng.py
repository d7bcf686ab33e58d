"""CPU: the threefry / jax.random restatement (oracle/threefry_oracle.py) against published known answers.

 * Random123's known-answer vectors for Threefry-2x32-20 (kat_vectors; the three jax's own test-suite checks);
 * values the JAX documentation prints for PRNGKey(0): split -> [[4146024105, 967050713], [2718843009, 1272950319]] and
   normal(key, (1,)) -> -0.20584226 under the legacy layout (jax < 0.5 default); [[1797259609, 2579123966],
   [928981903, 3453687069]] and normal(key) -> 1.6226422 under the partitionable layout (jax >= 0.5 default, the
   reference's jax 0.5.3).  Quoted from the documentation, not generated here: jax is not installed.
"""
import numpy as np

import threefry_oracle as tf


def test_threefry_random123_vectors():
    kat = [((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)),
           ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
           ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]
    for key, ctr, want in kat:
        o0, o1 = tf.threefry2x32(key[0], key[1], ctr[0], ctr[1])
        assert (int(o0), int(o1)) == want


def test_documented_values_legacy():
    k = tf.prng_key(0)
    assert k.tolist() == [0, 0]
    assert tf.split(k, 2, "legacy").tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    z = tf.normal(k, (1,), "legacy")
    assert z.dtype == np.float32 and abs(float(z[0]) - (-0.20584226)) < 5e-8


def test_documented_values_partitionable():
    k = tf.prng_key(0)
    assert tf.split(k, 2, "partitionable").tolist() == [[1797259609, 2579123966], [928981903, 3453687069]]
    assert abs(float(tf.normal(k, (), "partitionable")) - 1.6226422) < 2e-7


def test_seed_key_layout():
    assert tf.prng_key(42).tolist() == [0, 42]
    assert tf.prng_key((7 << 32) | 9).tolist() == [7, 9]


def test_layouts_are_shape_consistent():
    k = tf.prng_key(3)
    for impl in ("partitionable", "legacy"):
        a = tf.random_bits(k, (6, 5), impl)
        b = tf.random_bits(k, (30,), impl)
        assert np.array_equal(a.ravel(), b)           # only the flat index matters
        assert tf.random_bits(k, (7,), impl).shape == (7,)   # odd length (legacy pads the counter)
    # partitionable: a prefix of a longer draw is the shorter draw; legacy: it is not (the halves move)
    assert np.array_equal(tf.random_bits(k, (10,), "partitionable"), tf.random_bits(k, (20,), "partitionable")[:10])
    assert not np.array_equal(tf.random_bits(k, (10,), "legacy"), tf.random_bits(k, (20,), "legacy")[:10])


def test_normal_moments_and_range():
    z = tf.normal(tf.prng_key(1), (400000,))
    assert np.isfinite(z).all()
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z ** 3).mean())) < 2e-2 and abs(float((z ** 4).mean()) - 3.0) < 5e-2
    # the extreme bit patterns: u = lo (clamped) and u = 1 - 2^-23 * 2 ... both finite
    e = tf.normal_from_bits(np.array([0, 0xFFFFFFFF], np.uint32))
    assert np.isfinite(e).all() and e[0] < -5 and e[1] > 5


def test_sampler_key_chain_protocol():
    k = tf.prng_key(5)
    for impl in ("partitionable", "legacy"):
        # DDPM ddpm solver: two splits per iteration (ddpm.py:188,204)
        keys, rng = tf.sampler_key_chain("ddpm", "ddpm", 3, k, impl)
        r = k
        for j in range(3):
            r = tf.split(r, 2, impl)[0]
            r, kj = tf.split(r, 2, impl)
            assert np.array_equal(keys[j], kj)
        assert np.array_equal(rng, r)
        # ddim: one split, no noise key
        keys, rng = tf.sampler_key_chain("ddpm", "ddim", 3, k, impl)
        r = k
        for j in range(3):
            r = tf.split(r, 2, impl)[0]
        assert np.array_equal(rng, r) and not keys.any()
        # continuous: split(rng, 3) (continuous_ddpm.py:161); dpm2 one more split (:190)
        keys, rng = tf.sampler_key_chain("cddpm", "ddpm", 2, k, impl)
        r = k
        for j in range(2):
            r, _, kj = tf.split(r, 3, impl)
            assert np.array_equal(keys[j], kj)
        assert np.array_equal(rng, r)
        _, rng2 = tf.sampler_key_chain("cddpm", "dpm2", 2, k, impl)
        r = k
        for j in range(2):
            r = tf.split(r, 3, impl)[0]
            r = tf.split(r, 2, impl)[0]
        assert np.array_equal(rng2, r)
        _, rng3 = tf.sampler_key_chain("cnf", "euler", 4, k, impl)
        r = k
        for j in range(4):
            r = tf.split(r, 2, impl)[0]
        assert np.array_equal(rng3, r)


# ---- uniform / randint (round 2: the t draws of add_noise / linear_interpolation) --------------------------------
def test_uniform_and_randint_restatements():
    """PARITY UNPINNED above random_bits (no published vector quoted).  Structural known answers of the published
    algorithm: a power-of-two span makes the multiplier 0, so randint reduces to (lower bits mod span); every value is in
    range; uniform is the mantissa trick scaled into [minval, maxval)."""
    import threefry_oracle as tf
    key = tf.prng_key(7)
    for impl in ("partitionable", "legacy"):
        r = tf.randint(key, (1000,), 3, 3 + 8, impl)
        k1, k2 = tf.split(key, 2, impl)
        assert np.array_equal(r, 3 + (tf.random_bits(k2, (1000,), impl) % np.uint32(8)).astype(np.int32))
        r5 = tf.randint(key, (4000, 1), 1, 6, impl)           # DDPM.add_noise with steps = 5
        assert r5.dtype == np.int32 and r5.min() == 1 and r5.max() == 5
        counts = np.bincount(r5.ravel(), minlength=6)[1:]
        assert (np.abs(counts - 800) < 120).all()
        hi, lo = tf.random_bits(k1, (4000, 1), impl).astype(np.uint64), tf.random_bits(k2, (4000, 1), impl).astype(np.uint64)
        want = 1 + (((hi % 5) * ((2 ** 32) % 5) + lo % 5) % 5)     # exact integer arithmetic: no wrap happens for span 5
        assert np.array_equal(r5.astype(np.uint64), want)
        assert (tf.randint(key, (10,), 4, 4, impl) == 4).all()      # maxval <= minval -> minval
        u = tf.uniform(key, (4000,), 1e-3, 0.9946, impl)
        assert u.dtype == np.float32 and u.min() >= np.float32(1e-3) and u.max() < np.float32(0.9946)
        f = ((tf.random_bits(key, (4000,), impl) >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1)
        assert np.array_equal(u, np.maximum(np.float32(1e-3), f * np.float32(np.float32(0.9946) - np.float32(1e-3)) + np.float32(1e-3)))
