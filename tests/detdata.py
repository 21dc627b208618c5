"""Deterministic, version-independent test data (counter-based splitmix64 in NumPy).

The golden fixtures store only seeds and expected outputs; inputs are regenerated with these
functions here, on the GPU box and in tests/golden/gen_golden.py, bit-identically.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def u64(seed: int, n: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + _splitmix64(
            np.asarray([seed], dtype=np.uint64))[0])


def uniform01(seed: int, shape) -> np.ndarray:
    """float64 in [0, 1) with 53 random bits."""
    n = int(np.prod(shape, dtype=np.int64))
    return ((u64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))).reshape(shape)


def int8(seed: int, shape, lo: int = -128, hi: int = 128) -> np.ndarray:
    """integers in [lo, hi) as int8."""
    n = int(np.prod(shape, dtype=np.int64))
    r = (u64(seed, n) >> np.uint64(33)).astype(np.int64) % (hi - lo) + lo
    return r.astype(np.int8).reshape(shape)


def f16(seed: int, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    return (uniform01(seed, shape) * (hi - lo) + lo).astype(np.float16)


def f32(seed: int, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    return (uniform01(seed, shape) * (hi - lo) + lo).astype(np.float32)


def normal_f16(seed: int, shape, std: float = 1.0) -> np.ndarray:
    """Box-Muller on two uniform streams."""
    u1 = uniform01(seed, shape)
    u2 = uniform01(seed ^ 0x5DEECE66D, shape)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (z * std).astype(np.float16)
