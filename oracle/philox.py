"""CPU ORACLE (test infrastructure only) for the build-owned step randomness of ddpm_noise_prologue.

Philox4x32-10 (Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3", SC'11 — the published algorithm and
constants; known-answer vectors from the Random123 distribution are checked in tests/test_oracle_golden.py) and the
Box–Muller / integer-range mapping that csrc/ddpm_loss.hip applies.  Integer streams are bit-exact with the GPU;
the normals differ only by libm vs GPU logf/sincosf rounding.  There is no reference file to cite: the reference
draws its noise from torch's global generator (train_lora_dreambooth.py:824-833); this replaces that draw with a
counter-based stream that is identical on every rank and on CPU and GPU (SURVEY §8 d, f-3).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (counters) with scalar keys."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _u01(x):
    return ((x.astype(np.float32) + np.float32(0.5)) * np.float32(2.3283064365386963e-10)).astype(np.float32)


def step_randomness(batch, per_row, n_timesteps, seed, step):
    """(eps [batch, per_row] float32, t [batch] int64) exactly as the kernel draws them."""
    n = batch * per_row
    groups = (n + 3) // 4
    g = np.arange(groups, dtype=np.uint64)
    r0, r1, r2, r3 = philox4x32_10((g & np.uint64(0xFFFFFFFF)).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32),
                                   np.zeros(groups, np.uint32), np.zeros(groups, np.uint32), seed & 0xFFFFFFFF,
                                   step & 0xFFFFFFFF)
    rad0 = np.sqrt(np.float32(-2.0) * np.log(_u01(r0))).astype(np.float32)
    rad1 = np.sqrt(np.float32(-2.0) * np.log(_u01(r2))).astype(np.float32)
    a0 = np.float32(6.283185307179586) * _u01(r1)
    a1 = np.float32(6.283185307179586) * _u01(r3)
    z = np.stack([rad0 * np.cos(a0), rad0 * np.sin(a0), rad1 * np.cos(a1), rad1 * np.sin(a1)], axis=1).astype(np.float32)
    eps = z.reshape(-1)[:n].reshape(batch, per_row)
    b = np.arange(batch, dtype=np.uint32)
    t0, _, _, _ = philox4x32_10(b, np.zeros(batch, np.uint32), np.ones(batch, np.uint32), np.zeros(batch, np.uint32),
                                seed & 0xFFFFFFFF, step & 0xFFFFFFFF)
    t = ((t0.astype(np.uint64) * np.uint64(n_timesteps)) >> np.uint64(32)).astype(np.int64)
    return eps, t
