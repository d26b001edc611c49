"""Integer-hash synthetic operands (TEST INFRASTRUCTURE — only tests/, oracle/make_golden.py and bench tooling import it).

The big inputs of the reference-produced operator matrix (tests/golden/operator_matrix.safetensors: a 1280→10240 weight is
52 MB in fp32) are not stored: they are functions of (seed, element index) in 32-bit integer arithmetic — a murmur3-style
finaliser — mapped onto the grid k/1024, |k| < 1024, times a power of two.  Every value is exactly representable in fp16
(and in fp32 / float64), every platform computes the same integers, so the fixture generator (which feeds them to the
REFERENCE's module in this container) and the tests on the GPU box see bit-identical operands without sharing an RNG."""
import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix(h):
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def grid_values(shape, seed: int, exp2: int = 0) -> torch.Tensor:
    """fp32 tensor of `shape`; element i = (k_i / 1024)·2^exp2 with k_i in [-1024, 1024) from the hash of (seed, i)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = _mix(i * np.uint64(0x9E3779B1) + np.uint64((seed * 0x7F4A7C15 + 0x1B873593) & 0xFFFFFFFF))
    k = (h >> np.uint64(21)).astype(np.int64) - 1024  # top 11 bits
    v = k.astype(np.float64) * (2.0 ** (exp2 - 10))
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def weight_exp2(K: int) -> int:
    """Power of two nearest to 1/sqrt(K) (nn.Linear's default init range), as an exponent."""
    return -int(round(0.5 * np.log2(K)))


# SD1.5's LoRA-wrapped linear layers: the 9 distinct (K, N) pairs, the square ones both ways — attention
# to_q/to_k/to_v carry no bias, to_out.0 and GEGLU.proj do (diffusers' CrossAttention / GEGLU; pinned by the build-owned
# harness model, harness/unet.py).  M = 16 rows (77 = one caption for the layers that read the text encoder's output).
SD_LAYER_KINDS = [
    # K, N, bias, M
    (320, 320, False, 16), (320, 320, True, 16), (320, 2560, True, 16), (768, 320, False, 77),
    (640, 640, False, 16), (640, 640, True, 16), (640, 5120, True, 16), (768, 640, False, 77),
    (1280, 1280, False, 16), (1280, 1280, True, 16), (1280, 10240, True, 16), (768, 1280, False, 77),
]
MATRIX_RANKS = (1, 4, 8, 16)
MATRIX_SCALES = (1.0, 0.7)
N_PROBES = 16


def matrix_cases():
    """(tag, K, N, bias, M, r, scale, seed) of every case of the operator matrix, in file order."""
    out = []
    for li, (K, N, bias, M) in enumerate(SD_LAYER_KINDS):
        for r in MATRIX_RANKS:
            for si, s in enumerate(MATRIX_SCALES):
                out.append((f"k{li}.r{r}.s{si}", K, N, bias, M, r, s, 1000 * li + 10 * r + si))
    return out


def matrix_inputs(K, N, bias, M, r, seed):
    """x [M,K], w [N,K], b [N] | None, dy [M,N], down [r,K], up [N,r] — all fp32 holding fp16-exact values."""
    x = grid_values((M, K), seed * 8 + 0, 1)                  # |x| < 2
    w = grid_values((N, K), seed * 8 + 1, weight_exp2(K))
    b = grid_values((N,), seed * 8 + 2, -3) if bias else None
    dy = grid_values((M, N), seed * 8 + 3, 0)
    down = grid_values((r, K), seed * 8 + 4, -2)              # ≈ N(0, 1/r²)-sized
    up = grid_values((N, r), seed * 8 + 5, -4)                # a warm-started `up` (zero at init: lora.py:46)
    return x, w, b, dy, down, up


def probes(n: int, seed: int) -> torch.Tensor:
    """[n, N_PROBES] float64 probe matrix for the projections the fixture stores instead of the wide outputs."""
    return grid_values((n, N_PROBES), 777_000 + seed, 0).double()
