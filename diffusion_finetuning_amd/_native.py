"""ctypes binding of liblora_hip.so (C-ABI declared in include/lora_hip.h).

Thin by design: tensors in, raw device pointers + sizes + the current HIP stream out.  There is no
fallback: if the library is missing, or a call returns a non-zero status, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import torch

_LIB_PATH = os.environ.get("DFA_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liblora_hip.so")  # (DFA_LIB_PATH: dev A/B builds)
_lib = None
_lock = threading.Lock()

ABI_VERSION = 9
PROF_KINDS = 18

_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


class GradProblem(ctypes.Structure):
    """lora_grad_problem of include/lora_hip.h (host memory, consumed during the call)."""
    _fields_ = [
        ("S", _vp), ("P", _vp), ("out", _vp * 4),
        ("s_stride", _i64), ("p_stride", _i64), ("part_stride", _i64), ("M", _i64),
        ("C", _i32), ("r", _i32), ("rg", _i32), ("out_kn", _i32), ("n_blocks", _i32),
        ("scale", _f32),
    ]


GRAD_MAX_BLOCKS = 64


class ProfTotals(ctypes.Structure):
    _fields_ = [
        ("launches", _i64 * PROF_KINDS),
        ("ms", ctypes.c_double * PROF_KINDS),
        ("bytes", ctypes.c_double * PROF_KINDS),
        ("flops", ctypes.c_double * PROF_KINDS),
    ]


# symbol -> (restype, argtypes); must list every function include/lora_hip.h declares
SIGNATURES = {
    "lora_version": (_i32, []),
    "lora_status_string": (ctypes.c_char_p, [_i32]),
    "lora_pack_factors": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "lora_pack_factors_batched": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "lora_linear_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lora_linear_fwd_ws": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _i64,
                                  _vp]),
    "lora_linear_geglu_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lora_linear_bwd_input": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lora_linear_bwd_params": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lora_reduce_partials": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    "lora_grad_row_blocks": (_i32, [_i64]),
    "lora_grad_batched": (_i32, [ctypes.POINTER(GradProblem), _i32, _i32, _vp]),
    "lora_grad_plan_bytes": (_i64, [ctypes.POINTER(GradProblem), _i32]),
    "lora_grad_plan": (_i32, [ctypes.POINTER(GradProblem), _i32, _i32, _vp, _i64, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "lora_grad_planned": (_i32, [_vp, _i32, _i32, _i32, ctypes.c_double, ctypes.c_double, _vp]),
    "lora_fold_partials": (_i32, [_vp, _i32, _i64, _vp, _i64, _vp, _i32, _vp]),
    "lora_gemm_packed": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _f32,
                                _i64, _vp, _i64, _i32, _vp]),
    "lora_gemm_parts": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lora_gemm_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "lora_linear_bwd_input_ws": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp,
                                        _i64, _vp]),
    "lora_pack_items": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "ddpm_mse_fwd_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _f32, _f32, _vp, _vp, _vp, _i32, _vp]),
    "lora_mse_workspace_bytes": (_i64, []),
    "lora_mask_prepare": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lora_merge_weight": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "lora_merge_weight_batched": (_i32, [_vp, _i32, _i64, _f32, _vp]),
    "lora_lerp": (_i32, [_vp, _vp, _i64, _f32, _f32, _i32, _vp]),
    "lora_cast_matrix": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "lora_grad_sqnorm": (_i32, [_vp, _i64, _f32, _vp, _vp, _vp]),
    "lora_sqnorm_workspace_bytes": (_i64, []),
    "lora_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "ddpm_add_noise": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "ddpm_noise_prologue": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, ctypes.c_uint64, ctypes.c_uint64,
                                   _i32, _i32, _vp]),
    "embed_rows_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp]),
    "embed_rows_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp]),
    "lora_adamw_rows": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "geglu_linear_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "geglu_gate_fwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "geglu_gate_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "attn_split_heads": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "attn_merge_heads": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "attn_merge_heads_strided": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp]),
    "attn_ctx_supported": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "attn_ctx_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_ctx_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "attn_ctx_fwd_strided": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_ctx_bwd_strided": (_i32, [_vp] * 8 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_flash_fwd_strided": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_flash_bwd_strided": (_i32, [_vp] * 10 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_ctx_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_flash_supported": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "attn_flash_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "attn_flash_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "attn_flash_bwd": (_i32, [_vp] * 10 + [_i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lora_prof_enable": (_i32, [_i32]),
    "lora_prof_collect": (_i32, [ctypes.POINTER(ProfTotals)]),
    "lora_prof_null_mode": (_i32, [_i32]),
    "lora_prof_kernel_name": (ctypes.c_char_p, [_i32]),
}


def library_path() -> str:
    return _LIB_PATH


def lib():
    """Loads the library once.  Raises RuntimeError (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"diffusion_finetuning_amd: native library {_LIB_PATH} not found. Build it with "
                "`python -m diffusion_finetuning_amd.build_native` (needs hipcc, targets gfx950). "
                "There is no CPU or PyTorch fallback for the LoRA hot path."
            )
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.lora_version() != ABI_VERSION:
            raise RuntimeError(
                f"diffusion_finetuning_amd: {_LIB_PATH} has ABI {handle.lora_version()}, expected {ABI_VERSION}; rebuild it"
            )
        _lib = handle
    return _lib


def _check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().lora_status_string(status).decode()
        if status == -2:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed with status {status}: {msg}")


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise RuntimeError(f"diffusion_finetuning_amd: unsupported dtype {dtype} (float32, float16, bfloat16 only)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor) -> int:
    """hipStream_t of the calling thread's current stream on t's device (the raw-handle query when this PyTorch has
    it: building a torch.cuda.Stream object per launch costs ~2 µs of host time, ~700 times per step)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "diffusion_finetuning_amd: the LoRA hot path runs only on a HIP device (MI355X); got a "
                f"{t.device} tensor. Move the model and inputs to 'cuda'. There is no CPU fallback."
            )


def staging_device(*tensors) -> torch.device:
    """The HIP device host-resident operands are staged through: the device of the first device tensor given, else the
    current HIP device.  Raises (never computes on the CPU) when there is none."""
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError(
            "diffusion_finetuning_amd: this operation runs as a HIP kernel and no HIP device (MI355X) is visible; "
            "host tensors are staged through the device, there is no CPU implementation."
        )
    return torch.device("cuda", torch.cuda.current_device())


def lora_pack_factors(a, b, dtype: torch.dtype):
    """a [r,K], b [N,r] fp32 → (Apack [2,16·K], Bpack [2,16·N]) in `dtype` (both orientations, see lora_hip.h);
    (None, None) for r > 16."""
    _require_device(a, b)
    r, K = a.shape
    N = b.shape[0]
    if r > 16:
        return None, None
    apack = torch.empty(32 * K, dtype=dtype, device=a.device)
    bpack = torch.empty(32 * N, dtype=dtype, device=a.device)
    _check(lib().lora_pack_factors(_ptr(a), _ptr(b), _ptr(apack), _ptr(bpack), K, N, r, dtype_code(dtype), _stream(a)),
           "lora_pack_factors")
    return apack, bpack


def lora_pack_factors_batched(table, n_layers: int, max_len: int, params, packed) -> None:
    _require_device(table, params, packed)
    _check(lib().lora_pack_factors_batched(_ptr(table), n_layers, max_len, _ptr(params), _ptr(packed),
                                           dtype_code(packed.dtype), _stream(params)), "lora_pack_factors_batched")


def lora_linear_fwd(x2, w, bias, a, b, scale: float, packs=None):
    """x2 [M,K], w [N,K], bias [N]|None (dtype of x2); a [r,K], b [N,r] fp32 masters; packs = (Apack, Bpack)
    (made here when not given). Returns (y [M,N], T [M,r] fp32)."""
    _require_device(x2, w, bias, a, b)
    M, K = x2.shape
    N, r = b.shape
    if packs is None:
        packs = lora_pack_factors(a, b, x2.dtype)
    y = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
    t = torch.empty((M, r), dtype=torch.float32, device=x2.device)
    ws = _splitk_workspace(M, K, N, x2)
    _check(
        lib().lora_linear_fwd_ws(_ptr(x2), _ptr(w), _ptr(bias), _ptr(a), _ptr(b), _ptr(packs[0]), _ptr(packs[1]), _ptr(y),
                                 _ptr(t), M, K, N, r, float(scale), dtype_code(x2.dtype), _ptr(ws),
                                 0 if ws is None else ws.numel() * 4, _stream(x2)),
        "lora_linear_fwd",
    )
    return y, t


def lora_linear_geglu_fwd(x2, w, bias, r: int, scale: float, packs, want_y: bool):
    """`proj` forward with the GEGLU gate in the epilogue: x2 [M,K], w [2F,K], packs = (Apack, Bpack).
    Returns (out [M,F], y [M,2F] | None, T [M,r] fp32), or None when the library has no fused kernel for the shape/dtype."""
    _require_device(x2, w, bias)
    M, K = x2.shape
    N = w.shape[0]
    out = torch.empty((M, N // 2), dtype=x2.dtype, device=x2.device)
    y = torch.empty((M, N), dtype=x2.dtype, device=x2.device) if want_y else None
    t = torch.empty((M, r), dtype=torch.float32, device=x2.device)
    st = lib().lora_linear_geglu_fwd(_ptr(x2), _ptr(w), _ptr(bias), _ptr(packs[0]), _ptr(packs[1]), _ptr(y), _ptr(out),
                                     _ptr(t), M, K, N, r, float(scale), dtype_code(x2.dtype), _stream(x2))
    if st == -5:
        return None
    _check(st, "lora_linear_geglu_fwd")
    return out, y, t


def lora_linear_bwd_input(dy2, wt, a, b, scale: float, need_dx: bool, packs=None):
    """dy2 [M,N]; wt [K,N] (= Wᵀ) or None when need_dx is False; packs = (Apack, Bpack) (made here when not given).
    Returns (dx [M,K]|None, U [M,r] fp32)."""
    _require_device(dy2, wt, a, b)
    M, N = dy2.shape
    r, K = a.shape
    if packs is None:
        packs = lora_pack_factors(a, b, dy2.dtype)
    dx = torch.empty((M, K), dtype=dy2.dtype, device=dy2.device) if need_dx else None
    u = torch.empty((M, r), dtype=torch.float32, device=dy2.device)
    ws = _splitk_workspace(M, N, K, dy2) if need_dx else None
    _check(
        lib().lora_linear_bwd_input_ws(_ptr(dy2), _ptr(wt), _ptr(a), _ptr(b), _ptr(packs[0]), _ptr(packs[1]), _ptr(dx),
                                       _ptr(u), M, K, N, r, float(scale), dtype_code(dy2.dtype), _ptr(ws),
                                       0 if ws is None else ws.numel() * 4, _stream(dy2)),
        "lora_linear_bwd_input",
    )
    return dx, u


_ws_bytes_cache = {}
_splitk_ws = {}        # device -> persistent workspace (ticket header zeroed ONCE: every launch leaves it zero again)
_splitk_ws_retired = []  # outgrown workspaces stay alive: a recorded hipGraph may still hold their address


def _splitk_workspace(M: int, Kc: int, Nc: int, like):
    """Scratch for a contraction the library wants to split over K (None when it does not): one buffer per device, grown
    on demand, shared by every call on the device — the launches are stream-ordered, and the hot path runs on one stream
    per process (two streams splitting concurrently would need a workspace each: lora_hip.h)."""
    key = (M, Kc, Nc, like.dtype)
    nbytes = _ws_bytes_cache.get(key)
    if nbytes is None:
        nbytes = _ws_bytes_cache[key] = int(lib().lora_gemm_workspace_bytes(M, Kc, Nc, dtype_code(like.dtype)))
    if nbytes <= 0:
        return None
    ws = _splitk_ws.get(like.device)
    if ws is None or ws.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            return None  # never allocate-and-zero inside a recording: this launch simply runs unsplit
        if ws is not None:
            _splitk_ws_retired.append(ws)
        ws = _splitk_ws[like.device] = torch.zeros(max(nbytes, 32 << 20) // 4, dtype=torch.float32, device=like.device)
    return ws


def grad_blocks_for(M: int) -> int:
    """Row blocks for the factor-gradient kernel when the caller has no fixed layout (plain autograd mode)."""
    forced = os.environ.get("LORA_GRAD_BLOCKS")  # tuning knob for tools/gemm_bench.py
    if forced:
        return int(forced)
    return max(1, min(128, M // 32))


def lora_linear_bwd_params_partial(dy2, x2, t, u, ga_part, gb_part, part_stride: int, n_blocks: int, scale: float):
    """Stores per-row-block partial sums: block b at ga_part + b·part_stride ([r,K]) / gb_part + b·part_stride ([N,r])."""
    _require_device(dy2, x2, t, u, ga_part, gb_part)
    M, N = dy2.shape
    K = x2.shape[1]
    r = t.shape[1]
    _check(
        lib().lora_linear_bwd_params(_ptr(dy2), _ptr(x2), _ptr(t), _ptr(u), _ptr(ga_part), _ptr(gb_part),
                                     int(part_stride), int(n_blocks), M, K, N, r, float(scale),
                                     dtype_code(dy2.dtype), _stream(dy2)),
        "lora_linear_bwd_params",
    )


_row_blocks_cache = {}


def grad_row_blocks(M: int) -> int:
    """Row blocks lora_grad_batched uses for a problem of M rows when n_blocks is left to the library."""
    nb = _row_blocks_cache.get(M)
    if nb is None:
        nb = _row_blocks_cache[M] = int(lib().lora_grad_row_blocks(int(M)))
    return nb


def grad_problem(S, s_ptr_off: int, s_stride: int, C: int, P, p_ptr_off: int, p_stride: int, r: int, outs, rg: int,
                 out_kn: bool, part_stride: int, M: int, scale: float, n_blocks: int = 0):
    """One lora_grad_problem: G[c,j] = scale·Σ_m S[m,c]·P[m,j].  S / P are device tensors, *_ptr_off element offsets of
    the slice's first column; outs = device pointers (ints) of the rank groups' row-block-0 partials."""
    q = GradProblem()
    q.S = S.data_ptr() + s_ptr_off * S.element_size()
    q.P = P.data_ptr() + p_ptr_off * 4
    for i, o in enumerate(outs):
        q.out[i] = o
    q.s_stride, q.p_stride, q.part_stride, q.M = s_stride, p_stride, part_stride, M
    q.C, q.r, q.rg, q.out_kn, q.n_blocks, q.scale = C, r, rg, int(out_kn), n_blocks, scale
    return q


def lora_grad_batched(problems, dtype: torch.dtype, device) -> None:
    """Launches every problem of the list (a few launches whatever its length; tables travel as kernel arguments)."""
    n = len(problems)
    if n == 0:
        return
    lora_grad_batched_array((GradProblem * n)(*problems), n, dtype, device)


def lora_grad_batched_array(arr, n: int, dtype: torch.dtype, device) -> None:
    """The same on a `(GradProblem * n)` array the caller keeps (ops._SinkPlan re-uses one from step to step)."""
    stream = _raw_stream(device.index) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
    _check(lib().lora_grad_batched(arr, n, dtype_code(dtype), stream), "lora_grad_batched")


def lora_grad_plan_bytes(problems) -> int:
    n = len(problems)
    return int(lib().lora_grad_plan_bytes((GradProblem * n)(*problems), n)) if n else 0


def lora_grad_one_launch(problems, dtype: torch.dtype, device, host_plan=None):
    """All problems in ONE launch through a plan in device memory (include/lora_hip.h: lora_grad_plan / lora_grad_planned).
    host_plan: a pinned uint8 tensor to write the plan into (a recording hands over a buffer it owns — the copy node reads it
    on every replay); None allocates one (host-launched steps: torch's pinned allocator keeps the block until the copy has
    run).  Returns the (host, device) plan tensors — the caller keeps them alive until the launch has run (a recording: for
    its life) — or None when the library declines (fp32 / unaligned operands / rank > 16: use lora_grad_batched)."""
    n = len(problems)
    if n == 0 or dtype == torch.float32:
        return None
    arr = (GradProblem * n)(*problems)
    need = int(lib().lora_grad_plan_bytes(arr, n))
    if host_plan is None:
        host_plan = torch.empty(need, dtype=torch.uint8, pin_memory=True)
    elif host_plan.numel() < need or not host_plan.is_pinned():
        return None
    n_items, n_blocks = _i32(0), _i32(0)
    st = lib().lora_grad_plan(arr, n, dtype_code(dtype), host_plan.data_ptr(), host_plan.numel(), ctypes.byref(n_items),
                              ctypes.byref(n_blocks))
    if st == -5:
        return None
    _check(st, "lora_grad_plan")
    used = n_items.value * 128 + n_blocks.value * 4
    if used == 0:
        return (host_plan, None)
    dev_plan = torch.empty(used, dtype=torch.uint8, device=device)
    dev_plan.copy_(host_plan[:used], non_blocking=True)
    e = float(torch.empty(0, dtype=dtype).element_size())
    nbytes = sum(e * q.M * q.C + 4.0 * q.M * q.r + 4.0 * q.r * q.C for q in problems)
    flops = sum(2.0 * q.M * q.r * q.C for q in problems)
    stream = _raw_stream(device.index) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
    _check(lib().lora_grad_planned(dev_plan.data_ptr(), n_items.value, n_blocks.value, dtype_code(dtype), nbytes, flops, stream),
           "lora_grad_planned")
    return (host_plan, dev_plan)


def lora_fold_partials(ranges, n_ranges: int, max_len: int, partials, part_stride: int, grads, accumulate: bool) -> None:
    """grads[off:off+len] (+)= Σ_{b<blocks} partials[b·part_stride + off : …] for every row {off, len, blocks, 0} of the
    int64 device table `ranges`."""
    _require_device(ranges, partials, grads)
    _check(lib().lora_fold_partials(_ptr(ranges), int(n_ranges), int(max_len), _ptr(partials), int(part_stride),
                                    _ptr(grads), int(accumulate), _stream(grads)), "lora_fold_partials")


def lora_pack_items(table, n_items: int, max_len: int, params, packed) -> None:
    _require_device(table, params, packed)
    _check(lib().lora_pack_items(_ptr(table), n_items, max_len, _ptr(params), _ptr(packed), dtype_code(packed.dtype),
                                 _stream(params)), "lora_pack_items")


def lora_gemm_packed(am, lda: int, bm, bias, fp, qp, tile_part, part_table, n_parts: int, c, p_out, M: int, Kc: int,
                     Nc: int, r: int, scale: float, work_cols: int = 0) -> None:
    """C = Am·Bmᵀ + bias + s·P·Qᵀ, P = Am·Fᵀ on packed factors (include/lora_hip.h: lora_gemm_packed).  All operands
    are device tensors (or None); nothing is allocated here."""
    _require_device(am, bm, bias, fp, qp, tile_part, part_table, c, p_out)
    ws = _splitk_workspace(M, Kc, Nc, am) if (c is not None and tile_part is None) else None
    _check(lib().lora_gemm_packed(_ptr(am), int(lda), _ptr(bm), _ptr(bias), _ptr(fp), _ptr(qp), _ptr(tile_part),
                                  _ptr(part_table), int(n_parts), _ptr(c), _ptr(p_out), int(M), int(Kc), int(Nc), int(r),
                                  float(scale), int(work_cols), _ptr(ws), 0 if ws is None else ws.numel() * 4,
                                  dtype_code(am.dtype), _stream(am)), "lora_gemm_packed")


def lora_gemm_parts(am, bm, bias, fp, qp, c, p_out, ldp: int, M: int, Kc: int, Nc: int, r: int, n_parts: int,
                    parts_on_k: bool, scale: float) -> bool:
    """n_parts equal LoRA layers that share an input in one launch at any rank <= 16 (include/lora_hip.h: lora_gemm_parts);
    False when the library has no kernel for the shape / dtype (the caller runs the layers one by one)."""
    _require_device(am, bm, bias, fp, qp, c, p_out)
    st = lib().lora_gemm_parts(_ptr(am), _ptr(bm), _ptr(bias), _ptr(fp), _ptr(qp), _ptr(c), _ptr(p_out), int(ldp), int(M),
                               int(Kc), int(Nc), int(r), int(n_parts), int(bool(parts_on_k)), float(scale),
                               dtype_code(am.dtype), _stream(am))
    if st == -5:
        return False
    _check(st, "lora_gemm_parts")
    return True


def lora_reduce_partials(partials, part_stride: int, n_blocks: int, grads, n: int, accumulate: bool) -> None:
    _require_device(partials, grads)
    _check(lib().lora_reduce_partials(_ptr(partials), int(part_stride), int(n_blocks), _ptr(grads), int(n),
                                      int(accumulate), _stream(grads)), "lora_reduce_partials")


def lora_linear_bwd_params(dy2, x2, t, u, ga, gb, scale: float):
    """Accumulates into ga [r,K], gb [N,r] (fp32): partial sums per row block + ordered reduction."""
    _require_device(dy2, x2, t, u, ga, gb)
    M, N = dy2.shape
    r, K = ga.shape
    nb = grad_blocks_for(M)
    size = r * (K + N)
    stride = (size + 3) // 4 * 4
    ws = torch.empty((nb + 1, stride), dtype=torch.float32, device=dy2.device)
    lora_linear_bwd_params_partial(dy2, x2, t, u, ws[0], ws[0, r * K:], stride, nb, scale)
    out = ws[nb]
    lora_reduce_partials(ws, stride, nb, out, size, False)
    ga += out[: r * K].view(r, K)
    gb += out[r * K: size].view(N, r)


_ws_cache = {}


def _workspace(device, nbytes: int, tag: str):
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def ddpm_mse_fwd_bwd(pred, target, mask, n_inst: int, n_prior: int, prior_weight: float, grad_scale: float,
                     want_grad: bool = True):
    """pred/target [rows, C, H, W] contiguous, same dtype; mask fp32 [rows,1,H,W] (normalised) or None.
    Returns (loss fp32 scalar tensor, dpred|None)."""
    _require_device(pred, target, mask)
    rows = pred.shape[0]
    assert rows == n_inst + n_prior
    per_row = pred[0].numel()
    hw = pred.shape[-1] * pred.shape[-2]
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    ws = _workspace(pred.device, int(lib().lora_mse_workspace_bytes()), "mse")
    _check(
        lib().ddpm_mse_fwd_bwd(_ptr(pred), _ptr(target), _ptr(mask), n_inst, n_prior, per_row, hw,
                               float(prior_weight), float(grad_scale), _ptr(loss), _ptr(dpred), _ptr(ws),
                               dtype_code(pred.dtype), _stream(pred)),
        "ddpm_mse_fwd_bwd",
    )
    return loss, dpred


def lora_mask_prepare(mask_in, h: int, w: int):
    _require_device(mask_in)
    B, _, hin, win = mask_in.shape
    out = torch.empty((B, 1, h, w), dtype=torch.float32, device=mask_in.device)
    _check(lib().lora_mask_prepare(_ptr(mask_in), _ptr(out), B, hin, win, h, w, _stream(mask_in)), "lora_mask_prepare")
    return out


def lora_merge_weight(w, a, b, alpha: float, factor_dtype: torch.dtype = torch.float32) -> None:
    """In place W += alpha*(b@a); a [r,K], b [N,r] fp32 copies of factors held in `factor_dtype`."""
    _require_device(w, a, b)
    N, K = w.shape
    r = a.shape[0]
    _check(lib().lora_merge_weight(_ptr(w), _ptr(a), _ptr(b), K, N, r, float(alpha), dtype_code(w.dtype),
                                   dtype_code(factor_dtype), _stream(w)),
           "lora_merge_weight")


def lora_merge_weight_batched(entries, alpha: float) -> None:
    """entries: [(W [N,K] device tensor (merged in place), a [r,K] fp32, b [N,r] fp32, factor_dtype), ...] — one launch."""
    if not entries:
        return
    rows, max_elems = [], 1
    for w, a, b, fdt in entries:
        _require_device(w, a, b)
        N, K = w.shape
        rows.append([w.data_ptr(), a.data_ptr(), b.data_ptr(), K, N, a.shape[0], dtype_code(w.dtype), dtype_code(fdt)])
        max_elems = max(max_elems, N * K)
    dev = entries[0][0].device
    table = torch.tensor(rows, dtype=torch.int64).to(dev)
    _check(lib().lora_merge_weight_batched(_ptr(table), len(rows), max_elems, float(alpha), _stream(entries[0][0])),
           "lora_merge_weight_batched")


def lora_lerp_(x1, x2, alpha: float) -> None:
    """In place x1 ← T(T(alpha·x1) + T((1-alpha)·x2)) on flat device tensors of one dtype (cli_lora_add.py:52-55)."""
    _require_device(x1, x2)
    assert x1.dtype == x2.dtype and x1.numel() == x2.numel() and x1.is_contiguous() and x2.is_contiguous()
    _check(lib().lora_lerp(_ptr(x1), _ptr(x2), x1.numel(), float(alpha), float(1 - alpha), dtype_code(x1.dtype),
                           _stream(x1)), "lora_lerp")


def lora_cast_matrix(src, dst_dtype: torch.dtype, transpose: bool):
    _require_device(src)
    rows, cols = src.shape
    dst = torch.empty((cols, rows) if transpose else (rows, cols), dtype=dst_dtype, device=src.device)
    _check(lib().lora_cast_matrix(_ptr(src), _ptr(dst), rows, cols, dtype_code(src.dtype), dtype_code(dst_dtype),
                                  int(transpose), _stream(src)), "lora_cast_matrix")
    return dst


def lora_grad_sqnorm(grad, grad_mul: float, norm_out) -> None:
    _require_device(grad, norm_out)
    ws = _workspace(grad.device, int(lib().lora_sqnorm_workspace_bytes()), "sqnorm")
    _check(lib().lora_grad_sqnorm(_ptr(grad), grad.numel(), float(grad_mul), _ptr(norm_out), _ptr(ws), _stream(grad)),
           "lora_grad_sqnorm")


def lora_adamw_step(param, grad, exp_avg, exp_avg_sq, norm_in, grad_mul, max_norm, lr, beta1, beta2, eps,
                    weight_decay, step: int) -> None:
    _require_device(param, grad, exp_avg, exp_avg_sq, norm_in)
    _check(lib().lora_adamw_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                                 _ptr(norm_in), float(grad_mul), float(max_norm), float(lr), float(beta1),
                                 float(beta2), float(eps), float(weight_decay), int(step), _stream(param)),
           "lora_adamw_step")


def ddpm_add_noise(x0, eps, t, sqrt_acp, sqrt_1macp, out_dtype: torch.dtype, v_prediction: bool, want_target=True):
    _require_device(x0, eps, t, sqrt_acp, sqrt_1macp)
    B = x0.shape[0]
    per_row = x0[0].numel()
    noisy = torch.empty(x0.shape, dtype=out_dtype, device=x0.device)
    target = torch.empty(x0.shape, dtype=out_dtype, device=x0.device) if want_target else None
    _check(lib().ddpm_add_noise(_ptr(x0), _ptr(eps), _ptr(t), _ptr(sqrt_acp), _ptr(sqrt_1macp), _ptr(noisy),
                                _ptr(target), B, per_row, int(v_prediction), dtype_code(out_dtype), _stream(x0)),
           "ddpm_add_noise")
    return noisy, target


def ddpm_noise_prologue(x0, sqrt_acp, sqrt_1macp, out_dtype: torch.dtype, seed: int, step: int, v_prediction: bool,
                        n_timesteps: int = 1000, want_draw: bool = False):
    """Draws eps/t on the device (Philox keyed by seed, step) and returns (noisy, target, t[, eps])."""
    _require_device(x0, sqrt_acp, sqrt_1macp)
    B = x0.shape[0]
    per_row = x0[0].numel()
    noisy = torch.empty(x0.shape, dtype=out_dtype, device=x0.device)
    target = torch.empty(x0.shape, dtype=out_dtype, device=x0.device)
    t = torch.empty(B, dtype=torch.int64, device=x0.device)
    eps = torch.empty(x0.shape, dtype=torch.float32, device=x0.device) if want_draw else None
    _check(lib().ddpm_noise_prologue(_ptr(x0), _ptr(sqrt_acp), _ptr(sqrt_1macp), _ptr(noisy), _ptr(target), _ptr(eps),
                                     _ptr(t), B, per_row, int(n_timesteps), int(seed) & (2**64 - 1),
                                     int(step) & (2**64 - 1), int(v_prediction), dtype_code(out_dtype), _stream(x0)),
           "ddpm_noise_prologue")
    return (noisy, target, t, eps) if want_draw else (noisy, target, t)


def embed_rows_fwd(table, ids, out_dtype: torch.dtype):
    """table [V,D] fp32, ids int64 (any shape) → rows [*ids.shape, D] in out_dtype (include/lora_hip.h: embed_rows_fwd)."""
    _require_device(table, ids)
    V, D = table.shape
    flat = ids.reshape(-1)
    flat = flat if flat.is_contiguous() else flat.contiguous()
    out = torch.empty((flat.numel(), D), dtype=out_dtype, device=table.device)
    _check(lib().embed_rows_fwd(_ptr(table), _ptr(flat), _ptr(out), flat.numel(), D, V, dtype_code(out_dtype), _stream(table)),
           "embed_rows_fwd")
    return out.view(*ids.shape, D)


def embed_rows_bwd(d_rows, ids, grad_table, accumulate: bool = False, active=None) -> None:
    """grad_table[t] (+)= Σ of the rows d_rows[p] with ids[p] == t, positions in ascending order (deterministic);
    active [V] uint8 (optional): set to 1 for every token that occurs."""
    _require_device(d_rows, ids, grad_table, active)
    V, D = grad_table.shape
    assert d_rows.is_contiguous() and ids.is_contiguous() and d_rows.numel() == ids.numel() * D
    assert active is None or (active.dtype == torch.uint8 and active.numel() == V)
    _check(lib().embed_rows_bwd(_ptr(d_rows), _ptr(ids), _ptr(grad_table), _ptr(active), ids.numel(), D, V,
                                dtype_code(d_rows.dtype), int(accumulate), _stream(d_rows)), "embed_rows_bwd")


def lora_adamw_rows(param, grad, exp_avg, exp_avg_sq, active, norm_in, grad_mul, max_norm, lr, beta1, beta2, eps,
                    weight_decay, step: int) -> None:
    """AdamW over a [V, D] table whose rows outside `active` never had a gradient (include/lora_hip.h: lora_adamw_rows)."""
    _require_device(param, grad, exp_avg, exp_avg_sq, active, norm_in)
    V, D = param.shape
    _check(lib().lora_adamw_rows(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(active), V, D, _ptr(norm_in),
                                 float(grad_mul), float(max_norm), float(lr), float(beta1), float(beta2), float(eps),
                                 float(weight_decay), int(step), _stream(param)), "lora_adamw_rows")


def geglu_gate_fwd(y2):
    """y2 [M, 2C] contiguous → h·gelu(g) [M, C]."""
    _require_device(y2)
    M, C2 = y2.shape
    out = torch.empty((M, C2 // 2), dtype=y2.dtype, device=y2.device)
    _check(lib().geglu_gate_fwd(_ptr(y2), _ptr(out), M, C2 // 2, dtype_code(y2.dtype), _stream(y2)), "geglu_gate_fwd")
    return out


def geglu_gate_bwd(y2, dout2):
    _require_device(y2, dout2)
    M, C2 = y2.shape
    dy = torch.empty_like(y2)
    _check(lib().geglu_gate_bwd(_ptr(y2), _ptr(dout2), _ptr(dy), M, C2 // 2, dtype_code(y2.dtype), _stream(y2)),
           "geglu_gate_bwd")
    return dy


_zero_factors = {}
_zero_factors_retired = []


def zero_factor_buffer(n_elems: int, dtype: torch.dtype, device):
    """A zeroed buffer of >= n_elems elements for launches without a rank-r term (a zero [16, Kc] factor tile and a zero
    [Nc, 16] epilogue factor); one per (device, dtype), grown on demand.  None inside a recording when it would have to be
    allocated (never allocate-and-zero under capture); an outgrown buffer stays alive — a recorded hipGraph may still read it."""
    key = (device, dtype)
    z = _zero_factors.get(key)
    if z is None or z.numel() < n_elems:
        if torch.cuda.is_current_stream_capturing():
            return None
        if z is not None:
            _zero_factors_retired.append(z)
        z = _zero_factors[key] = torch.zeros(max(n_elems, 16 * 10240), dtype=dtype, device=device)
    return z


def geglu_linear_bwd(dz2, w2t, y2):
    """Backward of `z = (h·gelu(g)) @ W2ᵀ + b2` w.r.t. y = [h | g] in ONE launch (the gate's backward rides in the epilogue
    of dout = dz·W2): dz2 [M,Nz], w2t = W2ᵀ [F,Nz], y2 [M,2F] → dY [M,2F]; None when the library has no fused kernel."""
    _require_device(dz2, w2t, y2)
    M, Nz = dz2.shape
    F = w2t.shape[0]
    z = zero_factor_buffer(16 * max(Nz, F), dz2.dtype, dz2.device)
    if z is None:
        return None  # (the caller runs the two-launch form)
    dy = torch.empty_like(y2)
    st = lib().geglu_linear_bwd(_ptr(dz2), _ptr(w2t), _ptr(y2), _ptr(dy), _ptr(z), M, Nz, F, dtype_code(dz2.dtype) if dz2.dtype != torch.float32 else 0,
                                _stream(dz2))
    if st == -5:
        return None
    _check(st, "geglu_linear_bwd")
    return dy


def attn_split_heads(x3, heads: int, D: int):
    """x3 [B, N, H·d] contiguous → [B, H, N, D] (zero-padded columns)."""
    _require_device(x3)
    B, N, HD = x3.shape
    d = HD // heads
    out = torch.empty((B, heads, N, D), dtype=x3.dtype, device=x3.device)
    _check(lib().attn_split_heads(_ptr(x3), _ptr(out), B, N, heads, d, D, dtype_code(x3.dtype), _stream(x3)),
           "attn_split_heads")
    return out


def attn_merge_heads(x4, d: int):
    """x4 [B, H, N, D] → [B, N, H·d].  x4 may be a strided view with a contiguous last dim (what the attention core
    returns); anything else is made contiguous first."""
    _require_device(x4)
    B, H, N, D = x4.shape
    vec = 16 // x4.element_size()
    sB, sH, sN, sD = x4.stride()
    if sD != 1 or sB % vec or sH % vec or sN % vec or x4.data_ptr() % 16:
        x4 = x4.contiguous()
        sB, sH, sN, sD = x4.stride()
    out = torch.empty((B, N, H * d), dtype=x4.dtype, device=x4.device)
    _check(lib().attn_merge_heads_strided(_ptr(x4), _ptr(out), B, N, H, d, D, sB, sH, sN, dtype_code(x4.dtype),
                                          _stream(x4)), "attn_merge_heads_strided")
    return out


_attn_supported = {}  # (core, shape, dtype) → bool: a pure function of the library, asked 2–3 times per attention call


def attn_ctx_supported(B: int, Tq: int, Tk: int, H: int, d: int, dtype) -> bool:
    key = (0, B, Tq, Tk, H, d, dtype)
    ok = _attn_supported.get(key)
    if ok is None:
        ok = _attn_supported[key] = (dtype in (torch.float16, torch.bfloat16) and
                                     bool(lib().attn_ctx_supported(B, Tq, Tk, H, d, dtype_code(dtype))))
    return ok


def attn_ctx_fwd(q, k, v, heads: int, scale: float):
    """q [B,Tq,H·d], k/v [B,Tk,H·d] contiguous → softmax(q·kᵀ·scale)·v per head, [B,Tq,H·d]."""
    _require_device(q, k, v)
    B, Tq, HD = q.shape
    Tk = k.shape[1]
    out = torch.empty_like(q)
    _check(lib().attn_ctx_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Tq, Tk, heads, HD // heads, float(scale),
                              dtype_code(q.dtype), _stream(q)), "attn_ctx_fwd")
    return out


def attn_ctx_bwd(q, k, v, dout, heads: int, scale: float):
    """→ (dq, dk, dv), same layouts as the inputs."""
    _require_device(q, k, v, dout)
    B, Tq, HD = q.shape
    Tk = k.shape[1]
    d = HD // heads
    nbytes = lib().attn_ctx_bwd_workspace_bytes(B, Tq, Tk, heads, d)
    if nbytes < 0:
        raise RuntimeError("attn_ctx_bwd: unsupported shape")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _check(lib().attn_ctx_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(dout), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), B, Tq, Tk,
                              heads, d, float(scale), dtype_code(q.dtype), _stream(q)), "attn_ctx_bwd")
    return dq, dk, dv


def attn_ctx_fwd_kv(q, kv, off_k: int, off_v: int, heads: int, scale: float):
    """Cross-attention whose K and V are the column slices [off_k, off_k+H·d) / [off_v, …) of `kv` [B·Tk, ld] (the output
    of a grouped to_k/to_v projection); q [B,Tq,H·d] contiguous → out [B,Tq,H·d]."""
    _require_device(q, kv)
    B, Tq, HD = q.shape
    ld = kv.shape[-1]
    Tk = kv.shape[0] // B
    es = kv.element_size()
    out = torch.empty_like(q)
    _check(lib().attn_ctx_fwd_strided(_ptr(q), kv.data_ptr() + off_k * es, kv.data_ptr() + off_v * es, _ptr(out), ld, B,
                                      Tq, Tk, heads, HD // heads, float(scale), dtype_code(q.dtype), _stream(q)),
           "attn_ctx_fwd_strided")
    return out


def attn_ctx_bwd_kv(q, kv, dkv, off_k: int, off_v: int, dout, heads: int, scale: float):
    """Backward of attn_ctx_fwd_kv: returns dq; dK / dV are WRITTEN into the same column slices of `dkv` [B·Tk, ld]."""
    _require_device(q, kv, dkv, dout)
    B, Tq, HD = q.shape
    ld = kv.shape[-1]
    Tk = kv.shape[0] // B
    d = HD // heads
    es = kv.element_size()
    nbytes = lib().attn_ctx_bwd_workspace_bytes(B, Tq, Tk, heads, d)
    if nbytes < 0:
        raise RuntimeError("attn_ctx_bwd: unsupported shape")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q)
    _check(lib().attn_ctx_bwd_strided(_ptr(q), kv.data_ptr() + off_k * es, kv.data_ptr() + off_v * es, _ptr(dout),
                                      _ptr(dq), dkv.data_ptr() + off_k * es, dkv.data_ptr() + off_v * es, _ptr(ws), ld,
                                      dkv.shape[-1], B, Tq, Tk, heads, d, float(scale), dtype_code(q.dtype), _stream(q)),
           "attn_ctx_bwd_strided")
    return dq


def attn_flash_fwd_qkv(qkv, heads: int, scale: float, want_lse: bool = True):
    """Self-attention on a grouped projection's output qkv [B, T, 3·H·d] (q | k | v column slices) → (o [B,T,H·d], lse)."""
    _require_device(qkv)
    B, T, W = qkv.shape
    HD = W // 3
    es = qkv.element_size()
    out = torch.empty((B, T, HD), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, heads, T), dtype=torch.float32, device=qkv.device) if want_lse else None
    base = qkv.data_ptr()
    _check(lib().attn_flash_fwd_strided(base, base + HD * es, base + 2 * HD * es, _ptr(out), _ptr(lse), W, B, T, T, heads,
                                        HD // heads, float(scale), dtype_code(qkv.dtype), _stream(qkv)),
           "attn_flash_fwd_strided")
    return out, lse


def attn_flash_bwd_qkv(qkv, out, dout, lse, heads: int, scale: float):
    """→ dqkv [B, T, 3·H·d] (dq | dk | dv written as column slices of one buffer: the dY of the grouped projection)."""
    _require_device(qkv, out, dout, lse)
    B, T, W = qkv.shape
    HD = W // 3
    es = qkv.element_size()
    ws = torch.empty(lib().attn_flash_bwd_workspace_bytes(B, T, heads) // 4, dtype=torch.float32, device=qkv.device)
    dqkv = torch.empty_like(qkv)
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    _check(lib().attn_flash_bwd_strided(base, base + HD * es, base + 2 * HD * es, _ptr(out), _ptr(dout), _ptr(lse), dbase,
                                        dbase + HD * es, dbase + 2 * HD * es, _ptr(ws), W, W, B, T, T, heads, HD // heads,
                                        float(scale), dtype_code(qkv.dtype), _stream(qkv)), "attn_flash_bwd_strided")
    return dqkv


def attn_flash_supported(B: int, Tq: int, Tk: int, H: int, d: int, dtype) -> bool:
    key = (1, B, Tq, Tk, H, d, dtype)
    ok = _attn_supported.get(key)
    if ok is None:
        ok = _attn_supported[key] = (dtype in (torch.float16, torch.bfloat16) and
                                     bool(lib().attn_flash_supported(B, Tq, Tk, H, d, dtype_code(dtype))))
    return ok


def attn_flash_fwd(q, k, v, heads: int, scale: float, want_lse: bool = True):
    """q [B,Tq,H·d], k/v [B,Tk,H·d] contiguous → (o [B,Tq,H·d], lse [B,H,Tq] fp32 | None)."""
    _require_device(q, k, v)
    B, Tq, HD = q.shape
    Tk = k.shape[1]
    out = torch.empty_like(q)
    lse = torch.empty((B, heads, Tq), dtype=torch.float32, device=q.device) if want_lse else None
    _check(lib().attn_flash_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, Tq, Tk, heads, HD // heads,
                                float(scale), dtype_code(q.dtype), _stream(q)), "attn_flash_fwd")
    return out, lse


def attn_flash_bwd(q, k, v, out, dout, lse, heads: int, scale: float):
    """→ (dq, dk, dv), same layouts as the inputs."""
    _require_device(q, k, v, out, dout, lse)
    B, Tq, HD = q.shape
    Tk = k.shape[1]
    ws = torch.empty(lib().attn_flash_bwd_workspace_bytes(B, Tq, heads) // 4, dtype=torch.float32, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _check(lib().attn_flash_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dq), _ptr(dk),
                                _ptr(dv), _ptr(ws), B, Tq, Tk, heads, HD // heads, float(scale), dtype_code(q.dtype),
                                _stream(q)), "attn_flash_bwd")
    return dq, dk, dv


def prof_enable(capacity: int) -> None:
    _check(lib().lora_prof_enable(int(capacity)), "lora_prof_enable")


def prof_null_mode(on: bool) -> None:
    """Launch-floor mode: every profiled launch site dispatches an empty kernel of the same shape (timing only)."""
    _check(lib().lora_prof_null_mode(1 if on else 0), "lora_prof_null_mode")


def prof_collect():
    tot = ProfTotals()
    _check(lib().lora_prof_collect(ctypes.byref(tot)), "lora_prof_collect")
    return {
        lib().lora_prof_kernel_name(k).decode(): {"launches": int(tot.launches[k]), "ms": float(tot.ms[k]),
                                                  "bytes": float(tot.bytes[k]), "flops": float(tot.flops[k])}
        for k in range(PROF_KINDS) if tot.launches[k] > 0
    }
