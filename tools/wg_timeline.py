"""Dev tool: where does a workgroup of the fused GEMM spend its time?

Builds a DIAGNOSTIC copy of the library with -DLORA_STAMPS (per-workgroup cycle-counter stamps at the phase
boundaries of lora_gemm_kernel), launches one forward per shape and prints the phase durations (median over
workgroups, in µs of the 100 MHz wall clock / shader cycles) plus the launch ramp (first/last start, last end).
Never used by the product path.  usage: python tools/wg_timeline.py [M K N]..."""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diffusion_finetuning_amd import build_native as bn

OUT = os.path.join(bn.LIB_DIR, "liblora_hip_stamps.so")


def build():
    srcs = [os.path.join(bn.CSRC, s) for s in bn.SOURCES]
    cmd = [bn.HIPCC, *bn.FLAGS, "-DLORA_STAMPS", "-shared", "-o", OUT, *srcs]
    subprocess.run(cmd, check=True)


def main():
    if not os.path.exists(OUT) or "--rebuild" in sys.argv:
        build()
    lib = ctypes.CDLL(OUT)
    vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    lib.lora_linear_fwd.argtypes = [vp] * 9 + [i64, ci, ci, ci, cf, ci, vp]
    lib.lora_pack_factors.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.lora_debug_stamps.argtypes = [vp, ci]
    lib.lora_linear_geglu_fwd.argtypes = [vp] * 8 + [i64, ci, ci, ci, cf, ci, vp]
    gated = "--geglu" in sys.argv  # N = 2·F: the `proj` forward with the gate in its epilogue
    splitk = "--splitk" in sys.argv  # long contractions: the launch cuts K into slices (workspace handed over); LORA_SPLIT_AFFINITY=0|1
    lib.lora_linear_fwd_ws.argtypes = [vp] * 9 + [i64, ci, ci, ci, cf, ci, vp, i64, vp]
    lib.lora_gemm_workspace_bytes.restype = i64
    lib.lora_gemm_workspace_bytes.argtypes = [i64, ci, ci, ci]
    args = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [
        (16384, 320, 320), (4096, 640, 640), (16384, 320, 2560), (16384, 1280, 320), (1024, 1280, 1280)]
    dev = "cuda"
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        a = torch.randn(4, K, device=dev) / 4
        b = torch.randn(N, 4, device=dev) * 0.05
        ap = torch.zeros(2 * 16 * K, device=dev, dtype=torch.float16)
        bp = torch.zeros(2 * 16 * N, device=dev, dtype=torch.float16)
        y = torch.empty(M, N, device=dev, dtype=torch.float16)
        t = torch.empty(M, 4, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        assert lib.lora_pack_factors(a.data_ptr(), b.data_ptr(), ap.data_ptr(), bp.data_ptr(), K, N, 4, 1, st) == 0

        out = torch.empty(M, N // 2, device=dev, dtype=torch.float16)

        ws = None
        if splitk:
            nb = lib.lora_gemm_workspace_bytes(M, K, N, 1)
            assert nb > 0, "the library does not split this shape"
            ws = torch.zeros(nb // 4 + 16, device=dev)

        def launch():
            if splitk:
                assert lib.lora_linear_fwd_ws(x.data_ptr(), w.data_ptr(), None, a.data_ptr(), b.data_ptr(), ap.data_ptr(),
                                              bp.data_ptr(), y.data_ptr(), t.data_ptr(), M, K, N, 4, 1.0, 1, ws.data_ptr(),
                                              ws.numel() * 4, st) == 0
                return
            if gated:
                assert lib.lora_linear_geglu_fwd(x.data_ptr(), w.data_ptr(), None, ap.data_ptr(), bp.data_ptr(), y.data_ptr(),
                                                 out.data_ptr(), t.data_ptr(), M, K, N, 4, 1.0, 1, st) == 0
                return
            rc = lib.lora_linear_fwd(x.data_ptr(), w.data_ptr(), None, a.data_ptr(), b.data_ptr(), ap.data_ptr(),
                                     bp.data_ptr(), y.data_ptr(), t.data_ptr(), M, K, N, 4, 1.0, 1, st)
            assert rc == 0

        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        assert lib.lora_debug_stamps_reset() == 0
        launch()
        torch.cuda.synchronize()
        buf = np.zeros(8192 * 16, dtype=np.uint64)
        assert lib.lora_debug_stamps(buf.ctypes.data, buf.size) == 0
        s = buf.reshape(8192, 16).astype(np.int64)
        live = s[:, 0] > 0
        s = s[live]
        if splitk:
            # slices that were NOT their tile's last arriver leave after the ticket (stamp 14): main loop, slab store + ticket,
            # life; the last arrivers also add the slices and run the epilogue (stamps 5..8)
            last = s[:, 5] > 0
            wall0 = s[:, 0].min()
            life = (s[:, 9] - s[:, 0]) / 100.0
            kk = np.median((s[:, 8] - s[:, 1])[life > 0] / life[life > 0])
            def med(v): return f"median {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}"
            print(f"--- split-K {M}x{K}x{N} (affinity {os.environ.get('LORA_SPLIT_AFFINITY', '0')}): {len(s)} workgroups, {int(last.sum())} last "
                  f"arrivers; kernel span {((s[:, 9] - wall0) / 100.0).max():.2f} us; starts up to {((s[:, 0] - wall0) / 100.0).max():.2f} us")
            print(f"    to first stage landed   {med((s[:, 3] - s[:, 1]) / kk)}")
            print(f"    main loop               {med((s[:, 4] - s[:, 3]) / kk)}")
            print(f"    slab store + ticket     {med((s[:, 14] - s[:, 4]) / kk)}")
            print(f"    life, other slices      {med(life[~last])}")
            print(f"    last arriver: sum of the slices + packed P  {med(((s[:, 5] - s[:, 14]) / kk)[last])}")
            print(f"    last arriver: rank step, C tile, stores     {med(((s[:, 8] - s[:, 5]) / kk)[last])}")
            print(f"    life, last arrivers     {med(life[last])}")
            continue
        n = len(s)
        wall0 = s[:, 0].min()
        start_us = (s[:, 0] - wall0) / 100.0
        end_us = (s[:, 9] - wall0) / 100.0
        cyc = s[:, [1, 12, 13, 2, 3, 4, 5, 6, 7, 8]]
        ph = np.diff(cyc, axis=1)  # cycles per phase
        # calibrate cycles -> µs with the wall clock of the same workgroups
        dur_cyc = (cyc[:, -1] - cyc[:, 0]).astype(float)
        dur_us = (s[:, 9] - s[:, 0]) / 100.0
        k = np.median(dur_cyc[dur_us > 0] / dur_us[dur_us > 0])
        names = ["kernarg+tile index", "addresses+Q dma+bias", "issue stages", "first stage lands", "main loop", "P combine", "LoRA mfma", "C->LDS", "C stores"]
        print(f"--- {M}x{K}x{N}: {n} workgroups, {k:.0f} cycles/us; kernel span {end_us.max():.2f} us "
              f"(starts {start_us.min():.2f}..{np.percentile(start_us, 50):.2f}..{start_us.max():.2f}, "
              f"WG life median {np.median(dur_us):.2f} max {dur_us.max():.2f})")
        for i, nm in enumerate(names):
            v = ph[:, i] / k
            print(f"    {nm:18s} median {np.median(v):6.2f} us   p90 {np.percentile(v, 90):6.2f}   max {v.max():6.2f}")
        xcc = s[:, 11] & 0xF
        cu = s[:, 10]
        print(f"    distinct (xcc,hw_id>>8&0xff) slots: {len(set(zip(xcc.tolist(), ((cu >> 8) & 0xFF).tolist())))}")


main()
