"""Host-side logic of the step harness that needs no GPU: the fixed-lag loss-scale schedule and the collective sequence
of the data-parallel exchange in every launch mode (world 4 over gloo)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_finetuning_amd.trainer import SCHEDULER_NAMES, LoraTrainer, LossScaler, SlabExchange, lr_lambda


@pytest.mark.parametrize("name", SCHEDULER_NAMES)
def test_lr_lambda_inside_torch_lambdalr_gives_the_schedules_the_trainers_ask_for(name):
    """`lr_lambda(name, warm-up, total)` under torch's LambdaLR (the mechanism get_scheduler wraps,
    train_lora_dreambooth.py:737-743): ramp over the warm-up steps, then the named decay; and the two placements of
    `lr_scheduler.step()` the reference has — after the optimizer (train_lora_dreambooth.py:885-886: step k at λ(k)) and
    before it (cli_lora_pti.py:434: step k at λ(k + 1)) — as LoraTrainer._scheduled_lr_factor walks them."""
    warm, total, base = 3, 12, 2e-4
    lam = lr_lambda(name, warm, total, lr_init=base)
    p = torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.AdamW([p], lr=base)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    seen = []
    for k in range(total + 2):
        seen.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    assert all(abs(v - base * lam(k)) < 1e-18 for k, v in enumerate(seen))
    assert [lam(k) for k in range(warm)] == ([1.0] * warm if name == "constant" else [k / warm for k in range(warm)])
    assert lam(warm) == pytest.approx(1.0)
    if name in ("linear", "cosine", "cosine_with_restarts"):
        assert lam(total) == pytest.approx(0.0, abs=1e-12) and all(lam(k) >= lam(k + 1) for k in range(warm, total))
    if name == "linear":
        assert lam(warm + 3) == pytest.approx(1.0 - 3.0 / (total - warm))
    for first in (False, True):
        tr_ = LoraTrainer.__new__(LoraTrainer)  # the schedule bookkeeping alone (no device)
        tr_.lr_lambda, tr_.scheduler_steps_first, tr_.scheduler_epoch, tr_.scheduler_steps_per_call = lam, first, 0, 1
        got = [tr_._scheduled_lr_factor() for _ in range(5)]
        assert got == [lam(k + 1 if first else k) for k in range(5)] and tr_.scheduler_epoch == 5
    # the dreambooth route under accelerate on N processes — AcceleratedScheduler steps the wrapped scheduler N times
    # per call (split_batches=False): optimizer step k runs at λ(N·k); restated here with torch's LambdaLR stepped N times
    for world in (2, 8):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([w], lr=base)
        sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
        want = []
        for _ in range(4):
            want.append(opt.param_groups[0]["lr"] / base)
            opt.step()
            for _ in range(world):
                sch.step()
        tr_ = LoraTrainer.__new__(LoraTrainer)
        tr_.lr_lambda, tr_.scheduler_steps_first, tr_.scheduler_epoch, tr_.scheduler_steps_per_call = lam, False, 0, world
        got = [tr_._scheduled_lr_factor() for _ in range(4)]
        assert got == pytest.approx(want, abs=1e-15) and tr_.scheduler_epoch == 4 * world


def test_lr_lambda_rejects_what_get_scheduler_rejects():
    with pytest.raises(ValueError):
        lr_lambda("exponential")
    with pytest.raises(ValueError):
        lr_lambda("linear", 0, None)  # needs num_training_steps


def test_loss_scaler_applies_flags_at_a_fixed_lag():
    """The flag of step k changes the scale at the start of step k + LAG, never earlier or later: that is what keeps
    data-parallel ranks on one scale (GradScaler semantics otherwise: halve on overflow, double after a clean interval,
    never above the initial value)."""
    sc = LossScaler(1024.0, growth_interval=3)
    flags = [1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]
    seen = []
    for k, f in enumerate(flags):
        changed = sc.begin_step()
        seen.append(sc.scale)
        assert changed == (k >= 1 and seen[k] != seen[k - 1])
        sc.watch(lambda f=f: float(f))
    # step k uses the decisions of steps <= k-2
    assert seen == [1024, 1024, 512, 256, 256, 256, 512, 512, 256, 256, 256, 512]
    assert LossScaler.LAG == 2


def test_loss_scaler_never_reads_a_flag_early():
    sc = LossScaler(8.0)
    reads = []
    for k in range(5):
        sc.begin_step()
        sc.watch(lambda k=k: reads.append(k) or 0.0)
        assert reads == list(range(max(0, k - 1)))  # at step k only the flags of steps <= k-2 have been awaited


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeTrainer:
    """The exchange calls of LoraTrainer._step_eager / _step_graph, nothing else (trainer.py)."""

    def __init__(self, n, capture_graph, early_range, capture_ok=True):
        self.grads = torch.ones(n)
        self.exchange = SlabExchange(self.grads, n)
        self.exchange.single = capture_graph          # as LoraTrainer.__init__
        self.exchange.early_range = early_range       # as _install_bucket_hook
        self.capture_graph, self.capture_ok = capture_graph, capture_ok

    def step(self, recordable=True):
        if self.capture_graph and recordable:
            if not self.capture_ok:                   # recording failed on this rank: host-launched from now on
                self.capture_graph = False
                return self._eager()
            return self.exchange.finish()             # replay, then the one all-reduce
        return self._eager()

    def _eager(self):
        self.exchange.arm()
        self.exchange.launch_early()                  # the mid-block backward hook
        self.exchange.finish()


def _sequence_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    real = dist.all_reduce

    def logged(t, *a, **kw):
        log.append(int(t.numel()))
        return real(t, *a, **kw)

    dist.all_reduce = logged
    n = 1000
    results = {}
    scenarios = {
        "host-launched, two buckets": dict(capture_graph=False, capture_ok=True),
        "recorded on every rank": dict(capture_graph=True, capture_ok=True),
        "recording fails on rank 1 only": dict(capture_graph=True, capture_ok=rank != 1),
    }
    for name, kw in scenarios.items():
        del log[:]
        t = _FakeTrainer(n, early_range=(300, 900), **kw)
        for recordable in (True, True, False, True):  # (a step that cannot be recorded in between: text encoder without ids)
            t.step(recordable)
        results[name] = list(log)
        assert torch.equal(t.grads, torch.full((n,), float(world) ** 4)), name  # four SUM all-reduces over every element
    gathered = [None] * world
    dist.all_gather_object(gathered, results)
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_collective_sequence_is_the_same_on_every_rank_in_every_launch_mode():
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sequence_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    for name in gathered[0]:
        seqs = [g[name] for g in gathered]
        assert all(s == seqs[0] for s in seqs), (name, seqs)
    assert gathered[0]["host-launched, two buckets"] == [600, 300, 100] * 4      # early [300,900), then [0,300), [900,1000)
    assert gathered[0]["recorded on every rank"] == [1000] * 4
    assert gathered[0]["recording fails on rank 1 only"] == [1000] * 4            # also on the rank that fell back
