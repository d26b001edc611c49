"""Two data-parallel ranks driving the real HIP path on ONE GPU (gloo transports the slab; RCCL itself is covered by the
1-rank test in test_gpu_parity.py): 2 ranks × batch 2 must equal 1 rank × batch 4 — broadcast of the initial slab,
early [up|mid] bucket from the backward hook, per-range folding of the partial sums, 1/world in the optimizer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, batch, out, graph=False, early_groups=False):
    import itertools

    import diffusion_finetuning_amd as dfa
    from diffusion_finetuning_amd import trainer as tr
    from oracle import lora_oracle as orc
    from tests.conftest import build_tiny_unet

    torch.set_num_threads(2)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ctx_dim = 32
    if early_groups:  # three levels, 64-wide context: two cross-attentions in the down blocks, four in [up|mid] — groupable
        from diffusion_finetuning_amd.attention import set_use_memory_efficient_attention_xformers
        from harness.unet import UNet2DConditionModel, tiny_config

        torch.manual_seed(3)
        unet = UNet2DConditionModel(tiny_config(64, 64, 3))
        unet.requires_grad_(False)
        unet = unet.to(dev)
        ctx_dim = 64
    else:
        unet = build_tiny_unet(seed=3).to(dev)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    g = torch.Generator().manual_seed(11 + rank)  # ranks start DIFFERENT: the broadcast must fix that
    with torch.no_grad():
        for i, p in enumerate(plist):
            if i % 2 == 0:
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(dev))
    if early_groups:
        # grouped projections AND an early bucket: the down blocks' cross-attentions get a K/V group of their own, so the
        # [up|mid] range of the slab is final — launched, folded, all-reduced — while backward is still in the down blocks
        set_use_memory_efficient_attention_xformers(unet, True)
        trainer = tr.LoraTrainer(unet, lr=1e-3, group_projections=True, early_bucket=True)
        assert len(trainer.slab.ctx_groups) == 2 and len(trainer.slab.qkv_groups) > 0
        names = {id(m): n for n, m in unet.named_modules()}
        sides = [{names[id(m)].split(".")[0] for m in g.modules} for g in trainer.slab.ctx_groups]
        assert sorted(map(sorted, sides)) == [["down_blocks"], ["mid_block", "up_blocks"]]
        sent = []
        send = trainer.exchange._send
        trainer.exchange._send = lambda a, b: (sent.append((a, b)), send(a, b))[1]
    else:
        trainer = tr.LoraTrainer(unet, lr=1e-3, group_projections=False, capture_graph=graph)  # ungrouped: the bucketed exchange is in play
    if world > 1:
        assert trainer.exchange.active and trainer.exchange.early_range is not None
        assert trainer.exchange.single == graph  # a requested recording pins the exchange to ONE all-reduce per step
    for step in range(3):
        latents, noise, ts, ctx = orc.synthetic_batch(step, batch * world, 8, 6, ctx_dim)
        sl = slice(rank * batch, (rank + 1) * batch)
        trainer.step(latents[sl].to(dev), noise[sl].to(dev), ts[sl].to(dev), ctx[sl].to(dev))
    if early_groups and world > 1:  # every step: the early [up|mid] bucket first (from the hook), then the rest of the slab
        a0, b1 = trainer.exchange.early_range
        assert b1 == trainer.slab.numel and sent == [(a0, b1), (0, a0), (b1, b1)] * 3, sent
    if graph:
        assert trainer._graph is not None  # the steps really were replays
    state = trainer.slab.params[: trainer.slab.numel].cpu()
    if world > 1:
        gathered = [torch.zeros_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        assert all(torch.equal(gathered[0], t) for t in gathered)  # replicas stay identical
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out.put(state.numpy().copy())  # by value: the producer may exit before the consumer reads


def test_two_ranks_equal_one_rank_with_double_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    two = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single = ctx.Process(target=_run, args=(0, 1, port, 4, q))
    single.start()
    one = torch.from_numpy(q.get(timeout=300))
    single.join(timeout=300)
    assert single.exitcode == 0
    err = ((two - one).norm() / one.norm()).item()
    assert err < 1e-4, err


def test_two_ranks_with_grouped_projections_and_an_early_bucket_equal_one_rank_with_double_batch():
    """LoraTrainer(early_bucket=True): the context K/V group is cut by block range, so WITH grouped
    projections the [up|mid] bucket is exchanged from the mid block's backward hook while the down blocks still run backward
    (reference: DDP's buckets fire inside backward, train_lora_dreambooth.py:744-757,877) — 2 ranks × batch 2 equal 1 rank
    × batch 4, replicas identical, the bucket sequence is [early, head of the slab] on every step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, 2, q, False, True)) for r in range(2)]
    for p in procs:
        p.start()
    two = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single = ctx.Process(target=_run, args=(0, 1, port, 4, q, False, True))
    single.start()
    one = torch.from_numpy(q.get(timeout=300))
    single.join(timeout=300)
    assert single.exitcode == 0
    err = ((two - one).norm() / one.norm()).item()
    assert err < 1e-4, err


def test_two_ranks_replaying_recorded_steps_equal_one_rank_with_double_batch():
    """The same equivalence with `capture_graph=True` on both ranks: forward + backward replayed from each rank's hipGraph
    with a process group of two alive, the slab exchanged in one all-reduce between replay and optimizer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, 2, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    two = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single = ctx.Process(target=_run, args=(0, 1, port, 4, q, False))
    single.start()
    one = torch.from_numpy(q.get(timeout=300))
    single.join(timeout=300)
    assert single.exitcode == 0
    err = ((two - one).norm() / one.norm()).item()
    assert err < 1e-4, err


def _run_overflow(rank, world, port, out):
    """fp16, 2 ranks: rank 1's batch is poisoned with an inf at step 2 — after the SUM all-reduce BOTH ranks see a
    non-finite slab, skip the step, and must lower their loss scale at the SAME later step (LossScaler's fixed lag); a rank
    that moved one step earlier or later would mix gradients scaled by S and S/2 and the replicas would drift apart."""
    import itertools

    import diffusion_finetuning_amd as dfa
    from diffusion_finetuning_amd import trainer as tr
    from oracle import lora_oracle as orc
    from tests.conftest import build_tiny_unet

    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    unet = build_tiny_unet(seed=3).to(dev).half()
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for i, p in enumerate(itertools.chain(*params)):
            if i % 2 == 0:
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(dev))
    trainer = tr.LoraTrainer(unet, lr=1e-3, group_projections=False, loss_scale=256.0)
    scales = []
    for step in range(8):
        latents, noise, ts, ctx = orc.synthetic_batch(step, 2 * world, 8, 6, 32)
        sl = slice(rank * 2, (rank + 1) * 2)
        lat = latents[sl].clone()
        if step == 2 and rank == 1:
            lat[0, 0, 0, 0] = float("inf")
        if rank == 1:
            torch.cuda.synchronize()  # the ranks' host/GPU timing differs on purpose: rank 1 always sees its flag copy landed
        trainer.step(lat.to(dev), noise[sl].to(dev), ts[sl].to(dev), ctx[sl].to(dev))
        if rank == 1:
            torch.cuda.synchronize()
        scales.append(trainer.loss_scale)
    state = trainer.slab.params[: trainer.slab.numel].cpu()
    assert torch.isfinite(state).all()
    assert trainer.opt.skipped_steps() == 1 and trainer.opt.applied_steps() == 7
    gathered = [None] * world
    dist.all_gather_object(gathered, (scales, state.numpy().copy()))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_back_the_loss_scale_off_at_the_same_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_overflow, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    (s0, p0), (s1, p1) = gathered
    assert s0 == s1, (s0, s1)                                   # the same scale at every step on both ranks
    assert s0 == [256.0] * 4 + [128.0] * 4, s0                  # step 2 overflowed → applied at the start of step 2 + LAG
    assert (p0 == p1).all()                                     # and the replicas stayed bit-identical


def _run_pti(rank, world, port, batch, out):
    """The PTI tuning step (trainable token table, cli_lora_pti.py:706-722) under data parallelism: every rank runs the text
    encoder on ITS captions; the table gradient is exchanged as (ids, gradient rows) — an all-gather in rank order followed by
    the ordered per-token sum — not as a 200-MB dense all-reduce."""
    import itertools
    import json

    import diffusion_finetuning_amd as dfa
    from diffusion_finetuning_amd import trainer as tr
    from diffusion_finetuning_amd.attention import set_use_memory_efficient_attention_xformers
    from oracle import lora_oracle as orc
    from tests.conftest import _load
    from tests.test_oracle_golden import build_pti_models

    torch.set_num_threads(2)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    t, meta = _load("pti_trajectory.safetensors")
    cfg = json.loads(meta["cfg"])
    unet, te = build_pti_models(t, cfg, dev, torch.float32)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    with torch.no_grad():
        for p, v in zip(plist, torch.split(t["lora.init"], [q.numel() for q in plist])):
            p.copy_(v.view(p.shape).to(dev))
    set_use_memory_efficient_attention_xformers(unet, True)
    orc.freeze_all_but_token_embeddings(te)
    trainer = tr.LoraTrainer(unet, te, lr=1e-3, lr_embed=5e-3, weight_decay=1e-3, v_prediction=True)
    for step in range(3):
        latents, noise, ts, _ = orc.synthetic_batch(step, batch * world, 8, cfg["ctx_len"], cfg["hidden"], t_max=800)
        ids = orc.synthetic_token_ids(step, batch * world, cfg["ctx_len"], cfg["vocab"])
        sl = slice(rank * batch, (rank + 1) * batch)
        trainer.step(latents[sl].to(dev), noise[sl].to(dev), ts[sl].to(dev), input_ids=ids[sl].to(dev))
    a, b = trainer.token_table.range
    state = torch.cat([trainer.slab.params[: trainer.slab.numel], trainer.slab.params[a:b]]).cpu()
    if world > 1:
        gathered = [torch.zeros_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        assert all(torch.equal(gathered[0], g_) for g_ in gathered)  # replicas — token table included — stay BIT-identical
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out.put((state.numpy().copy(), trainer.slab.numel))


def test_two_ranks_train_the_token_table_like_one_rank_with_double_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_pti, args=(r, 2, port, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    two, n = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single = ctx.Process(target=_run_pti, args=(0, 1, port, 2, q))
    single.start()
    one, _ = q.get(timeout=300)
    single.join(timeout=300)
    assert single.exitcode == 0
    two, one = torch.from_numpy(two), torch.from_numpy(one)
    assert ((two[:n] - one[:n]).norm() / one[:n].norm()).item() < 1e-4
    assert ((two[n:] - one[n:]).norm() / one[n:].norm()).item() < 1e-4
