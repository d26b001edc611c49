"""BASELINE config 5's second half — cli_lora_pti.py's tuning phase with continue_inversion (the default, :528): the token
embedding table of the text encoder trains next to the UNet's LoRA factors (:706-722), the step runs the text encoder itself
(:199-206), every attn2 to_k/to_v therefore produces a dX into the context, timesteps are drawn below int(1000·0.8) (:444),
v-prediction target (:217-218).  Checked through the C-ABI against float64 math, against the trajectory the reference's own
modules produced (tests/golden/pti_trajectory.safetensors) and, at full SD2.1-768 size, against the CPU oracle."""
import itertools
import json

import pytest
import torch

import diffusion_finetuning_amd as dfa
from diffusion_finetuning_amd import _native as nat
from diffusion_finetuning_amd import trainer as tr
from diffusion_finetuning_amd.attention import set_use_hip_geglu, set_use_memory_efficient_attention_xformers
from oracle import lora_oracle as orc
from tests.test_oracle_golden import build_pti_models

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_token_rows_gather_and_ordered_row_sum(dtype):
    """embed_rows_fwd = table[ids] cast; embed_rows_bwd = per-token sum of the gradient rows in position order — against
    float64 index_add (the sum of ≤ 60 rows of 16-bit values is exact in fp32 up to rounding: 1e-6), rows of absent tokens
    untouched, accumulate on top of an earlier pass, bit-reproducible run to run."""
    g = torch.Generator().manual_seed(3)
    V, D, n = 500, 1024, 2 * 77
    table = torch.randn(V, D, generator=g).to(DEV)
    ids = torch.randint(0, V, (2, 77), generator=g)
    ids[:, 40:] = 7            # padding: one token in 74 positions
    ids[0, 3] = ids[1, 5] = 11  # a token that occurs in both rows
    ids = ids.to(DEV)
    rows = nat.embed_rows_fwd(table, ids, dtype)
    assert rows.shape == (2, 77, D) and rows.dtype == dtype
    assert torch.equal(rows, table[ids].to(dtype))
    d = torch.randn(n, D, generator=g).to(dtype).to(DEV)
    grad = torch.full((V, D), 3.0, device=DEV)
    nat.embed_rows_bwd(d, ids.reshape(-1), grad)
    want = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.reshape(-1).cpu(), d.double().cpu())
    hit = torch.zeros(V, dtype=torch.bool)
    hit[ids.reshape(-1).cpu()] = True
    assert (grad[hit.to(DEV)].double().cpu() - want[hit]).abs().max() < 2e-5 * want[hit].abs().max()
    assert torch.equal(grad[(~hit).to(DEV)], torch.full_like(grad[(~hit).to(DEV)], 3.0))  # absent tokens: untouched
    again = torch.full((V, D), 3.0, device=DEV)
    nat.embed_rows_bwd(d, ids.reshape(-1), again)
    assert torch.equal(again, grad)  # one owner per token, fixed order
    nat.embed_rows_bwd(d, ids.reshape(-1), again, accumulate=True)
    assert (again[hit.to(DEV)].double().cpu() - 2 * want[hit]).abs().max() < 4e-5 * want[hit].abs().max()


def _tiny_trainer(t, cfg, dtype, graph=False, grouped=True, hook=True):
    unet, te = build_pti_models(t, cfg, DEV, dtype)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    with torch.no_grad():
        for p, v in zip(plist, torch.split(t["lora.init"], [q.numel() for q in plist])):
            p.copy_(v.view(p.shape).to(DEV))
    if hook:
        set_use_memory_efficient_attention_xformers(unet, True)
    orc.freeze_all_but_token_embeddings(te)  # what cli_lora_pti.py:704-722 leaves trainable
    trainer = tr.LoraTrainer(unet, te, lr=cfg["lr_unet"], lr_embed=cfg["lr_embed"], weight_decay=cfg["weight_decay"],
                             v_prediction=cfg["v_prediction"], capture_graph=graph, group_projections=grouped)
    assert trainer.token_table is not None and trainer.trains_text_encoder
    return trainer, unet, te


def _run_tiny(t, cfg, trainer, steps=None):
    losses = []
    for s in range(steps or cfg["steps"]):
        lat, noise, ts, _ = orc.synthetic_batch(s, cfg["batch"], cfg["latent_hw"], cfg["ctx_len"], cfg["hidden"],
                                                t_max=int(1000 * cfg["t_multiplier"]))
        losses.append(trainer.step(lat.to(DEV), noise.to(DEV), ts.to(DEV), input_ids=t["ids"][s].to(DEV)))
    return torch.stack(losses).reshape(-1).cpu()


@pytest.mark.parametrize("graph", [False, True])
def test_pti_tuning_trajectory_matches_the_reference_produced_one(golden_pti, relerr, graph):
    """fp32, 8 steps: LoRA factors AND token table within 1e-3 of the trajectory produced by the reference's LoRA modules +
    torch.optim.AdamW (two param groups, one weight decay) + clip_grad_norm_ over both models — host-launched and as a
    recorded step (text encoder forward/backward, the gather and the stashed gradient rows inside the recording)."""
    t, meta = golden_pti
    cfg = json.loads(meta["cfg"])
    trainer, unet, te = _tiny_trainer(t, cfg, torch.float32, graph=graph)
    losses = _run_tiny(t, cfg, trainer)
    if graph:
        assert trainer._graph is not None
    table = te.get_input_embeddings().weight.detach().cpu()
    assert relerr(losses, t["losses"]) < 1e-3, relerr(losses, t["losses"])
    assert relerr(tr.flat_lora_state(unet).cpu(), t["lora.final"]) < 1e-3
    assert relerr(tr.flat_lora_state(unet).cpu() - t["lora.init"], t["lora.final"] - t["lora.init"]) < 2e-2
    assert relerr(table, t["table.final"]) < 1e-3
    assert relerr(table - t["table.init"], t["table.final"] - t["table.init"]) < 2e-2, \
        relerr(table - t["table.init"], t["table.final"] - t["table.init"])
    # the Parameter the caller holds IS the trained table (save_all reads it: cli_lora_pti.py:461-470)
    assert te.get_input_embeddings().weight.data_ptr() == trainer.slab.params[trainer.token_table.range[0]:].data_ptr()


@pytest.mark.parametrize("graph", [False, True])
def test_pti_tuning_trajectory_under_the_default_linear_schedule(golden_pti, golden_pti_linear, relerr, graph):
    """perform_tuning's default schedule — get_scheduler("linear", 0 warm-up steps, max_train_steps_tuning), stepped BEFORE
    every batch (cli_lora_pti.py:434,534-535,746-751): the fused trainer's `lr_scheduler` / `max_train_steps` /
    `scheduler_steps_first` against the trajectory torch's LambdaLR + AdamW produced around the reference's LoRA modules
    (tests/golden/pti_trajectory_linear.safetensors), fp32: state within 1e-5, the accumulated update within 2e-2 (the
    constant-rate test's bound), the logged learning rates equal to the fixture's."""
    t, meta = golden_pti
    lin, lmeta = golden_pti_linear
    cfg, sch = json.loads(meta["cfg"]), json.loads(lmeta["schedule"])
    unet, te = build_pti_models(t, cfg, DEV, torch.float32)
    params, _ = dfa.inject_trainable_lora(unet, r=4)
    plist = list(itertools.chain(*params))
    with torch.no_grad():
        for p, v in zip(plist, torch.split(t["lora.init"], [q.numel() for q in plist])):
            p.copy_(v.view(p.shape).to(DEV))
    set_use_memory_efficient_attention_xformers(unet, True)
    orc.freeze_all_but_token_embeddings(te)
    trainer = tr.LoraTrainer(unet, te, lr=cfg["lr_unet"], lr_embed=cfg["lr_embed"], weight_decay=cfg["weight_decay"],
                             v_prediction=cfg["v_prediction"], capture_graph=graph, lr_scheduler=sch["name"],
                             lr_warmup_steps=sch["num_warmup_steps"], max_train_steps=sch["num_training_steps"],
                             scheduler_steps_first=True)
    losses = []
    for s in range(cfg["steps"]):
        lat, noise, ts, _ = orc.synthetic_batch(s, cfg["batch"], cfg["latent_hw"], cfg["ctx_len"], cfg["hidden"],
                                                t_max=int(1000 * cfg["t_multiplier"]))
        losses.append(trainer.step(lat.to(DEV), noise.to(DEV), ts.to(DEV), input_ids=t["ids"][s].to(DEV)))
        assert max(abs(a - b) for a, b in zip(trainer.get_last_lr(), lin["lrs"][s].tolist())) < 1e-12
    losses = torch.stack(losses).reshape(-1).cpu()
    table = te.get_input_embeddings().weight.detach().cpu()
    assert relerr(losses, lin["losses"]) < 1e-4, relerr(losses, lin["losses"])
    assert relerr(tr.flat_lora_state(unet).cpu(), lin["lora.final"]) < 1e-5, relerr(tr.flat_lora_state(unet).cpu(), lin["lora.final"])
    assert relerr(table, lin["table.final"]) < 1e-5, relerr(table, lin["table.final"])
    assert relerr(tr.flat_lora_state(unet).cpu() - t["lora.init"], lin["lora.final"] - t["lora.init"]) < 2e-2
    assert relerr(table - t["table.init"], lin["table.final"] - t["table.init"]) < 2e-2
    # the constant-rate trajectory is somewhere else entirely
    assert relerr(tr.flat_lora_state(unet).cpu() - t["lora.init"], t["lora.final"] - t["lora.init"]) > 0.2


def test_replay_after_a_host_launched_step_still_sums_the_recordings_gradient_rows(golden_pti, relerr):
    """A recorded step's (ids, gradient rows) buffers belong to the RECORDING: a host-launched step in between (the caller
    toggles `capture_graph`, or hands over a step that is not recordable) resets the table's own list, and the next replay
    must still sum the rows its kernels just wrote — graph, eager, graph, graph … equals the all-eager trajectory."""
    t, meta = golden_pti
    cfg = json.loads(meta["cfg"])
    want_tr, want_unet, want_te = _tiny_trainer(t, cfg, torch.float32, graph=False)
    want_losses = _run_tiny(t, cfg, want_tr, steps=6)
    trainer, unet, te = _tiny_trainer(t, cfg, torch.float32, graph=True)
    losses = []
    for s in range(6):
        trainer.capture_graph = s != 2  # steps 0, 1 replayed; 2 host-launched; 3.. replayed from the SAME recording
        lat, noise, ts, _ = orc.synthetic_batch(s, cfg["batch"], cfg["latent_hw"], cfg["ctx_len"], cfg["hidden"],
                                                t_max=int(1000 * cfg["t_multiplier"]))
        losses.append(trainer.step(lat.to(DEV), noise.to(DEV), ts.to(DEV), input_ids=t["ids"][s].to(DEV)))
        if s == 1:
            recording = trainer._graph["graph"]
    assert trainer._graph["graph"] is recording  # not re-recorded: the replay after the eager step is the old recording
    losses = torch.stack(losses).reshape(-1).cpu()
    assert relerr(losses, want_losses) < 1e-5, relerr(losses, want_losses)
    got, want = te.get_input_embeddings().weight.detach(), want_te.get_input_embeddings().weight.detach()
    assert relerr(got - t["table.init"].to(DEV), want - t["table.init"].to(DEV)) < 1e-4, \
        relerr(got - t["table.init"].to(DEV), want - t["table.init"].to(DEV))
    assert relerr(tr.flat_lora_state(unet), tr.flat_lora_state(want_unet)) < 1e-5


def test_grouped_context_projection_produces_the_context_gradient(relerr):
    """With a context that carries a gradient the attn2 to_k/to_v of all blocks still run as ONE forward launch, and their dX
    comes from one launch over the concatenated contraction (groups._CtxProjFn.backward) — the f16 trajectory, token table
    included, follows the per-layer one (2·blocks dX launches + their accumulations) and the fp32 CPU oracle's.  (A model
    whose context width is a multiple of 64: the tiny fixture model's 32-wide context is not groupable.)"""
    from harness.unet import UNet2DConditionModel, tiny_config
    from transformers import CLIPTextConfig, CLIPTextModel

    vocab, L, steps, batch = 80, 8, 4, 2
    ccfg = CLIPTextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=vocab,
                          max_position_embeddings=L, bos_token_id=1, eos_token_id=2, pad_token_id=0)

    def make():
        torch.manual_seed(3)
        u = UNet2DConditionModel(tiny_config(64, 64, 2))
        u.requires_grad_(False)
        torch.manual_seed(4)
        e = CLIPTextModel(ccfg)
        return u, e

    def warm(params):
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for i, p in enumerate(params):
                if i % 2 == 0:
                    p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))

    ref_unet, ref_te = make()
    ref_params, _ = orc.inject(ref_unet, r=4)
    warm(ref_params)
    ref_table = orc.freeze_all_but_token_embeddings(ref_te)
    table_init = ref_table.detach().clone()
    ref_losses = orc.pti_tuning_steps(ref_unet, ref_te, ref_params, steps, batch, 8, L, vocab, lr_unet=1e-3, lr_embed=5e-3)

    def train(grouped):
        unet, te = make()
        unet, te = unet.to(DEV).half(), te.to(DEV).half()
        params, _ = dfa.inject_trainable_lora(unet, r=4)
        warm(list(itertools.chain(*params)))
        with torch.no_grad():
            te.get_input_embeddings().weight.data = table_init.to(DEV)  # fp32 master, not the f16-rounded module copy
        set_use_memory_efficient_attention_xformers(unet, True)
        orc.freeze_all_but_token_embeddings(te)
        trainer = tr.LoraTrainer(unet, te, lr=1e-3, lr_embed=5e-3, weight_decay=1e-3, v_prediction=True, group_projections=grouped)
        losses, g1 = [], None
        for s_ in range(steps):
            lat, noise, ts, _ = orc.synthetic_batch(s_, batch, 8, L, 64, t_max=800)
            ids = orc.synthetic_token_ids(s_, batch, L, vocab)
            losses.append(trainer.step(lat.to(DEV), noise.to(DEV), ts.to(DEV), input_ids=ids.to(DEV)))
            if s_ == 0:  # the table gradient of the first step: what the context's dX turned into, before Adam touches it
                a, b = trainer.token_table.range
                g1 = trainer.slab.grads[a:b].view(vocab, 64).cpu().clone()
        return (trainer, tr.flat_lora_state(unet).cpu(), te.get_input_embeddings().weight.detach().float().cpu(),
                torch.stack(losses).reshape(-1).cpu(), g1)

    tg, lora_g, tab_g, lg, g1_g = train(True)
    grp = tg.slab.ctx_groups[0]
    assert grp._pass is not None and grp._pass.consumers == len(grp.modules) == 4  # the group ran, context gradient and all
    tu, lora_u, tab_u, lu, g1_u = train(False)
    assert not tu.slab.ctx_groups
    # the gradient that went through the grouped dX launch against the per-layer launches (+ accumulations), and against the
    # oracle's (the oracle's is clipped: compare directions)
    assert relerr(g1_g, g1_u) < 1e-2, relerr(g1_g, g1_u)
    ref1_u, ref1_te = make()
    ref1_params, _ = orc.inject(ref1_u, r=4)
    warm(ref1_params)
    ref1_table = orc.freeze_all_but_token_embeddings(ref1_te)
    orc.pti_tuning_steps(ref1_u, ref1_te, ref1_params, 1, batch, 8, L, vocab, lr_unet=1e-3, lr_embed=5e-3)
    assert relerr(g1_g / g1_g.norm(), ref1_table.grad / ref1_table.grad.norm()) < 2e-2
    assert relerr(lg, lu) < 2e-3, relerr(lg, lu)
    assert relerr(lora_g, lora_u) < 2e-3
    assert relerr(lg, torch.tensor(ref_losses)) < 5e-3
    assert relerr(lora_g, orc.flat_params(ref_params)) < 5e-3
    # the table after 4 Adam steps at lr 5e-3 (≈ sign steps: elements whose gradient is within f16 noise of zero flip)
    assert relerr(tab_g - table_init, tab_u - table_init) < 0.15, relerr(tab_g - table_init, tab_u - table_init)
    assert relerr(tab_g - table_init, ref_table.detach() - table_init) < 0.15, relerr(tab_g - table_init, ref_table.detach() - table_init)


def test_full_size_cfg5_pti_step_with_token_embeddings_vs_cpu_oracle(relerr):
    """BASELINE config 5 as cli_lora_pti.py runs it, at full size: SD2.1-768-shaped UNet (LoRA r = 16, 96×96 latents, batch 1)
    + an OpenCLIP-H-shaped text encoder (hidden 1024, 23 layers, 16 heads, MLP 4096, vocabulary 49408, 77 positions; random
    init) whose token table trains; the step runs the encoder from token ids, attn2 to_k/to_v produce dX at ctx 1024 / r = 16,
    timestep drawn below 800, v-prediction.  ONE f16 step against the fp32 CPU oracle (oracle.pti_tuning_steps): loss, the
    direction of the LoRA gradient slab and of the table-gradient rows of the tokens that occurred, the update of both."""
    import bench
    from harness.unet import UNet2DConditionModel, sd21_768_config
    from transformers import CLIPTextConfig, CLIPTextModel

    torch.set_num_threads(bench.usable_cpus())
    vocab, L = 49408, 77
    ccfg = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                          vocab_size=vocab, max_position_embeddings=L, bos_token_id=49406, eos_token_id=49407, pad_token_id=0,
                          hidden_act="gelu")

    def make():
        torch.manual_seed(0)
        u = UNet2DConditionModel(sd21_768_config())
        u.requires_grad_(False)
        torch.manual_seed(2)
        e = CLIPTextModel(ccfg)
        e.requires_grad_(False)
        return u, e

    lr_u, lr_e, wd = 1e-4, 5e-4, 1e-3   # cli_lora_pti.py:525-537 defaults (lr_unet, learning_rate_ti, weight_decay_lora)
    ref_unet, ref_te = make()
    ref_params, _ = orc.inject(ref_unet, r=16)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for i, p in enumerate(ref_params):
            if i % 2 == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.01)
    init_state = orc.flat_params(ref_params).clone()
    u_state = {k: v.clone() for k, v in ref_unet.state_dict().items() if "lora_" not in k}
    t_state = {k: v.clone() for k, v in ref_te.state_dict().items()}
    table = orc.freeze_all_but_token_embeddings(ref_te)
    table_init = table.detach().clone()
    ids = orc.synthetic_token_ids(0, 1, L, vocab, bos=49406, eos=49407)
    ref_losses = orc.pti_tuning_steps(ref_unet, ref_te, ref_params, 1, 1, 96, L, vocab, lr_unet=lr_u, lr_embed=lr_e,
                                      weight_decay=wd, bos=49406, eos=49407)
    ref_grad = torch.cat([p.grad.reshape(-1) for p in ref_params])
    ref_tgrad = table.grad.clone()
    want, want_table = orc.flat_params(ref_params), table.detach().clone()
    del ref_unet, ref_te

    unet, te = make()
    unet.load_state_dict({k.replace(".linear.", "."): v for k, v in u_state.items()})
    te.load_state_dict(t_state)
    unet, te = unet.half().to(DEV), te.half().to(DEV)
    params, _ = dfa.inject_trainable_lora(unet, r=16)
    plist = list(itertools.chain(*params))
    with torch.no_grad():
        for p, rp in zip(plist, torch.split(init_state, [q.numel() for q in plist])):
            p.copy_(rp.view(p.shape).to(DEV))
        te.get_input_embeddings().weight.data = table_init.to(DEV)  # the fp32 master (the module copy above was rounded to f16)
    set_use_memory_efficient_attention_xformers(unet, True)
    set_use_hip_geglu(unet, True)
    orc.freeze_all_but_token_embeddings(te)
    trainer = tr.LoraTrainer(unet, te, lr=lr_u, lr_embed=lr_e, weight_decay=wd, v_prediction=True)
    assert trainer.token_table is not None and trainer.slab.ctx_groups[0].G == 32 and len(trainer.slab.qkv_groups) == 16
    lat, noise, ts, _ = orc.synthetic_batch(0, 1, 96, L, 1024, t_max=800)
    loss = trainer.step(lat.to(DEV), noise.to(DEV), ts.to(DEV), input_ids=ids.to(DEV)).item()
    assert not trainer.opt.overflowed()
    assert trainer.slab.ctx_groups[0]._pass is not None  # the grouped projection ran — with a context that wants a gradient
    assert abs(loss - ref_losses[0]) / abs(ref_losses[0]) < 3e-3, (loss, ref_losses[0])
    # LoRA slab: direction of the gradient (whole slab, worst layer) and the update
    grad = trainer.slab.grads[: trainer.slab.numel].cpu()
    gn, rn = grad / grad.norm(), ref_grad / ref_grad.norm()
    worst = max(relerr(gn[o:o + n], rn[o:o + n]) for o, n in trainer.slab.offsets)
    print(f"cfg-5 PTI step: loss {loss:.5f} vs {ref_losses[0]:.5f}; LoRA grad direction err {relerr(gn, rn):.2e} (worst layer {worst:.2e})")
    assert relerr(gn, rn) < 1.5e-2 and worst < 8e-2, (relerr(gn, rn), worst)
    got = tr.flat_lora_state(unet).cpu()
    agree = (((got - init_state) * (want - init_state)) > 0).float().mean().item()
    assert agree > 0.96, agree
    # token table: gradient rows exist exactly for the tokens that occurred, their direction matches the oracle's
    a, b = trainer.token_table.range
    tgrad = trainer.slab.grads[a:b].view(vocab, 1024).cpu()
    used = torch.zeros(vocab, dtype=torch.bool)
    used[ids.reshape(-1)] = True
    assert float(tgrad[~used].abs().max()) == 0.0 and float(ref_tgrad[~used].abs().max()) == 0.0
    # (compare like with like: the oracle's rows are clipped with the global coefficient, ours are not yet scaled → directions)
    tn, trn = tgrad[used] / tgrad[used].norm(), ref_tgrad[used] / ref_tgrad[used].norm()
    print(f"  table-gradient rows ({int(used.sum())} tokens): direction err {relerr(tn, trn):.2e}")
    assert relerr(tn, trn) < 3e-2, relerr(tn, trn)
    # relative size of the two gradient blocks (what the shared clip norm sees)
    ratio = (tgrad.norm() / grad.norm()).item() / (ref_tgrad.norm() / ref_grad.norm()).item()
    assert abs(ratio - 1) < 2e-2, ratio
    got_table = te.get_input_embeddings().weight.detach().float().cpu()
    upd, wupd = got_table[used] - table_init[used], want_table[used] - table_init[used]
    t_agree = ((upd * wupd) > 0).float().mean().item()
    print(f"  update sign agreement: LoRA {agree:.4f}, table rows {t_agree:.4f}")
    assert t_agree > 0.95, t_agree
    # rows of absent tokens: decoupled weight decay only, bit-for-bit AdamW semantics on a zero gradient
    assert relerr(got_table[~used][:2000], want_table[~used][:2000]) < 1e-6


def test_new_entry_points_edge_cases():
    """Empty inputs, out-of-range ids and the combinations the library declines (include/lora_hip.h: lora_gemm_parts,
    embed_rows_*): no launch for empty work, LORA_E_UNSUPPORTED (→ False: the caller runs the layers one by one) for f32 and
    for a part-wise backward that is not three parts, LORA_E_BADARG for a width that does not split evenly."""
    table = torch.arange(12, dtype=torch.float32, device=DEV).view(3, 4)
    empty = torch.empty(0, dtype=torch.int64, device=DEV)
    assert nat.embed_rows_fwd(table, empty, torch.float16).shape == (0, 4)
    grad = torch.full((3, 4), 7.0, device=DEV)
    nat.embed_rows_bwd(torch.empty(0, 4, dtype=torch.float16, device=DEV), empty, grad)
    assert torch.equal(grad, torch.full_like(grad, 7.0))
    # an id outside the table (torch.nn.Embedding raises): the C entry points treat it the SAME way in both directions — the
    # forward row is poisoned with NaN (never another token's row), the backward drops the position — and the binding layer
    # raises like torch where looking at the ids is free: on the host (TokenTable.check_ids)
    ids = torch.tensor([5, -1, 1], device=DEV)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        rows = nat.embed_rows_fwd(table, ids, dt)
        assert torch.isnan(rows[:2]).all() and torch.equal(rows[2], table[1].to(dt))
    nat.embed_rows_bwd(torch.ones(3, 4, device=DEV), ids, grad)
    assert torch.equal(grad[1], torch.ones(4, device=DEV)) and torch.equal(grad[0], torch.full((4,), 7.0, device=DEV))
    assert torch.equal(grad[2], torch.full((4,), 7.0, device=DEV))
    tt = tr.TokenTable.__new__(tr.TokenTable)
    tt.V = 3
    tt.check_ids(torch.tensor([[0, 2]]))
    for bad in (ids.cpu(), torch.tensor([3])):
        with pytest.raises(IndexError):
            tt.check_ids(bad)
    tt.check_ids(ids)  # device-resident ids are not looked at (no sync, no one-rank-only exception): the NaN row is the signal

    K = N = 64
    x = torch.randn(16, K, device=DEV).half()
    w = torch.randn(3 * N, K, device=DEV).half()
    fa = torch.zeros(3 * 16 * K, device=DEV).half()
    qb = torch.zeros(3 * N * 16, device=DEV).half()
    y = torch.empty(16, 3 * N, device=DEV).half()
    t = torch.empty(16, 24, device=DEV)
    assert nat.lora_gemm_parts(x[:0], w, None, fa, qb, y[:0], t[:0], 24, 0, K, 3 * N, 8, 3, False, 1.0)   # M = 0: nothing to do
    assert nat.lora_gemm_parts(x, w, None, fa, qb, y, t, 24, 16, K, 3 * N, 8, 3, False, 1.0)
    assert torch.allclose(y.float(), x.float() @ w.float().t(), atol=2e-2, rtol=2e-3)                      # zero factors: the base GEMM
    assert not nat.lora_gemm_parts(x.float(), w.float(), None, fa.float(), qb.float(), y.float(), t, 24, 16, K, 3 * N, 8, 3,
                                   False, 1.0)                                                              # f32: unsupported
    assert not nat.lora_gemm_parts(x, w[:2 * N], None, fa, qb, y, t, 24, 16, K, 2 * N, 8, 2, True, 1.0)     # part-wise backward: 3 parts only
    with pytest.raises(RuntimeError):
        nat.lora_gemm_parts(x, w, None, fa, qb, y, t, 24, 16, K, 3 * N - 8, 8, 3, False, 1.0)              # width not divisible


def test_row_aware_adamw_equals_the_dense_update_bit_for_bit():
    """lora_adamw_rows — full AdamW on the rows that ever had a gradient, p·(1 − lr·wd) on the others — against lora_adamw_step
    over the whole table (torch.optim.AdamW's arithmetic, pinned elsewhere): for rows with g = m = v = 0 the dense update IS the
    pure decay, so the two must agree BIT FOR BIT — parameters and both moments — over several steps, including a skipped
    (overflowed) one and rows that were touched earlier but get no gradient now."""
    g = torch.Generator().manual_seed(9)
    V, D = 300, 64
    p0 = torch.randn(V, D, generator=g).to(DEV)
    state = [dict(p=p0.clone(), m=torch.zeros(V, D, device=DEV), v=torch.zeros(V, D, device=DEV)) for _ in range(2)]
    active = torch.zeros(V, dtype=torch.uint8, device=DEV)
    norm = [torch.zeros(4, device=DEV) for _ in range(2)]
    for step in range(5):
        ids = torch.randint(0, V, (40,), generator=g).to(DEV)
        rows = torch.randn(40, D, generator=g).to(DEV)
        if step == 2:
            rows[3, 5] = float("inf")  # an overflowed step: skipped by both, the rows it touched still count as active
        grads = [torch.zeros(V, D, device=DEV) for _ in range(2)]
        nat.embed_rows_bwd(rows, ids, grads[0], active=None)
        nat.embed_rows_bwd(rows, ids, grads[1], active=active)
        assert torch.equal(grads[0], grads[1])
        for k in range(2):
            nat.lora_grad_sqnorm(grads[k].view(-1), 1.0, norm[k])
        nat.lora_adamw_step(state[0]["p"].view(-1), grads[0].view(-1), state[0]["m"].view(-1), state[0]["v"].view(-1), norm[0],
                            1.0, 1.0, 5e-3, 0.9, 0.999, 1e-8, 1e-3, 0)
        nat.lora_adamw_rows(state[1]["p"], grads[1], state[1]["m"], state[1]["v"], active, norm[1], 1.0, 1.0, 5e-3, 0.9, 0.999, 1e-8,
                            1e-3, 0)
        for key in ("p", "m", "v"):
            assert torch.equal(state[0][key], state[1][key]), (step, key)
    assert 0 < int(active.sum()) < V and torch.equal(norm[0], norm[1])
    assert float(norm[0][3]) == 1.0 and float(norm[0][2]) == 4.0  # one skipped, four applied steps
    untouched = active == 0
    assert torch.equal(state[1]["m"][untouched], torch.zeros_like(state[1]["m"][untouched]))
    assert not torch.equal(state[1]["p"][untouched], p0[untouched])  # ... and they did decay
