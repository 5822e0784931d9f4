"""GPU suite: the nn.Module flavour of the trainer step (trainer/module_engine.py — the MoE / expert-parallel path of train()):

  * on a DENSE model it must reproduce the flat-buffer engine (the path every parity gate runs on) — loss, gradient norm, parameters after
    optimizer steps — from the same weights and rollouts: same kernels underneath, different schedule (autograd graph + one fused AdamW launch
    over flat buffers instead of the hand-scheduled backward);
  * train(config) on a Qwen3-MoE model: the reference's launch grammar, metric keys, HF-named filesystem broadcast with per-expert tensors."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mbs(vocab, seq_len, n_mb, shift=0):
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=n_mb, generate_samples=True), seq_len=seq_len)
    dl.batch_counter = shift
    mbs = dl.get_batch()
    for mb in mbs:
        mb["input_ids"] = mb["input_ids"] % vocab
        mb["temperatures"] = torch.where(torch.arange(seq_len)[None] % 3 == 0, 0.7, 1.0)
    return mbs


@pytest.mark.parametrize("family", ["qwen3", "qwen2"])
def test_module_engine_equals_flat_engine_on_a_dense_model(family):
    import dataclasses

    from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims, OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
    from prime_rl_amd.trainer.models.llama import Qwen2Config, Qwen2ForCausalLM
    from prime_rl_amd.trainer.models.qwen3 import Qwen3Config, Qwen3ForCausalLM
    from prime_rl_amd.trainer.module_engine import ModuleEngine

    tied = family == "qwen2"
    dims = dataclasses.replace(ModelDims.tiny(layers=2, vocab=512), qk_norm=family == "qwen3", attention_bias=family == "qwen2", tie_word_embeddings=tied)
    optim = OptimConfig(lr=1e-3, weight_decay=0.01, max_norm=1.0)
    flat = GRPOEngine(dims, device=DEV, optim=optim, seed=5)
    cfg_cls, model_cls = (Qwen3Config, Qwen3ForCausalLM) if family == "qwen3" else (Qwen2Config, Qwen2ForCausalLM)
    cfg = cfg_cls(vocab_size=512, hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size, num_hidden_layers=2, num_attention_heads=dims.num_attention_heads,
                  num_key_value_heads=dims.num_key_value_heads, head_dim=128, tie_word_embeddings=tied)
    model = model_cls(cfg).to(DEV).to(torch.bfloat16)
    model.load_state_dict({k: v.clone() for k, v in flat.store.state_dict().items()})
    inject_prime_lm_head(model, chunk_size=128)
    mod = ModuleEngine(model, DEV, optim=optim)
    sd0 = {k: v.float().clone() for k, v in flat.store.state_dict().items()}
    # parameters now live in the flat buffers: one bf16 buffer holds every bf16 parameter
    assert all(p.data_ptr() >= mod.params_bf16.data_ptr() and p.data_ptr() < mod.params_bf16.data_ptr() + mod.params_bf16.numel() * 2 for p in model.parameters())
    assert mod.n_replicated == mod.total  # no expert parallelism: everything is replicated
    for step in range(3):
        mbs = _mbs(512, 192, 4, shift=step)
        passes = [pack_micro_batches(mbs[:2], DEV), pack_micro_batches(mbs[2:], DEV)]
        a = flat.step(passes)
        b = mod.step(passes)
        flat.wait_params_ready()
        la, lb = sum(x.item() for x in a["loss"]), sum(x.item() for x in b["loss"])
        # step 0: identical weights -> the only differences are GEMM shapes (fused qkv / gate|up vs separate projections).  Later steps start
        # from parameters that differ by a few flipped AdamW updates (|delta| = lr per element at step 1)
        assert abs(la - lb) < (1e-4 if step == 0 else 5e-4), (step, la, lb)
        na, nb = a["grad_norm"].item(), b["grad_norm"].item()
        assert abs(na - nb) / na < (2e-2 if step == 0 else 5e-2), (step, na, nb)
        assert a["scales"] == b["scales"]
        for x, y in zip(a["logprobs"], b["logprobs"]):
            d = (x - y).abs()
            assert d.mean() < (4e-3 if step == 0 else 2e-2), (step, d.mean().item())
        for key in ("sample_stats", "sample_metrics"):
            assert all(x.shape == y.shape for x, y in zip(a[key], b[key]))
    sd_flat, sd_mod = flat.store.state_dict(), mod.state_dict()
    assert set(sd_flat) == set(sd_mod)
    for k in sd_flat:
        # three AdamW steps of |update| <= lr each: the two engines move every matrix the same way (the module engine takes bf16 per-pass gradients
        # from autograd where the flat engine accumulates f32 weight gradients, so near-zero gradient elements may step in opposite directions)
        da, db = (sd_flat[k].float() - sd0[k]).flatten(), (sd_mod[k].float() - sd0[k]).flatten()
        cos = torch.nn.functional.cosine_similarity(da, db, dim=0).item()
        # (the k_proj bias barely matters to the loss — a constant added to every key moves no softmax except through RoPE — so its Adam-normalised
        # update is mostly the sign of noise: a looser bound for vectors)
        assert float((da - db).abs().max()) <= 6.5 * 1e-3 and cos > (0.9 if sd0[k].dim() > 1 else 0.7), (k, float((da - db).abs().max()), cos)
    # the fused optimizer launch zeroes the f32 gradients it consumed — except over DIRECT parameters (here the LM head's weight), whose first weight-gradient
    # launch of the next step overwrites the slice: those are only flagged "not written yet"
    direct = {id(p) for p in mod._direct}
    assert direct, "the injected LM head's weight takes its gradient straight from the weight-gradient GEMM"
    for (name, p, off), gv in zip(mod.entries, mod._grad_views):
        if id(p) in direct:
            assert p.main_grad_written == [False], name
        else:
            assert torch.count_nonzero(gv) == 0, name


def test_train_loop_on_a_qwen3_moe_model(tmp_path, capsys):
    """`train(config)` picks the module engine for an MoE model: reference launch grammar, metric keys, loss that moves under a large lr, and a
    filesystem broadcast in HF names (per-expert gate / up / down projections, `mlp.gate.weight` router)."""
    from safetensors.torch import load_file

    from prime_rl_amd.configs.trainer import parse_argv
    from prime_rl_amd.trainer.module_engine import ModuleDims
    from prime_rl_amd.trainer.rl.train import train

    toml = tmp_path / "rl.toml"
    toml.write_text('max_steps = 3\n\n[data.fake]\nbatch_size = 4\ngenerate_samples = true\n\n[model]\nseq_len = 128\n\n[optim]\nlr = 1e-3\n')
    cfg = parse_argv(["@", str(toml), "--model.name", "tiny-moe", "--model.debug.random-init", "--output-dir", str(tmp_path), "--tokens-per-pass", "256",
                      "--monitors.file.path", "metrics.jsonl"])
    hist = train(cfg)
    assert len(hist) == 3
    for key in ("perf/throughput", "perf/mfu", "optim/lr", "optim/grad_norm", "loss/mean", "entropy/all/mean", "mismatch_kl/all/max", "is_masked/mean"):
        assert key in hist[0] and math.isfinite(hist[2][key]), key
    assert hist[0]["optim/grad_norm"] > 0 and hist[0]["perf/mfu"] > 0
    # MoE load-balance statistics in the step record and on the console line (train.py:531-546, 644-645)
    assert hist[0]["max_vio/mean"] >= 0 and 0 < hist[0]["routing_confidence/mean"] <= 1.0 + 1e-5 and "max_vio/max" in hist[0]
    assert "| Max Vio " in capsys.readouterr().out
    assert any(abs(hist[0][k] - hist[2][k]) > 0 for k in ("loss/mean", "entropy/all/mean"))  # the parameters moved
    sd = load_file(str(tmp_path / "broadcasts" / "step_3" / "model.safetensors"))
    assert "model.layers.0.mlp.gate.weight" in sd and "model.layers.1.mlp.experts.7.down_proj.weight" in sd and "model.layers.0.mlp.experts.0.gate_proj.weight" in sd
    assert not any(".experts.w1" in k or "router.gate" in k for k in sd)
    assert sd["model.layers.0.mlp.experts.3.up_proj.weight"].shape == (128, 256) and all(v.dtype == torch.bfloat16 for v in sd.values())
    # the reference's flop model for an MoE config (perf.py:110-175): top-k experts + router + attention + head per token
    from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig

    c = Qwen3MoeConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, num_experts=8,
                       num_experts_per_tok=2, moe_intermediate_size=128)
    want = 2 * (2 * 256 * 4 * 128 + 2 * 256 * 2 * 128) + 2 * (2 * 3 * 128 * 256 + 8 * 256) + 512 * 256
    assert ModuleDims(c).active_mm_params() == want


def test_expert_gradients_written_in_f32_by_the_weight_gradient_gemm(monkeypatch):
    """PM_MOE_MAIN_GRAD (default on): the grouped weight-gradient GEMM writes / accumulates the expert gradients in f32 straight into the engine's
    flat main-gradient buffer (no bf16 `.grad`, no bf16 -> f32 accumulation pass).  Against the autograd path: same gradients up to the ONE bf16
    rounding that path applies per pass, over two accumulated passes, and the same optimizer step."""
    from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
    from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
    from prime_rl_amd.trainer.module_engine import ModuleEngine
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    cfg = Qwen3MoeConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, num_experts=8,
                         num_experts_per_tok=2, moe_intermediate_size=256, norm_topk_prob=True)
    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=4), seq_len=256)
    mbs = dl.get_batch()
    for mb in mbs:
        mb["input_ids"] = mb["input_ids"] % cfg.vocab_size
    passes = [pack_micro_batches(mbs[:2], DEV), pack_micro_batches(mbs[2:], DEV)]
    grads, params, norms = {}, {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PM_MOE_MAIN_GRAD", mode)
        torch.manual_seed(0)
        with torch.device(DEV):
            model = Qwen3MoeForCausalLM(cfg)
        model = model.to(torch.bfloat16)
        gen = torch.Generator(device=DEV).manual_seed(0)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.fill_(1.0) if p.dim() == 1 else p.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.05).to(p.dtype))
        inject_prime_lm_head(model, chunk_size=8192)
        eng = ModuleEngine(model, DEV, optim=OptimConfig(lr=1e-3))
        # w1, w2, w3 of every layer + the LM head + the q / k / v / o projections (layers/linear.py nodes; all four are multiples of 128 at this width;
        # the 8-expert router gate is not and keeps the `.grad` path)
        assert len(eng._direct) == ((3 + 4) * cfg.num_hidden_layers + 1 if mode == "1" else 0)
        scales = eng.global_scales(tuple(sum(p.local_counts[k] for p in passes) for k in range(3)))
        for p in passes:
            ctx = eng.forward_loss(p, scales)
            eng.backward(ctx)
        grads[mode] = eng.main_grad.clone()
        if mode == "1":
            assert all(p.grad is None and p.main_grad_written[0] for p in eng._direct)
            assert len(eng._fused_norm) == 2 * cfg.num_hidden_layers and all(valid[0] for _, valid, _, _ in eng._fused_norm)
        norms[mode] = float(eng.optimizer_step())
        params[mode] = eng.params_bf16.clone()
        if mode == "1":
            # consumed: direct slices are flagged "not written" (the next step's first weight-gradient launch overwrites them — the optimizer launch does
            # not spend 4 B per parameter zeroing them), every other slice is zero
            assert not any(p.main_grad_written[0] for p in eng._direct)
            direct = {id(p) for p in eng._direct}
            assert all(float(gv.abs().max()) == 0.0 for (nm, p, off), gv in zip(eng.entries, eng._grad_views) if id(p) not in direct)
            assert [z for _, _, z in eng._update_spans].count(False) >= cfg.num_hidden_layers  # the experts of a layer form one span
    # the expert parameters' share of the norm came out of the weight-gradient epilogues (ModuleEngine._fused_norm): same number as the pass over the buffer
    monkeypatch.setenv("PM_MOE_MAIN_GRAD", "1")
    monkeypatch.setenv("PM_MOE_FUSED_NORM", "0")
    torch.manual_seed(0)
    with torch.device(DEV):
        model = Qwen3MoeForCausalLM(cfg)
    model = model.to(torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.fill_(1.0) if p.dim() == 1 else p.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.05).to(p.dtype))
    inject_prime_lm_head(model, chunk_size=8192)
    eng = ModuleEngine(model, DEV, optim=OptimConfig(lr=1e-3))
    assert not eng._fused_norm
    scales = eng.global_scales(tuple(sum(p.local_counts[k] for p in passes) for k in range(3)))
    for p in passes:
        eng.backward(eng.forward_loss(p, scales))
    assert torch.equal(eng.main_grad, grads["1"])
    plain = float(eng.optimizer_step())
    assert abs(plain - norms["1"]) <= 1e-6 * plain, (plain, norms["1"])
    assert torch.equal(eng.params_bf16, params["1"]) or (eng.params_bf16 != params["1"]).float().mean().item() < 1e-5
    rel = ((grads["1"] - grads["0"]).norm() / grads["0"].norm()).item()
    assert rel < 4e-3, rel  # the autograd path rounds every pass's expert gradient to bf16 before accumulating
    assert abs(norms["1"] - norms["0"]) <= 2e-3 * norms["0"]
    assert (params["1"] != params["0"]).float().mean().item() < 0.02


def test_direct_parameters_whose_gradient_arrives_through_autograd_after_all(monkeypatch):
    """the optimizer launch does not zero the gradient slice of a DIRECT parameter (its first weight-gradient launch of a step overwrites it).  When
    such a parameter's gradient arrives as a bf16 `.grad` after all — experts whose width the f32-writing grouped GEMM does not take (128 is not a multiple of
    256) — the engine must zero the slice before accumulating, in EVERY step, or the second step would add to the first step's consumed gradient.  Two optimizer
    steps with PM_MOE_MAIN_GRAD=1 against the engine that accumulates every gradient from `.grad` (=0): same gradients, norms and parameters."""
    from prime_rl_amd import ops
    from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
    from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
    from prime_rl_amd.trainer.module_engine import ModuleEngine
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    cfg = Qwen3MoeConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, num_experts=8,
                         num_experts_per_tok=2, moe_intermediate_size=128, norm_topk_prob=True)
    assert not ops.grouped_gemm_tn_f32_supported(128, 256, 256)
    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=2), seq_len=256)
    steps = []
    for _ in range(2):
        mbs = dl.get_batch()
        for mb in mbs:
            mb["input_ids"] = mb["input_ids"] % cfg.vocab_size
        steps.append(pack_micro_batches(mbs, DEV))
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PM_MOE_MAIN_GRAD", mode)
        torch.manual_seed(0)
        with torch.device(DEV):
            model = Qwen3MoeForCausalLM(cfg)
        model = model.to(torch.bfloat16)
        gen = torch.Generator(device=DEV).manual_seed(0)
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.fill_(1.0) if p.dim() == 1 else p.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.05).to(p.dtype))
        inject_prime_lm_head(model, chunk_size=8192)
        eng = ModuleEngine(model, DEV, optim=OptimConfig(lr=1e-3))
        if mode == "1":
            experts = [p for p in eng._direct if p.dim() == 3]
            assert len(experts) == 3 * cfg.num_hidden_layers and not eng._fused_norm  # direct, but their width keeps them off the f32-writing GEMM
        norms, grads = [], []
        for pb in steps:
            scales = eng.global_scales(tuple(pb.local_counts[k] for k in range(3)))
            eng.backward(eng.forward_loss(pb, scales))
            grads.append(eng.main_grad.clone())
            norms.append(float(eng.optimizer_step()))
        # a third optimizer step that no backward pass preceded (every direct parameter unwritten: an unused / frozen path): the slices still hold step 2's
        # consumed gradients and must be zeroed, not applied again
        assert float(eng.optimizer_step()) == 0.0 and not eng.main_grad.any().item()
        res[mode] = (norms, grads, eng.params_bf16.clone())
    # (the LM head and the attention projections take the f32-writing GEMMs in mode 1 and one bf16 rounding per pass in mode 0; a slice that kept the first
    # step's gradient would be off by its whole norm)
    for a, b in zip(res["1"][0], res["0"][0]):
        assert abs(a - b) <= 2e-3 * b, (res["1"][0], res["0"][0])
    for ga, gb in zip(res["1"][1], res["0"][1]):
        # measured 0.0097 in step 2 (bf16 rounding of every pass's gradient in mode 0, on parameters that already differ by an ulp here and there);
        # a slice that kept step 1's gradient would be off by the experts' share of the whole norm (> 0.3)
        assert ((ga - gb).norm() / gb.norm()).item() < 0.05
    # bf16 parameters after three updates whose gradients differ by ~1 %: 5.9 % of the elements round to the neighbouring bf16 value (measured)
    diff = res["1"][2].float() - res["0"][2].float()
    # (an element whose tiny gradient changes sign between the two runs moves by lr in opposite directions in each of the three updates)
    assert (diff != 0).float().mean().item() < 0.10 and diff.abs().max().item() <= 3 * 2 * 1e-3 + 2.0**-7


def test_router_replay_through_rollout_files_and_train(tmp_path):
    """`enable_router_replay` end to end (train.py:349-360): rollouts written by the orchestrator-side sender carry `routed_experts` (what vLLM returns
    with --enable-return-routed-experts), travel the msgpack wire, are packed by prepare_batch, reach the MoE layers as forced expert indices.
    Replaying the experts the SAME weights would pick gives the free run's numbers exactly; different experts give different numbers; a missing
    payload with the flag on is an error; a payload with the flag off is ignored."""
    import asyncio

    import numpy as np

    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer.batch import build_bin_cost, prepare_batch
    from prime_rl_amd.trainer.engine import pack_micro_batches
    from prime_rl_amd.trainer.model import setup_module_engine
    from prime_rl_amd.trainer.parallel_dims import get_parallel_dims
    from prime_rl_amd.trainer.rl.data import DataLoader
    from prime_rl_amd.trainer.rl.train import train
    from prime_rl_amd.transports.rollouts import FileSystemMicroBatchSender
    from prime_rl_amd.transports.rollouts.types import RoutedExperts, TrainingSample

    def cfg_for(out, replay, steps=1):
        return TrainerConfig.model_validate(dict(model=dict(name="tiny-moe", seq_len=256, debug=dict(random_init=True)), max_steps=steps, output_dir=str(out),
                                                 tokens_per_pass=1024, enable_router_replay=replay, optim=dict(lr=1e-3),
                                                 rollout_transport=dict(type="filesystem")))

    rng = np.random.default_rng(5)

    def samples(routing_for=None):
        out = []
        r2 = np.random.default_rng(6)
        for i, n in enumerate((200, 120, 256, 77)):
            ids = r2.integers(0, 512, size=n).tolist()
            routed = None
            if routing_for is not None:
                arr = routing_for(i, n)
                routed = RoutedExperts(data=arr.astype("uint8").tobytes(), shape=[n, 2, 2], dtype="uint8")
            out.append(TrainingSample(token_ids=ids, mask=[j >= 16 for j in range(n)], logprobs=(-5.0 - r2.random(n)).tolist(), temperatures=[1.0] * n,
                                      env_name="math", advantages=r2.standard_normal(n).tolist(), routed_experts=routed))
        return out

    # --- the routing the model itself would choose for these samples: one free forward per sample through the same module engine
    free_cfg = cfg_for(tmp_path / "probe", False)
    eng = setup_module_engine(free_cfg, get_parallel_dims(free_cfg.model, 256, world_size=1), device=DEV)
    own = {}
    for i, s in enumerate(samples()):
        (mb,), = prepare_batch([s], 256, 1, build_bin_cost(None), 1)
        pb = pack_micro_batches([DataLoader._micro_batch_to_tensor(mb)], DEV)
        recorded = []
        hooks = [m.router.register_forward_hook(lambda mod, inp, out_: recorded.append(out_[1].clone())) for m in eng.model.moe_layers()]
        with torch.no_grad():
            eng.forward_loss(pb, (1, 1, 1))
        for h in hooks:
            h.remove()
        own[i] = torch.stack(recorded, dim=1).cpu().numpy()  # [n, layers, top_k]
    del eng
    torch.cuda.empty_cache()

    def run(name, routing_for, replay):
        out = tmp_path / name
        sender = FileSystemMicroBatchSender(out, data_world_size=1, current_step=1)
        asyncio.run(sender.send(prepare_batch(samples(routing_for), 256, 1, build_bin_cost(None), 1)))
        return train(cfg_for(out, replay))[0]

    free = run("free", None, False)
    same = run("same", lambda i, n: own[i][:n], True)
    other = run("other", lambda i, n: (own[i][:n] + 1 + rng.integers(0, 3, size=own[i][:n].shape)) % 8, True)
    ignored = run("ignored", lambda i, n: (own[i][:n] + 3) % 8, False)
    for key in ("loss/mean", "entropy/all/mean", "mismatch_kl/all/mean", "optim/grad_norm"):
        assert same[key] == free[key], (key, same[key], free[key])  # replaying the model's own choices changes nothing, bit for bit
        assert ignored[key] == free[key], key  # payload present, flag off: not used
    assert other["loss/mean"] != free["loss/mean"] and other["entropy/all/mean"] != free["entropy/all/mean"]
    with pytest.raises(Exception, match="enable_router_replay needs routed_experts"):
        run("missing", None, True)
    dense = TrainerConfig.model_validate(dict(model=dict(name="tiny", seq_len=256, debug=dict(random_init=True)), enable_router_replay=True))
    from prime_rl_amd.trainer.model import setup_engine

    with pytest.raises(NotImplementedError, match="no router"):
        setup_engine(dense, get_parallel_dims(dense.model, 256, world_size=1), device=DEV)
