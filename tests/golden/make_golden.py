"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own code.

Run in the build container only (`/root/reference` does not exist on the GPU box):

    python tests/golden/make_golden.py

It file-loads (never copies) `/root/reference/src/prime_rl/trainer/models/layers/lm_head.py` and
`/root/reference/src/prime_rl/trainer/rl/loss.py`, with stub modules for the absent third-party imports
(jaxtyping, beartype, prime_rl.configs.trainer, prime_rl.utils.utils), runs them on seeded CPU inputs and stores
inputs + outputs as small .npz files.  The tests compare the oracle (CPU suite) and the HIP kernels (GPU suite)
against these files.
"""

from __future__ import annotations

import importlib.util
import sys
import types
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent


def _stub_modules() -> None:
    jt = types.ModuleType("jaxtyping")

    class _Sub:
        def __class_getitem__(cls, item):
            return object

    jt.Float = jt.Int = jt.Bool = _Sub
    jt.jaxtyped = lambda typechecker=None: (lambda f: f)
    sys.modules["jaxtyping"] = jt
    bt = types.ModuleType("beartype")
    bt.beartype = lambda f: f
    sys.modules["beartype"] = bt

    cfg = types.ModuleType("prime_rl.configs.trainer")

    @dataclass
    class DefaultLossConfig:  # mirrors packages/prime-rl-configs/src/prime_rl/configs/trainer.py:494-507
        type: str = "default"
        dppo_mask_low: float = 0.2
        dppo_mask_high: float = 0.2
        adv_tau: float = 1.0
        kl_tau: float = 1e-3

    @dataclass
    class IPOLossConfig:  # configs/trainer.py:509-520
        type: str = "ipo"
        ipo_threshold: float = 0.1
        adv_tau: float = 1.0
        kl_tau: float = 1e-3

    @dataclass
    class CustomLossConfig:
        type: str = "custom"
        import_path: str = ""
        kwargs: dict = field(default_factory=dict)

    cfg.DefaultLossConfig, cfg.IPOLossConfig, cfg.CustomLossConfig = DefaultLossConfig, IPOLossConfig, CustomLossConfig
    cfg.LossConfig = object
    for name in ("prime_rl", "prime_rl.configs", "prime_rl.utils", "prime_rl.trainer", "prime_rl.trainer.rl", "prime_rl.trainer.models",
                 "prime_rl.trainer.models.layers"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []  # mark as package
            sys.modules[name] = m
    sys.modules["prime_rl.configs.trainer"] = cfg
    uu = types.ModuleType("prime_rl.utils.utils")
    uu.import_object = lambda path: None
    sys.modules["prime_rl.utils.utils"] = uu
    lg = types.ModuleType("prime_rl.utils.logger")

    class _L:
        def info(self, *a, **k):
            pass

        debug = warning = info

    lg.get_logger = lambda: _L()
    sys.modules["prime_rl.utils.logger"] = lg
    vlm = types.ModuleType("prime_rl.utils.vlm")
    vlm.get_final_logit_softcapping = lambda config: None
    sys.modules["prime_rl.utils.vlm"] = vlm
    torch.compile = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))  # eager reference


def _load(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main() -> None:
    assert REF.exists(), "run in the build container: /root/reference is required"
    _stub_modules()
    lm = _load("ref_lm_head", "prime_rl/trainer/models/layers/lm_head.py")
    loss = _load("ref_loss", "prime_rl/trainer/rl/loss.py")
    cfgm = sys.modules["prime_rl.configs.trainer"]

    # ---- 1. fused LM head, fp32, the reference's own unit-test shape (tests/unit/train/rl/test_fused_lm_head.py:25-61)
    torch.manual_seed(0)
    b, s, h, v = 2, 4, 8, 37
    temperature = torch.full((b, s), 1.7)
    hidden = torch.randn(b, s, h, requires_grad=True)
    labels = torch.randint(0, v, (b, s))
    weight = torch.randn(v, h)
    head = lm.FusedOutputLinear(in_features=h, out_features=v, chunk_size=3)
    head.weight = torch.nn.Parameter(weight.clone())
    out = head(hidden, labels, temperature=temperature)
    gl = torch.randn(b, s)
    (out["logprobs"] * gl).sum().backward()
    np.savez_compressed(OUT / "lm_head_fp32.npz", hidden=hidden.detach().numpy(), weight=weight.numpy(), labels=labels.numpy(),
             temperature=temperature.numpy(), logprobs=out["logprobs"].detach().numpy(), entropy=out["entropy"].detach().numpy(),
             grad_logprobs=gl.numpy(), grad_hidden=hidden.grad.numpy(), grad_weight=head.weight.grad.numpy())

    # ---- 2. fused LM head, bf16, larger vocab (crosses the 8192 vocab-chunk boundary), mixed temperatures
    torch.manual_seed(1)
    n, h, v = 48, 64, 9000
    hidden = (torch.randn(1, n, h) * 1.0).to(torch.bfloat16).requires_grad_(True)
    weight = (torch.randn(v, h) * 0.2).to(torch.bfloat16)
    labels = torch.randint(0, v, (1, n))
    temperature = torch.where(torch.rand(1, n) < 0.5, torch.tensor(0.7), torch.tensor(1.0))
    head = lm.FusedOutputLinear(in_features=h, out_features=v, chunk_size=8192).to(torch.bfloat16)
    head.weight = torch.nn.Parameter(weight.clone())
    out = head(hidden, labels, temperature=temperature)
    gl = torch.randn(1, n)
    (out["logprobs"] * gl).sum().backward()
    np.savez_compressed(OUT / "lm_head_bf16.npz", hidden=hidden.detach().float().numpy(), weight=weight.float().numpy(), labels=labels.numpy(),
             temperature=temperature.numpy(), logprobs=out["logprobs"].detach().numpy(), entropy=out["entropy"].detach().numpy(),
             grad_logprobs=gl.numpy(), grad_hidden=hidden.grad.float().numpy(), grad_weight=head.weight.grad.float().numpy())

    # ---- 3. vanilla path: selective_log_softmax + compute_entropy on materialised f32 logits / T (loss.py:48-63, train.py:441)
    torch.manual_seed(2)
    n, v = 24, 3001  # odd vocab: exercises the unaligned head/tail path of the kernel
    logits = torch.randn(1, n, v) * 3.0
    labels = torch.randint(0, v, (1, n))
    temperature = torch.where(torch.rand(1, n) < 0.5, torch.tensor(0.7), torch.tensor(1.0))
    scaled = logits / temperature.unsqueeze(-1)
    lp = loss.selective_log_softmax(scaled, labels)
    ent = loss.compute_entropy(scaled)
    np.savez_compressed(OUT / "logprob_vanilla.npz", logits=logits[0].numpy(), labels=labels[0].numpy(), temperature=temperature[0].numpy(),
             logprobs=lp[0].numpy(), entropy=ent[0].numpy())

    # ---- 4. compute_loss fixtures (loss.py:285-403) incl. autograd gradient w.r.t. trainer_logprobs
    def run_loss(tag, seed, lens, cfg, with_weights, dppo_seed_scale=0.3):
        g = torch.Generator().manual_seed(seed)
        n = sum(lens)
        inf = -torch.rand(n, generator=g) * 2.0
        tr = (inf + torch.randn(n, generator=g) * dppo_seed_scale).clamp(max=-1e-3).requires_grad_(True)
        ref = -torch.rand(n, generator=g) * 2.0
        adv = torch.randn(n, generator=g)
        adv[torch.rand(n, generator=g) < 0.1] = 0.0
        mask = torch.rand(n, generator=g) > 0.3
        rl_w = ce_w = rk_w = None
        if with_weights:
            rl_w = torch.where(torch.rand(n, generator=g) < 0.6, torch.tensor(1.0), torch.tensor(0.0))
            ce_w = torch.where(torch.rand(n, generator=g) < 0.3, torch.tensor(0.5), torch.tensor(0.0))
            rk_w = torch.where(torch.rand(n, generator=g) < 0.3, torch.tensor(1.0), torch.tensor(0.0))
        fn = loss.setup_rl_loss_fn(cfg)
        sp = lambda t: None if t is None else list(t.split(lens))
        scales = (max(int((mask & (rl_w != 0)).sum()) if rl_w is not None else int(mask.sum()), 1),
                  max(int((ce_w != 0).sum()), 1) if ce_w is not None else 1, max(int((rk_w != 0).sum()), 1) if rk_w is not None else 1)
        val, met = loss.compute_loss(list(tr.split(lens)), sp(inf), sp(ref) if with_weights else None, sp(adv), sp(mask), sp(rl_w), sp(ce_w),
                                     sp(rk_w), fn, *scales)
        val.backward()
        arrs = dict(trainer_logprobs=tr.detach().numpy(), inference_logprobs=inf.numpy(), ref_logprobs=ref.numpy(), advantages=adv.numpy(),
                    loss_mask=mask.numpy(), lens=np.asarray(lens), scales=np.asarray(scales), loss=val.detach().numpy(), grad=tr.grad.numpy())
        if with_weights:
            arrs.update(rl_weights=rl_w.numpy(), ce_weights=ce_w.numpy(), ref_kl_weights=rk_w.numpy())
        for k, t in met.items():
            arrs["metric__" + k.replace("/", "__")] = t.detach().numpy()
        np.savez_compressed(OUT / f"loss_{tag}.npz", **arrs)

    run_loss("default", 3, [50, 30, 17], cfgm.DefaultLossConfig(), False)
    run_loss("default_weights", 4, [40, 25, 9, 31], cfgm.DefaultLossConfig(), True)
    run_loss("ipo", 5, [64, 33], cfgm.IPOLossConfig(), False)

    # ---- 5. FakeDataLoader micro-batches (trainer/rl/data.py:55-159): the synthetic rollout batches of the benchmark
    world = types.ModuleType("prime_rl.trainer.world")
    world.get_world = lambda: types.SimpleNamespace(world_size=2, rank=1)
    sys.modules["prime_rl.trainer.world"] = world
    tr = types.ModuleType("prime_rl.transports.rollouts")
    tr.MicroBatch = tr.MicroBatchReceiver = tr.TransportConfig = object
    tr.setup_micro_batch_receiver = lambda *a, **k: None
    sys.modules["prime_rl.transports"] = types.ModuleType("prime_rl.transports")
    sys.modules["prime_rl.transports.rollouts"] = tr
    cfgm.FakeDataLoaderConfig = object
    data = _load("ref_data", "prime_rl/trainer/rl/data.py")
    arrs = {}
    for gen_samples in (False, True):
        dl = data.FakeDataLoader(types.SimpleNamespace(batch_size=4, generate_samples=gen_samples), seq_len=96, dp_world_size=2)
        for step in range(2):
            for mi, mb in enumerate(dl.get_batch()):
                tag = f"g{int(gen_samples)}_s{step}_m{mi}_"
                for k in ("input_ids", "position_ids", "advantages", "inference_logprobs", "temperatures", "loss_mask", "seq_lens"):
                    arrs[tag + k] = mb[k].numpy()
                arrs[tag + "sequence_lengths"] = np.asarray(mb["sequence_lengths"])
    np.savez_compressed(OUT / "fake_data.npz", **arrs)

    # ---- 6. packing: the reference's prepare_batch (trainer/batch.py:796-835) on random text samples -> packing.json
    import json
    import random

    ms = types.ModuleType("msgspec")

    class Struct:  # minimal msgspec.Struct: keyword/positional init from annotations, class kwargs swallowed
        def __init_subclass__(cls, **kwargs):
            super().__init_subclass__()

        def __init__(self, *args, **kwargs):
            names = list(self.__class__.__annotations__)
            for n_, v_ in zip(names, args):
                setattr(self, n_, v_)
            for n_ in names[len(args):]:
                if n_ in kwargs:
                    setattr(self, n_, kwargs.pop(n_))
                elif hasattr(self.__class__, n_):
                    setattr(self, n_, getattr(self.__class__, n_))
                else:
                    raise TypeError(f"missing field {n_}")
            assert not kwargs, kwargs

    ms.Struct = Struct
    sys.modules["msgspec"] = ms
    del sys.modules["prime_rl.transports.rollouts"]
    rt = types.ModuleType("prime_rl.transports.rollouts")
    rt.__path__ = []
    sys.modules["prime_rl.transports.rollouts"] = rt
    rtypes = _load("prime_rl.transports.rollouts.types", "prime_rl/transports/rollouts/types.py")
    rt.TrainingSample, rt.MicroBatch = rtypes.TrainingSample, rtypes.MicroBatch
    batch = _load("ref_batch", "prime_rl/trainer/batch.py")
    qcfg = types.SimpleNamespace(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, head_dim=128, intermediate_size=12288,
                                 num_hidden_layers=36, vocab_size=151936)
    sys.path.insert(0, str(OUT))
    import packing_cases as PC

    cases = []
    for case_id, (n_samples, seq_len, workers, pad, max_len, use_cfg, streams) in enumerate(PC.CASES):
        samples = PC.make_samples(case_id)
        grid = batch.prepare_batch([rtypes.TrainingSample(**s) for s in samples], seq_len, workers, batch.build_bin_cost(qcfg if use_cfg else None), pad)
        cases.append(dict(case_id=case_id, grid=PC.digest_grid(grid)))
    weights_cases = []
    for seed in range(6):
        rng = random.Random(100 + seed)
        k = rng.choice([2, 3, 4, 8])
        w = [rng.randint(1, 10**6) for _ in range(k * rng.randint(1, 6))]
        weights_cases.append(dict(weights=w, k=k, parts=batch.balanced_partition(w, k)))
    bc = batch.build_bin_cost(qcfg)
    (OUT / "packing.json").write_text(json.dumps(dict(cases=cases, partitions=weights_cases, bin_cost_probe=[bc([2048]), bc([100, 200, 300])])))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
