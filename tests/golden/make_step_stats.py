"""Generate tests/golden/step_stats.json from the REFERENCE's own `Tensors.compute_stats` + `flexible_all_gather`
(/root/reference/src/prime_rl/trainer/utils.py:205-281), at world_size 1 and at world_size 2 (gloo, two processes).

Run in the build container only (`/root/reference` does not exist on the GPU box):

    python tests/golden/make_step_stats.py

The reference module is file-loaded (never copied) with stub modules for its absent imports (torchtitan, prime_rl.trainer.world,
...).  `compute_stats` moves every vector `.to("cuda")` unconditionally (utils.py:257) and this container has no GPU, so
`Tensor.to` / `torch.empty` / `torch.tensor` / `torch.zeros` are wrapped for the run to map the "cuda" device onto the CPU — the
arithmetic (cat, all_gather, mean / median / std / min / max, the empty-key NaNs) is untouched.
"""

from __future__ import annotations

import importlib.util
import json
import math
import os
import sys
import types
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent

# per-rank inputs: key -> list of 1-D vectors appended over the "micro-batches" of a step
CASES = {
    "world1": [
        {"loss": [[0.25], [-0.5], [0.125]], "entropy/all": [[1.0, 2.0, 4.0], [8.0]], "single": [[3.5]], "empty": [[]],
         "is_masked": [[0.0, 1.0], [0.0, 0.0, 1.0, 1.0]]},
    ],
    "world2": [
        {"loss": [[0.25], [-0.5]], "entropy/all": [[1.0, 2.0, 4.0], [8.0]], "only_rank0": [[7.0, -1.0]], "empty_everywhere": [[]],
         "mismatch_kl/all": [[0.001, 0.004, 0.002]]},
        {"loss": [[0.75]], "entropy/all": [[16.0, 32.0, 64.0, 128.0, 256.0]], "only_rank1": [[2.5]], "empty_everywhere": [[]],
         "mismatch_kl/all": [[]]},
    ],
}


def _stubs() -> None:
    for name in ("prime_rl", "prime_rl.trainer", "prime_rl.utils", "torchtitan", "torchtitan.distributed"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules.setdefault(name, m)
    tt = types.ModuleType("torchtitan.distributed.utils")
    tt.clip_grad_norm_ = lambda *a, **k: None
    sys.modules["torchtitan.distributed.utils"] = tt
    w = types.ModuleType("prime_rl.trainer.world")
    w.get_world = lambda: None
    sys.modules["prime_rl.trainer.world"] = w
    lg = types.ModuleType("prime_rl.utils.logger")
    lg.get_logger = lambda: None
    sys.modules["prime_rl.utils.logger"] = lg
    pa = types.ModuleType("prime_rl.utils.pathing")
    pa.get_ckpt_dir = lambda *a, **k: None
    sys.modules["prime_rl.utils.pathing"] = pa
    uu = types.ModuleType("prime_rl.utils.utils")
    uu.get_step_path = lambda *a, **k: None
    sys.modules["prime_rl.utils.utils"] = uu


def _cuda_is_cpu() -> None:
    def strip(args, kwargs):
        args = tuple("cpu" if (isinstance(a, str) and a == "cuda") else a for a in args)
        if kwargs.get("device") in ("cuda", torch.device("cuda")):
            kwargs = dict(kwargs, device="cpu")
        return args, kwargs

    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a, k = strip(a, k)
        return orig_to(self, *a, **k)

    torch.Tensor.to = to
    for fn in ("empty", "tensor", "zeros"):
        orig = getattr(torch, fn)

        def wrapped(*a, _orig=orig, **k):
            a, k = strip(a, k)
            return _orig(*a, **k)

        setattr(torch, fn, wrapped)


def _run(rank: int, world: int, port: int, inputs: list[dict], out_q) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _stubs()
    _cuda_is_cpu()
    spec = importlib.util.spec_from_file_location("ref_trainer_utils", REF / "prime_rl/trainer/utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tensors = mod.Tensors()
    for key, vecs in inputs[rank].items():
        for v in vecs:
            tensors[key].append(torch.tensor(v, dtype=torch.float32))
    stats = tensors.compute_stats()
    if rank == 0:
        out_q.put({k: (None if isinstance(v, float) and math.isnan(v) else v) for k, v in stats.items()})
    dist.barrier()
    dist.destroy_process_group()


def main() -> None:
    assert REF.exists(), "run in the build container: /root/reference is required"
    ctx = mp.get_context("spawn")
    result = {"_doc": "expected = the reference's Tensors.compute_stats() on rank 0; null stands for NaN", "cases": {}}
    for i, (name, inputs) in enumerate(CASES.items()):
        q = ctx.Queue()
        procs = [ctx.Process(target=_run, args=(r, len(inputs), 29640 + i, inputs, q)) for r in range(len(inputs))]
        for p in procs:
            p.start()
        expected = q.get(timeout=120)
        for p in procs:
            p.join()
        result["cases"][name] = {"inputs": inputs, "expected": expected}
    (OUT / "step_stats.json").write_text(json.dumps(result, indent=1, sort_keys=True) + "\n")
    print("wrote", OUT / "step_stats.json")


if __name__ == "__main__":
    main()
