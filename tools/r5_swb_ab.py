"""Round 5 A/B inside one process: the Qwen3-8B step (8 x 2048 tokens, bench.py's default workload) with the SwiGLU backward in the row-gradient GEMM's
epilogue vs as its own kernel — interleaved rounds on ONE engine (the switch is a run-time attribute), so box and clock state are shared."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches  # noqa: E402
from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    eng = GRPOEngine(ModelDims.qwen3_8b(), device=dev, optim=OptimConfig(), loss=LossConfig(), seed=0)
    loader = FakeDataLoader(FakeDataLoaderConfig(batch_size=n_mb), seq_len=2048)
    batches = [[pack_micro_batches(loader.get_batch(), dev)] for _ in range(4)]
    res = {True: [], False: []}
    for _ in range(4):
        for fused in (True, False):
            eng.fused_swiglu_bwd = fused
            eng.step(batches[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches[1:]:
                eng.step(b)
            torch.cuda.synchronize()
            res[fused].append((time.perf_counter() - t0) / 3 * 1e3)
    for fused, ts in res.items():
        print(json.dumps(dict(micro_batches=n_mb, fused_swiglu_bwd=fused, ms_per_step=[round(t, 2) for t in ts], median=round(sorted(ts)[len(ts) // 2], 2))))


main()
