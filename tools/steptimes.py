import sys, time, torch
sys.path.insert(0, '.')
from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims, pack_micro_batches
from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig
dev = torch.device('cuda', 0)
dims = ModelDims.qwen3_8b()
eng = GRPOEngine(dims, device=dev, seed=0)
dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=8), seq_len=2048)
batches = [[pack_micro_batches(dl.get_batch(), dev)] for _ in range(8)]
torch.cuda.synchronize()
for i, b in enumerate(batches):
    t0 = time.perf_counter(); eng.step(b); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"step {i}: enqueue {1e3*(t1-t0):.0f} ms  total {1e3*(t2-t0):.0f} ms  reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB  allocs {torch.cuda.memory_stats()['num_device_alloc']}", flush=True)
