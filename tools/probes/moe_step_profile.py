import os, sys, torch
sys.path.insert(0, os.getcwd())
from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
from prime_rl_amd.trainer.module_engine import ModuleEngine
from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig
DEV = "cuda:0"
cfg = Qwen3MoeConfig(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=4,
                     num_experts=128, num_experts_per_tok=8, moe_intermediate_size=768, norm_topk_prob=True)
with torch.device(DEV):
    model = Qwen3MoeForCausalLM(cfg)
model = model.to(torch.bfloat16)
gen = torch.Generator(device=DEV).manual_seed(0)
with torch.no_grad():
    for name, p in model.named_parameters():
        p.fill_(1.0) if p.dim() == 1 else p.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.02).to(p.dtype))
inject_prime_lm_head(model, chunk_size=8192)
eng = ModuleEngine(model, DEV, optim=OptimConfig())
dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=8), seq_len=2048)
batches = [[pack_micro_batches(dl.get_batch(), DEV)] for _ in range(3)]
eng.step(batches[0]); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.step(batches[1]); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
glue = [e for e in prof.key_averages(group_by_stack_n=6) if e.key in ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::zero_", "aten::mul", "aten::cat", "aten::index_select", "aten::clone") and e.self_device_time_total > 100]
glue.sort(key=lambda e: -e.self_device_time_total)
for e in glue[:24]:
    print(f"{e.key:18s} {e.self_device_time_total / 1e3:8.3f} ms  x{e.count:3d}  " + " <- ".join(f.strip().split('/')[-1] for f in e.stack[:5]))
