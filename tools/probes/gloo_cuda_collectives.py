"""Probe: which collectives does the gloo backend run on CUDA tensors when two ranks share one GPU?  (Used to decide whether the multi-rank engine path
can be exercised with the real HIP kernels on a one-GPU box.)"""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    res = {}
    x = torch.ones(8, device=dev) * (rank + 1)
    for name, fn in (("all_reduce", lambda: dist.all_reduce(x.clone())),
                     ("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(torch.empty(4, device=dev), x.clone())),
                     ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(16, device=dev), x.clone())),
                     ("all_to_all_single", lambda: dist.all_to_all_single(torch.empty(8, device=dev), x.clone())),
                     ("broadcast", lambda: dist.broadcast(x.clone(), 0)),
                     ("bf16 reduce_scatter", lambda: dist.reduce_scatter_tensor(torch.empty(4, device=dev, dtype=torch.bfloat16), x.clone().bfloat16()))):
        try:
            fn()
            torch.cuda.synchronize()
            res[name] = "ok"
        except Exception as e:  # noqa: BLE001
            res[name] = "FAIL " + repr(e)[:120]
    if rank == 0:
        for k, v in res.items():
            print(k, v)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(2, 29611), nprocs=2, join=True)
