"""debug: world-4 shared-GPU EP module-engine step (tests/test_multirank_one_gpu.py::_moe_worker) with per-rank logs"""
import faulthandler, os, sys, traceback
import torch.multiprocessing as mp
sys.path.insert(0, os.getcwd())

def w(rank, world, port):
    faulthandler.enable(open(f"gpurun_out/ep4_rank{rank}.log", "w"))
    try:
        from tests.test_multirank_one_gpu import _moe_worker
        _moe_worker(rank, world, port, "")
        open(f"gpurun_out/ep4_rank{rank}.ok", "w").write("ok")
    except BaseException:
        open(f"gpurun_out/ep4_rank{rank}.err", "w").write(traceback.format_exc())
        raise

if __name__ == "__main__":
    mp.start_processes(w, args=(4, 29777), nprocs=4, join=True, start_method="spawn")
