"""vLLM worker extension for in-place weight updates over RCCL — the MI355X counterpart of
`prime_rl.inference.vllm.worker.nccl.NCCLWeightUpdateWorker` (src/prime_rl/inference/vllm/worker/nccl.py:92-163).

Same RPC surface, so the reference's inference server drives it unchanged (inference/vllm/server.py:70-147 registers the class as
`worker_extension_cls`; utils/client.py:391-445 calls `/init_broadcaster` once and `/update_weights {weight_dir}` per step, the
orchestrator touching `NCCL_READY` in `weight_dir` right before):

    init_broadcaster(host, port, rank_offset, inference_world_size, timeout, quantize_in_weight_transfer=False, session_id="default")
    liveness_probe()
    update_weights_from_path(weight_dir)

Differences, all behind that surface:
  * the communicator is a raw RCCL communicator (`prime_rl_amd.comm.RCCLCommunicator`, C ABI `pm_comm_*`) bootstrapped from the same
    TCP rendezvous (`host:port`), not vllm's PyNcclCommunicator;
  * the push world holds ALL trainer ranks followed by the inference GPUs: inference GPU g is rank `n_trainers + rank_offset + g`
    (reference: trainer master = 0, GPU g = `rank_offset + g + 1`, i.e. the same formula with one trainer).  `n_trainers` travels as an
    extra keyword of `/init_broadcaster` (default: PRIME_RL_N_TRAINERS or 1);
  * the default wire protocol is the sharded multi-source push (transports/weights: every trainer rank sends its slice of every layer
    bucket straight to this GPU over its own xGMI link; `LAYOUT.json` in `weight_dir` names the tensors).  With
    `protocol="reference"` (PRIME_PUSH_PROTOCOL) it receives the reference's byte stream instead: count, pickled
    {dtype: [(key, shape, numel)]} per state dict, one flat tensor per dtype (inference/vllm/worker/nccl.py:28-61).
Either way `(hf_name, tensor)` pairs stream into the model in checkpoint order, layer by layer, through vLLM's layer-wise reload
(`initialize_layerwise_reload -> model.load_weights(iter) -> finalize_layerwise_reload`, worker/weight_transfer.py:13-25) when
vLLM is importable, and through `model.load_weights(iter)` alone otherwise.
"""

from __future__ import annotations

import os
from pathlib import Path

import torch

from ....transports.weights import RCCLWeightReceiver, init_push_communicator


def load_weights_checkpoint_layerwise(model, state_iter, model_config=None, vllm_config=None) -> None:
    """weight_transfer.py:13-25; degrades to a plain `load_weights` outside a vLLM process."""
    try:
        from vllm.config import set_current_vllm_config
        from vllm.model_executor.model_loader.reload import finalize_layerwise_reload, initialize_layerwise_reload
    except ImportError:
        model.load_weights(state_iter)
        return
    device = next(model.parameters()).device
    with torch.device(device), set_current_vllm_config(vllm_config):
        initialize_layerwise_reload(model)
        model.load_weights(state_iter)
        finalize_layerwise_reload(model, model_config)


class RCCLWeightUpdateWorker:
    """Mixed into vLLM's `Worker` by `worker_extension_cls` (it reads `self.device`, `self.model_runner`, `self.vllm_config`)."""

    def init_broadcaster(self, host: str, port: int, rank_offset: int, inference_world_size: int, timeout: int,
                         quantize_in_weight_transfer: bool = False, session_id: str = "default", n_trainers: int | None = None,
                         protocol: str | None = None) -> None:
        del session_id
        if quantize_in_weight_transfer:
            raise NotImplementedError("FP8 weight transfer is outside the MI355X GRPO hot path (SURVEY.md section 2)")
        self.n_trainers = int(n_trainers if n_trainers is not None else os.environ.get("PRIME_RL_N_TRAINERS", "1"))
        self.push_protocol = protocol or os.environ.get("PRIME_PUSH_PROTOCOL", "sharded")
        local_rank = self.device.index  # the worker's device index is its local rank (nccl.py:110-116)
        global_rank_inference = rank_offset + local_rank
        comm = init_push_communicator(host, port, self.n_trainers + global_rank_inference, self.n_trainers, inference_world_size, self.device,
                                      timeout, self.push_protocol)
        self.rccl_weight_receiver = RCCLWeightReceiver(comm, self.n_trainers, self.device, self.push_protocol)

    def liveness_probe(self) -> None:
        """No-op RPC used by the API server liveness endpoint."""
        return None

    def update_weights_from_path(self, weight_dir: str) -> None:
        runner = self.model_runner
        model = runner.model.runnable if hasattr(runner.model, "runnable") else runner.model
        state_iter = self.rccl_weight_receiver.receive_state_dict(Path(weight_dir))
        load_weights_checkpoint_layerwise(model, state_iter, getattr(runner, "model_config", None), getattr(self, "vllm_config", None))
