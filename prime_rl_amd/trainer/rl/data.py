"""Micro-batch sources for the trainer — mirrors `prime_rl.trainer.rl.data` (src/prime_rl/trainer/rl/data.py).

`TensorMicroBatch` keeps the reference's field names and shapes ([1, seq] CPU tensors; data.py:18-52) so the engine consumes
what `DataLoader.get_batch()` hands over.  `FakeDataLoader` reproduces the reference's synthetic batches value-for-value
(same torch.Generator seeds and draw order, data.py:55-159): it is the input of the benchmark and of the parity tests
("identical rollout batches").
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import TypedDict

import numpy as np
import torch
from torch import Tensor


class TensorMicroBatch(TypedDict, total=False):
    input_ids: Tensor  # i64 [1, seq]
    position_ids: Tensor  # i64 [1, seq]
    advantages: Tensor  # f32 [1, seq]
    inference_logprobs: Tensor  # f32 [1, seq]
    ref_logprobs: Tensor | None
    loss_mask: Tensor  # bool [1, seq]
    temperatures: Tensor  # f32 [1, seq]
    env_names: list[str]
    sequence_lengths: list[int]
    lora_num_tokens: Tensor
    seq_lens: Tensor  # i64 [docs]
    routed_experts: Tensor | None
    mm_kwargs: dict | None
    mm_token_type_ids: Tensor | None
    rl_weights: Tensor | None
    ce_weights: Tensor | None
    ref_kl_weights: Tensor | None


@dataclass
class FakeDataLoaderConfig:
    """FakeDataLoaderConfig (packages/prime-rl-configs/src/prime_rl/configs/trainer.py): global batch size in micro-batches."""

    batch_size: int = 2
    generate_samples: bool = False


class FakeDataLoader:
    """Deterministic synthetic micro-batches; seed = dp_rank*1_000_000 + batch_counter*1000 + micro_idx (data.py:79-82)."""

    def __init__(self, config: FakeDataLoaderConfig, seq_len: int, dp_world_size: int = 1, dp_rank: int = 0):
        self.dp_rank = dp_rank
        self.num_micro_batches = config.batch_size // dp_world_size
        self.seq_len = seq_len
        self.generate_samples = config.generate_samples
        self.batch_counter = 0

    def wait_for_batch(self) -> None:
        return

    def get_batch(self) -> list[TensorMicroBatch]:
        make = self._packed_documents if self.generate_samples else self._single_document
        out = []
        for micro_idx in range(self.num_micro_batches):
            gen = torch.Generator().manual_seed(self.dp_rank * 1_000_000 + self.batch_counter * 1000 + micro_idx)
            out.append(make(gen))
        self.batch_counter += 1
        return out

    @staticmethod
    def _finish(ids: Tensor, pos: Tensor, doc_lens: list[int], gen: torch.Generator) -> TensorMicroBatch:
        n = ids.numel()
        adv = torch.randn(n, generator=gen)  # draw order matters: advantages before inference_logprobs
        inf = torch.randn(n, generator=gen)
        return TensorMicroBatch(
            input_ids=ids.view(1, n), position_ids=pos.view(1, n), advantages=adv.view(1, n), inference_logprobs=inf.view(1, n), ref_logprobs=None,
            temperatures=torch.ones(1, n), env_names=["fake"] * n, sequence_lengths=list(doc_lens), loss_mask=torch.ones(1, n, dtype=torch.bool),
            lora_num_tokens=torch.tensor([n], dtype=torch.int32), seq_lens=torch.tensor(doc_lens, dtype=torch.long), routed_experts=None,
            mm_kwargs=None, mm_token_type_ids=None, rl_weights=None, ce_weights=None, ref_kl_weights=None,
        )

    def _single_document(self, gen: torch.Generator) -> TensorMicroBatch:
        ids = torch.randint(0, 100, (1, self.seq_len), generator=gen).view(-1)
        return self._finish(ids, torch.arange(self.seq_len), [self.seq_len], gen)

    def _packed_documents(self, gen: torch.Generator) -> TensorMicroBatch:
        total, ids, pos, lens = 0, [], [], []
        while total < self.seq_len:
            ln = int(torch.randint(1, self.seq_len // 8, (1,), generator=gen).item())
            ln = min(ln, self.seq_len - total)
            total += ln
            lens.append(ln)
            ids.append(torch.randint(0, 120000, (ln,), generator=gen).long())
            pos.append(torch.arange(ln))
        return self._finish(torch.cat(ids), torch.cat(pos), lens, gen)


class DataLoader:
    """Receives packed micro batches from the orchestrator, one stream per DP rank (data.py:162-238): list fields -> `[1, seq]`
    CPU tensors with the reference's dtypes."""

    def __init__(self, output_dir, start_step: int, dp_world_size: int, config=None, dp_rank: int = 0):
        from ...transports.rollouts import setup_micro_batch_receiver
        from ...transports.rollouts.binary import BinaryStepReceiver

        # `format = "binary"` selects the raw-array step blobs + pinned read-ahead (transports/rollouts/binary.py);
        # the default is the reference's msgpack-of-lists wire
        self.binary = getattr(config, "format", "msgpack") == "binary"
        self.receiver = BinaryStepReceiver(output_dir, dp_rank, start_step) if self.binary else setup_micro_batch_receiver(output_dir, dp_rank, start_step, config)
        self.last_blob = None

    def wait_for_batch(self) -> None:
        self.receiver.wait()

    def get_batch(self) -> list[TensorMicroBatch]:
        if self.binary:
            from ...transports.rollouts.binary import blob_to_tensor_micro_batches

            self.last_blob = self.receiver.receive()
            return blob_to_tensor_micro_batches(self.last_blob)
        return [self._micro_batch_to_tensor(mb) for mb in self.receiver.receive()]

    def get_packed_batch(self, device):
        """Binary wire only: the whole step as one device-resident `PackedBatch` (one H2D copy), plus the host-side micro-batch view."""
        from ...transports.rollouts.binary import blob_to_packed_batch

        if not self.binary:
            raise RuntimeError("get_packed_batch needs the binary rollout wire (config.format = 'binary')")
        mbs = self.get_batch()
        return blob_to_packed_batch(self.last_blob, device), mbs

    @staticmethod
    def _micro_batch_to_tensor(mb) -> TensorMicroBatch:
        def opt(v, dtype):
            return torch.tensor(v, dtype=dtype).unsqueeze(0) if v is not None else None

        if mb.mm_kwargs:
            raise NotImplementedError("multimodal payloads are outside the GRPO hot path (SURVEY.md section 2 row 12)")
        routed = None
        if mb.routed_experts is not None:  # router replay (data.py:197-208): raw bytes -> i32 [1, tokens, layers, top_k]
            r = mb.routed_experts
            routed = torch.from_numpy(np.frombuffer(r.data, dtype=np.dtype(r.dtype)).reshape(r.shape).astype(np.int32)).unsqueeze(0)
        return TensorMicroBatch(
            input_ids=torch.tensor(mb.input_ids, dtype=torch.long).unsqueeze(0), position_ids=torch.tensor(mb.position_ids, dtype=torch.long).unsqueeze(0),
            advantages=torch.tensor(mb.advantages, dtype=torch.float).unsqueeze(0),
            inference_logprobs=torch.tensor(mb.inference_logprobs, dtype=torch.float).unsqueeze(0), ref_logprobs=opt(mb.ref_logprobs, torch.float),
            loss_mask=torch.tensor(mb.loss_mask, dtype=torch.bool).unsqueeze(0), temperatures=torch.tensor(mb.temperatures, dtype=torch.float).unsqueeze(0),
            env_names=mb.env_names, sequence_lengths=mb.sequence_lengths, lora_num_tokens=torch.tensor([len(mb.input_ids)], dtype=torch.int32),
            seq_lens=torch.tensor(mb.seq_lens, dtype=torch.long), mm_kwargs=None, mm_token_type_ids=None, routed_experts=routed,
            rl_weights=opt(mb.rl_weights, torch.float), ce_weights=opt(mb.ce_weights, torch.float), ref_kl_weights=opt(mb.ref_kl_weights, torch.float),
        )
