"""Packing cases WITH router-replay payloads, shared by make_packing_routed_golden.py (runs the REFERENCE on them) and tests/test_batch_cpu.py."""

import hashlib
import json
import random

import numpy as np

CASES = [  # (n_samples, seq_len, workers, pad_to_multiple_of, max_sample_len, layers, top_k, dtype, routed_every)
    (12, 64, 2, 8, 70, 2, 2, "uint8", 1), (21, 128, 3, 16, 150, 3, 4, "int16", 2), (9, 96, 1, 1, 60, 1, 8, "int32", 3), (30, 256, 4, 64, 300, 2, 2, "uint8", 1)]
FIELDS = ("input_ids", "loss_mask", "advantages", "inference_logprobs", "position_ids", "sequence_lengths", "temperatures", "env_names", "seq_lens")


def make_samples(case_id: int) -> list[dict]:
    """Plain dicts; `routed_experts` (when present) = dict(data=bytes, shape=[tokens, layers, top_k], dtype=str) — wrap it in the RoutedExperts type."""
    n_samples, _, _, _, max_len, layers, top_k, dtype, every = CASES[case_id]
    rng = random.Random(1000 + case_id)
    nrng = np.random.default_rng(1000 + case_id)
    samples = []
    for i in range(n_samples):
        ln = rng.randint(1, max_len)
        kw = {}
        if i % every == 0:  # every > 1: samples with and without a payload in one batch — they must not share a micro-batch
            arr = nrng.integers(0, 64, size=(ln, layers, top_k)).astype(dtype)
            kw["routed_experts"] = dict(data=arr.tobytes(), shape=[ln, layers, top_k], dtype=dtype)
        samples.append(dict(token_ids=[rng.randrange(1000) for _ in range(ln)], mask=[rng.random() < 0.7 for _ in range(ln)],
                            logprobs=[-rng.random() for _ in range(ln)], temperatures=[1.0] * ln, env_name=f"env{i % 2}",
                            advantages=[rng.random() - 0.5 for _ in range(ln)], **kw))
    return samples


def digest_grid(grid) -> list[list[dict]]:
    out = []
    for rank in grid:
        row = []
        for mb in rank:
            blob = json.dumps({k: getattr(mb, k) for k in FIELDS}, sort_keys=True)
            r = mb.routed_experts
            routed = None if r is None else dict(shape=list(r.shape), dtype=r.dtype, sha1=hashlib.sha1(bytes(r.data)).hexdigest())
            row.append(dict(n=len(mb.input_ids), sequence_lengths=list(mb.sequence_lengths), sha1=hashlib.sha1(blob.encode()).hexdigest(), routed=routed))
        out.append(row)
    return out
