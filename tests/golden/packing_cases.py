"""Deterministic packing test cases shared by make_golden.py (which runs the REFERENCE on them) and tests/test_batch_cpu.py."""

import hashlib
import json
import random

CASES = [  # (n_samples, seq_len, workers, pad_to_multiple_of, max_sample_len, use_model_cost, with_weight_streams)
    (16, 64, 2, 1, 70, False, False), (37, 128, 4, 8, 150, True, False), (9, 256, 8, 16, 300, True, True), (50, 512, 3, 1, 400, False, True),
    (5, 32, 1, 4, 40, True, False), (64, 2048, 8, 64, 1500, True, False)]
FIELDS = ("input_ids", "loss_mask", "advantages", "inference_logprobs", "position_ids", "sequence_lengths", "temperatures", "env_names", "seq_lens",
          "ref_logprobs", "rl_weights", "ce_weights", "ref_kl_weights")


def make_samples(case_id: int) -> list[dict]:
    n_samples, _, _, _, max_len, _, streams = CASES[case_id]
    rng = random.Random(case_id)
    samples = []
    for i in range(n_samples):
        ln = rng.randint(1, max_len)
        kw = {}
        if streams and i % 3 == 0:
            kw["ce_weights"] = [float(rng.random() < 0.3) for _ in range(ln)]
        if streams and i % 4 == 1:
            kw["ref_logprobs"] = [-rng.random() for _ in range(ln)]
        samples.append(dict(token_ids=[rng.randrange(1000) for _ in range(ln)], mask=[rng.random() < 0.7 for _ in range(ln)],
                            logprobs=[-rng.random() for _ in range(ln)], temperatures=[1.0] * ln, env_name=f"env{i % 3}",
                            advantages=[rng.random() - 0.5 for _ in range(ln)], **kw))
    return samples


def digest_grid(grid) -> list[list[dict]]:
    """Per micro-batch: token count, sample lengths and a sha1 over every wire field (exact float repr)."""
    out = []
    for rank in grid:
        row = []
        for mb in rank:
            blob = json.dumps({k: getattr(mb, k) for k in FIELDS}, sort_keys=True)
            row.append(dict(n=len(mb.input_ids), sequence_lengths=list(mb.sequence_lengths), sha1=hashlib.sha1(blob.encode()).hexdigest()))
        out.append(row)
    return out
