"""CPU suite: the orchestrator-side advantage plugins on their DEFAULT placement — host arithmetic, like the reference's orchestrator
(src/prime_rl/orchestrator/algo/grpo.py:25-45, max_rl.py:22-28, hierarchical_grpo.py:33-42) — must give the same known answers as the
HIP kernel placement (`device="cuda"`, tests/test_seams_gpu.py): the very same test bodies run here with device="cpu".  Pins of the
reference's own tests: tests/unit/orchestrator/test_advantage.py:173-240."""

import asyncio
import importlib.util
from pathlib import Path

import pytest

_spec = importlib.util.spec_from_file_location("_seams_gpu_bodies", Path(__file__).parent / "test_seams_gpu.py")
_bodies = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bodies)


def test_grpo_and_maxrl_plugins_on_the_host():
    _bodies.test_grpo_algorithm_plugin(device="cpu")


def test_length_penalty_terms_on_the_host():
    _bodies.test_length_penalty_terms_like_the_reference(device="cpu")


def test_hierarchical_grpo_on_the_host():
    _bodies.test_hierarchical_grpo_peer_sets(device="cpu")


def test_default_placement_needs_no_accelerator():
    """The async `score_group` hook must not touch the GPU library by default (it runs on the orchestrator's event loop)."""
    from prime_rl_amd.orchestrator.algo import GRPOAlgoConfig, build_algorithm

    algo = build_algorithm(GRPOAlgoConfig())
    assert algo.device == "cpu"
    eps = _bodies._episode([1.0, 0.0])
    asyncio.run(algo.score_group(eps))
    assert [e.traces[0].nodes[0].advantages[0] for e in eps] == pytest.approx([0.5, -0.5])
