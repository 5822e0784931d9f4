"""Algorithm hook surface — mirrors `prime_rl.orchestrator.algo.base` (src/prime_rl/orchestrator/algo/base.py:35-81).
Objects are duck-typed (`episode.traces`, `trace.{has_error, agent.trainable, nodes, reward, num_output_tokens,
num_total_tokens, num_turns}`), so verifiers.v1 episodes plug in unchanged; `SimpleTrace` & co. are the minimal shapes used
by the tests (verifiers is an un-vendored submodule of the reference)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import ClassVar


@dataclass
class SimpleNode:
    token_ids: list[int]
    mask: list[bool]
    sampled: bool = True
    advantages: list[float] | None = None
    reference_logprobs: list[float] | None = None
    loss_weights: dict | None = None


@dataclass
class SimpleAgent:
    trainable: bool = True
    name: str = "agent"


@dataclass
class SimpleBranch:
    """A linear path through the trace graph as one flat token sequence — the slice of `verifiers.v1.Branch` that
    `trace_to_samples` reads (orchestrator/trajectories.py:111-160): `advantages` expands the nodes' compact sampled-token credit to
    one value per token, 0.0 where the mask is False."""

    nodes: list[SimpleNode]
    logprobs: list[float] | None = None
    reference_logprobs: list[float] | None = None
    multi_modal_data: object | None = None
    routed_experts: object | None = None

    @property
    def token_ids(self) -> list[int]:
        return [t for n in self.nodes for t in n.token_ids]

    @property
    def sampled_mask(self) -> list[bool]:
        return [m for n in self.nodes for m in n.mask]

    @property
    def advantages(self) -> list[float] | None:
        if all(n.advantages is None for n in self.nodes):
            return None
        out: list[float] = []
        for n in self.nodes:
            compact = iter(n.advantages or [])
            out += [next(compact, 0.0) if m else 0.0 for m in n.mask]
        return out


@dataclass
class SimpleTrace:
    id: str
    reward: float
    nodes: list[SimpleNode]
    has_error: bool = False
    agent: SimpleAgent = field(default_factory=SimpleAgent)
    num_output_tokens: int = 0
    num_total_tokens: int = 0
    num_turns: int = 1

    @property
    def branches(self) -> list[SimpleBranch]:
        """A trace without forks is one branch over all of its nodes."""
        return [SimpleBranch(self.nodes, logprobs=[0.0] * sum(len(n.token_ids) for n in self.nodes))]


@dataclass
class SimpleEpisode:
    traces: list[SimpleTrace]
    id: str = ""


def iter_trainable_traces(episodes):
    """Yield clean trainable traces that contain sampled tokens (base.py:35-42)."""
    for episode in episodes:
        for trace in episode.traces:
            if trace.has_error or not trace.agent.trainable:
                continue
            if any(any(node.mask) for node in trace.nodes):
                yield episode, trace


class Algorithm:
    """Override `score_episode` for rollout-local work and `score_group` for cohort-relative work (base.py:45-81)."""

    action_loss_type: ClassVar[str] = "rl"

    def __init__(self, config=None, policy_pool=None):
        self.config = config
        self.policy_pool = policy_pool
        self.connected_pools: list = []

    async def setup(self) -> None:
        pass

    async def score_episode(self, episode) -> None:
        pass

    async def score_group(self, episodes) -> None:
        pass

    async def finalize_episode(self, episode) -> None:
        if any(True for _ in iter_trainable_traces([episode])):
            await self.score_episode(episode)

    async def finalize_group(self, episodes) -> None:
        await self.score_group(episodes)
