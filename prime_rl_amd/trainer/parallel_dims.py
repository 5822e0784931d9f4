"""`ParallelDims` / `get_parallel_dims` — the degrees the trainer runs with (src/prime_rl/trainer/parallel_dims.py:36-75,215-262,319-345).

Same fields, same validation rule (`dp_replicate * dp_shard * cp * pp == world_size`, `dp_shard = -1` = fill), same derived
properties.  The reference builds a torch DeviceMesh from it for FSDP2; here the degrees map onto plain process groups over
RCCL: `dp` = the ranks that hold different micro-batches (ZeRO-2 shards, trainer/zero.py), `cp` = the ranks that share one packed row
(Ulysses all-to-all, trainer/ulysses.py).  Rank layout: cp is the fastest-varying dimension (neighbouring GPUs, i.e. direct xGMI
links), so `dp_rank = rank // cp` exactly like `data.py:174-175` of the reference (`dp_rank = rank // non_dp_size`)."""

from __future__ import annotations

from dataclasses import dataclass, field

import torch.distributed as dist


@dataclass
class ParallelDims:
    dp_replicate: int
    dp_shard: int
    cp: int
    pp: int
    ep: int
    world_size: int
    _groups: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        for d in (self.dp_replicate, self.cp, self.pp, self.ep):
            if d < 1:
                raise ValueError("Parallelism degree should be >= 1, except for dp_shard")
        if not (self.dp_shard == -1 or self.dp_shard >= 1):
            raise ValueError("dp_shard must be -1 or >= 1")
        if self.dp_shard < 0:
            self.dp_shard = self.world_size // (self.dp_replicate * self.cp * self.pp)
        if self.dp_shard < 1 or self.dp_replicate * self.dp_shard * self.cp * self.pp != self.world_size:
            raise ValueError(f"Invalid parallel dims: dp_replicate({self.dp_replicate}) * dp_shard({self.dp_shard}) * cp({self.cp}) * pp({self.pp}) "
                             f"!= WORLD_SIZE({self.world_size})")
        if self.ep > 1 and not (self.ep % self.cp == 0 and (self.dp_shard * self.cp) % self.ep == 0):
            raise ValueError("ep must be a multiple of cp and divide dp_shard * cp")

    # -- derived (parallel_dims.py:221-262) --------------------------------------------------------------------------------
    @property
    def dp_size(self) -> int:
        return self.dp_replicate * self.dp_shard

    @property
    def dp_enabled(self) -> bool:
        return self.dp_replicate > 1 or self.dp_shard > 1

    @property
    def cp_enabled(self) -> bool:
        return self.cp > 1

    @property
    def dp_cp_enabled(self) -> bool:
        return self.dp_enabled or self.cp_enabled

    @property
    def fsdp_enabled(self) -> bool:
        return self.dp_shard > 1 or self.cp_enabled

    @property
    def pp_enabled(self) -> bool:
        return self.pp > 1

    @property
    def ep_enabled(self) -> bool:
        return self.ep > 1

    @property
    def fsdp_gradient_divide_factor(self) -> int:
        return self.dp_replicate * self.dp_shard * self.cp

    @property
    def non_data_parallel_size(self) -> int:
        return self.cp * self.pp

    @property
    def seq_len_divisor(self) -> int:
        return self.cp * 2

    # -- rank layout + process groups ------------------------------------------------------------------------------------------
    def dp_rank_of(self, rank: int) -> int:
        return rank // self.non_data_parallel_size

    def cp_rank_of(self, rank: int) -> int:
        return rank % self.cp

    def get_group(self, name: str):
        """"dp" (same cp position, different micro-batches), "cp" (one packed row), "dp_cp" (gradient reduction = everyone), "ep_replica" (same ep position in every ep group), "ep" (`ep`
        consecutive ranks of dp_cp that share one set of experts, parallel_dims.py:150-190 of the reference: ep borrows from dp_shard x cp).
        Every rank must call this for the same names in the same order (new_group is collective).  None = the default group / no group."""
        if name in self._groups:
            return self._groups[name]
        if not dist.is_initialized() or self.world_size == 1:
            self._groups[name] = None
            return None
        rank = dist.get_rank()
        mine = None
        if name == "dp_cp":
            mine = None  # the default group
        elif name == "cp":
            for d in range(self.dp_size):
                ranks = list(range(d * self.cp, (d + 1) * self.cp))
                g = dist.new_group(ranks) if self.cp > 1 else None
                if rank in ranks:
                    mine = g
        elif name == "dp":
            if self.cp == 1:
                mine = None
            else:
                for c in range(self.cp):
                    ranks = list(range(c, self.world_size, self.cp))
                    g = dist.new_group(ranks)
                    if rank in ranks:
                        mine = g
        elif name == "ep":
            if self.ep > 1:
                for first in range(0, self.world_size, self.ep):
                    ranks = list(range(first, first + self.ep))
                    g = dist.new_group(ranks)
                    if rank in ranks:
                        mine = g
        elif name == "ep_replica":  # the ranks that hold the SAME experts in different ep groups (they sum their expert gradients)
            if 1 < self.ep < self.world_size:
                for pos in range(self.ep):
                    ranks = list(range(pos, self.world_size, self.ep))
                    g = dist.new_group(ranks)
                    if rank in ranks:
                        mine = g
        else:
            raise KeyError(name)
        self._groups[name] = mine
        return mine


def resolve_ep(config, world_size: int) -> int:
    """`model.ep = "auto"` -> an integer, the reference's rule (parallel_dims.py:281-317): MoE models get min(world_size // dp_replicate, 8) — without it
    the default config would replicate every expert with its f32 master, moments and gradient on every rank (~18 B/param: Qwen3-30B-A3B does not fit a
    288 GB part that way) — dense models 1, and 1 when `model.impl` is not "custom" / "auto"; a non-torch `ep_comm_backend` with a resolved ep of 1
    is an error, as in the reference.  An explicit integer is returned unchanged."""
    if isinstance(config.ep, int):
        return config.ep
    import logging

    log = logging.getLogger("prime_rl_amd")
    impl = getattr(config, "impl", "auto")
    if impl not in ("custom", "auto"):  # parallel_dims.py:291-295: expert parallelism needs the custom implementation
        log.info("EP auto: impl='%s' does not support EP, resolving ep=1", impl)
        return 1
    from .model import resolve_moe_config  # local import: model.py imports this module

    if resolve_moe_config(config) is None:
        log.info("EP auto: model is not MoE, resolving ep=1")
        return 1
    ep = max(1, min(world_size // max(int(config.dp_replicate), 1), 8))
    log.info("EP auto: world_size=%d, dp_replicate=%d -> resolved ep=%d", world_size, config.dp_replicate, ep)
    backend = getattr(config, "ep_comm_backend", "torch")
    if backend != "torch" and ep <= 1:  # parallel_dims.py:311-316
        raise ValueError(f"model.ep_comm_backend='{backend}' requires ep > 1, but auto-resolved ep=1 (world_size={world_size}). "
                         "Set ep explicitly or use ep_comm_backend='torch'.")
    return ep


def get_parallel_dims(config, seq_len: int | None = None, world_size: int | None = None) -> ParallelDims:
    """parallel_dims.py:319-345: dp_shard fills whatever dp_replicate and cp leave of the world; `config` is a ModelConfig."""
    world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    ep = resolve_ep(config, world)
    dims = ParallelDims(dp_replicate=config.dp_replicate, dp_shard=-1, cp=config.cp, pp=1, ep=ep, world_size=world)
    if seq_len is not None and seq_len % dims.seq_len_divisor != 0:
        raise ValueError(f"Sequence length ({seq_len}) must be divisible by seq_len_divisor ({dims.seq_len_divisor}) for the given parallel "
                         f"dimensions. This requirement comes from context parallel (CP={config.cp}).")
    return dims
