"""Run-directory layout shared by orchestrator, trainer and inference — mirrors `prime_rl.utils.pathing`
(src/prime_rl/utils/pathing.py:101-114, 217-230): `rollouts/step_{k}/rank_{r}.bin`, `broadcasts/step_{n}/{STABLE,NCCL_READY}`."""

from __future__ import annotations

import time
from pathlib import Path


def get_rollout_dir(output_dir: Path) -> Path:
    return Path(output_dir) / "rollouts"


def get_broadcast_dir(output_dir: Path) -> Path:
    return Path(output_dir) / "broadcasts"


def get_step_path(path: Path, step: int) -> Path:
    return Path(path) / f"step_{step}"


def sync_wait_for_path(path: Path, interval: float = 1.0, timeout: float | None = None) -> None:
    waited = 0.0
    while not Path(path).exists():
        time.sleep(interval)
        waited += interval
        if timeout is not None and waited >= timeout:
            raise TimeoutError(f"timed out waiting for {path}")
