"""Config surface of the reference (packages/prime-rl-configs/src/prime_rl/configs) for the trainer hot path."""

from .trainer import TrainerConfig, parse_argv  # noqa: F401
