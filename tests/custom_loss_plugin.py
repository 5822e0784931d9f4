"""A `[trainer.loss] type = "custom"` plugin for the tests: the reference's default loss restated in plain torch on the LossFn contract
(LossInputs -> LossOutputs, loss.py:13-45,109-161), with a `tau` kwarg and one metric of its own."""

import torch

from prime_rl_amd.trainer.rl.loss import LossInputs, LossOutputs


def torch_default_loss(inputs: LossInputs, dppo_mask_low: float = 0.2, dppo_mask_high: float = 0.2, adv_tau: float = 1.0, kl_tau: float = 1e-3) -> LossOutputs:
    lp, li, adv, mask = inputs.trainer_logprobs, inputs.inference_logprobs, inputs.advantages, inputs.loss_mask
    log_ratio = lp - li
    ratio = torch.exp(log_ratio)
    mismatch_kl = ratio - log_ratio - 1
    prob_diff = torch.exp(lp) - torch.exp(li)
    invalid = torch.where(adv > 0, prob_diff > dppo_mask_high, prob_diff < -dppo_mask_low)
    is_masked = invalid & mask
    keep = mask & ~invalid
    pg = keep * (adv_tau * adv) * ratio
    kl = mask * log_ratio**2
    loss = (-pg + kl_tau * kl).sum()
    denom = mask.sum().clamp(min=1)
    return LossOutputs(loss=loss, metrics={"mismatch_kl": (mismatch_kl * mask).sum() / denom, "is_masked": is_masked.sum() / denom,
                                           "plugin_tokens": mask.sum().float()})
