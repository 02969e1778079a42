"""Learning-rate probe (reference: PyGPR/hp_update.py:6-28): one loss_and_grad and two loss
evaluations along -grad give a quadratic-fit step size  gamma = -1/2 C1 / C2."""
import torch
from torch import Tensor

from .loss import Loss


def get_learn_rate(current_param: Tensor, loss_new: Loss, eps: float) -> float:
    old = torch.clone(current_param).numpy()
    f0, jac = loss_new.loss_and_grad(old)
    f_plus = loss_new.loss(old - eps * jac)
    f_minus = loss_new.loss(old + eps * jac)
    c1 = (f_plus - f_minus) / (2.0 * eps)
    c2 = (f_plus + f_minus - 2 * f0) / (2.0 * eps ** 2)
    return -0.5 * (c1 / c2)
