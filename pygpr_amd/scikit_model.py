"""scikit-learn facade (reference: PyGPR/scikit_model.py:15-35)."""
from typing import Any

from sklearn.base import BaseEstimator, RegressorMixin
from torch import Tensor

from .gpr import GPR


class SK_WRAP(RegressorMixin, BaseEstimator):
    """Scikit model wrapper for GPR: fit stores x, y on the model, predict returns the mean."""

    def __init__(self, model: GPR) -> None:
        self.model: GPR = model
        return None

    def fit(self, x: Tensor, y: Tensor) -> Any:
        print("Fitting", x.shape, y.shape)
        self.model.x = x
        self.model.y = y
        return self

    def predict(self, xp: Tensor) -> Tensor:
        print("Predicting", xp.shape)
        self.need_upd = True
        yp, covar = self.model.predict(xp, var="none")
        return yp
