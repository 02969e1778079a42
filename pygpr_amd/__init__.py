"""pygpr_amd -- MI355X-native dense Gaussian-process hot path behind PyGPR's class surface
(reference export list: PyGPR/__init__.py:1-7; `get_learn_rate`, `Matern52` and `GRBCM_MLE` are added)."""
from .gpr import GPR, Exact_GP
from .covar import Squared_exponential, Matern52, Covar, Compose, White_noise
from .loss import Loss, MLE
from .opt import Opt, CG, Nelder_Mead, BFGS_Quad, CG_Quad, hessian
from .gr_bcm import GRBCM, GRBCM_MLE, log_likelihood_batched
from .hp_update import get_learn_rate
from .scikit_model import SK_WRAP
from .sampler import UNIFORM, MATERN1, sample_gp, cluster_samples, euclidean_dist

__all__ = [
    "GPR", "Exact_GP", "Squared_exponential", "Matern52", "Covar", "Compose", "White_noise", "Loss", "MLE", "Opt",
    "CG", "Nelder_Mead", "BFGS_Quad", "CG_Quad", "hessian", "GRBCM", "GRBCM_MLE", "get_learn_rate", "SK_WRAP",
    "UNIFORM", "MATERN1", "cluster_samples", "euclidean_dist", "sample_gp", "log_likelihood_batched",
]
