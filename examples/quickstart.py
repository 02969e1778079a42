"""Drop-in use of the MI355X path behind PyGPR's class surface (needs one GPU and the built library:
`python __graft_entry__.py`).  The only change from a PyGPR script is the import line.

    python examples/quickstart.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as PyGPR          # instead of: import PyGPR

torch.manual_seed(0)
n, d, m = 2000, 4, 500
x = torch.rand(n, d, dtype=torch.float64)
y = torch.sin(-x.sum(-1)) + 0.05 * torch.randn(n, dtype=torch.float64)
xs = torch.rand(m, d, dtype=torch.float64)

# ---- exact GP: fit, hyper-parameter training with the stock CG driver, prediction
cov = PyGPR.Compose([PyGPR.Squared_exponential(), PyGPR.White_noise()])
gp = PyGPR.Exact_GP(x, y, cov)
gp.set_params(torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64))
loss = PyGPR.MLE(gp)
print("NLML before training:", float(loss.loss(gp.params.numpy())))
opt = PyGPR.CG(loss)
opt.args.update(maxiter=20, disp=False)
opt.minimize()                       # scipy CG on loss.loss_and_grad; writes the optimum back into gp (prints
                                     # "Optimizer Failed" when it stops at maxiter, as the reference does: opt.py:61-65)
print("NLML after %d evaluations: %.4f" % (opt.res.nfev, float(opt.res.fun)))
mean, var = gp.predict(xs, var="diag")
print("test RMSE %.4f, mean predictive std %.4f" % (float((mean - torch.sin(-xs.sum(-1))).pow(2).mean().sqrt()),
                                                    float(var.sqrt().mean())))

# ---- grBCM committee: 4 local experts + a global communication set, shared hyper-parameters
nc, nls, ng = 4, 400, 200
xl, yl = x[: nc * nls].reshape(nc, nls, d), y[: nc * nls].reshape(nc, nls)
xg, yg = x[nc * nls: nc * nls + ng], y[nc * nls: nc * nls + ng]
committee = PyGPR.GRBCM(xl, yl, xg, yg, cov)
committee.set_params(gp.params)
mu_c, var_c = committee.predict(xs, var="diag")
print("committee vs exact GP: max |mean difference| %.4f" % float((mu_c - mean).abs().max()))

# ---- scikit-learn facade
sk = PyGPR.SK_WRAP(PyGPR.Exact_GP(x, y, cov))
sk.model.set_params(gp.params)
print("R^2 on the test points: %.4f" % sk.score(xs, torch.sin(-xs.sum(-1))))
if os.path.exists("opt.dat"):       # CG's callback log (opt.py:69-78), as in the reference
    os.remove("opt.dat")
