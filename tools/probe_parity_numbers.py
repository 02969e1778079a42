"""Prints the measured errors behind the tolerances of tests/test_parity_gpu.py (default-hp gradient, get_learn_rate)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
from oracle import pygpr_oracle as orc
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gp.npz"))
r3 = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "round3.npz"))
cov = lambda: pg.Compose([pg.Squared_exponential(), pg.White_noise()])
for tag in ("a", "b"):
    x, y, hp0 = g[tag + "_x"], g[tag + "_y"], g[tag + "_hp0"]
    k = orc.kernel([orc.SE, orc.WN], hp0, x, form="direct") + 1e-7 * np.eye(x.shape[0])
    l0, g0 = pg.MLE(pg.Exact_GP(T(x), T(y), cov())).loss_and_grad(hp0.copy())
    lref = g[tag + "_loss0"] if tag == "a" else -3461.42170686
    gref = g[tag + "_grad0"]
    print("default hp", tag, "n", x.shape[0], "cond %.3e" % np.linalg.cond(k), "nlml rel", abs(float(l0) - float(lref)) / abs(float(lref)),
          "grad rel inf", np.abs(g0 - gref).max() / np.abs(gref).max(), "per comp", np.abs(g0 - gref) / np.abs(gref).max())
gp = pg.Exact_GP(T(g["a_x"]), T(g["a_y"]), cov())
for eps, ref in ((1e-6, g["a_gamma"]), (float(r3["lr_eps3"]), r3["lr_gamma3"]), (float(r3["lr_eps4"]), r3["lr_gamma4"])):
    gam = pg.get_learn_rate(T(g["a_hp"]), pg.MLE(gp), eps)
    print("learn rate eps", eps, "gamma", float(gam), "ref", float(ref), "rel", abs(float(gam) / float(ref) - 1))
