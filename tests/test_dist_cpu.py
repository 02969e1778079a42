"""gloo runs of the multi-GPU path's host side: experts sharded over ranks, ONE all-reduce of the [3, m] aggregation
buffer (+ one status word per rank) per test batch and ONE of [1 + nhp] (+ one status word per rank) per shared-hp evaluation.  Device ops are the
oracle-backed test double (tests/oracle_ops.py); the result must equal the single-process golden output."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir, cases=(0, 1, 2)):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pygpr_amd as pg
    from pygpr_amd import _ops
    from oracle_ops import OracleOps

    _ops._OPS = OracleOps()
    g = np.load(os.path.join(HERE, "golden", "grbcm.npz"))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    res = {}
    from pygpr_amd.gr_bcm import expert_block

    for case in cases:              # nc = 2, 4, 8 experts over the ranks (uneven blocks, and ranks without an expert, for world = 3)
        p = "g%d_" % case
        cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
        m = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), cov, distributed=True)
        nc = g[p + "xl"].shape[0]
        assert m.world == world and (m.lo, m.hi) == expert_block(nc, rank, world)
        assert (m.gpl is None) == (m.hi == m.lo)
        res["nloc%d" % case] = np.array(m.hi - m.lo)
        m.gpg.set_params(T(g[p + "hpg"]))
        m.set_local_params(T(g[p + "hpl"]))
        mu, var = m.predict(T(g[p + "xs"]), var="diag")
        loss, grad = pg.GRBCM_MLE(m).loss_and_grad(g[p + "hpg"].copy())
        res["mu%d" % case], res["var%d" % case] = mu.numpy(), var.numpy()
        res["loss%d" % case], res["grad%d" % case] = np.array(loss), grad
        # full-covariance committee over the ranks (shared hp as in the fixture): the [3, m] all-reduce, then the
        # [m_pad, m_pad] weighted-precision all-reduce with the 1/world padding diagonal
        m.set_params(T(g[p + "hpg"]))
        mu_f, cov_f = m.predict(T(g[p + "xs"]), var="full")
        res["muf%d" % case], res["covf%d" % case] = mu_f.numpy(), cov_f.numpy()
    # a non-positive-definite expert (the last one) on ONE rank only: every rank must raise, none may hang in the collective
    p = "g1_"
    xl = g[p + "xl"].copy()
    xl[-1, 3, 0] = np.nan
    owner = [r for r in range(world) if expert_block(xl.shape[0], r, world)[0] <= xl.shape[0] - 1 < expert_block(xl.shape[0], r, world)[1]][0]
    bad = pg.GRBCM(T(xl), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), pg.Compose([pg.Squared_exponential(), pg.White_noise()]),
                   distributed=True)
    bad.set_params(T(g[p + "hpg"]))
    raised = []
    for call in (lambda: bad.predict(T(g[p + "xs"]), var="diag"), lambda: pg.GRBCM_MLE(bad).loss_and_grad(g[p + "hpg"].copy()),
                 lambda: bad.predict(T(g[p + "xs"]), var="full")):
        try:
            call()
            raised.append(0)
        except torch.linalg.LinAlgError as err:
            raised.append(2 if "reported by rank %d" % owner in str(err) else 1)
    res["raised"] = np.array(raised)
    res["owner"] = np.array(owner)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def _run_and_check(tmp_path, golden, world, cases):
    from oracle import pygpr_oracle as orc

    port = 29500 + (os.getpid() + 7 * world) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), cases), nprocs=world, join=True)
    g = golden("grbcm")
    outs = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    for case in cases:
        p = "g%d_" % case
        assert sum(int(o["nloc%d" % case]) for o in outs) == g[p + "xl"].shape[0]
        for o in outs:                                  # every rank finishes with the full committee answer
            np.testing.assert_allclose(o["mu%d" % case], g[p + "mu"], atol=1e-10)
            np.testing.assert_allclose(o["var%d" % case], g[p + "var"], atol=1e-11)
        x, y = orc.grbcm_data(g[p + "xl"], g[p + "yl"], g[p + "xg"], g[p + "yg"])
        ref = [orc.mle_loss_and_grad([orc.SE, orc.WN], g[p + "hpg"], x[c], y[c], "kinv") for c in range(x.shape[0])]
        for o in outs:
            np.testing.assert_allclose(o["loss%d" % case], sum(r[0] for r in ref), rtol=1e-10)
            np.testing.assert_allclose(o["grad%d" % case], sum(r[1] for r in ref), rtol=1e-8, atol=1e-8)
            np.testing.assert_allclose(o["covf%d" % case], g[p + "cov_full"], rtol=1e-6, atol=1e-10)
            np.testing.assert_allclose(o["muf%d" % case], g[p + "mu_full"], rtol=1e-6, atol=1e-9)
    # failure on one rank: it raises its own error (1), every other rank the relayed one naming it (2) -- in all three calls
    owner = int(outs[0]["owner"])
    for r, o in enumerate(outs):
        assert o["raised"].tolist() == ([1, 1, 1] if r == owner else [2, 2, 2]), (r, owner, o["raised"])
    return outs


def test_grbcm_two_ranks_gloo(tmp_path, golden):
    """2 ranks: nc = 2 (1 each), 4 (2 each), 8 (2 x 4: the third layout of the scaling curve)."""
    outs = _run_and_check(tmp_path, golden, 2, (0, 1, 2))
    assert [int(o["nloc2"]) for o in outs] == [4, 4]


def test_grbcm_three_ranks_uneven_blocks_and_an_empty_rank(tmp_path, golden):
    """world = 3: nc = 8 gives blocks of 3, 3, 2 experts; nc = 2 leaves rank 2 WITHOUT a local expert (gpl = None: it still
    holds the global expert, joins every all-reduce with zero terms and finishes with the committee's answer); the failing
    expert of the nc = 4 case sits on the middle rank and the last rank owns nothing."""
    outs = _run_and_check(tmp_path, golden, 3, (0, 2))
    assert [int(o["nloc0"]) for o in outs] == [1, 1, 0] and [int(o["nloc2"]) for o in outs] == [3, 3, 2]
    assert int(outs[0]["owner"]) == 1


def test_grbcm_eight_ranks_one_expert_each(tmp_path, golden):
    """The layout north_star names: 8 ranks x 1 expert (nc = 8) -- diag and full-covariance committee and the shared-hp objective
    end on every rank with the single-process golden answer; in the failing-expert part (nc = 4 on 8 ranks) four ranks own no
    expert at all and the owner is rank 3."""
    outs = _run_and_check(tmp_path, golden, 8, (2,))
    assert [int(o["nloc2"]) for o in outs] == [1] * 8 and int(outs[0]["owner"]) == 3


def test_grbcm_four_ranks_two_experts_each(tmp_path, golden):
    """4 ranks x 2 experts (nc = 8: each rank's local experts take the experts-together path of Exact_GP / MLE) and 4 x 1 (nc = 4)."""
    outs = _run_and_check(tmp_path, golden, 4, (2, 1))
    assert [int(o["nloc2"]) for o in outs] == [2] * 4 and [int(o["nloc1"]) for o in outs] == [1] * 4


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): bench.py spawns the ranks itself, they rendezvous,
    all-gather their rank numbers and time the [1 + nhp + 1] all-reduce; rank 0's JSON line is relayed."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rep = json.loads(line)["dist"]
    assert rep["world"] == 2 and rep["ranks"] == [0, 1] and rep["backend"] == "gloo" and rep["allreduce_us"] > 0
