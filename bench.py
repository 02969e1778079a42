#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): NLML+gradient evaluations/sec at N=16384, D=8, RBF + white noise, fp64.

    python bench.py --gpus N --steps K --warmup W

N > 1 from a bare shell: bench.py starts its own N ranks (a child `python -m torch.distributed.run`, spawned before
anything touches the GPU) and relays rank 0's JSON line; under a launcher that already set WORLD_SIZE (the driver's
`python -m torch.distributed.run ... bench.py --gpus N`) it is simply one of the ranks.

One "step" = every rank evaluates `loss_and_grad` (PyGPR/loss.py:92-128) once for ITS expert -- covariance
build, Cholesky, L^-1, alpha, K^-1, fused gradient contraction at n = 16384, d = 8 -- followed, for N > 1, by
the shared-hyper-parameter all-reduce(sum) of [1 + nhp (+ status)] doubles over RCCL (grBCM co-training, GRBCM_MLE).
At N = 1 that is exactly one exact-GP NLML+grad evaluation.  value = N * K / t (weak scaling, whole job).

Extra objects on the same JSON line:
  roofline      the MFMA GEMM core (all instantiations of pg_gemm_kernel) over one evaluation: algorithmic
                flop of its launches / summed launch durations measured with HIP events inside the library
                (pg_profile; single-stream profile mode), against the fp64 matrix peak; `frac_end_to_end` is
                n^3 / ms_per_step / peak for the headline (multi-stream) schedule
  dist          backend, world size, the all-gathered rank list and the [1 + nhp] all-reduce latency (N > 1)
  nlml_check    the evaluation's NLML against the committed fp64 value for the default seed/size (asserted)
  cholesky, roofline_kernel_build, cfg2_fit_predict, cfg5_cotrain, grbcm_predict   secondary legs (N = 1: all of them)
  cpu_baseline  the CPU oracle's lean K^-1-route evaluation (oracle/pygpr_oracle.py, torch CPU + LAPACK) timed
                on this host's cores: min of 2 at N = 8192 and ONE measured evaluation at the full N = 16384
                (`value`), on the same data as the GPU's -- the oracle's NLML/gradient are compared with the GPU's
  cpu_baseline_as_written  the reference's algorithm as written (dK stack + batched cholesky_solve), N=4096, n^3-scaled
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix (v_mfma_f64_16x16x4_f64: 2048 flop / 64 clk / SIMD, 2.4 GHz)
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X fp32 matrix (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

# NLML of rank 0's expert at the default size and seeds (n=16384 = 1024 global [seed 99] + 15360 own [seed 1234], d=8,
# hp sigma=1 l=1 sigma_n=0.1).  Produced by the HIP path in round 1 (BENCH_r01.json) and confirmed by the CPU oracle's
# full-size evaluation in round 2 (profiles/r02_fullsize_oracle_check.json).
EXPECTED_NLML = {(16384, 8, 1024): -12646.821869277313}
NLML_RTOL = 1e-9


def synth_expert(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
    return x, y


def default_hp(d):
    return np.concatenate([[1.0], np.ones(d), [0.1]])            # sigma, l_1..l_d, sigma_n


def host_cpu_share():
    """What this process may actually use of the host: CPUs in its affinity mask, the cgroup's CPU quota (v2 cpu.max or v1
    cfs_quota), physical cores.  A box that shows 128 logical CPUs but grants a 16-CPU quota throttles 128 spinning LAPACK threads
    -- rounds 2-4 let torch pick its default thread count and the same evaluation took 40 / 64 / 86 s."""
    info = {"logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_cpu_quota": None, "physical_cores": None}
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                info["cgroup_cpu_quota"] = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    info["cgroup_cpu_quota"] = q / per
        except Exception:
            pass
    try:
        import psutil
        info["physical_cores"] = psutil.cpu_count(logical=False)
    except Exception:
        pass
    return info


def cpu_baseline(x_full, y_full, hp, n_small, gpu_loss, gpu_grad, full_size):
    """Oracle (kind = "port"), lean K^-1 route on the host cores this process is granted: min of 2 at n_small, and -- on the GPU's own
    data at full size -- min of 2 measured evaluations (both reported; the second is skipped when the first took more than 75 s),
    whose NLML / gradient are compared with the GPU's.  Reproducibility (round 5): the thread count is FIXED to what the process may
    use (min of affinity mask, cgroup quota, physical cores; PG_BENCH_CPU_THREADS overrides) and recorded with the load average."""
    from oracle import pygpr_oracle as orc

    n_full, d = x_full.shape
    share = host_cpu_share()
    cand = [v for v in (share["affinity_cpus"], share["cgroup_cpu_quota"], share["physical_cores"]) if v]
    threads_default = torch.get_num_threads()
    threads = int(os.environ.get("PG_BENCH_CPU_THREADS", "0"))
    if threads <= 0:
        threads = max(1, int(min(cand))) if cand else threads_default
    torch.set_num_threads(threads)
    cores = torch.get_num_threads()          # intra-op threads the baseline actually used
    load0 = os.getloadavg() if hasattr(os, "getloadavg") else None
    orc.mle_loss_and_grad_lean(hp, x_full[:512], y_full[:512])          # warm LAPACK threads
    xs, ys = synth_expert(n_small, d, 1234)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        orc.mle_loss_and_grad_lean(hp, xs, ys)
        ts.append(time.perf_counter() - t0)
    t_small = min(ts)
    scale = (n_full / n_small) ** 3
    out = {
        "unit": "evals/s", "cores": cores, "kind": "port",
        "threads": {"used": cores, "torch_default": threads_default, "host": share,
                    "rule": "min(affinity mask, cgroup CPU quota, physical cores); PG_BENCH_CPU_THREADS overrides"},
        "loadavg_before": list(load0) if load0 else None,
        "small_sample": {"n": n_small, "seconds_min_of_2": t_small, "seconds_all": ts,
                         "extrapolated_evals_per_s_at_full_n": 1.0 / (t_small * scale)},
    }
    if full_size:
        t_all = []
        for rep in range(2):
            t0 = time.perf_counter()
            l_ref, g_ref = orc.mle_loss_and_grad_lean(hp, x_full, y_full)
            t_all.append(time.perf_counter() - t0)
            if t_all[-1] > 75.0:          # keep the default run within a few minutes on a slow or loaded host
                break
        t_full = min(t_all)
        out["value"] = 1.0 / t_full
        out["seconds_measured"] = t_full
        out["seconds_all"] = t_all
        out["sample"] = ("min of %d measured lean K^-1-route NLML+grad evaluations at the full N=%d D=%d on the GPU's own data "
                         "(oracle.mle_loss_and_grad_lean: torch CPU fp64, LAPACK potrf + potri, %d intra-op threads): %s s; "
                         "beside it min of 2 at N=%d: %.2f s (x%.0f n^3-extrapolated: %.1f s)"
                         % (len(t_all), n_full, d, cores, " / ".join("%.1f" % t for t in t_all), n_small, t_small, scale, t_small * scale))
        out["parity_at_full_size"] = {
            "oracle_nlml": float(l_ref), "gpu_nlml": float(gpu_loss),
            "nlml_rel_err": abs(float(gpu_loss) - float(l_ref)) / abs(float(l_ref)),
            "grad_rel_err_inf": float(np.abs(np.asarray(gpu_grad) - g_ref).max() / np.abs(g_ref).max()),
            "tolerance": "NLML rtol 1e-9, gradient 1e-7 of |g|inf (sigma_n = 0.1 class, SURVEY 8c)",
        }
        assert out["parity_at_full_size"]["nlml_rel_err"] < 1e-9, out["parity_at_full_size"]
        assert out["parity_at_full_size"]["grad_rel_err_inf"] < 1e-7, out["parity_at_full_size"]
    else:
        out["value"] = 1.0 / (t_small * scale)
        out["seconds_measured"] = t_small
        out["sample"] = ("min of 2 lean K^-1-route NLML+grad evaluations (oracle.mle_loss_and_grad_lean: torch CPU fp64, LAPACK "
                         "potrf + potri, %d intra-op threads) at N=%d D=%d: %.2f s; n^3-scaled x%.0f to N=%d (extrapolated)"
                         % (cores, n_small, d, t_small, scale, n_full))
    out["loadavg_after"] = list(os.getloadavg()) if hasattr(os, "getloadavg") else None
    return out


def cpu_baseline_as_written(n_full, d, n_cpu):
    """Oracle (kind = "port") of the reference's algorithm AS WRITTEN (dK stack + batched cholesky_solve,
    n^3/3 + 2 nhp n^3 flop), n^3-scaled to n_full.  SURVEY 8d variant A; reported next to cpu_baseline."""
    from oracle import pygpr_oracle as orc

    x, y = synth_expert(n_cpu, d, 1234)
    hp = default_hp(d)
    t0 = time.perf_counter()
    orc.mle_loss_and_grad_as_written(hp, x, y)
    t = time.perf_counter() - t0
    scale = (n_full / n_cpu) ** 3
    cores = torch.get_num_threads()
    return {
        "value": 1.0 / (t * scale), "unit": "evals/s", "cores": cores, "kind": "port",
        "sample": "1 as-written NLML+grad eval (oracle.mle_loss_and_grad_as_written: torch CPU fp64, dK stack + batched "
                  "cholesky_solve, %d intra-op threads) at N=%d D=%d took %.2f s; n^3-scaled x%.0f to N=%d (extrapolated)"
                  % (cores, n_cpu, d, t, scale, n_full),
        "seconds_measured": t,
    }


def check_committee(g, xs, hp, n_train_check=1024):
    """Asserted check of a grBCM committee at whatever size it was built (bench leg and tests/test_parity_gpu.py):
      (i)  the device aggregation (pg_grbcm_local_terms -> all-reduce -> pg_grbcm_finish, PyGPR/gr_bcm.py:116-149) of one batch
           against the oracle's restatement of GRBCM.aggregate fed with the SAME per-expert device means / variances;
      (ii) every owned expert's factor solves its system: predicting an expert's own training inputs gives
           K alpha' = y - (sigma_n^2 + jitter) alpha (PyGPR/gpr.py:65-85), no O(n^3) oracle run needed.
    Returns the measured errors; raises AssertionError beyond the stated tolerances (mean 1e-9, variance rtol 1e-9, K alpha 1e-7)."""
    from oracle import pygpr_oracle as orc
    from pygpr_amd._ops import get_ops

    ops = get_ops()
    hp = np.asarray(hp, dtype=np.float64)
    xsd = ops.to_device(xs.reshape(-1, xs.shape[-1]), g.gpg.dtype)
    mg, vg = g.gpg._predict_device(xsd, "diag")
    ml, vl = g.gpl._predict_device(xsd, "diag") if g.gpl is not None else ([], [])
    mu, var = g.predict(xs, var="diag")
    out = {"batch": int(xsd.shape[0]), "experts_checked": len(ml)}
    if not g.distributed:
        cpu = lambda t: t.detach().cpu().double().numpy()  # noqa: E731
        mu_o, var_o, beta_o, prec_o = orc.grbcm_aggregate(cpu(mg[0]), cpu(vg[0]), np.stack([cpu(t) for t in ml]),
                                                          np.stack([cpu(t) for t in vl]))
        out["aggregate_mean_abs_err"] = float(np.abs(cpu(mu) - mu_o).max())
        out["aggregate_var_rel_err"] = float(np.abs(cpu(var) / var_o - 1.0).max())
        out["beta_abs_err"] = float(np.abs(cpu(g.beta) - beta_o).max())
        assert out["aggregate_mean_abs_err"] < 1e-9 and out["aggregate_var_rel_err"] < 1e-9 and out["beta_abs_err"] < 1e-9, out
        assert np.all(var_o > 0) and np.all(np.isfinite(mu_o))
    # K alpha = y per expert (the global expert too)
    noise = float(hp[-1] ** 2 + 1e-7)
    worst = 0.0
    for gp in ([g.gpg] + ([g.gpl] if g.gpl is not None else [])):
        x3 = gp._x.reshape(-1, gp._x.shape[-2], gp._x.shape[-1])
        y2 = gp._y.reshape(-1, gp._y.shape[-1])
        k = min(n_train_check, x3.shape[1])
        xq = ops.to_device(x3[:, -k:, :].contiguous(), gp.dtype)        # the LAST k points: an expert's own shard, not the shared global set
        means, _ = gp._predict_device(xq if x3.shape[0] > 1 else xq[0], "none")
        alpha = gp.wt.reshape(y2.shape[0], -1)
        for c, m_c in enumerate(means):
            resid = m_c.detach().cpu().double().numpy() - (y2[c, -k:].double().numpy() - noise * alpha[c, -k:].double().cpu().numpy())
            worst = max(worst, float(np.abs(resid).max()))
    out["k_alpha_eq_y_abs_err"] = worst
    assert worst < 1e-7, out
    return out


def launch_ranks(n):
    """`--gpus N` from a bare shell: start N fresh rank processes and relay their output.  Nothing in this process has
    touched the GPU yet (importing torch and counting devices do not), and it never execs: the launcher is a child."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=16384, help="points per expert (BASELINE: 16384)")
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--ng", type=int, default=1024, help="size of the grBCM global/communication set")
    ap.add_argument("--cpu-n", type=int, default=8192, help="size of the small CPU-baseline sample")
    ap.add_argument("--cpu-n-written", type=int, default=4096, help="size of the as-written CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the measured full-size CPU evaluation (~80 s)")
    ap.add_argument("--no-grbcm", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline + roofline only")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, run the collectives of the `dist` object and print it; no GPU work (launch self-test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one process per GPU; PG_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a
    # single-GPU box -- RCCL refuses two ranks on one device)
    backend = os.environ.get("PG_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, %d visible (PG_BENCH_BACKEND=gloo rehearses on fewer)"
                         % (world, world, ndev))
    dev_index = local_rank % ndev
    if not args.rendezvous_only:
        torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"      # where collectives take their buffers

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def dist_report(nwords, mine=None):
        """Who took part: every rank reports (rank, device index) and -- after the timed loop -- its own record `mine`
        (ms_per_step, device name, coupled-chain state, panels of its last factorisation that ran coupled, time-outs seen);
        the shared-gradient all-reduce is timed alone.  A rank that fell back to the classic chain or sits on a slower
        box shows up here instead of vanishing in the max-over-ranks time."""
        if world == 1:
            rep = {"backend": None, "world": 1, "ranks": [0], "devices": [dev_index]}
            if mine is not None:
                rep["per_rank"] = [mine]
            return rep
        me = torch.tensor([rank, dev_index], dtype=torch.int64, device=cdev)
        got = [torch.zeros_like(me) for _ in range(world)]
        dist.all_gather(got, me)
        buf = torch.zeros(nwords, dtype=torch.float64, device=cdev)
        for _ in range(5):
            dist.all_reduce(buf)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(buf)
        sync()
        t_ar = max_over_ranks((time.perf_counter() - t0) / 50)
        rep = {"backend": dist.get_backend(), "world": dist.get_world_size(),
               "ranks": [int(g[0]) for g in got], "devices": [int(g[1]) for g in got],
               "allreduce_us": 1e6 * t_ar, "allreduce_doubles": int(buf.numel()),
               "transport": "RCCL over xGMI" if backend == "nccl" else "gloo (rehearsal: ranks may share GPUs)"}
        if mine is not None:
            every = [None] * world
            dist.all_gather_object(every, mine)
            rep["per_rank"] = every
        return rep

    def allreduce_us(nwords, reps=50):
        """Latency of one all-reduce(sum) of nwords doubles, max over ranks (0.0 on one rank)."""
        if world == 1:
            return 0.0
        buf = torch.zeros(nwords, dtype=torch.float64, device=cdev)
        for _ in range(5):
            dist.all_reduce(buf)
        sync()
        dist.barrier()
        t0_ = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        sync()
        return 1e6 * max_over_ranks((time.perf_counter() - t0_) / reps)

    if args.rendezvous_only:
        rep = dist_report(1 + (args.d + 2) + world)
        if rank == 0:
            print(json.dumps({"dist": rep}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    import pygpr_amd as pg
    from pygpr_amd import _lib
    from pygpr_amd._ops import get_ops, make_spec

    ops = get_ops()
    n, d, ng = args.n, args.d, args.ng
    nls = n - ng
    cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
    hp = default_hp(d)

    # one expert per rank, n = ng + nls points: global set (shared) + own shard
    xg, yg = synth_expert(ng, d, 99)
    shards = [synth_expert(nls, d, 1234 + r) for r in range(world)]
    xl = torch.from_numpy(np.stack([s[0] for s in shards]))
    yl = torch.from_numpy(np.stack([s[1] for s in shards]))
    model = pg.GRBCM(xl, yl, torch.from_numpy(xg), torch.from_numpy(yg), cov, distributed=(world > 1))
    loss = pg.GRBCM_MLE(model)
    loss.memoize = False     # every step is a full evaluation (the library would otherwise return the cached result)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss.loss_and_grad(hp)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val, grad = loss.loss_and_grad(hp)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0          # this rank's own clock up to its last step (the steps themselves are in
    barrier()                                         # lockstep through the all-reduce; see ms_per_eval_alone below)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed)
    value = world * args.steps / elapsed
    # every rank alone, no collective: what tells a slow box or a rank on the classic chain from the others
    alone = pg.MLE(model.gpl)
    alone.memoize = False
    alone.loss_and_grad(hp[None, :].copy())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        alone.loss_and_grad(hp[None, :].copy())
    torch.cuda.synchronize()
    ms_alone = 1e3 * (time.perf_counter() - t1) / 3
    del alone

    out = {
        "metric": "NLML+grad evals/sec at N=%d D=%d RBF" % (n, d), "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "BASELINE config 3: one exact-GP NLML+gradient evaluation per GPU per step, N=%d D=%d, "
                        "Compose([Squared_exponential, White_noise]), hp sigma=1 l=1 sigma_n=0.1, jitter 1e-7; "
                        "N>1: grBCM shared-hp co-training (each rank's expert = %d global + %d own points), one "
                        "all-reduce of [1+nhp] per step" % (n, d, ng, nls),
            "experts_per_gpu": 1, "nlml_sum_over_ranks": float(val), "grad_inf": float(np.abs(grad).max()),
        },
        "build": _lib.build_id(),
    }

    mine = {"rank": rank, "device_index": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
            "ms_per_step": 1e3 * elapsed_local / args.steps, "ms_per_eval_alone": ms_alone, "coupled_chain": ops.coupled_chain(),
            "coupled_panels_last_potrf": ops.last_coupled_panels(), "chain_timeouts": ops.chain_timeouts(),
            "fallbacks": int(getattr(ops, "fallbacks", 0)), "chain_rearms": ops.chain_rearms()}
    out["dist"] = dist_report(1 + (d + 2) + world, mine)      # [NLML, gradient (nhp = d + 2), one status word per rank]

    local = exp = None
    if rank == 0:
        # ---- roofline of the dominant kernel: one profiled evaluation (events around every GEMM launch)
        local = pg.MLE(model.gpl)            # rank-local evaluation: no collective inside a rank-0-only section
        local.memoize = False
        l0, g0 = local.loss_and_grad(hp[None, :].copy())
        l0, g0 = float(np.atleast_1d(l0)[0]), np.atleast_2d(g0)[0]
        ops.profile(1)
        local.loss_and_grad(hp[None, :].copy())
        torch.cuda.synchronize()
        ops.profile(0)
        flops, ms, launches = ops.profile_read()
        # ALGORITHMIC flop of one evaluation (n^3/3 potrf + n^3/3 L^-1 + n^3/3 L^-T L^-1 = n^3) over the summed
        # durations of the GEMM-core launches that carry them.  The launches execute whole 128-wide diagonal tiles (2.5 % more
        # flop than n^3 at n = 16384): that tile-granular count is reported beside it, never as `achieved`.
        algo = float(n) ** 3
        achieved = algo / ms / 1e9 if ms > 0 else 0.0
        eval_tflops = float(n) ** 3 / (elapsed / args.steps) / 1e12
        out["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,   # filled from the committed PMC passes below
            "kernel": "pg_gemm_kernel<double,...> (MFMA GEMM core, all instantiations) over one evaluation",
            "profile_mode": "single-stream: look-ahead off, every GEMM-core launch timed alone with HIP events on its launch "
                            "stream (pg_profile); frac is the GEMM core's, frac_end_to_end the headline schedule's",
            "launches": launches, "avg_launch_ms": ms / max(launches, 1), "flops_per_launch": algo / max(launches, 1),
            "algorithmic_flops_per_eval": algo, "eval_tflops": eval_tflops,
            "frac_end_to_end": eval_tflops / FP64_MATRIX_PEAK_TFLOPS,
            "tile_granular": {"flops_per_eval": flops, "achieved": flops / ms / 1e9 if ms > 0 else 0.0,
                              "frac": (flops / ms / 1e9 if ms > 0 else 0.0) / FP64_MATRIX_PEAK_TFLOPS,
                              "note": "flop the launches execute, counting whole 128 x 128 tiles on the diagonal (pg_gemm_flops)"},
        }
        # HBM-side bytes of the GEMM-core launches: PMC counters cannot be read from inside this process; they come
        # from separate rocprofv3 --pmc passes over the same evaluation (tools/probe_eval_once.py, tools/pmc_summary.py),
        # committed under profiles/ together with the identity of the build they were taken on.  A summary taken on
        # another build (kernel sources changed since) is not reported.
        pmc_name = "r05_pmc_eval_traffic.json"
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if n == 16384 and os.path.exists(pmc):
            with open(pmc) as fh:
                js = json.load(fh)
            k = js["kernels"]["gemm_core"]
            if js.get("build", {}).get("src_sha16") == out["build"]["src_sha16"]:
                # bytes per EVALUATION (the counter passes and the headline run different launch schedules, so a per-launch figure
                # beside this object's `launches` would describe two different things: round-4 verdict)
                out["roofline"]["traffic"] = k["hbm_bytes"]
                out["roofline"]["traffic_unit"] = "B per evaluation (2 x FETCH_SIZE + WRITE_SIZE summed over the GEMM-core launches of one evaluation)"
                out["roofline"]["traffic_schedule"] = {
                    "launches": k["launches"], "bytes_per_launch": k["hbm_bytes_per_launch"],
                    "note": "the counter passes run the classic chain without look-ahead (a counter-collecting profiler runs one kernel at a "
                            "time: the coupled chain cannot run there); `launches` / `avg_launch_ms` of this object describe the single-stream "
                            "profile mode of THIS run -- do not multiply across the two"}
                out["roofline"]["traffic_source"] = "profiles/" + pmc_name
            else:
                out["roofline"]["traffic_note"] = ("profiles/%s was taken on build %s, this is %s: not reported"
                                                   % (pmc_name, js.get("build", {}).get("src_sha16"), out["build"]["src_sha16"]))
        # ---- the published size is checked, not only printed
        exp_nlml = EXPECTED_NLML.get((n, d, ng))
        if exp_nlml is not None:
            rel = abs(l0 - exp_nlml) / abs(exp_nlml)
            out["nlml_check"] = {"nlml": l0, "expected": exp_nlml, "rel_err": rel, "rtol": NLML_RTOL,
                                 "grad_inf": float(np.abs(g0).max())}
            assert rel < NLML_RTOL, "NLML at the published size moved: %r" % (out["nlml_check"],)

    legs = rank == 0 and world == 1 and not args.no_legs
    if legs:
        # potrf alone and the covariance build, HIP events on torch's current stream (the library's caller stream)
        exp = model.gpl._device_data()[0]
        npad = exp.n_pad
        spec = make_spec([0], [0], [d + 1])
        hpd = torch.from_numpy(hp).cuda()
        a = ops.empty(npad, npad)
        invd = ops.potrf_workspace(npad, torch.float64)
        info = torch.zeros(1, dtype=torch.int32, device="cuda")

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best

        # the covariance build: FOUR launches back to back between two events, so that the figure is the kernel's duration
        # (what rocprofv3 --kernel-trace reports: profiles/r05_kernel_build_hbm.json) and not one launch's latency on top of it
        def build_x4(lower):
            for _ in range(4):
                ops.kernel_build(spec, hpd, exp.x, None, a, lower_only=lower, jitter=1e-7)

        t_build = timed(lambda: build_x4(False), 5) / 4
        bytes_build = 8.0 * n * n + 8.0 * n * d
        out["roofline_kernel_build"] = {
            "bound": "hbm", "achieved": bytes_build / t_build / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_build / t_build / 1e6 / HBM_PEAK_GBS, "ms": t_build, "algorithmic_bytes": bytes_build,
            "kernel": "pg_kbuild_kernel<double, true, 2> (mirrored symmetric build); lower_only: <double, false, 2>, the build the "
                      "factorisation's path uses", "timing": "HIP events around 4 back-to-back launches / 4, best of 5",
            "method_changed_since": "r02 (one launch per event pair there: about 70 us of launch latency on top of the kernel; the "
                                    "single-launch figure of this run is lower_only.ms_single_launch)",
        }
        t_lower = timed(lambda: build_x4(True), 5) / 4
        bytes_lower = 4.0 * n * (n + 64) + 8.0 * n * d
        out["roofline_kernel_build"]["lower_only"] = {"ms": t_lower, "algorithmic_bytes": bytes_lower,
                                                      "achieved": bytes_lower / t_lower / 1e6,
                                                      "frac": bytes_lower / t_lower / 1e6 / HBM_PEAK_GBS}
        kbp = os.path.join(ROOT, "profiles", "r05_kernel_build_hbm.json")
        if n == 16384 and os.path.exists(kbp):
            with open(kbp) as fh:
                kj = json.load(fh)
            if kj.get("build", {}).get("src_sha16") == out["build"]["src_sha16"]:
                out["roofline_kernel_build"]["traffic"] = kj["mirrored"]["hbm_bytes_pmc"]
                out["roofline_kernel_build"]["lower_only"]["traffic"] = kj["lower_only"]["hbm_bytes_pmc"]
                out["roofline_kernel_build"]["traffic_source"] = "profiles/r05_kernel_build_hbm.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled)"
        t_lower_single = timed(lambda: ops.kernel_build(spec, hpd, exp.x, None, a, lower_only=True, jitter=1e-7), 3)   # one launch, for the legs below
        out["roofline_kernel_build"]["lower_only"]["ms_single_launch"] = t_lower_single

        def fac():
            ops.kernel_build(spec, hpd, exp.x, None, a, lower_only=True, jitter=1e-7)
            ops.potrf(a, invd, info)

        def chol_leg(nn):
            t = timed(fac, 3) - t_lower_single * (nn / float(n)) ** 2
            return {"n": nn, "ms": t, "tflops": nn ** 3 / 3.0 / t / 1e9,
                    "frac_of_fp64_matrix_peak": nn ** 3 / 3.0 / t / 1e9 / FP64_MATRIX_PEAK_TFLOPS}

        out["cholesky"] = chol_leg(n)
        out["cholesky"]["coupled_panels"] = ops.last_coupled_panels()
        del a, invd
        # BASELINE config 2: single GP, N=8192 D=8 fp64 -- kernel build + Cholesky + alpha (fit), then mean + diag variance
        # at 8192 test points (predict; includes L^-1 on the first call)
        x2, y2 = synth_expert(8192, d, 4242)
        if n == 16384:      # the Cholesky alone at config 2's size
            x2d = torch.from_numpy(x2).cuda()
            a = ops.empty(8192, 8192)
            invd = ops.potrf_workspace(8192, torch.float64)

            def fac2():
                ops.kernel_build(spec, hpd, x2d, None, a, lower_only=True, jitter=1e-7)
                ops.potrf(a, invd, info)

            t2l = timed(lambda: ops.kernel_build(spec, hpd, x2d, None, a, lower_only=True, jitter=1e-7), 3)
            t2 = timed(fac2, 5) - t2l
            out["cholesky_n8192"] = {"n": 8192, "ms": t2, "tflops": 8192 ** 3 / 3.0 / t2 / 1e9,
                                     "frac_of_fp64_matrix_peak": 8192 ** 3 / 3.0 / t2 / 1e9 / FP64_MATRIX_PEAK_TFLOPS,
                                     "coupled_panels": ops.last_coupled_panels()}
            del a, invd, x2d
            # a grBCM-expert-sized factorisation: all of it is the latency-bound chain (32 steps of 128 columns)
            x4d = torch.from_numpy(x2[:4096]).cuda()
            a = ops.empty(4096, 4096)
            invd = ops.potrf_workspace(4096, torch.float64)

            def fac4():
                ops.kernel_build(spec, hpd, x4d, None, a, lower_only=True, jitter=1e-7)
                ops.potrf(a, invd, info)

            t4l = timed(lambda: ops.kernel_build(spec, hpd, x4d, None, a, lower_only=True, jitter=1e-7), 3)
            t4 = timed(fac4, 5) - t4l
            out["cholesky_n4096"] = {"n": 4096, "ms": t4, "us_per_128_column_step": 1e3 * t4 / 32, "tflops": 4096 ** 3 / 3.0 / t4 / 1e9,
                                     "coupled_panels": ops.last_coupled_panels()}
            del a, invd, x4d
        gp2 = pg.Exact_GP(torch.from_numpy(x2), torch.from_numpy(y2), cov, eager_inverse=True)   # variances follow
        gp2.set_params(torch.from_numpy(hp))
        xs2 = torch.from_numpy(np.random.default_rng(4321).random((8192, d))).cuda()
        gp2.update()
        gp2.predict(xs2, var="diag")
        torch.cuda.synchronize()
        fit_ms, pred_ms = 1e30, 1e30
        for _ in range(3):
            gp2.need_upd = True
            t0 = time.perf_counter(); gp2.update(); torch.cuda.synchronize(); t1 = time.perf_counter()
            mu2, var2 = gp2.predict(xs2, var="diag"); torch.cuda.synchronize(); t2 = time.perf_counter()
            fit_ms, pred_ms = min(fit_ms, 1e3 * (t1 - t0)), min(pred_ms, 1e3 * (t2 - t1))
        out["cfg2_fit_predict"] = {"n": 8192, "d": d, "m": 8192, "fit_ms": fit_ms, "predict_ms": pred_ms,
                                   "predict_points_per_s": 8192 / (pred_ms * 1e-3),
                                   "note": "fit = covariance build + Cholesky fused with L^-1 + alpha (eager_inverse=True, since variances "
                                           "follow); predict = K* build + mean + diag variance"}
        del gp2
        if not args.no_cpu_baseline:
            e0 = model.gpl._x[0].numpy(), model.gpl._y[0].numpy()
            out["cpu_baseline"] = cpu_baseline(e0[0], e0[1], hp, min(args.cpu_n, n), l0, g0, not args.no_cpu_full)
            out["gpu_over_cpu"] = value / world / out["cpu_baseline"]["value"]
            out["cpu_baseline_as_written"] = cpu_baseline_as_written(n, d, min(args.cpu_n_written, n))

    if legs:
        # ---- BASELINE config 3 as stated: "hp_update/opt.py: 50 NLML-grad steps" through the optimiser -- scipy's CG driving
        # MLE.loss_and_grad (PyGPR/opt.py:45-67) with maxiter = 50 at N = 16384 on this build: evaluations, seconds, and the cost of
        # one evaluation INCLUDING scipy's line-search logic and the host round trip of hp / [1 + nhp] per call
        import tempfile
        cwd = os.getcwd()
        os.chdir(tempfile.mkdtemp())            # CG writes its trace file where it runs (opt.py:66)
        try:
            gp3 = pg.Exact_GP(model.gpl._x[0], model.gpl._y[0], cov)
            gp3.set_params(torch.from_numpy(hp))
            mle3 = pg.MLE(gp3)
            mle3.memoize = False
            nlml0 = float(mle3.loss(hp.copy()))
            cg = pg.CG(mle3)
            cg.args.update(maxiter=50, disp=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # the driver prints "Optimizer Failed" at maxiter like the reference (opt.py:63);
                cg.minimize()                                 # stdout carries the ONE JSON line
            torch.cuda.synchronize()
            tcg = time.perf_counter() - t0
            nlml1 = float(cg.res.fun)
            assert nlml1 < nlml0, "CG did not decrease the NLML: %r -> %r" % (nlml0, nlml1)
            out["cfg3_cg_run"] = {
                "config": "pg.CG(pg.MLE(Exact_GP)).minimize(), maxiter = 50, N = %d D = %d, start hp sigma = 1 l = 1 sigma_n = 0.1" % (n, d),
                "iterations": int(cg.res.nit), "evaluations": int(cg.res.nfev), "seconds": tcg, "ms_per_evaluation": 1e3 * tcg / max(int(cg.res.nfev), 1),
                "evals_per_s": int(cg.res.nfev) / tcg, "nlml_start": nlml0, "nlml_end": nlml1, "scipy_success": bool(cg.res.success),
                "note": "every evaluation is a full NLML + gradient (memo off); ms_per_evaluation includes scipy and the host round trip; "
                        "scipy_success = false means scipy stopped at maxiter = 50 (status %d: %s), which the reference reports the same way -- it "
                        "prints 'Optimizer Failed' and still writes res.x back (PyGPR/opt.py:61-65); the NLML decrease is asserted"
                        % (int(getattr(cg.res, "status", -1)), str(getattr(cg.res, "message", ""))[:60]),
            }
            del gp3, mle3, cg
        finally:
            os.chdir(cwd)
        torch.cuda.empty_cache()
        # ---- row f-1: the full predictive covariance (gpr.py:108-120) at config 2's size: K* build, V = L^-1 K* (n^2 m flop, K
        # ranges), C = K** - V^T V (n m^2 flop on lower tiles, mirrored afterwards)
        x2f, y2f = synth_expert(8192, d, 4242)
        gpf = pg.Exact_GP(torch.from_numpy(x2f), torch.from_numpy(y2f), cov, eager_inverse=True)
        gpf.set_params(torch.from_numpy(hp))
        mf = 4096
        xsf = torch.from_numpy(np.random.default_rng(11).random((mf, d))).cuda()
        gpf.update()
        gpf.predict(xsf, var="full")
        torch.cuda.synchronize()
        tf_ = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            muf, covf = gpf.predict(xsf, var="full")
            torch.cuda.synchronize()
            tf_ = min(tf_, time.perf_counter() - t0)
        flopf = float(8192) ** 2 * mf + 8192.0 * float(mf) ** 2
        out["predict_full"] = {
            "config": "Exact_GP.predict(var='full'), n = 8192, D = %d, m = %d test points (factor and L^-1 resident)" % (d, mf),
            "ms": 1e3 * tf_, "points_per_s": mf / tf_, "cov_diag_mean": float(torch.diagonal(covf).mean()), "cov_symmetric": bool(torch.equal(covf, covf.T)),
            "roofline": {"bound": "mfma", "achieved": flopf / tf_ / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flopf / tf_ / 1e12 / FP64_MATRIX_PEAK_TFLOPS, "algorithmic_flops": flopf,
                         "kernel": "pg_gemm_kernel<double, NT, 128x128> (Vt = K* L^-T from the test-point-major K*, K ranges) + <double, NT> (K** - Vt Vt^T, lower tiles)",
                         "note": "flop = n^2 m (triangular product) + n m^2 (symmetric update); the K* / K** builds and the mirror are HBM-bound extras inside the time"},
        }
        del gpf, xsf, muf, covf
        torch.cuda.empty_cache()

    # ---- secondary metric: grBCM committee prediction throughput (BASELINE config 4)
    del model, loss, local, exp
    torch.cuda.empty_cache()
    if not args.no_grbcm and not args.no_legs:
        nc, nls4, ng4, d4, m4, mb = 8, 8192, 1024, 16, 65536, 8192
        xg4, yg4 = synth_expert(ng4, d4, 7)
        sh = [synth_expert(nls4, d4, 100 + c) for c in range(nc)]
        g4 = pg.GRBCM(torch.from_numpy(np.stack([s[0] for s in sh])), torch.from_numpy(np.stack([s[1] for s in sh])),
                      torch.from_numpy(xg4), torch.from_numpy(yg4), cov, distributed=(world > 1))
        hp4 = torch.from_numpy(np.concatenate([[1.0], np.full(d4, 0.5), [0.1]]))
        g4.gpg.set_params(hp4)
        g4.set_local_params(hp4)
        xs = torch.from_numpy(np.random.default_rng(4321).random((m4, d4))).cuda()
        g4.predict(xs[:mb])                                     # fit (factorise, invert) + warm-up batch
        barrier()
        t0 = time.perf_counter()
        for s in range(0, m4, mb):
            mu, var = g4.predict(xs[s: s + mb])
        torch.cuda.synchronize()
        tp_local = time.perf_counter() - t0
        barrier()
        tp = time.perf_counter() - t0
        tp = max_over_ranks(tp)
        flop4 = float(m4) * (nc * float(ng4 + nls4) ** 2 + float(ng4) ** 2)     # n^2 flop per test point per expert
        per_rank = [tp]
        if world > 1:
            got = [None] * world
            dist.all_gather_object(got, float(tp_local))
            per_rank = got
        out["grbcm_predict"] = {
            "value": m4 / tp, "unit": "points/s", "scaling": "strong", "seconds": tp, "seconds_per_rank": per_rank,
            "config": "8 experts x (%d global + %d local) points, D=%d, RBF+noise fp64, %d test points in batches of %d, "
                      "diag variance; experts sharded over %d rank(s); one [3,m] all-reduce per batch" % (ng4, nls4, d4, m4, mb, world),
            "mean_abs": float(mu.abs().mean()), "var_mean": float(var.mean()),
            "allreduce_us_per_batch": allreduce_us(3 * mb + world), "allreduce_doubles_per_batch": 3 * mb + world, "batches": m4 // mb,
            "roofline": {"bound": "mfma", "achieved": flop4 / tp / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": flop4 / tp / 1e12 / (FP64_MATRIX_PEAK_TFLOPS * world), "algorithmic_flops": flop4,
                         "kernel": "pg_gemm_kernel<double,...> column-sum epilogue (variance product L^-1 K*^T), all experts",
                         "note": "flop = m (nc n^2 + ng^2): the triangular variance product per test point per expert (SURVEY 8d); "
                                 "K* build, mean and aggregation are lower order"},
            "check": check_committee(g4, xs[:mb].cpu(), hp4.numpy()),
        }
        if legs:
            # row f-1 at config 4's size: the FULL-covariance committee (aggregate_full_covar, gr_bcm.py:99-114) -- per expert K* build,
            # V = L^-1 K* and K** - V^T V (n^2 m + n m^2 flop), then nc + 2 inversions of m x m matrices (m^3 each: Cholesky, L^-1,
            # L^-T L^-1) around the weighted sum of the experts' precisions
            mf4 = 2048
            xf4 = xs[:mf4]
            g4.predict(xf4, var="full")
            torch.cuda.synchronize()
            tf4 = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                muf4, covf4 = g4.predict(xf4, var="full")
                torch.cuda.synchronize()
                tf4 = min(tf4, time.perf_counter() - t0)
            n4 = float(ng4 + nls4)
            flopf4 = nc * (n4 ** 2 * mf4 + n4 * float(mf4) ** 2) + (float(ng4) ** 2 * mf4 + float(ng4) * float(mf4) ** 2) + (nc + 2) * float(mf4) ** 3
            dg4 = torch.diagonal(covf4)
            assert bool(torch.isfinite(covf4).all()) and float(dg4.min()) > 0.0
            out["grbcm_predict_full"] = {
                "config": "GRBCM.predict(var='full'): 8 experts x (%d global + %d local) points, D=%d, m = %d test points, one GPU" % (ng4, nls4, d4, mf4),
                "ms": 1e3 * tf4, "points_per_s": mf4 / tf4, "cov_diag_mean": float(dg4.mean()), "cov_symmetric": bool(torch.equal(covf4, covf4.T)),
                "roofline": {"bound": "mfma", "achieved": flopf4 / tf4 / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": flopf4 / tf4 / 1e12 / FP64_MATRIX_PEAK_TFLOPS, "algorithmic_flops": flopf4,
                             "kernel": "pg_gemm_kernel<double,...>: Vt = K* L^-T (NT, K ranges), K** - Vt Vt^T for all experts in one launch (NT, lower tiles), and the "
                                       "m x m inversions (factor, L^-1, L^-T L^-1: the local experts' as one batched call per step)",
                             "note": "flop = nc (n^2 m + n m^2) + the global expert's share + (nc + 2) m^3"},
            }
            del muf4, covf4
        del g4, xs

    # ---- experts together (round 3): Exact_GP.update() of a batched model -- covariance build + Cholesky + L^-1 + alpha for all
    # experts -- as ONE batched call per step of the blocked algorithm against one expert after the other (the round-2 loop, each
    # expert on the single-model schedule with its coupled chain).  PyGPR/gpr.py:65-74 factorises the batch in one tc.cholesky.
    if legs:
        from pygpr_amd import gpr as _gpr

        def fit_ms(nc_, n_, d_, batched):
            keep = _gpr._BATCH_MAX_N
            _gpr._BATCH_MAX_N = keep if batched else 0
            try:
                rng_ = np.random.default_rng(3)
                x_ = rng_.random((nc_, n_, d_))
                y_ = np.sin(-x_.sum(-1)) + 0.1 * rng_.standard_normal((nc_, n_))
                gp_ = pg.Exact_GP(torch.from_numpy(x_), torch.from_numpy(y_), cov, eager_inverse=True)
                gp_.set_params(torch.from_numpy(np.tile(np.concatenate([[1.0], np.full(d_, 0.5), [0.1]]), (nc_, 1))))
                gp_.update()
                torch.cuda.synchronize()
                best = 1e30
                for _ in range(3):
                    gp_.need_upd = True
                    t0_ = time.perf_counter()
                    gp_.update()
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0_)
                assert (gp_._bat is not None) == batched
                del gp_
                torch.cuda.empty_cache()
                return 1e3 * best
            finally:
                _gpr._BATCH_MAX_N = keep

        et = {}
        for name, (nc_, n_, d_) in (("nc10_n100", (10, 100, 3)), ("nc8_n2048", (8, 2048, 16)), ("nc8_n4096", (8, 4096, 16)),
                                    ("cfg4_nc8_n9216", (8, 9216, 16))):
            tb_, ts_ = fit_ms(nc_, n_, d_, True), fit_ms(nc_, n_, d_, False)
            et[name] = {"batched_ms": tb_, "one_after_the_other_ms": ts_, "speedup": ts_ / tb_,
                        "tflops_batched": nc_ * 2.0 * float(n_) ** 3 / 3.0 / (tb_ * 1e-3) / 1e12}
        # the gradient path experts-together (round 4): MLE.loss_and_grad on a batched model, params [nc, nhp] (loss.py:92-128 of the
        # reference on x [nc, n, d]) -- one batched call per step (build + factor + L^-1, alpha + NLML, K^-1, contraction) and ONE
        # synchronisation, against the per-expert loop of rounds 1-3 (PG_MLE_SERIAL=1)
        def mle_ms(nc_, n_, d_, batched):
            rng_ = np.random.default_rng(3)
            x_ = rng_.random((nc_, n_, d_))
            y_ = np.sin(-x_.sum(-1)) + 0.1 * rng_.standard_normal((nc_, n_))
            gp_ = pg.Exact_GP(torch.from_numpy(x_), torch.from_numpy(y_), cov)
            hp_ = np.tile(np.concatenate([[1.0], np.full(d_, 0.5), [0.1]]), (nc_, 1))
            mle_ = pg.MLE(gp_)
            mle_.memoize = False
            if not batched:
                os.environ["PG_MLE_SERIAL"] = "1"
            try:
                l_, g_ = mle_.loss_and_grad(hp_.copy())
                torch.cuda.synchronize()
                best = 1e30
                for _ in range(3):
                    t0_ = time.perf_counter()
                    mle_.loss_and_grad(hp_.copy())
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0_)
                assert mle_.last_batched == batched
            finally:
                os.environ.pop("PG_MLE_SERIAL", None)
            del gp_, mle_
            torch.cuda.empty_cache()
            return 1e3 * best, l_, g_

        for name, (nc_, n_, d_) in (("mle_nc10_n100", (10, 100, 3)), ("mle_nc8_n2048", (8, 2048, 16)), ("mle_nc8_n4096", (8, 4096, 16))):
            (tb_, lb_, gb_), (ts_, ls_, gs_) = mle_ms(nc_, n_, d_, True), mle_ms(nc_, n_, d_, False)
            assert np.allclose(lb_, ls_, rtol=1e-10) and np.allclose(gb_, gs_, rtol=1e-7, atol=1e-9 * np.abs(gs_).max())
            et[name] = {"batched_ms": tb_, "one_after_the_other_ms": ts_, "speedup": ts_ / tb_,
                        "tflops_batched": nc_ * float(n_) ** 3 / (tb_ * 1e-3) / 1e12, "synchronisations": 1,
                        "what": "MLE.loss_and_grad, params [nc, nhp]: n^3 flop per expert (factor + L^-1 + K^-1); batched and looped "
                                "results agree (asserted: NLML 1e-10, gradient 1e-7)"}
        et["what"] = ("Exact_GP.update(), eager inverse: covariance build + Cholesky + L^-1 + alpha for nc experts of n points, best of 3; "
                      "batched = pg_build_potrf_trtri_batched + pg_alpha_batched (every launch covers all experts)")
        out["experts_together"] = et

    # ---- the diagonal prediction experts-together (round 5): the reference predicts a batched model with ONE batched kernel / bmm /
    # cholesky_solve (gpr.py:76-106 on x [nc, n, d]); rounds 1-4 walked the experts on the host.  (i) the reference's own test size
    # (tests/test_gpr.py:59-100: ten experts of 100 points, m = 100) batched against the loop (PG_PREDICT_SERIAL=1), results asserted
    # bit-identical; (ii) a committee of 64 experts of 1024 points (ng = 256 + 768 own), D = 16, 65 536 test points in batches of 8192.
    if legs:
        def predict_ms(gp_, xs_, serial, reps=5):
            if serial:
                os.environ["PG_PREDICT_SERIAL"] = "1"
            try:
                out_ = gp_.predict(xs_, var="diag")
                torch.cuda.synchronize()
                best = 1e30
                for _ in range(reps):
                    t0_ = time.perf_counter()
                    out_ = gp_.predict(xs_, var="diag")
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0_)
            finally:
                os.environ.pop("PG_PREDICT_SERIAL", None)
            return 1e3 * best, out_

        rng_ = np.random.default_rng(3)
        x_ = rng_.random((10, 100, 3)); y_ = np.sin(-x_.sum(-1)) + 0.1 * rng_.standard_normal((10, 100))
        gp_ = pg.Exact_GP(torch.from_numpy(x_), torch.from_numpy(y_), cov)
        gp_.set_params(torch.from_numpy(np.tile(np.concatenate([[1.0], np.full(3, 0.5), [0.1]]), (10, 1))))
        xs_ = torch.from_numpy(rng_.random((100, 3))).cuda()
        tb_, ob_ = predict_ms(gp_, xs_, False)
        assert gp_.last_predict_batched
        ts_, os_ = predict_ms(gp_, xs_, True)
        assert not gp_.last_predict_batched and torch.equal(ob_[0], os_[0]) and torch.equal(ob_[1], os_[1])
        out["predict_nc10_n100"] = {
            "config": "Exact_GP.predict(var='diag') of a batched model: 10 experts x 100 points, D = 3, m = 100 test points (the reference's "
                      "own test size, PyGPR/tests/test_gpr.py:59-100); factors resident",
            "batched_ms": tb_, "one_after_the_other_ms": ts_, "speedup": ts_ / tb_, "bit_identical_to_the_loop": True,
            "launches": "counted by rocprofv3: profiles/r05_predict_nc10_n100_kernel_stats.txt (tools/probe_predict_batched.py)",
        }
        del gp_
        nc6, ng6, nls6, d6, m6, mb6 = 64, 256, 768, 16, 65536, 8192
        xg6, yg6 = synth_expert(ng6, d6, 17)
        sh6 = [synth_expert(nls6, d6, 300 + c) for c in range(nc6)]
        g6 = pg.GRBCM(torch.from_numpy(np.stack([s_[0] for s_ in sh6])), torch.from_numpy(np.stack([s_[1] for s_ in sh6])),
                      torch.from_numpy(xg6), torch.from_numpy(yg6), cov)
        hp6 = torch.from_numpy(np.concatenate([[1.0], np.full(d6, 0.5), [0.1]]))
        g6.gpg.set_params(hp6)
        g6.set_local_params(hp6)
        xs6 = torch.from_numpy(np.random.default_rng(4321).random((m6, d6))).cuda()

        def committee_s(g_, xs__, mb_, serial):
            if serial:
                os.environ["PG_PREDICT_SERIAL"] = "1"
            try:
                g_.predict(xs__[:mb_])
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for s0_ in range(0, xs__.shape[0], mb_):
                    mu_, var_ = g_.predict(xs__[s0_: s0_ + mb_])
                torch.cuda.synchronize()
                return time.perf_counter() - t0_, mu_, var_
            finally:
                os.environ.pop("PG_PREDICT_SERIAL", None)

        t6b, mu6b, var6b = committee_s(g6, xs6, mb6, False)
        t6s, mu6s, var6s = committee_s(g6, xs6, mb6, True)
        assert torch.equal(mu6b, mu6s) and torch.equal(var6b, var6s)
        flop6 = float(m6) * (nc6 * float(ng6 + nls6) ** 2 + float(ng6) ** 2)
        out["grbcm_predict_nc64_n1024"] = {
            "value": m6 / t6b, "unit": "points/s", "seconds": t6b, "one_after_the_other_seconds": t6s, "speedup": t6s / t6b,
            "config": "GRBCM.predict(var='diag'): %d experts x (%d global + %d own) points, D=%d, RBF+noise fp64, %d test points in batches of %d, "
                      "one GPU; every expert's K* in one launch, means + variances in three, the committee's terms in one" % (nc6, ng6, nls6, d6, m6, mb6),
            "bit_identical_to_the_loop": True, "mean_abs": float(mu6b.abs().mean()), "var_mean": float(var6b.mean()),
            "roofline": {"bound": "mfma", "achieved": flop6 / t6b / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop6 / t6b / 1e12 / FP64_MATRIX_PEAK_TFLOPS, "algorithmic_flops": flop6,
                         "kernel": "pg_gemm_kernel<double, NT, 128x128, column-sum epilogue> with the expert as second batch level",
                         "note": "flop = m (nc n^2 + ng^2); per batch 4.3 GB of K* are written and read once (HBM-side: 8 nc n m B each way)"},
        }
        del g6, xs6, mu6b, mu6s, var6b, var6s
        torch.cuda.empty_cache()

    # ---- bound the 8-GPU efficiency on ONE GPU (round 5): the committee of config 4 built with k = 1, 2, 4 of its 8 local experts + the
    # global expert -- what each rank owns at 8 / 4 / 2 GPUs -- beside the 8-expert run, the same 65 536 test points in batches of 8192.
    # t_8 / (8 / k) / t_k is the strong-scaling efficiency that host overhead and per-batch fixed costs alone allow (no collective runs
    # here: the all-reduce of a [3, m] + world batch is added from the last measured rehearsal, named in `allreduce_source`).
    if legs and not args.no_grbcm:
        nc, nls4, ng4, d4, m4, mb = 8, 8192, 1024, 16, 65536, 8192
        xg4, yg4 = synth_expert(ng4, d4, 7)
        sh = [synth_expert(nls4, d4, 100 + c) for c in range(nc)]
        hp4 = torch.from_numpy(np.concatenate([[1.0], np.full(d4, 0.5), [0.1]]))
        xs = torch.from_numpy(np.random.default_rng(4321).random((m4, d4))).cuda()
        ar_us, ar_src = 0.0, None
        for name_ in ("r05_bench_n2_gloo_rehearsal.json", "r04_bench_n2_gloo_rehearsal.json"):
            pth = os.path.join(ROOT, "profiles", name_)
            if os.path.exists(pth):
                try:
                    with open(pth) as fh:
                        ar_us = float(json.load(fh)["grbcm_predict"]["allreduce_us_per_batch"])
                    ar_src = "profiles/" + name_ + " (two gloo ranks on one GPU; RCCL over xGMI is unmeasured)"
                    break
                except Exception:
                    pass
        share = {}
        for k_ in (8, 4, 2, 1):
            gk = pg.GRBCM(torch.from_numpy(np.stack([s_[0] for s_ in sh[:k_]])), torch.from_numpy(np.stack([s_[1] for s_ in sh[:k_]])),
                          torch.from_numpy(xg4), torch.from_numpy(yg4), cov)
            gk.gpg.set_params(hp4)
            gk.set_local_params(hp4)
            gk.predict(xs[:mb])
            torch.cuda.synchronize()
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0_ = time.perf_counter()
            e0_.record()
            for s0_ in range(0, m4, mb):
                gk.predict(xs[s0_: s0_ + mb])
            e1_.record()
            t_enq = time.perf_counter() - t0_          # host time to ENQUEUE all batches (nothing in predict() synchronises on one rank)
            torch.cuda.synchronize()
            t_k = time.perf_counter() - t0_
            share[k_] = {"seconds": t_k, "ms_per_batch": 1e3 * t_k / (m4 // mb), "host_enqueue_ms_per_batch": 1e3 * t_enq / (m4 // mb),
                         "gpu_ms_per_batch": e0_.elapsed_time(e1_) / (m4 // mb)}
            del gk
            torch.cuda.empty_cache()
        nb_ = m4 // mb
        proj = {}
        for k_ in (4, 2, 1):
            w_ = nc // k_
            t_rank = share[k_]["seconds"] + nb_ * ar_us * 1e-6
            proj["world_%d" % w_] = {"experts_per_rank": k_, "seconds_per_rank_projected": t_rank,
                                     "efficiency_projected": share[8]["seconds"] / (w_ * t_rank),
                                     "speedup_projected": share[8]["seconds"] / t_rank}
        out["grbcm_predict_rank_share"] = {
            "config": "config 4's committee (8 x (1024 + 8192) points, D = 16) built with k of its local experts + the global expert on ONE GPU; "
                      "%d test points in batches of %d, diag variance" % (m4, mb),
            "per_k": {str(k_): v_ for k_, v_ in share.items()}, "projection": proj,
            "allreduce_us_per_batch_assumed": ar_us, "allreduce_source": ar_src,
            "what": "efficiency_projected = t(8 experts) / (world x (t(k experts) + batches x all-reduce)): what per-batch fixed costs (the global "
                    "expert's prediction, launches, aggregation, the one synchronising status read of the distributed path) allow; the "
                    "north-star target is >= 6x at 8 GPUs (0.75)",
        }
        del xs

    # ---- BASELINE config 5: grBCM, 8 experts x (1024 + 32768) points, Matern-5/2, fp32, shared-hp co-training objective
    # sum_c NLML_c and its gradient (GRBCM_MLE): 1 warm-up + 3 evaluations.  At N = 1 the 8 experts run one after another
    # on the one GPU (with one expert per GPU each rank does 1/8 of this plus one [1 + nhp] all-reduce).
    if legs:
        torch.cuda.empty_cache()
        nc5, nls5, ng5, d5 = 8, 32768, 1024, 16
        rng = np.random.default_rng(1234)
        f5 = lambda x: np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal(x.shape[:-1])   # noqa: E731
        xl5 = rng.random((nc5, nls5, d5)); yl5 = f5(xl5)
        xg5 = rng.random((ng5, d5)); yg5 = f5(xg5)
        t32 = lambda a: torch.from_numpy(a).to(torch.float32)   # noqa: E731
        cov5 = pg.Compose([pg.Matern52(), pg.White_noise()])
        m5 = pg.GRBCM(t32(xl5), t32(yl5), t32(xg5), t32(yg5), cov5)
        obj5 = pg.GRBCM_MLE(m5)
        obj5.memoize = False
        hp5 = np.concatenate([[1.0], 0.5 * np.ones(d5), [0.1]])
        obj5.loss_and_grad(hp5.copy())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            l5, g5 = obj5.loss_and_grad(hp5 * (1.0 + 1e-3 * (i + 1)))
        torch.cuda.synchronize()
        t5 = (time.perf_counter() - t0) / 3
        n5 = ng5 + nls5
        flop5 = nc5 * float(n5) ** 3
        out["cfg5_cotrain"] = {
            "config": "grBCM co-training objective, %d experts x (%d global + %d own) = %d points each, D=%d, Matern-5/2 + noise, "
                      "fp32, shared hp; 1 warm-up + 3 evaluations of sum_c NLML_c and its gradient on ONE GPU" % (nc5, ng5, nls5, n5, d5),
            "dtype": "f32", "ms_per_eval": 1e3 * t5, "ms_per_expert": 1e3 * t5 / nc5, "evals_per_s": 1.0 / t5,
            "tflops": flop5 / t5 / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS, "frac_of_fp32_matrix_peak": flop5 / t5 / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "algorithmic_flops_per_eval": flop5, "loss": float(l5), "grad_inf": float(np.abs(g5).max()),
        }
        del m5, obj5

    # ---- BASELINE config 5 in its stated layout (world > 1): ONE fp32 Matern-5/2 expert of 1024 + 32768 points per GPU, shared-hp
    # co-training objective, one all-reduce of [1 + nhp] + world status words per evaluation (GRBCM_MLE).  1 warm-up + 3 evaluations.
    if world > 1 and not args.no_legs:
        torch.cuda.empty_cache()
        nls5, ng5, d5 = 32768, 1024, 16
        rng = np.random.default_rng(1234)
        f5 = lambda x: np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal(x.shape[:-1])   # noqa: E731
        xl5 = rng.random((world, nls5, d5)); yl5 = f5(xl5)
        xg5 = rng.random((ng5, d5)); yg5 = f5(xg5)
        t32 = lambda a: torch.from_numpy(a).to(torch.float32)   # noqa: E731
        m5 = pg.GRBCM(t32(xl5), t32(yl5), t32(xg5), t32(yg5), pg.Compose([pg.Matern52(), pg.White_noise()]), distributed=True)
        obj5 = pg.GRBCM_MLE(m5)
        obj5.memoize = False
        hp5 = np.concatenate([[1.0], 0.5 * np.ones(d5), [0.1]])
        obj5.loss_and_grad(hp5.copy())
        barrier()
        t0 = time.perf_counter()
        for i in range(3):
            l5, g5 = obj5.loss_and_grad(hp5 * (1.0 + 1e-3 * (i + 1)))
        torch.cuda.synchronize()
        t5_local = (time.perf_counter() - t0) / 3
        barrier()
        t5 = max_over_ranks((time.perf_counter() - t0) / 3)
        every5 = [None] * world
        dist.all_gather_object(every5, float(t5_local))
        n5 = ng5 + nls5
        flop5 = world * float(n5) ** 3
        out["cfg5_cotrain"] = {
            "config": "grBCM co-training objective, ONE expert of (%d global + %d own) = %d points per GPU on %d GPUs, D=%d, Matern-5/2 + "
                      "noise, fp32, shared hp; 1 warm-up + 3 evaluations of sum_c NLML_c and its gradient, one all-reduce each"
                      % (ng5, nls5, n5, world, d5),
            "dtype": "f32", "scaling": "weak", "ms_per_eval": 1e3 * t5, "evals_per_s": 1.0 / t5, "ms_per_eval_per_rank": [1e3 * v for v in every5],
            "tflops": flop5 / t5 / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS * world, "frac_of_fp32_matrix_peak": flop5 / t5 / 1e12 / (FP32_MATRIX_PEAK_TFLOPS * world),
            "algorithmic_flops_per_eval": flop5, "loss": float(l5), "grad_inf": float(np.abs(g5).max()),
            "allreduce_us": allreduce_us(1 + (d5 + 2) + world), "allreduce_doubles": 1 + (d5 + 2) + world,
        }
        del m5, obj5

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
