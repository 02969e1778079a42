#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): NLML+gradient evaluations/sec at N=16384, D=8, RBF + white noise, fp64.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = every rank evaluates `loss_and_grad` (PyGPR/loss.py:92-128) once for ITS expert -- covariance
build, Cholesky, L^-1, alpha, K^-1, fused gradient contraction at n = 16384, d = 8 -- followed, for N > 1, by
the shared-hyper-parameter all-reduce(sum) of [1 + nhp] doubles over RCCL (grBCM co-training, GRBCM_MLE).
At N = 1 that is exactly one exact-GP NLML+grad evaluation.  value = N * K / t (weak scaling, whole job).

Extra objects on the same JSON line:
  roofline      the MFMA GEMM core (all instantiations of pg_gemm_kernel) over one evaluation: algorithmic
                flop of its launches / summed launch durations measured with HIP events inside the library
                (pg_profile), against the fp64 matrix peak; plus potrf alone and the covariance build (HBM)
  cpu_baseline_as_written  the reference's algorithm as written (dK stack + batched cholesky_solve), N=4096, n^3-scaled
  cpu_baseline  the CPU oracle's lean K^-1-route evaluation (oracle/pygpr_oracle.py, torch CPU + LAPACK) timed
                on this host's cores at a bounded size and n^3-scaled to N = 16384
  grbcm_predict BASELINE config 4 (8 experts x (1024 + 8192) points, D = 16, 65536 test points): committee
                predictions/sec, experts sharded over the ranks, one [3, m] all-reduce per test batch
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix (v_mfma_f64_16x16x4_f64: 2048 flop / 64 clk / SIMD, 2.4 GHz)
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def synth_expert(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
    return x, y


def cpu_baseline(n_full, d, n_cpu):
    """Oracle (kind = "port"), lean K^-1 route, all host cores, n^3-scaled to n_full."""
    from oracle import pygpr_oracle as orc

    x, y = synth_expert(n_cpu, d, 1234)
    hp = np.concatenate([[1.0], np.ones(d), [0.1]])
    orc.mle_loss_and_grad_lean(hp, x[:512], y[:512])          # warm LAPACK threads
    t0 = time.perf_counter()
    orc.mle_loss_and_grad_lean(hp, x, y)
    t = time.perf_counter() - t0
    scale = (n_full / n_cpu) ** 3
    cores = torch.get_num_threads()          # intra-op threads the baseline actually used
    return {
        "value": 1.0 / (t * scale), "unit": "evals/s", "cores": cores, "kind": "port",
        "sample": "1 lean K^-1-route NLML+grad eval (oracle.mle_loss_and_grad_lean: torch CPU fp64, LAPACK potrf + potri, "
                  "%d intra-op threads) at N=%d D=%d took %.2f s; n^3-scaled x%.0f to N=%d" % (cores, n_cpu, d, t, scale, n_full),
        "seconds_measured": t,
    }


def cpu_baseline_as_written(n_full, d, n_cpu):
    """Oracle (kind = "port") of the reference's algorithm AS WRITTEN (dK stack + batched cholesky_solve,
    n^3/3 + 2 nhp n^3 flop), n^3-scaled to n_full.  SURVEY 8d variant A; reported next to cpu_baseline."""
    from oracle import pygpr_oracle as orc

    x, y = synth_expert(n_cpu, d, 1234)
    hp = np.concatenate([[1.0], np.ones(d), [0.1]])
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=16384, help="points per expert (BASELINE: 16384)")
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--ng", type=int, default=1024, help="size of the grBCM global/communication set")
    ap.add_argument("--cpu-n", type=int, default=8192, help="size of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-n-written", type=int, default=4096, help="size of the as-written CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grbcm", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    # one process per GPU; PG_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a
    # single-GPU box -- RCCL refuses two ranks on one device)
    backend = os.environ.get("PG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import pygpr_amd as pg
    from pygpr_amd._ops import get_ops, make_spec

    ops = get_ops()
    n, d, ng = args.n, args.d, args.ng
    nls = n - ng
    cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
    hp = np.concatenate([[1.0], np.ones(d), [0.1]])            # sigma, l_1..l_d, sigma_n

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
    barrier()
    elapsed = time.perf_counter() - t0
    def max_over_ranks(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    elapsed = max_over_ranks(elapsed)
    value = world * args.steps / elapsed

    out = {
        "metric": "NLML+grad evals/sec at N=%d D=%d RBF" % (n, d), "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "BASELINE config 3: one exact-GP NLML+gradient evaluation per GPU per step, N=%d D=%d, "
                        "Compose([Squared_exponential, White_noise]), hp sigma=1 l=1 sigma_n=0.1, jitter 1e-7; "
                        "N>1: grBCM shared-hp co-training (each rank's expert = %d global + %d own points), one "
                        "all-reduce of [1+nhp] per step" % (n, d, ng, nls),
            "experts_per_gpu": 1, "nlml": float(val), "grad_inf": float(np.abs(grad).max()),
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: one profiled evaluation (events around every GEMM launch)
        local = pg.MLE(model.gpl)            # rank-local evaluation: no collective inside a rank-0-only section
        local.memoize = False
        local.loss_and_grad(hp[None, :].copy())
        ops.profile(1)
        local.loss_and_grad(hp[None, :].copy())
        torch.cuda.synchronize()
        ops.profile(0)
        flops, ms, launches = ops.profile_read()
        achieved = flops / ms / 1e9 if ms > 0 else 0.0
        out["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,   # filled from the committed PMC passes below
            "kernel": "pg_gemm_kernel<double,...> (MFMA GEMM core, all instantiations) over one evaluation",
            "launches": launches, "avg_launch_ms": ms / max(launches, 1), "flops_per_launch": flops / max(launches, 1),
            "algorithmic_flops_per_eval": float(n) ** 3, "eval_tflops": float(n) ** 3 / (elapsed / args.steps) / 1e12,
        }
        # HBM-side bytes of the GEMM-core launches: PMC counters cannot be read from inside this process; they come
        # from separate rocprofv3 --pmc passes over the same evaluation (tools/probe_eval_once.py, tools/pmc_summary.py),
        # committed under profiles/.  Only valid for the default problem size.
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_eval_traffic.json")
        if n == 16384 and os.path.exists(pmc):
            with open(pmc) as fh:
                k = json.load(fh)["kernels"]["gemm_core"]
            out["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "B per launch (2 x FETCH_SIZE + WRITE_SIZE, %d launches, %.1f GB per evaluation)" % (
                k["launches"], k["hbm_bytes"] / 1e9)
            out["roofline"]["traffic_source"] = "profiles/r01_pmc_eval_traffic.json"
        # potrf alone and the covariance build, HIP events on torch's current stream (the library's stream)
        exp = model.gpl._device_experts()[0]
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

        t_build = timed(lambda: ops.kernel_build(spec, hpd, exp.x, None, a, jitter=1e-7), 5)
        bytes_build = 8.0 * n * n + 8.0 * n * d
        out["roofline_kernel_build"] = {
            "bound": "hbm", "achieved": bytes_build / t_build / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_build / t_build / 1e6 / HBM_PEAK_GBS, "ms": t_build, "algorithmic_bytes": bytes_build,
        }
        t_lower = timed(lambda: ops.kernel_build(spec, hpd, exp.x, None, a, lower_only=True, jitter=1e-7), 3)

        def fac():
            ops.kernel_build(spec, hpd, exp.x, None, a, lower_only=True, jitter=1e-7)
            ops.potrf(a, invd, info)

        t_potrf = timed(fac, 3) - t_lower
        out["cholesky"] = {"ms": t_potrf, "tflops": n ** 3 / 3.0 / t_potrf / 1e9,
                           "frac_of_fp64_matrix_peak": n ** 3 / 3.0 / t_potrf / 1e9 / FP64_MATRIX_PEAK_TFLOPS}
        del a, invd
        # BASELINE config 2: single GP, N=8192 D=8 fp64 -- kernel build + Cholesky + alpha (fit), then mean + diag variance
        # at 8192 test points (predict; includes L^-1 on the first call)
        x2, y2 = synth_expert(8192, d, 4242)
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
            out["cpu_baseline"] = cpu_baseline(n, d, min(args.cpu_n, n))
            out["gpu_over_cpu"] = value / world / out["cpu_baseline"]["value"]
            out["cpu_baseline_as_written"] = cpu_baseline_as_written(n, d, min(args.cpu_n_written, n))

    # ---- secondary metric: grBCM committee prediction throughput (BASELINE config 4)
    if not args.no_grbcm:
        del model, loss
        torch.cuda.empty_cache()
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
        barrier()
        tp = time.perf_counter() - t0
        tp = max_over_ranks(tp)
        out["grbcm_predict"] = {
            "value": m4 / tp, "unit": "points/s", "scaling": "strong", "seconds": tp,
            "config": "8 experts x (%d global + %d local) points, D=%d, RBF+noise fp64, %d test points in batches of %d, "
                      "diag variance; experts sharded over %d rank(s); one [3,m] all-reduce per batch" % (ng4, nls4, d4, m4, mb, world),
            "mean_abs": float(mu.abs().mean()), "var_mean": float(var.mean()),
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
