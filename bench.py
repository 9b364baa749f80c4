#!/usr/bin/env python
"""bench.py — QP-subproblem throughput of the MI355X-native batched ADMM solver.

A "step" = one pass of the hot path over one synthetic batch: `setup(); solve()` (fused, one
kernel launch) for every QP of this rank's shard, inputs already resident in HBM.  Default
workload = BASELINE.json configs[2]: the batch of 65,536 dense QPs (n=50, m=100, fp64), sharded over the GPUs of the run (strong
scaling: one launch of the whole batch on one GPU, 8,192 QPs per GPU on eight); `--global-batch 0 --batch-per-gpu B` gives every
GPU B QPs instead (weak scaling); `--mode fixed` runs exactly `--iters` ADMM iterations per QP
(check_termination=0), `--mode default` uses the reference's default settings (eps 1e-3, check
every 25, max_iter 1000), `--mode sqp` the settings the reference's SQP driver gives its QP solver (src/sqp.cpp:15-23: eps 1e-4,
check every 10, max_iter 100, adaptive rho every 50, alpha 1.6 — short solves, the regime where the HBM fraction means something).
`--global-batch N` fixes the TOTAL batch (strong scaling: rank r solves block shard_bounds(N, world, r); with one GPU that is the whole
BASELINE configs[2] batch of 65,536 in one launch).

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (HBM, from HIP
events recorded on the launch stream around every timed launch) and `cpu_baseline` (the CPU oracle,
a port of the reference's src/qp.cpp, timed on this host's cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c3", "c2", "c5"], default="c3",
                    help="c3 (default): BASELINE configs[2] shard 8192 x (50,100); c2: configs[1] 4096 x (20,40); "
                         "c5: configs[4] 8192 x (200,400) with a 5 %% dense CSR A")
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--m", type=int, default=100)
    ap.add_argument("--batch-per-gpu", type=int, default=8192)
    ap.add_argument("--global-batch", type=int, default=-1,
                    help="total batch over all GPUs (strong scaling); 0 = --batch-per-gpu on every GPU (weak); default: 65536 for the "
                         "c3 workload at its BASELINE shape (configs[2] is that batch), weak otherwise")
    ap.add_argument("--p-density", type=float, default=0.0,
                    help="with --workload c5: P sparse as well — a diagonally dominant P with this fraction of its off-diagonal entries, handed "
                         "over in compressed columns (sqph_setup_solve_csr_sp: read in place by the block-row kernel)")
    ap.add_argument("--mode", choices=["fixed", "default", "sqp"], default="fixed")
    ap.add_argument("--iters", type=int, default=200, help="ADMM iterations per QP in --mode fixed")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--f32-arith", action="store_true", help="with --dtype f32: true fp32 arithmetic where a kernel exists (SQPH_FLAG_F32_ARITH)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the final RCCL gather of results (N>1)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the `extra` object (the other BASELINE configs, measured after the headline; default run at N=1 only)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend: nccl (= RCCL over xGMI, the product path) or gloo with every rank on device 0 — the N > 1 "
                         "code path (rank seeds, shard bounds, padded gather, rank 0's re-solve check of every rank's shard) with real "
                         "kernels on a ONE-GPU box, where RCCL refuses two ranks on one device (tests/test_bench_contract.py)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (the N>1 code path — process group, "
                         "RCCL gather — on one device); --gpus N > 1 without RANK in the environment always does")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` started WITHOUT a launcher: re-run this script under torch.distributed.run, one rank per GPU
    (what the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` does), and pass rank 0's JSON
    line through.  Fails loudly when fewer than N devices are visible — never a silent one-GPU run labelled N."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.backend == "gloo" and have >= 1:
        have = args.gpus  # every rank shares device 0
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible (there is no CPU path and no over-subscription)" % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SQPH_BENCH_DIST1="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn"]
    raise SystemExit(subprocess.call(cmd, env=env))


def apply_mode(st, mode, iters):
    """The three settings the bench lines are quoted on (see the module docstring)."""
    if mode == "fixed":
        st.max_iter = iters
        st.check_termination = 0
    elif mode == "sqp":  # SQP constructor, src/sqp.cpp:15-23
        st.warm_start, st.check_termination, st.eps_abs, st.eps_rel = 1, 10, 1e-4, 1e-4
        st.max_iter, st.adaptive_rho, st.adaptive_rho_interval, st.alpha = 100, 1, 50, 1.6


def oracle_settings(st):
    import oracle

    return oracle.default_settings(
        rho=st.rho, sigma=st.sigma, alpha=st.alpha, eps_rel=st.eps_rel, eps_abs=st.eps_abs,
        max_iter=st.max_iter, check_termination=st.check_termination, warm_start=st.warm_start,
        adaptive_rho=st.adaptive_rho, adaptive_rho_tolerance=st.adaptive_rho_tolerance,
        adaptive_rho_interval=st.adaptive_rho_interval)


def extra_line(name, n, m, B, mode, data, dev, steps=5, warmup=2, csr=None, nnz_avg=0.0, oracle_k=64, A_dense=None, P_sparse=None, pnnz_avg=0.0):
    """One of the other BASELINE configs, measured like the headline (device-resident inputs, HIP events around every launch on the
    launch stream, wall clock around `steps` launches) on a short run, with the GPU results of the first `oracle_k` QPs checked
    against the CPU oracle.  Returns the record that goes into the JSON line's `extra` object."""
    import numpy as np
    import torch

    import oracle
    from sqp_solver_amd import QPSolverBatch

    P, q, A_cm, l, u = data
    solver = QPSolverBatch(n, m, B, dtype=np.float64, device=dev.index or 0)
    st = solver.settings
    apply_mode(st, mode, 200)
    solver.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        if csr is not None:  # (P_sparse: P handed over in compressed columns as well, sqph_setup_solve_csr_sp)
            solver.setup_solve_csr(P if P_sparse is None else P_sparse, q, csr[0], csr[1], csr[2], l, u, colmajor=True)
        else:
            solver.setup_solve(P, q, A_cm, l, u, colmajor=True)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    solver.enable_timing(True)
    # two timed repetitions of `steps` launches, the faster one reported (a short run right behind another workload's tear-down
    # has shown one-off host stalls of tens of ms: the kernel events did not move, the wall clock of that repetition did)
    reps = []
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t0)
    elapsed = min(reps)
    kernel_ms = solver.collect_kernel_ms()[-steps:]
    solver.enable_timing(False)
    info = solver.info()
    iters = float(np.minimum(info.iter, st.max_iter).mean())
    bytes_per_qp = solver.algorithmic_bytes_per_qp()
    rec = {}
    if csr is not None:
        bytes_per_qp = int(8 * (n * n + n + 2 * m) + 12 * nnz_avg + 4 * (m + 1) + 8 * (n + m) + 40)
        # what the no-check kernel needs: the lower triangle of P only (the full P is read by the residual checks alone)
        rec["needed_bytes_per_qp"] = int(8 * (n * (n + 1) // 2 + n + 2 * m) + 12 * nnz_avg + 4 * (m + 1) + 8 * (n + m) + 40)
        if P_sparse is not None:  # 12 nnz(P) + 4 (n + 1) bytes of compressed columns instead of 8 n^2
            bytes_per_qp = int(12 * pnnz_avg + 4 * (n + 1) + 8 * (n + 2 * m) + 12 * nnz_avg + 4 * (m + 1) + 8 * (n + m) + 40)
            rec["needed_bytes_per_qp"] = bytes_per_qp
            rec["nnz_P_avg"] = pnnz_avg
    kavg = float(np.mean(kernel_ms))
    achieved = bytes_per_qp * B / (kavg * 1e-3) / 1e9
    rec.update({
        "workload": "%s: %d x (n=%d, m=%d) %s, %s" % (name, B, n, m, ("CSR A, CSC P" if P_sparse is not None else "CSR A") if csr is not None else "dense", mode),
        "ms_per_step": elapsed / steps * 1e3, "kernel_ms_avg": kavg, "value": B * steps / elapsed, "unit": "QP/s",
        "admm_iters_per_qp": iters, "kernel": solver.kernel_name(), "steps": steps,
        # both repetitions are printed (the headline is ONE timed run of --steps launches: compare it with ms_per_step_repetitions[0])
        "timing": "faster of two repetitions of `steps` launches", "ms_per_step_repetitions": [r / steps * 1e3 for r in reps],
        "algorithmic_bytes_per_qp": bytes_per_qp, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc_traffic(solver.kernel_name(), n, m, B, mode),
    })
    if "needed_bytes_per_qp" in rec:
        rec["frac_needed"] = rec["needed_bytes_per_qp"] * B / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBS
    if oracle_k > 0:
        k = min(B, oracle_k)
        if csr is not None:
            Ah = A_dense[:k].cpu().numpy()
        else:
            Ah = A_cm[:k].cpu().numpy().transpose(0, 2, 1)
        xo, yo, zo, io = oracle.solve_batch(P[:k].cpu().numpy().transpose(0, 2, 1), q[:k].cpu().numpy(), Ah, l[:k].cpu().numpy(),
                                            u[:k].cpu().numpy(), settings=oracle_settings(st), nthreads=oracle.max_threads(), dtype=np.float64)
        xg, yg, zg, ig = solver.solution()

        def rel(a, b):
            den = np.maximum(np.max(np.abs(b), axis=1), 1e-300)
            return float(np.max(np.max(np.abs(a - b), axis=1) / den))

        rec["parity"] = {"oracle_sample": k, "max_rel_err_x": rel(xg[:k], xo), "max_rel_err_y": rel(yg[:k], yo),
                         "status_equal": bool((ig.status[:k] == io["status"]).all()), "iter_equal": bool((ig.iter[:k] == io["iter"]).all())}
        # (a parity sample, not a CPU timing: 64-256 QPs cannot load this host's threads; the timed CPU baseline is the headline's)
    solver.close()
    return rec


def extra_configs(dev, c3_data):
    """The BASELINE configs besides the headline, each as a short measured line: configs[1] (C2), configs[2] under the reference's
    default settings and under the SQP driver's settings and fixed-200 on the 8,192-QP shard one of eight GPUs solves, configs[4] (C5,
    CSR A), configs[3] (the batched SQP driver, a C++ host program: tests/cpp/sqp_batch_test.bin).  configs[0] is the CPU plumbing case."""
    import torch

    from sqp_solver_amd.problems import random_qp_batch_torch

    out = {}
    out["c3_default"] = extra_line("configs[2] shard", 50, 100, c3_data[0].shape[0], "default", c3_data, dev, steps=10, oracle_k=256)
    out["c3_sqp"] = extra_line("configs[2] shard", 50, 100, c3_data[0].shape[0], "sqp", c3_data, dev, steps=10, oracle_k=256)
    d = random_qp_batch_torch(4096, 20, 40, seed=20250228 + 2, dtype=torch.float64, device=dev)
    out["c2"] = extra_line("configs[1]", 20, 40, 4096, "fixed", d, dev, steps=20, oracle_k=256)
    del d
    out["c3_shard_8192"] = extra_line("configs[2] shard (what one of eight GPUs solves)", 50, 100, c3_data[0].shape[0], "fixed", c3_data, dev,
                                      steps=20, warmup=5)  # (the headline's launch pattern)
    torch.cuda.empty_cache()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_csr

    P, q, rp, ci, v, l, u, A_dense, nnz_avg = bench_csr.make(8192, 200, 400, 0.05, 20250228 + 5, dev)
    out["c5"] = extra_line("configs[4]", 200, 400, 8192, "fixed", (P, q, None, l, u), dev, steps=3, warmup=1, csr=(rp, ci, v),
                           nnz_avg=nnz_avg, oracle_k=128, A_dense=A_dense)
    # ... under the reference's default settings (termination checks, 383 iterations per QP here) and under the SQP driver's
    out["c5_default"] = extra_line("configs[4]", 200, 400, 8192, "default", (P, q, None, l, u), dev, steps=2, warmup=1, csr=(rp, ci, v),
                                   nnz_avg=nnz_avg, oracle_k=128, A_dense=A_dense)
    out["c5_sqp"] = extra_line("configs[4]", 200, 400, 8192, "sqp", (P, q, None, l, u), dev, steps=3, warmup=1, csr=(rp, ci, v),
                               nnz_avg=nnz_avg, oracle_k=128, A_dense=A_dense)
    # the same shape with P sparse as well (the legacy sparse class keeps P as Eigen::SparseMatrix, unsupported/qp_solver.hpp:24-25):
    # a 3 %-dense diagonally dominant P in compressed columns, read in place by the block-row kernel's sparse-P instantiations
    del P
    torch.cuda.empty_cache()
    Pd, Psp, pnnz = bench_csr.make_sparse_P(8192, 200, 0.03, 20250228 + 6, dev)
    out["c5_sparse_P"] = extra_line("configs[4] shape, P sparse too", 200, 400, 8192, "fixed", (Pd, q, None, l, u), dev, steps=3, warmup=1,
                                    csr=(rp, ci, v), nnz_avg=nnz_avg, oracle_k=128, A_dense=A_dense, P_sparse=Psp, pnnz_avg=pnnz)
    # (the dual half of a check walks the compressed columns instead of streaming 8 n^2 bytes of P: the check traffic of this route)
    out["c5_sparse_P_default"] = extra_line("configs[4] shape, P sparse too", 200, 400, 8192, "default", (Pd, q, None, l, u), dev, steps=1, warmup=1,
                                            csr=(rp, ci, v), nnz_avg=nnz_avg, oracle_k=128, A_dense=A_dense, P_sparse=Psp, pnnz_avg=pnnz)
    out["c5_sparse_P_sqp"] = extra_line("configs[4] shape, P sparse too", 200, 400, 8192, "sqp", (Pd, q, None, l, u), dev, steps=3, warmup=1,
                                        csr=(rp, ci, v), nnz_avg=nnz_avg, oracle_k=128, A_dense=A_dense, P_sparse=Psp, pnnz_avg=pnnz)
    del Pd, Psp, q, rp, ci, v, l, u, A_dense
    torch.cuda.empty_cache()
    out["c4"] = sqp_driver_line()
    return out


def sqp_driver_line():
    """configs[3]: the batched SQP host driver (include/sqp_hip/sqp.hpp, C++) on 1,024 SimpleNLP instances — measured by the test
    binary's `bench` mode (tests/cpp/sqp_batch_test.bin, built by __graft_entry__.build() / tests/test_cpp_sqp.py): wall per batch
    with the reference's cold subproblems and with sqp_settings_t::warm_start_qp, launches, ADMM iterations, end points equal to the
    serial CPU oracle's."""
    import subprocess

    exe = os.path.join(ROOT, "tests", "cpp", "sqp_batch_test.bin")
    if not os.path.exists(exe):  # built by __graft_entry__.build(); otherwise here (g++ is in the image)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import test_cpp_sqp

            test_cpp_sqp.build()
        except Exception as e:  # noqa: BLE001
            return {"error": "tests/cpp/sqp_batch_test.bin not built: %r" % (e,)}
    try:
        p = subprocess.run([exe, "bench"], capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "sqp_batch_test.bin bench failed (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
        return json.loads(line[0])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from sqp_solver_amd import QPSolverBatch
    from sqp_solver_amd.problems import random_qp_batch_torch

    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        self_launch(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (torch.distributed.run --nproc-per-node %d), or "
                         "run bench.py without a launcher and let it start them" % (args.gpus, world, args.gpus))
    if args.backend == "gloo":
        local_rank = 0  # every rank on device 0 (the gloo leg exists for one-GPU boxes)
    if torch.cuda.is_available() and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: local rank %d but only %d HIP device(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    # stdout carries exactly ONE line, the JSON record: libraries that write banners there (RCCL prints its version block on
    # stdout when the first communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("SQPH_BENCH_DIST1") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cdev = torch.device("cpu") if args.backend == "gloo" else dev  # where the few scalar collectives of this script live

    if args.workload == "c2":
        args.n, args.m, args.batch_per_gpu = 20, 40, 4096
    elif args.workload == "c5":
        args.n, args.m = 200, 400
    n, m, B = args.n, args.m, args.batch_per_gpu
    if args.global_batch < 0:
        args.global_batch = 65536 if (args.workload == "c3" and (n, m) == (50, 100) and B == 8192) else 0
    strong = args.global_batch > 0
    from sqp_solver_amd.dist import shard_bounds

    if strong:
        lo, hi = shard_bounds(args.global_batch, world, rank)
        B = hi - lo
        args.batch_per_gpu = B
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ndt = np.float64 if args.dtype == "f64" else np.float32

    # synthetic, device-resident, per-QP column-major (the C-ABI layout); rank-dependent seed
    csr = None
    if args.workload == "c5":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_csr

        P, q, rp, ci, v, l, u, A_dense, nnz_avg = bench_csr.make(B, n, m, 0.05, 20250228 + 5 + 1000 * rank, dev)
        csr = (rp, ci, v)
        A_cm = None
        P_arg, pnnz_avg = P, 0.0
        if args.p_density > 0:
            del P
            P, P_arg, pnnz_avg = bench_csr.make_sparse_P(B, n, args.p_density, 20250228 + 6 + 1000 * rank, dev)
    else:
        P, q, A_cm, l, u = random_qp_batch_torch(B, n, m, seed=20250228 + 3 + 1000 * rank, dtype=tdt, device=dev)
    torch.cuda.synchronize()

    solver = QPSolverBatch(n, m, B, dtype=ndt, device=local_rank, force_generic=args.force_generic, f32_arith=args.f32_arith)
    st = solver.settings
    apply_mode(st, args.mode, args.iters)
    solver.set_stream(torch.cuda.current_stream().cuda_stream)

    # device views of the resident results for the gather
    xp, yp, zp, ip = solver.device_state_ptrs()

    def step():
        if csr is not None:
            solver.setup_solve_csr(P_arg, q, csr[0], csr[1], csr[2], l, u, colmajor=True)  # P handed over as it lies (column-major per QP, or its compressed columns)
        else:
            solver.setup_solve(P, q, A_cm, l, u, colmajor=True)

    gather_bufs = None
    if use_dist and not args.no_gather:
        # final collection of (x, y, info) records on rank 0 over xGMI (RCCL gather)
        from sqp_solver_amd.dist import ResultGather

        gather_bufs = ResultGather(solver, world, rank, dev)

    def full_step():
        step()
        if gather_bufs is not None:
            gather_bufs.gather()

    for _ in range(args.warmup):
        full_step()
    if gather_bufs is not None:
        gather_bufs.flush()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()

    solver.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    if gather_bufs is not None:
        gather_bufs.flush()  # the last batches' gathers are still in flight (they overlap the following solve)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = solver.collect_kernel_ms()
    solver.enable_timing(False)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1 diagnostics (outside the timed region): what a step costs without the gather, what a gather costs on its own, the
    # kernel time of every rank, how many ranks RCCL sees — so that a scaling run that disappoints can be read
    multi = None
    if use_dist:
        dsteps = max(2, min(args.steps, 5))
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(dsteps):
            step()
        torch.cuda.synchronize()
        solve_ms = (time.perf_counter() - t0) / dsteps * 1e3
        gather_ms = None
        if gather_bufs is not None:
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(dsteps):
                gather_bufs.gather()
                gather_bufs.flush()
                torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - t0) / dsteps * 1e3
        kavg = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        per = [torch.zeros(3, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(per, torch.tensor([kavg, solve_ms, gather_ms if gather_ms is not None else float("nan")], dtype=torch.float64, device=cdev))
        multi = {
            "rccl_ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
            "kernel_ms_avg_per_rank": [float(p[0]) for p in per],
            "solve_ms_per_rank": [float(p[1]) for p in per],            # a step without the gather (wall, synchronised)
            "gather_ms_sync_per_rank": [float(p[2]) for p in per],      # one gather on its own, joined before the next (not overlapped)
            "record_bytes_per_rank": int(B * (8 * (n + max(m, 1)) + 40)),
        }

    # total ADMM iterations of the last step (info.iter counts max_iter+1 when exhausted, qp.cpp:147-150)
    info = solver.info()
    iters_local = int(np.minimum(info.iter, st.max_iter).sum())
    n_solved = int((info.status == 0).sum())
    if use_dist:
        t = torch.tensor([iters_local, n_solved], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        iters_total, solved_total = int(t[0].item()), int(t[1].item())
    else:
        iters_total, solved_total = iters_local, n_solved

    total_batch = args.global_batch if strong else B * world
    total_qps = total_batch * args.steps
    value = total_qps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    admm_iters_per_sec = iters_total * args.steps / elapsed

    out = None
    if rank == 0:
        bytes_per_qp = solver.algorithmic_bytes_per_qp()
        flops_per_iter = 2.0 * (2 * m * n + n * n)  # Schur-ordered iteration
        if csr is not None:
            # CSR A: 8(n^2 + n + 2m) + 12 nnz + 4(m+1) read, 8(n+m) + 40 written
            bytes_per_qp = int(8 * (n * n + n + 2 * m) + 12 * nnz_avg + 4 * (m + 1) + 8 * (n + m) + 40)
            if args.p_density > 0:  # 12 nnz(P) + 4 (n + 1) bytes of compressed columns instead of 8 n^2
                bytes_per_qp = int(12 * pnnz_avg + 4 * (n + 1) + 8 * (n + 2 * m) + 12 * nnz_avg + 4 * (m + 1) + 8 * (n + m) + 40)
            flops_per_iter = 2.0 * (2 * nnz_avg + n * n)
            kernel_ms = kernel_ms[-args.steps:]  # one solver launch per step (the structural pre-check is not an event pair)
        avg_kernel_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = bytes_per_qp * B / (avg_kernel_ms * 1e-3) / 1e9
        iters_per_qp = iters_local / B
        # the arithmetic type of the path: fp64 unless the true-fp32 kernel variant ran (fp32 at the interface alone is not fp32 arithmetic)
        arith_dtype = "f32" if solver.kernel_name().endswith("_f32") else "f64"
        if args.workload == "c2":
            label = "BASELINE configs[1]"
        elif args.workload == "c5":
            label = "BASELINE configs[4]" if args.p_density <= 0 else "BASELINE configs[4]'s shape with P in compressed columns (%.0f entries per QP)" % pnnz_avg
        elif (n, m) == (50, 100):
            label = "BASELINE configs[2] whole batch" if total_batch >= 65536 else "BASELINE configs[2] shard"
        else:
            label = "custom shape (not a BASELINE config)"
        out = {
            "metric": "qp_solves_per_sec",
            "value": value,
            "unit": "QP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": arith_dtype,
            "data": "synthetic",
            "config": {
                "workload": "%s: %d %s QPs n=%d m=%d per GPU (%d total), %s" % (
                    label,
                    B, "CSR-A (5 % dense)" if csr is not None else "dense", n, m, total_batch,
                    ("fixed %d ADMM iterations (check_termination=0)" % args.iters) if args.mode == "fixed"
                    else "reference default settings (eps 1e-3, check 25, max_iter 1000)" if args.mode == "default"
                    else "the SQP driver's QP settings (src/sqp.cpp:15-23: eps 1e-4, check 10, max_iter 100, adaptive rho / 50, alpha 1.6)"),
                "n": n, "m": m, "batch_per_gpu": B, "global_batch": total_batch, "mode": args.mode, "interface_dtype": args.dtype,
                "admm_iters_per_qp": iters_per_qp, "kernel": solver.kernel_name(),
                # calls that never check (fixed mode) run the kernel instantiation without the residual-check block
                "kernel_variant": "no-check" if (args.mode == "fixed" and solver.kernel_name().startswith(("wg", "csr"))) else "checking",
                "parallelism": "batch-sharded x%d" % world,
                "gather": bool(gather_bufs is not None),
            },
            **({"multi_gpu": multi} if multi is not None else {}),
            "admm_iters_per_sec": admm_iters_per_sec,
            "solved_fraction": solved_total / float(total_batch),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(solver.kernel_name(), n, m, B, args.mode),
                "algorithmic_bytes_per_qp": bytes_per_qp,
                "kernel_ms_avg": avg_kernel_ms,
                "kernel_launches_timed": len(kernel_ms),
                "onchip_f64_tflops": flops_per_iter * iters_local / (avg_kernel_ms * 1e-3) / 1e12,
                # the loop is on-chip bound: iteration flops against the fp64 vector peak (78.6 TFLOP/s public
                # spec = half the guide's 157.3 TFLOP/s fp32 vector rate)
                "onchip_f64_frac": flops_per_iter * iters_local / (avg_kernel_ms * 1e-3) / 1e12 / 78.6,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            if csr is not None:
                k = min(B, 512)  # the oracle factors a dense 600 x 600 KKT matrix per QP
                A_cm = A_dense[:k].transpose(1, 2).contiguous()
                out["cpu_baseline"] = cpu_baseline(args, solver, P[:k], q[:k], A_cm, l[:k], u[:k], st, ndt)
            else:
                out["cpu_baseline"] = cpu_baseline(args, solver, P, q, A_cm, l, u, st, ndt)
        # the other BASELINE configs in the same driver-run line (default workload only: the headline stays what it is)
        default_headline = (args.workload == "c3" and (n, m) == (50, 100) and B >= 8192 and args.mode == "fixed" and args.iters == 200 and
                            args.dtype == "f64" and not args.force_generic)
        if world == 1 and not use_dist and not args.no_cpu_baseline and default_headline:
            out["pcie_inclusive"] = pcie_inclusive(solver, P, q, A_cm, l, u, min(B, 8192))
        if world == 1 and not use_dist and not args.no_extra and default_headline:
            t0 = time.perf_counter()
            k = 8192
            out["extra"] = extra_configs(dev, (P[:k], q[:k], A_cm[:k], l[:k], u[:k]))
            out["extra"]["seconds"] = time.perf_counter() - t0
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        if gather_bufs is not None and rank == 0:
            # sanity: the gathered record of rank 0's own shard equals its resident state
            xs, ys, infos = gather_bufs.stacked()
            loc = [t.to(xs.device) for t in gather_bufs.local]  # (the gloo leg gathers through host buffers)
            assert torch.equal(xs[:B], loc[0]) and torch.equal(ys[:B], loc[1]) and torch.equal(infos[:B], loc[2]), "gather mismatch"
            assert xs.shape[0] == total_batch and gather_bufs.rows == [shard_bounds(total_batch, world, r)[1] - shard_bounds(total_batch, world, r)[0] for r in range(world)]
            # ... and the records of EVERY rank's shard equal a re-solve, on this rank, of sampled QPs of that shard (its inputs are
            # regenerated here from the rank's seed): a gather that delivers the wrong rank's or a stale buffer fails this
            if csr is None:
                off = 0
                for r in range(world):
                    Br = gather_bufs.rows[r]
                    if Br > 0:
                        Pr, qr, Ar, lr_, ur = random_qp_batch_torch(Br, n, m, seed=20250228 + 3 + 1000 * r, dtype=tdt, device=dev)
                        idx = torch.linspace(0, Br - 1, steps=min(32, Br), device=dev).long()
                        chk = QPSolverBatch(n, m, int(idx.numel()), dtype=ndt, device=local_rank, force_generic=args.force_generic, f32_arith=args.f32_arith)
                        apply_mode(chk.settings, args.mode, args.iters)
                        chk.setup_solve(Pr[idx].contiguous(), qr[idx].contiguous(), Ar[idx].contiguous(), lr_[idx].contiguous(), ur[idx].contiguous(), colmajor=True)
                        xc, yc, _, _ = chk.solution()
                        same_kernel = chk.kernel_name() == solver.kernel_name()
                        chk.close()
                        gx, gy = xs[off + idx.to(xs.device)].cpu().numpy(), ys[off + idx.to(xs.device)].cpu().numpy()
                        # bit-identical when the re-solve ran the shard's kernel; a small sample of a tiny shape may be dispatched to
                        # a variant that sums in another order (admm_dispatch.h: the quad form of the one-QP-per-lane kernel)
                        ok = (np.array_equal(gx, xc) and np.array_equal(gy[:, :m], yc[:, :m])) if same_kernel else \
                            (np.allclose(gx, xc, rtol=1e-9, atol=1e-12) and np.allclose(gy[:, :m], yc[:, :m], rtol=1e-9, atol=1e-12))
                        assert ok, "rank %d: gathered records differ from a re-solve of its shard" % r
                        del Pr, qr, Ar, lr_, ur
                    off += Br
        dist.barrier()
        dist.destroy_process_group()
    return out


def kernel_source_hash():
    """sha256 over the kernel sources (sqp_solver_amd/csrc/*): a PMC traffic record is only valid for the code it was measured on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "sqp_solver_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(kernel, n, m, batch, mode):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in separate passes;
    FETCH_SIZE x2 per the gfx950 calibration in profiles/r01_fetch_size_calibration.txt).  The table (profiles/pmc_traffic.json,
    written by tools/record_traffic.py from a tools/profile_gpu.sh run) is keyed by kernel, workload AND the hash of the kernel
    sources: None when this exact code has not been profiled on this workload (rocprofv3 cannot run inside this process)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None
    return table.get("%s|n=%d|m=%d|batch=%d|%s|src=%s" % (kernel, n, m, batch, mode, kernel_source_hash()))


def pcie_inclusive(solver, P, q, A_cm, l, u, k):
    """the boundary also takes HOST buffers — the same fused call on the first k QPs with the problem in pageable host
    memory and the results fetched back (H2D + kernel + D2H through the C-ABI's host memspace), best of three.  Never `value`."""
    import numpy as np
    import torch

    from sqp_solver_amd import QPSolverBatch

    n, m = P.shape[1], l.shape[1]
    h = [a[:k].cpu().numpy() for a in (P, q, A_cm, l, u)]
    s2 = QPSolverBatch(n, m, k, dtype=np.float64, device=P.device.index or 0)
    for f in ("max_iter", "check_termination"):
        setattr(s2.settings, f, getattr(solver.settings, f))
    best = float("inf")
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s2.setup_solve(*h, colmajor=True)
        s2.solution()
        best = min(best, time.perf_counter() - t0)
    s2.close()
    nbytes = sum(a.nbytes for a in h) + k * (8 * (n + 2 * m) + 40)
    return {"batch": k, "ms": best * 1e3, "value": k / best, "unit": "QP/s", "host_bytes_moved": int(nbytes),
            "note": "pageable host buffers in, x / y / z / info out, same settings as the headline; best of 4 (the first call allocates the staging)"}


def cpu_baseline(args, solver, P, q, A_cm, l, u, st, ndt):
    """Time the CPU oracle (port of the reference's src/qp.cpp) on a bounded sample of the same
    batch on this host's cores, and check the GPU results of that sample against it."""
    import numpy as np

    import oracle

    B, n, m = P.shape[0], args.n, args.m
    ost = oracle_settings(st)
    cores = oracle.max_threads()

    def host(k):
        # oracle.solve_batch takes math-indexed [b,i,j]; the device arrays are column-major per QP
        return (P[:k].cpu().numpy().transpose(0, 2, 1), q[:k].cpu().numpy(),
                A_cm[:k].cpu().numpy().transpose(0, 2, 1), l[:k].cpu().numpy(), u[:k].cpu().numpy())

    # the TIMED runs use the oracle built with the flags BASELINE.md states for the reference's own build (-O3 -march=native, compiled on
    # this host: oracle.native_lib()) where that build succeeds; same sources and results as the parity copy
    native = oracle.native_lib() is not None
    pilot = min(B, max(cores * 4, 32))
    hp = host(pilot)
    # thread count: every hardware thread, or one per SMT pair where that is faster (the oracle's per-QP working set — the (n+m)^2 KKT
    # matrix and its factor, 360 KB at C3 — competes for the L2 the siblings share); the faster of the two pilots is the baseline
    best = None
    for nt in ([cores, cores // 2] if cores >= 16 else [cores]):
        t0 = time.perf_counter()
        oracle.solve_batch(*hp, settings=ost, nthreads=nt, dtype=ndt, native=native)
        dtp = max(time.perf_counter() - t0, 1e-6)
        if best is None or dtp < best[0]:
            best = (dtp, nt)
    dt, all_threads = best[0], cores
    cores = best[1]
    sample = int(min(B, max(pilot, args.cpu_seconds / dt * pilot)))
    hs = host(sample)
    t0 = time.perf_counter()
    xo, yo, zo, io = oracle.solve_batch(*hs, settings=ost, nthreads=cores, dtype=ndt, native=native)
    dt = time.perf_counter() - t0
    # one thread, for the per-core figure: a 48-QP sample
    k1 = min(sample, 48)
    t0 = time.perf_counter()
    oracle.solve_batch(*[a[:k1] for a in hs], settings=ost, nthreads=1, dtype=ndt, native=native)
    dt1 = max(time.perf_counter() - t0, 1e-9)
    xg, yg, zg, ig = solver.solution()

    def rel(a, b, floor=1e-300):
        den = np.maximum(np.max(np.abs(b), axis=1), floor)
        return float(np.max(np.max(np.abs(a - b), axis=1) / den))  # a NaN in a GPU result propagates and fails the parity gate

    return {
        "value": sample / dt,
        "unit": "QP/s",
        "cores": cores,
        "host_threads_available": all_threads,
        "kind": "port",
        "sample": "first %d QPs of rank 0's batch, same settings, oracle/qp_oracle.c with OpenMP over QPs, >= 4 QPs per thread (%.1f s)" % (sample, dt),
        "build": "-O3 -march=native -ffp-contract=off, compiled on this host" if native else "-O2 (portable copy: the native build failed here)",
        "admm_iters_per_sec": float(np.minimum(io["iter"], st.max_iter).sum()) / dt,
        "single_thread_value": k1 / dt1,
        "threads_over_one_thread": (sample / dt) / (k1 / dt1),
        "parity_max_rel_err_x": rel(xg[:sample], xo),
        # tiny QPs can have every constraint inactive (y = 0): relative to max(1, |y|) there; the plain relative error otherwise
        "parity_max_rel_err_y": rel(yg[:sample], yo, 1.0 if n <= 4 else 1e-300),
        "parity_status_equal": bool((ig.status[:sample] == io["status"]).all()),
        "parity_iter_equal": bool((ig.iter[:sample] == io["iter"]).all()),
    }


if __name__ == "__main__":
    main()
