#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X backend (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] — dgRMatrix 1M x 100k, 32 nnz/row (nnz = 32M),
f64, %*% dense 100k x 128 — one "step" = one SpMM over the whole matrix, inputs resident in HBM.
N GPUs (launched by torch.distributed.run, one rank per GPU): every rank owns a row block of that size
(weak scaling: global A has N x 1M rows, B replicated), computes its block of C and all-gathers the
blocks over RCCL/xGMI so every rank holds the full C (north_star's exchange step).  value = total
GFLOP/s over all ranks, 2*nnz*n flops per rank-step, max-over-ranks time, all-gather included.

Prints ONE JSON line on rank 0; see DESIGN.md §Measurement for how roofline / cpu_baseline are defined.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SETUP_CALLS = 12          # untimed products before the warm-up: workspace allocation + clock ramp (see main)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--cols", type=int, default=100_000)
    ap.add_argument("--nnz-row", type=int, default=32)
    ap.add_argument("--n", type=int, default=128, help="dense columns")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="colmajor", choices=["colmajor", "rowmajor"],
                    help="C layout at N=1 (colmajor = what tcrossprod_csr_dense returns to R)")
    ap.add_argument("--algo", type=int, default=0,
                    help="0 auto, 1 row-wave kernel, 2 slab/panel kernel, 3 planned kernel (plan cached), "
                         "4 planned kernel with the plan rebuilt inside every timed step")
    ap.add_argument("--sync", type=int, default=-1, help="planned kernel: 0 no barrier, 1 per row block, 2 per panel")
    ap.add_argument("--panels", type=int, default=0)
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget for the CPU baseline sample")
    ap.add_argument("--extras", action="store_true", help="also time SpMV / gather / merge (configs 3, 4 scaled)")
    return ap.parse_args()


def cpu_baseline(args, p, j, x, B_host):
    """Reference algorithm restated (oracle/mx_oracle.c), timed on this box's host cores; bounded sample."""
    from oracle import oracle as O
    threads = O.max_threads()
    m, n = p.size - 1, B_host.shape[1]
    # bounded sample: the first `rows_s` rows of the same matrix, sized from a probe so the leg stays ~cpu-seconds
    probe_rows = min(m, 20_000)
    dt = np.float64 if args.dtype == "f64" else np.float32
    Bflat = np.ascontiguousarray(B_host, dtype=dt).reshape(-1)

    def run(rows):
        pp = p[: rows + 1]
        C_out = np.zeros(rows * n, dtype=dt)
        t0 = time.perf_counter()
        O.gemm_csr_drm_as_dcm(rows, n, pp, j, x, Bflat, n, C_out, rows, threads, False)
        return time.perf_counter() - t0
    run(probe_rows)
    t_probe = run(probe_rows)
    rate = probe_rows / max(t_probe, 1e-9)
    rows_s = int(min(m, max(probe_rows, rate * args.cpu_seconds / 3)))
    best = min(run(rows_s) for _ in range(3))
    nnz_s = int(p[rows_s] - p[0])
    return {"value": round(2.0 * nnz_s * n / best / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"first {rows_s} of {m} rows of the same CSR x the same dense {B_host.shape[0]}x{n} "
                      f"({args.dtype}), gemm_csr_drm_as_dcm restated with OpenMP schedule(dynamic), best of 3"}


def committed_traffic(kernel_sub, default_workload):
    """HBM-side bytes per launch from the committed PMC summary (profiles/*_pmc.json, written by
    tools/prof_summary.py from separate rocprofv3 --pmc passes of this same command), or None when there is
    no summary for the kernel / workload that just ran.  A measured-offline number, labelled as such."""
    import glob
    if not default_workload:
        return None
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == "cfg2-default" and kernel_sub in d.get("kernel", "") and "hbm_traffic_bytes_per_launch" in d:
            best = (int(d["hbm_traffic_bytes_per_launch"]["total_corrected"]), os.path.relpath(f, ROOT))
    return best


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from matrixextra_amd import _lib, device as D, synth
    _lib.load()                      # fails loudly if libmxgpu.so is missing

    m, K, n = args.rows, args.cols, args.n
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ndt = np.float64 if args.dtype == "f64" else np.float32
    s_dense = 8 if args.dtype == "f64" else 4

    # synthetic inputs: seeds A=1 (+rank for the other row blocks), B=2
    p, j, x = synth.csr_fixed(m, K, args.nnz_row, seed=synth.SEED_A + 1000 * rank)
    B_host = synth.dense_normal(K, n, dtype=ndt)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.from_numpy(B_host).cuda()
    nnz = A.nnz
    A.rows_sorted()                  # once per matrix, outside the timed region (cached on the DeviceCSR)
    colmajor = (args.layout == "colmajor") and world == 1
    overlap = world > 1 and os.environ.get("MXGPU_BENCH_OVERLAP", "1") != "0"
    if overlap:
        C_full = C_loc = None                                                # the pipeline owns two gathered buffers
    elif world > 1:
        C_full = torch.empty((world * m, n), dtype=tdt, device="cuda")     # gathered row-major blocks
        C_loc = C_full[rank * m:(rank + 1) * m]                              # compute straight into my slot
    else:
        C_full = None
        C_loc = torch.empty((n, m) if colmajor else (m, n), dtype=tdt, device="cuda")

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def run_spmm(A_, B_, out_, colmajor_):
        if args.algo in (3, 4):
            D.spmm_planned(A_, B_, out=out_, colmajor=colmajor_, npanels=args.panels, wg_per_cu=args.wg_per_cu,
                           sync_mode=args.sync, rebuild_plan=(args.algo == 4))
        else:
            D.spmm(A_, B_, out=out_, colmajor=colmajor_, algo=args.algo, npanels=args.panels, wg_per_cu=args.wg_per_cu)

    sharded = None
    if world > 1:
        # matrixextra_amd.distributed: equal row blocks -> compute into my slot of C_full, one RCCL all-gather in place
        from matrixextra_amd import distributed as MD

        def timed_local(local_A, Bt, out, _k=[None]):
            if _k[0] is not None:
                ev[_k[0]][0].record()
            run_spmm(local_A, Bt, out, False)
            if _k[0] is not None:
                ev[_k[0]][1].record()
        timed_local.k = timed_local.__defaults__[0]
        sharded = MD.RowShardedSpMM(A, [(r * m, (r + 1) * m) for r in range(world)], timed_local)

    # N > 1: the all-gather of step k runs under the product of step k + 1 (two gathered buffers alternate;
    # MXGPU_BENCH_OVERLAP=0 gathers in line instead).  Everything is complete before the timed region closes.
    pipe = None
    if overlap:
        pipe = MD.PipelinedRowShardedSpMM(sharded, n, tdt, "cuda")

    def step(k=None):
        if world > 1:
            timed_local.k[0] = k
            if pipe is not None:
                pipe.step(B)
            else:
                sharded(B, out=C_full)
            return
        run_spmm(A, B, C_loc, colmajor)     # no per-step events here: each one is a packet the queue drains between kernels

    import ctypes
    lib = _lib.load()
    # one-time setup, not steps: the library's grow-only workspaces (plan arrays, packed copy of B, pinned read-back
    # buffer, timing events) are allocated on first use, and after the idle seconds of input generation the GPU needs
    # ~10 products (25 ms) to reach its steady clocks (tools/ramp_probe.py: 2.32 -> 2.09 ms per call).  SETUP_CALLS
    # untimed calls here, reported in the JSON line, keep a short --warmup/--steps run from measuring that ramp.
    lib.mxd_spmm_kernel_timing(1)
    for _ in range(SETUP_CALLS):
        step()
    torch.cuda.synchronize()
    lib.mxd_spmm_kernel_timing(0)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.mxd_spmm_kernel_timing(1)           # HIP events right around the dominant kernel of every launch
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    if pipe is not None:
        C_last = pipe.finish()            # waits for the gathers still in flight
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # local SpMM call (plan build + repack + kernel) of every step; only recorded when N > 1 (beside the all-gather)
    step_ms = np.array([a.elapsed_time(b) for a, b in ev]) if world > 1 else np.array([elapsed / args.steps * 1e3])
    kt = (ctypes.c_float * 256)()
    kcount = ctypes.c_int(0)
    _lib.check(lib.mxd_spmm_kernel_times(kt, 256, ctypes.byref(kcount)))
    lib.mxd_spmm_kernel_timing(0)
    kern_ms = np.array(kt[:kcount.value], dtype=np.float64) if kcount.value else step_ms
    kern_avg_s = float(kern_ms.mean()) / 1e3                         # dominant kernel only: the roofline figure
    flops_rank_step = 2.0 * nnz * n
    alg_bytes = synth.spmm_algorithmic_bytes(m, K, n, nnz, s_dense)
    achieved = alg_bytes / kern_avg_s / 1e9

    # parity spot check of the timed output against the CPU restatement (checker only; not timed)
    parity = None
    if rank == 0:
        from oracle import oracle as O
        rows_chk = 2048
        ref = np.zeros(rows_chk * n, dtype=ndt)
        O.gemm_csr_drm_as_drm(rows_chk, n, p[: rows_chk + 1], j, x, B_host.reshape(-1), n, ref, n, O.max_threads(), True)
        if pipe is not None:
            got = C_last[rank * m:rank * m + rows_chk].cpu().numpy()
        else:
            got = (C_loc[:, :rows_chk].t() if colmajor else C_loc[:rows_chk]).cpu().numpy()
        # normalised max error: |got - ref| / max|ref| over the checked block (element-wise relative error is
        # meaningless for entries that cancel to ~0)
        parity = float(np.max(np.abs(got.astype(np.float64) - ref.reshape(rows_chk, n))) / np.max(np.abs(ref)))
        if world > 1:
            Cg = C_last if pipe is not None else C_full
            blk = Cg[(world - 1) * m:(world - 1) * m + 4].cpu().numpy()
            assert np.isfinite(blk).all()

    kernel_name = _lib.load().mxd_spmm_last_kernel().decode()      # which kernel AUTO / --algo actually launched
    default_workload = (m, K, n, args.nnz_row, args.dtype, args.layout, args.algo, args.panels, args.wg_per_cu) == \
        (1_000_000, 100_000, 128, 32, "f64", "colmajor", 0, 0, 0) and world == 1
    traffic = committed_traffic(kernel_name, default_workload) if rank == 0 else None
    out = None
    if rank == 0:
        out = {
            "metric": "CSR x dense SpMM GFLOP/s (fp64, 1M x 100k, 32 nnz/row, k=128) + achieved HBM BW% vs CPU ref",
            "value": round(world * flops_rank_step * args.steps / elapsed / 1e9, 2),
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_calls": SETUP_CALLS,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"dgRMatrix {m}x{K} nnz/row={args.nnz_row} {args.dtype} %*% dense {K}x{n} "
                                   f"(BASELINE configs[1]); C {'col' if colmajor else 'row'}-major"
                                   + (f"; row-block per GPU + RCCL all-gather of C ({world}x{m} rows)"
                                      + (", gather of step k under the product of step k+1" if pipe is not None else "")
                                      if world > 1 else ""),
                       "rows_per_gpu": m, "cols": K, "nnz_per_row": args.nnz_row, "dense_cols": n,
                       "parallelism": f"rowshard{world}" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": kernel_name, "kernel_avg_ms": round(kern_avg_s * 1e3, 4),
                         "kernel_min_ms": round(float(kern_ms.min()), 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         # secondary, non-scoring: what actually bounds the kernel.  Every nonzero gathers one B row:
                         # nnz * n * s bytes move from L2 into the CUs' L1 whatever the schedule
                         "l2_to_l1_gather": {"bytes_per_launch": int(nnz) * n * s_dense,
                                             "achieved_GBps": round(nnz * n * s_dense / kern_avg_s / 1e9, 0),
                                             "guide_ceiling_GBps": [16000, 22000]}},
            "kernel_gflops": round(flops_rank_step / kern_avg_s / 1e9, 1),
            "parity_max_err_over_max_abs_vs_oracle": parity,
            "device": _lib.device_name(),
        }
        if world > 1:
            gather_s = max(elapsed / args.steps - kern_avg_s, 1e-9)
            out["allgather"] = {"bytes_received_per_gpu": int((world - 1) * m * n * s_dense),
                                "approx_ms": round(gather_s * 1e3, 3),
                                "approx_GBps_in_per_gpu": round((world - 1) * m * n * s_dense / gather_s / 1e9, 1)}
        out["spmm_call_avg_ms"] = round(float(step_ms.mean()), 4)
        if kernel_name == "spmm_plan_kernel" and args.algo in (0, 4) and world == 1:
            # `value` above pays for building the plan from plain CSR inside every step.  A caller that multiplies the
            # same matrix repeatedly keeps the plan (it depends on A only): steady-state figure, reported separately.
            for _ in range(2):
                D.spmm_planned(A, B, out=C_loc, colmajor=colmajor)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                D.spmm_planned(A, B, out=C_loc, colmajor=colmajor)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
            out["steady_state_cached_plan"] = {"ms_per_step": round(dt * 1e3, 4),
                                               "GFLOP/s": round(flops_rank_step / dt / 1e9, 1),
                                               "plan": A.plan_info()}
        if args.extras:
            out["extras"] = extras(args, A, B, torch, D, synth, p, j, x)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, p, j, x, B_host)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def extras(args, A, B, torch, D, synth, p, j, x):
    """configs[2]: SpMV + 200k-row gather on the same CSR; configs[3] scaled to fit beside it: CSR+CSR / CSR*CSR."""
    from matrixextra_amd import _lib

    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps / 1e3
    res = {}
    m, K, nnz = A.m, A.K, A.nnz
    v = torch.from_numpy(synth.dense_normal(K, 1).reshape(-1)).cuda()
    t = timeit(lambda: D.spmv(A, v))
    byts = 4 * (m + 1) + 12 * nnz + 8 * K + 8 * m
    res["spmv"] = {"ms": round(t * 1e3, 4), "GFLOP/s": round(2 * nnz / t / 1e9, 1), "GB/s": round(byts / t / 1e9, 1)}
    rows = torch.from_numpy(synth.rows_with_replacement(200_000, m)).cuda()
    g = D.csr_gather_rows(A, rows)
    t = timeit(lambda: D.csr_gather_rows(A, rows))
    byts = 4 * 200_000 * 4 + 2 * 12 * g.nnz
    res["gather_200k_rows"] = {"ms": round(t * 1e3, 4), "nnz_out": g.nnz, "GB/s": round(byts / t / 1e9, 1)}
    p2, j2, x2 = synth.csr_overlapping(p, j, K, args.nnz_row)
    A2 = D.DeviceCSR.from_host(p2, j2, x2, K)
    for name, op in (("add", _lib.MX_OP_ADD), ("mul", _lib.MX_OP_MUL)):
        o = D.csr_elemwise(op, A, A2)
        t = timeit(lambda: D.csr_elemwise(op, A, A2), reps=5)
        byts = 12 * (A.nnz + A2.nnz) + 8 * (m + 1) + 12 * o.nnz + 4 * (m + 1)
        res[f"csr_{name}_csr"] = {"ms": round(t * 1e3, 4), "nnz_out": o.nnz, "GB/s": round(byts / t / 1e9, 1),
                                  "Mnnz_in/s": round((A.nnz + A2.nnz) / t / 1e6, 1)}
    # end-to-end through the export-level C-ABI (host pointers in, host matrix out: pageable H2D + kernel + D2H),
    # i.e. what one .Call from R costs; never the headline `value`
    from matrixextra_amd import exports as G
    Yc = np.asfortranarray(B.cpu().numpy().T)
    G.tcrossprod_csr_dense_numeric(p[:1001], j, x, Yc, 1)
    t0 = time.perf_counter()
    out = G.tcrossprod_csr_dense_numeric(p, j, x, Yc, 1) if args.dtype == "f64" else \
        G.tcrossprod_csr_dense_float32(p, j, x, Yc, 1)
    t = time.perf_counter() - t0
    res["export_call_end_to_end"] = {"ms": round(t * 1e3, 2), "GFLOP/s": round(2 * nnz * out.shape[1] / t / 1e9, 1),
                                     "note": "pageable host buffers in and out, hipMalloc/hipFree per call, transfers pipelined through pinned slots (xfer.hip); includes the Python-side allocation of the 1 GB result"}
    return res


if __name__ == "__main__":
    main()
