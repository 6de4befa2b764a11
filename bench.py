#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X backend (BASELINE.json metric).

Workloads (config.workload):
  cfg2  BASELINE configs[1] — dgRMatrix 1M x 100k, 32 nnz/row (nnz 32M), f64, %*% dense 100k x 128.  The metric's
        configuration: the default for EVERY N.  At N > 1 it is the same matrix (same seed, same bytes) cut into N
        nnz-balanced row blocks (strong scaling): `python bench.py --gpus 1/2/4/8` is one curve.
  cfg5  BASELINE configs[4] — 8M x 200k, 64 nnz/row (CSR values f64), f32 dense 200k x 256.  At N > 1 the whole matrix,
        strong-scaled, rides on the same line as `extras.cfg5_strong`; at N = 1 one GPU's 1M-row block is timed briefly
        (`extras.cfg5_shard`) and `--config cfg5-full` runs the whole matrix on one GPU.
One "step" = one SpMM over the rank's whole matrix, inputs resident in HBM.  `value` (--algo 0): the plan AUTO uses — a
regrouping of the CSR's entries that depends on the matrix alone — is kept on the DeviceCSR and built once per matrix
(untimed, like rows_sorted()); `plan.rebuild_every_step` in the line is the same loop with the plan rebuilt from plain CSR
inside every step (--algo 4 times that form as `value`).
N GPUs (one rank per GPU; `python bench.py --gpus N` starts torch.distributed.run itself when it was not started by
it): every rank owns one nnz-balanced row block of the matrix, B replicated, blocks of C exchanged with one RCCL
all-gather so that every rank holds the full C.  value = total GFLOP/s of the whole product (2*nnz*n flops per step,
max-over-ranks time, all-gather included); `compute_only_gflops` = the same flops over the slowest rank's local product;
`cpu_baseline` (rank 0's host cores, a bounded sample) and rank 0's `roofline` are carried at every N.
`--scaling weak` is the other experiment: every rank owns a --rows row block (the matrix grows with N).

At N = 1 the line also carries (`extras`) configs[2] (SpMV + gather of 200k rows), configs[3] (CSR + CSR, CSR * CSR
on 2M x 2M, nnz 1e8 each) and one export-level call from host memory, each with its own roofline and CPU baseline.

Prints ONE JSON line on rank 0; see DESIGN.md §Measurement for how roofline / cpu_baseline are defined.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SETUP_CALLS = 12               # untimed products before the warm-up: workspace allocation + clock ramp (see main)

FORCE_DIST = os.environ.get("MXGPU_BENCH_FORCE_DIST") == "1"

WORKLOADS = {
    "cfg2": dict(rows=1_000_000, cols=100_000, nnz_row=32, n=128, dtype="f64",
                 label="BASELINE configs[1]"),
    "cfg5": dict(rows=1_000_000, cols=200_000, nnz_row=64, n=256, dtype="f32",
                 label="BASELINE configs[4], one GPU's row block of the 8M x 200k matrix"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS) + ["cfg5-full"],
                    help="workload; default cfg2 (the metric's configuration) for every N")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--nnz-row", type=int, default=None)
    ap.add_argument("--n", type=int, default=None, help="dense columns")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--layout", default="colmajor", choices=["colmajor", "rowmajor"],
                    help="C layout at N=1 (colmajor = what tcrossprod_csr_dense returns to R)")
    ap.add_argument("--algo", type=int, default=0,
                    help="0 auto, 1 row-wave kernel, 2 slab/panel kernel, 3 planned kernel (plan cached), "
                         "4 planned kernel with the plan rebuilt inside every timed step")
    ap.add_argument("--sync", type=int, default=-1, help="planned kernel: 0 no barrier, 1 per row block, 2 per panel")
    ap.add_argument("--panels", type=int, default=0)
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: strong (default) = --rows is the WHOLE matrix (cfg2: the N = 1 matrix itself; cfg5: configs[4]'s "
                         "8M rows), cut into N nnz-balanced row blocks; weak = every rank owns a --rows row block")
    ap.add_argument("--cfg5-strong-rows", type=int, default=8_000_000,
                    help="rows of the configs[4] matrix of `extras.cfg5_strong` (N > 1; a multiple of 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the SpMM CPU baseline sample")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the SpMV / gather / merge / export-call / cfg5-shard legs (N = 1 only)")
    ap.add_argument("--extras", action="store_true", help="(default at N = 1; kept for older command lines)")
    args = ap.parse_args()
    args.config_given = args.config is not None
    if args.config is None:
        args.config = "cfg2"
    if args.scaling is None:
        args.scaling = "strong" if (args.gpus > 1 or FORCE_DIST) else "weak"      # (one GPU: nothing to scale; reported as weak)
    w = WORKLOADS["cfg5" if args.config == "cfg5-full" else args.config]
    args.custom = any(v is not None for v in (args.rows, args.cols, args.nnz_row, args.n, args.dtype))
    for k in ("rows", "cols", "nnz_row", "n", "dtype"):
        if getattr(args, k) is None:
            setattr(args, k, w[k])
    if args.scaling == "strong" and not args.custom and args.config == "cfg5":
        args.rows = 8_000_000                                       # BASELINE configs[4] whole: 8M x 200k, 64 / row
    return args


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without torch.distributed.run around it: start it as a child process — before
    anything here touches the GPU — and leave with its return code.  Never falls back to one GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline_spmm(args, p, j, x, B_host, dtype):
    """Reference algorithm restated (oracle/mx_oracle.c gemm_csr_drm_as_dcm), timed on this box's host cores on a
    bounded sample (the first rows of the same matrix); all threads, plus the 1-thread figure."""
    from oracle import oracle as O
    threads = O.max_threads()
    m, n = p.size - 1, B_host.shape[1]
    dt = np.float64 if dtype == "f64" else np.float32
    Bflat = np.ascontiguousarray(B_host, dtype=dt).reshape(-1)

    def run(rows, nthreads):
        pp = p[: rows + 1]
        C_out = np.zeros(rows * n, dtype=dt)
        t0 = time.perf_counter()
        O.gemm_csr_drm_as_dcm(rows, n, pp, j, x, Bflat, n, C_out, rows, nthreads, False)
        return time.perf_counter() - t0

    def sample(nthreads, budget):
        probe_rows = min(m, 20_000 if nthreads > 1 else 2_000)
        run(probe_rows, nthreads)
        rate = probe_rows / max(run(probe_rows, nthreads), 1e-9)
        rows_s = int(min(m, max(probe_rows, rate * budget / 3)))
        best = min(run(rows_s, nthreads) for _ in range(3))
        return rows_s, 2.0 * int(p[rows_s] - p[0]) * n / best / 1e9
    rows_all, gf_all = sample(threads, args.cpu_seconds * 0.7)
    rows_one, gf_one = sample(1, args.cpu_seconds * 0.3)
    return {"value": round(gf_all, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "single_thread": {"value": round(gf_one, 3), "unit": "GFLOP/s", "cores": 1,
                              "sample": f"first {rows_one} rows, best of 3"},
            "sample": f"first {rows_all} of {m} rows of the same CSR x the same dense {B_host.shape[0]}x{n} "
                      f"({dtype}), gemm_csr_drm_as_dcm restated with OpenMP schedule(dynamic), -march=native, best of 3"}


def committed_traffic(kernel_sub, workload_tag, kernel_avg_ms):
    """HBM-side bytes per launch from the committed PMC summary (profiles/rNN_*_pmc.json, written by tools/prof_summary.py
    from separate rocprofv3 --pmc passes of this same command).  Only the NEWEST round's summaries are read, and one is
    quoted only when (a) it is for the kernel and workload that just ran, (b) the source files that kernel is built from
    still have the git blob hashes the summary recorded (tools/prof_common.py) and (c) its kernel duration agrees with
    the one measured in this run within 5 % — otherwise `traffic` is null with the reason beside it, never last round's
    counters.  Returns (bytes, file, None) or (None, None, why)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prof_common as PC
    why = "no summary of the newest round for this kernel and workload"
    for f in PC.newest_round("*_pmc.json"):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload_tag or kernel_sub not in d.get("kernel", ""):
            continue
        if "hbm_traffic_bytes_per_launch" not in d or "avg_ns" not in d:
            continue
        ok, reason = PC.kernel_unchanged(d["kernel"], d.get("source_blobs"))
        if not ok:
            why = f"{os.path.relpath(f, ROOT)}: {reason}"
            continue
        if abs(d["avg_ns"] / 1e6 - kernel_avg_ms) > 0.05 * kernel_avg_ms:
            why = f"{os.path.relpath(f, ROOT)}: kernel {d['avg_ns'] / 1e6:.4f} ms there, {kernel_avg_ms:.4f} ms in this run"
            continue
        return int(d["hbm_traffic_bytes_per_launch"]["total_corrected"]), os.path.relpath(f, ROOT), None
    return None, None, why


def committed_kernels_traffic(kernels, call_ms):
    """For a neighbouring operation timed as a whole call: the summed HBM-side bytes per call of the kernels it launches,
    from the NEWEST round's multi-kernel PMC summary (profiles/rNN_extras_pmc.json, tools/prof_summary_multi.py: per
    kernel the launches of one workload).  `kernels` = [(name in the summary, launches per call)].  Refused (traffic
    null + `traffic_refused`) when a kernel is missing, when a kernel's source files no longer hash to what the summary
    recorded, or when the kernels' summed duration in that profile does not fit the call just timed (more than 30 % above
    it, or under half of it)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prof_common as PC
    whys = []
    for f in PC.newest_round("*extras_pmc.json"):
        try:
            doc = json.load(open(f))
            ks = doc["kernels"]
        except Exception:
            continue
        if any(k not in ks or ks[k].get("avg_ns") is None or "total_corrected" not in ks[k] for k, _ in kernels):
            whys.append(f"{os.path.relpath(f, ROOT)}: a kernel of this call is not in the summary")
            continue
        bad = [r for r in (PC.kernel_unchanged(ks[k]["kernel"], doc.get("source_blobs")) for k, _ in kernels) if not r[0]]
        if bad:
            whys.append(f"{os.path.relpath(f, ROOT)}: {bad[0][1]}")
            continue
        ns = sum(ks[k]["avg_ns"] * c for k, c in kernels)
        # (the source hashes above are what tells a stale profile; this window only catches a summary of another workload —
        # kernels of a few microseconds run up to ~20 % slower under the profiler than inside the timed loop)
        if not (0.5 * call_ms <= ns / 1e6 <= 1.3 * call_ms):
            whys.append(f"{os.path.relpath(f, ROOT)}: kernels {ns / 1e6:.4f} ms there, the call {call_ms:.4f} ms in this run")
            continue
        return {"traffic": int(sum(ks[k]["total_corrected"] * c for k, c in kernels)),
                "traffic_source": os.path.relpath(f, ROOT), "traffic_kernels_ms": round(ns / 1e6, 4)}
    return {"traffic_refused": "; ".join(whys) if whys else "no multi-kernel summary of the newest round"}


STREAM = {"GBps": None}          # measured once per run (stream_copy_probe): the box's own float4-copy rate


def stream_copy_probe(torch, lib, _lib, nbytes=1 << 30, reps=10):
    """every fraction is quoted against the nominal 8 TB/s AND against a device-copy STREAM probe measured on
    the box the benchmark runs on: one 1 GiB float4 copy kernel (csrc/stream.hip), read + write bytes / time."""
    import ctypes
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.zero_()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def copy():
        _lib.check(lib.mxd_stream_copy(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(nbytes), st))
    for _ in range(3):
        copy()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        copy()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    STREAM["GBps"] = round(2 * nbytes / (ms / 1e3) / 1e9, 1)
    del a, b
    return {"GBps": STREAM["GBps"], "ms_per_GiB_copied": round(ms, 4), "bytes_copied": nbytes,
            "kernel": "stream_copy_kernel (16 B per lane, nontemporal stores), read + write bytes / time",
            "guide_figure_GBps": 6290}


def roofline(alg_bytes, seconds, **extra):
    """call- or kernel-level fraction of `alg_bytes` (SURVEY §8d algorithmic bytes) moved in `seconds`, against the nominal
    HBM peak (`frac`) and against the STREAM-copy rate measured in this run (`frac_of_stream_copy`)"""
    ach = alg_bytes / seconds / 1e9
    d = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(alg_bytes)}
    if STREAM["GBps"]:
        d["frac_of_stream_copy"] = round(ach / STREAM["GBps"], 4)
    d.update(extra)
    km = d.get("traffic_kernels_ms")
    if km:                                                          # kernel-level figure beside the call-level one
        ka = alg_bytes / (km / 1e3) / 1e9
        d["kernels"] = {"ms": km, "achieved": round(ka, 1), "frac": round(ka / HBM_PEAK_GBS, 4),
                        "source": "rocprofv3 kernel durations of the committed profile (checked against this run's call time)"}
        if STREAM["GBps"]:
            d["kernels"]["frac_of_stream_copy"] = round(ka / STREAM["GBps"], 4)
    return d


STRONG_CHUNKS = 64


def strong_rows(synth, torch, m_total, K, nnz_row, r0, r1):
    """Rows [r0, r1) of the --scaling strong matrix, on the device.  The matrix is the same whatever N is: 64 row chunks,
    chunk c drawn on the device from seed SEED_A + c (synth.device_csr_fixed); a rank draws the chunks its block touches."""
    assert m_total % STRONG_CHUNKS == 0, "--scaling strong: --rows must be a multiple of 64"
    cr = m_total // STRONG_CHUNKS
    js, xs = [], []
    for c in range(r0 // cr, (max(r1, r0 + 1) - 1) // cr + 1):
        _, j, x = synth.device_csr_fixed(cr, K, nnz_row, seed=synth.SEED_A + c)
        a, b = max(r0, c * cr) - c * cr, min(r1, (c + 1) * cr) - c * cr
        js.append(j[a * nnz_row:b * nnz_row])
        xs.append(x[a * nnz_row:b * nnz_row])
    m = r1 - r0
    p = (torch.arange(m + 1, dtype=torch.int64, device="cuda") * nnz_row).to(torch.int32)
    return p, torch.cat(js).contiguous(), torch.cat(xs).contiguous()


# ------------------------------------------------------------------------------------------------ the SpMM leg
def spmm_leg(args, torch, dist, D, synth, lib, _lib, cfg, world, rank, steps, warmup, want_cpu, want_steady):
    """Times `steps` SpMM steps of workload `cfg` (dict rows/cols/nnz_row/n/dtype) on this rank (+ all-gather for
    dist_on).  Returns the result dict on rank 0 (None elsewhere)."""
    import ctypes
    # MXGPU_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, in-place RCCL all-gather, pipelined buffers,
    # gathered-buffer parity check) with whatever WORLD_SIZE is — lets a ONE-GPU box exercise it (tests/test_gpu_rccl.py)
    dist_on = world > 1 or FORCE_DIST
    m, K, n, nnz_row, dtype = cfg["rows"], cfg["cols"], cfg["n"], cfg["nnz_row"], cfg["dtype"]
    tdt = torch.float64 if dtype == "f64" else torch.float32
    ndt = np.float64 if dtype == "f64" else np.float32
    s_dense = 8 if dtype == "f64" else 4

    strong = dist_on and args.scaling == "strong"
    B_host = synth.dense_normal(K, n, dtype=ndt)
    if strong:
        # --rows is the WHOLE matrix; this rank owns one of `world` nnz-balanced row blocks (ragged when the cut is)
        from matrixextra_amd import distributed as MDs
        m_total = m
        blocks = MDs.nnz_balanced_row_blocks(np.arange(m_total + 1, dtype=np.int64) * nnz_row, world)
        r0g, r1g = blocks[rank]
        m = r1g - r0g
        host_whole = cfg["name"] == "cfg2" and not args.custom
        if host_whole:
            # the metric's matrix: the SAME bytes as the N = 1 run (synth.csr_fixed, seed A = 1), drawn by every rank and cut
            pw, jw, xw = synth.csr_fixed(m_total, K, nnz_row, seed=synth.SEED_A)
            p, j, x = MDs.shard_csr(pw, jw, xw, r0g, r1g)
            A = D.DeviceCSR.from_host(p, j, x, K)
            last_rows = MDs.shard_csr(pw, jw, xw, m_total - 2048, m_total) if rank == 0 else None
            del pw, jw, xw
        else:
            dp, dj, dx = strong_rows(synth, torch, m_total, K, nnz_row, r0g, r1g)
            A = D.DeviceCSR(dp, dj, dx, m, K, int(dj.numel()))
            p = j = x = None
    else:
        # synthetic inputs: seeds A=1 (+1000*rank for the other row blocks), B=2
        m_total = world * m
        blocks = [(r * m, (r + 1) * m) for r in range(world)]
        p, j, x = synth.csr_fixed(m, K, nnz_row, seed=synth.SEED_A + 1000 * rank)
        A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.from_numpy(B_host).cuda()
    nnz = A.nnz
    A.rows_sorted()                  # once per matrix, outside the timed region (cached on the DeviceCSR)
    colmajor = (args.layout == "colmajor") and not dist_on
    overlap = dist_on and os.environ.get("MXGPU_BENCH_OVERLAP", "1") != "0"
    C_full = C_loc = None
    if dist_on and not overlap:
        C_full = torch.full((m_total, n), float("nan"), dtype=tdt, device="cuda")       # gathered row-major blocks
        C_loc = C_full[blocks[rank][0]:blocks[rank][1]]                                  # compute straight into my slot
    elif not dist_on:
        C_loc = torch.empty((n, m) if colmajor else (m, n), dtype=tdt, device="cuda")

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] \
        if dist_on else []

    keep_plan = [True]              # --algo 0: AUTO's plan is kept on the DeviceCSR (built in the untimed setup calls)

    def run_spmm(A_, B_, out_, colmajor_):
        if args.algo in (3, 4):
            D.spmm_planned(A_, B_, out=out_, colmajor=colmajor_, npanels=args.panels, wg_per_cu=args.wg_per_cu,
                           sync_mode=args.sync, rebuild_plan=(args.algo == 4))
        else:
            D.spmm(A_, B_, out=out_, colmajor=colmajor_, algo=args.algo, npanels=args.panels, wg_per_cu=args.wg_per_cu,
                   keep_plan=keep_plan[0])

    pipe = None
    cur = [None]
    if dist_on:
        # matrixextra_amd.distributed: equal row blocks -> compute into my slot, one RCCL all-gather in place
        from matrixextra_amd import distributed as MD

        def timed_local(local_A, Bt, out):
            k = cur[0]
            if k is not None:
                ev[k][0].record()
            run_spmm(local_A, Bt, out, False)
            if k is not None:
                ev[k][1].record()
        sharded = MD.RowShardedSpMM(A, blocks, timed_local)
        # the all-gather of step k runs under the product of step k + 1 (two gathered buffers alternate;
        # MXGPU_BENCH_OVERLAP=0 gathers in line instead).  Everything is complete before the timed region closes.
        if overlap:
            pipe = MD.PipelinedRowShardedSpMM(sharded, n, tdt, "cuda")
            for b in pipe.bufs:
                b.fill_(float("nan"))          # anything not written by a product or a completed gather stays NaN

    def step(k=None):
        if dist_on:
            cur[0] = k
            if pipe is not None:
                pipe.step(B)
            else:
                sharded(B, out=C_full)
            return
        run_spmm(A, B, C_loc, colmajor)     # no per-step events here: each one is a packet the queue drains between kernels

    # one-time setup, not steps: the library's grow-only workspaces (plan arrays, packed copy of B, pinned read-back
    # buffer, timing events) are allocated on first use, and after the idle seconds of input generation the GPU needs
    # ~10 products (25 ms) to reach its steady clocks (tools/ramp_probe.py: 2.32 -> 2.09 ms per call).  SETUP_CALLS
    # untimed calls here, reported in the JSON line, keep a short --warmup/--steps run from measuring that ramp.
    lib.mxd_spmm_kernel_timing(1)
    for _ in range(SETUP_CALLS):
        step()
    torch.cuda.synchronize()
    lib.mxd_spmm_kernel_timing(0)
    for _ in range(warmup):
        step()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    lib.mxd_spmm_kernel_timing(1)           # HIP events right around the dominant kernel of every launch
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    if pipe is not None:
        pipe.finish()                     # waits for the gathers still in flight
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # local SpMM call (plan build + repack + kernel) of every step; only recorded when N > 1 (beside the all-gather)
    step_ms = np.array([a.elapsed_time(b) for a, b in ev]) if dist_on else np.array([elapsed / steps * 1e3])
    kt = (ctypes.c_float * 256)()
    kcount = ctypes.c_int(0)
    _lib.check(lib.mxd_spmm_kernel_times(kt, 256, ctypes.byref(kcount)))
    lib.mxd_spmm_kernel_timing(0)
    kern_ms = np.array(kt[:kcount.value], dtype=np.float64) if kcount.value else step_ms
    kern_avg_s = float(kern_ms.mean()) / 1e3                         # dominant kernel only: the roofline figure
    flops_rank_step = 2.0 * nnz * n
    flops_all_step = 2.0 * m_total * nnz_row * n if strong else world * flops_rank_step
    alg_bytes = synth.spmm_algorithmic_bytes(m, K, n, nnz, s_dense)
    nranks_seen = None
    slowest_local_ms = float(step_ms.mean())
    if dist_on:
        # how many ranks really took part: an RCCL all-reduce of ones (beside dist.get_world_size(), for the driver's check)
        ones = torch.ones(1, dtype=torch.float32, device="cuda")
        dist.all_reduce(ones)
        nranks_seen = int(round(float(ones.item())))
        tl = torch.tensor([slowest_local_ms], dtype=torch.float64, device="cuda")     # the slowest rank's local product
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        slowest_local_ms = float(tl.item())
    if rank != 0:
        return None

    # parity of the timed output against the CPU restatement (checker only; not timed)
    from oracle import oracle as O
    rows_chk = 2048

    def oracle_rows(pp, jj, xx, r0):
        ref = np.zeros(rows_chk * n, dtype=ndt)
        lo, hi = int(pp[r0]), int(pp[r0 + rows_chk])
        O.gemm_csr_drm_as_drm(rows_chk, n, (pp[r0:r0 + rows_chk + 1] - pp[r0]).astype(np.int32), jj[lo:hi].copy(),
                              xx[lo:hi].copy(), B_host.reshape(-1), n, ref, n, O.max_threads(), True)
        return ref.reshape(rows_chk, n)

    def max_err(got, ref):
        # normalised max error: |got - ref| / max|ref| over the checked block (element-wise relative error is
        # meaningless for entries that cancel to ~0); NaN (unwritten / half-gathered data) propagates
        return float(np.max(np.abs(got.astype(np.float64) - ref)) / np.max(np.abs(ref)))
    if strong and p is None:                          # my block's first rows, read back from the device
        e = rows_chk * nnz_row
        p = (np.arange(rows_chk + 1, dtype=np.int64) * nnz_row).astype(np.int32)
        j, x = A.indices[:e].cpu().numpy(), A.values[:e].cpu().numpy()
    ref0 = oracle_rows(p, j, x, 0)
    if not dist_on:
        got = (C_loc[:, :rows_chk].t() if colmajor else C_loc[:rows_chk]).cpu().numpy()
        parity = max_err(got, ref0)
    else:
        # every gathered buffer, my own block AND the last rank's block (regenerated here from its seed): data that
        # never arrived, or arrived from an unfinished product, shows up as NaN or as a wrong value
        ml = blocks[-1][1] - blocks[-1][0]              # the last rank's block: its last rows_chk rows
        if strong and host_whole:
            refl = oracle_rows(*last_rows, 0)
        elif strong:
            _, jl_d, xl_d = strong_rows(synth, torch, m_total, K, nnz_row, m_total - rows_chk, m_total)
            pl = (np.arange(rows_chk + 1, dtype=np.int64) * nnz_row).astype(np.int32)
            refl = oracle_rows(pl, jl_d.cpu().numpy(), xl_d.cpu().numpy(), 0)
        else:
            pl, jl, xl = synth.csr_fixed(m, K, nnz_row, seed=synth.SEED_A + 1000 * (world - 1))
            refl = oracle_rows(pl, jl, xl, m - rows_chk)
        bufs = pipe.bufs if pipe is not None else [C_full]
        parity = 0.0
        for Cg in bufs:
            mine_blk = pipe.block(Cg, rank) if pipe is not None else Cg[blocks[rank][0]:blocks[rank][1]]
            last_blk = pipe.block(Cg, world - 1) if pipe is not None else Cg[blocks[-1][0]:blocks[-1][1]]
            parity = max(parity, max_err(mine_blk[:rows_chk].cpu().numpy(), ref0))
            parity = max(parity, max_err(last_blk[ml - rows_chk:ml].cpu().numpy(), refl))
            for r in range(world):                    # (padding rows of a ragged slot are never written: only the blocks are checked)
                blk = pipe.block(Cg, r) if pipe is not None else Cg[blocks[r][0]:blocks[r][1]]
                assert bool(torch.isfinite(blk).all()), "gathered C holds unwritten (NaN) entries"
    tol = 1e-10 if dtype == "f64" else 2e-5
    assert parity <= tol, f"bench output differs from the oracle: {parity}"

    kernel_name = lib.mxd_spmm_last_kernel().decode()      # which kernel AUTO / --algo actually launched
    tag = None
    if not args.custom and (args.layout, args.algo, args.panels, args.wg_per_cu) == ("colmajor", 0, 0, 0) and not dist_on:
        tag = {"cfg2": "cfg2-default", "cfg5": "cfg5-shard-default"}.get(cfg["name"])   # the `workload` string of profiles/*_pmc.json
    traffic = committed_traffic(kernel_name, tag, kern_avg_s * 1e3) if tag else (None, None, "not the profiled command line")
    gb = nnz * n * s_dense
    res = {
        "value": round(flops_all_step * steps / elapsed / 1e9, 2),
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "workload": f"dgRMatrix {m_total if strong else m}x{K} nnz/row={nnz_row} (CSR values f64) %*% dense {K}x{n} {dtype} "
                    f"({cfg['label']}); C {'col' if colmajor else 'row'}-major"
                    + ((f"; the WHOLE matrix, cut into {world} nnz-balanced row blocks (rank 0: {m} rows), one per GPU, + RCCL "
                        f"all-gather of C" if strong else f"; one such row block per GPU + RCCL all-gather of C ({world}x{m} rows)")
                       + (", gather of step k under the product of step k+1" if pipe is not None else "")
                       if dist_on else ""),
        "roofline": roofline(alg_bytes, kern_avg_s, traffic=traffic[0], traffic_source=traffic[1],
                             **({"traffic_refused": traffic[2]} if traffic[2] else {}), kernel=kernel_name,
                             kernel_avg_ms=round(kern_avg_s * 1e3, 4), kernel_min_ms=round(float(kern_ms.min()), 4),
                             # secondary, non-scoring: what actually bounds the kernel.  Every nonzero gathers one B
                             # row: nnz * n * s bytes move from L2 into the CUs' L1 whatever the schedule
                             l2_to_l1_gather={"bytes_per_launch": int(gb),
                                              "achieved_GBps": round(gb / kern_avg_s / 1e9, 0),
                                              "guide_ceiling_GBps": [16000, 22000]}),
        "kernel_gflops": round(flops_rank_step / kern_avg_s / 1e9, 1),
        "parity_max_err_over_max_abs_vs_oracle": parity,
        "spmm_call_avg_ms": round(float(step_ms.mean()), 4),
        "dims": {"rows_per_gpu": m, "cols": K, "nnz_per_row": nnz_row, "dense_cols": n},
    }
    if dist_on:
        gather_s = max(elapsed / steps - float(step_ms.mean()) / 1e3, 1e-9) if pipe is None else elapsed / steps
        slot = max(b - a for a, b in blocks)
        res["allgather"] = {"bytes_received_per_gpu": int((world - 1) * slot * n * s_dense),
                            "approx_ms": round(gather_s * 1e3, 3),
                            "approx_GBps_in_per_gpu": round((world - 1) * slot * n * s_dense / gather_s / 1e9, 1)}
        res["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                              "ranks_in_an_rccl_all_reduce_of_ones": nranks_seen, "scaling": args.scaling,
                              "row_blocks": [b - a for a, b in blocks], "rows_total": m_total}
        # the same flops over the slowest rank's local product alone (no exchange): what the GPUs do; `value` is what the job does
        res["compute_only_gflops"] = round(flops_all_step / (slowest_local_ms / 1e3) / 1e9, 2)
        res["roofline"]["scope"] = "rank 0's row block, its dominant kernel only (value is the whole job, all-gather included)"
    if kernel_name == "spmm_plan_kernel" and args.algo in (0, 4):
        res["plan"] = {"value_is": "plan kept on the DeviceCSR: built once per matrix in the untimed setup calls, every timed "
                                   "step = repack of B + the sweep" if args.algo == 0 else
                                   "plan rebuilt from plain CSR inside every timed step (--algo 4)",
                       "info": A.plan_info() if A._plan is not None else None}
    if want_steady and kernel_name == "spmm_plan_kernel" and args.algo in (0, 4) and not dist_on:
        # the other form, same loop: --algo 0 -> the C-ABI's own AUTO (plan rebuilt from plain CSR inside every step: what a
        # caller pays who brings a new matrix every time); --algo 4 -> the kept plan
        other_keeps = args.algo == 4

        def other():
            if other_keeps:
                D.spmm_planned(A, B, out=C_loc, colmajor=colmajor)
            else:
                D.spmm(A, B, out=C_loc, colmajor=colmajor, keep_plan=False)
        for _ in range(3):
            other()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            other()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        res["plan"]["kept_plan" if other_keeps else "rebuild_every_step"] = {
            "ms_per_step": round(dt * 1e3, 4), "GFLOP/s": round(flops_rank_step / dt / 1e9, 1)}
    if want_cpu:
        if p is None:            # a device-drawn block: the CPU sample is its first rows, read back
            rs = min(m, 60_000)
            e = rs * nnz_row
            p_c = (np.arange(rs + 1, dtype=np.int64) * nnz_row).astype(np.int32)
            res["cpu_baseline"] = cpu_baseline_spmm(args, p_c, A.indices[:e].cpu().numpy(), A.values[:e].cpu().numpy(), B_host, dtype)
        else:
            res["cpu_baseline"] = cpu_baseline_spmm(args, p, j, x, B_host, dtype)
    res["_host"] = (p, j, x, A, B, B_host)          # handed to the extras; removed before printing
    return res


# ------------------------------------------------------------------------------------------------------ extras
def extras(args, torch, D, synth, _lib, host, want_cpu):
    """configs[2]: SpMV + 200k-row gather on cfg2's CSR; configs[3]: CSR + CSR / CSR * CSR at full size (operands drawn
    on the device); one export-level call from host memory.  Each entry: time per call, roofline of its algorithmic
    bytes against HBM peak, a parity check against the oracle, and the oracle timed on the host."""
    from oracle import oracle as O
    p, j, x, A, B, B_host = host
    threads = O.max_threads()

    def timeit(fn, reps=10):
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps / 1e3

    def cpu_time(fn, reps=2):
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        return best
    res = {}
    m, K, nnz = A.m, A.K, A.nnz

    import ctypes as C
    from matrixextra_amd import exports as G
    lib = _lib.load()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]

    def section(name, fn):
        """one leg of the extras: a failure is recorded in the line, the other legs (and the headline) still report.
        MXGPU_BENCH_EXTRAS_SKIP / _ONLY (comma-separated leg names): tools/make_profiles.sh profiles legs that launch the same
        kernel on different workloads in separate runs"""
        skip = [q for q in os.environ.get("MXGPU_BENCH_EXTRAS_SKIP", "").split(",") if q]
        only = [q for q in os.environ.get("MXGPU_BENCH_EXTRAS_ONLY", "").split(",") if q]
        if name in skip or (only and name not in only):
            return
        try:
            fn()
        except Exception as exc:                         # noqa: BLE001 - anything: the JSON line must still come out
            import traceback
            res.setdefault("errors", {})[name] = (repr(exc)[:400], traceback.format_exc()[-1500:])
        finally:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

    # ---- configs[2] SpMV
    def _spmv_cfg3():
        v_host = synth.dense_normal(K, 1).reshape(-1)
        v = torch.from_numpy(v_host).cuda()
        y = D.spmv(A, v)
        t = timeit(lambda: D.spmv(A, v), reps=20)
        byts = 4 * (m + 1) + 12 * nnz + 8 * K + 8 * m
        ref = O.matmul_csr_dvec_numeric(p, j, x, v_host, threads)
        err = float(np.max(np.abs(y.cpu().numpy() - ref)) / np.max(np.abs(ref)))
        assert err <= 1e-12, f"SpMV differs from the oracle: {err}"
        e = {"ms": round(t * 1e3, 4), "GFLOP/s": round(2 * nnz / t / 1e9, 1),
             "roofline": roofline(byts, t, **committed_kernels_traffic([("spmv_flat_kernel", 1), ("slice_rows_kernel", 1)], t * 1e3)),
             "parity_max_err_over_max_abs_vs_oracle": err}
        if want_cpu:
            ta, t1 = cpu_time(lambda: O.matmul_csr_dvec_numeric(p, j, x, v_host, threads), 3), \
                cpu_time(lambda: O.matmul_csr_dvec_numeric(p, j, x, v_host, 1), 2)
            e["cpu_baseline"] = {"value": round(2 * nnz / ta / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
                                 "single_thread": {"value": round(2 * nnz / t1 / 1e9, 3), "unit": "GFLOP/s", "cores": 1},
                                 "stronger_baseline": "single_thread" if t1 < ta else "all_threads",
                                 "note": "the reference's loop is `schedule(dynamic)` with one row per grab (matmul.cpp:396-397); with "
                                         "32 entries per row the scheduling costs more than the row, so the restated loop is SLOWER on "
                                         "all threads than on one — compare with the single-thread figure",
                                 "sample": "the whole cfg3 SpMV, matmul_csr_dvec restated (OpenMP over rows), best of 3"}
        # the same product through a kept plan (entries regrouped by column panel so that v sits in LDS): what a solver that
        # multiplies by the same X every iteration gets; the plan build is reported beside it, never inside the figure
        A.spmv_plan(); torch.cuda.synchronize(); A.drop_spmv_plan()     # (the first build also pays for the allocator's first big blocks)
        t_build = None
        for _ in range(3):                                              # best of 3: one build in ~20 hits a 15 ms hipMalloc
            A.drop_spmv_plan(); torch.cuda.synchronize()
            t0 = time.perf_counter(); A.spmv_plan(); torch.cuda.synchronize(); tb = time.perf_counter() - t0
            t_build = tb if t_build is None else min(t_build, tb)
        yp = D.spmv_planned(A, v)
        errp = float(np.max(np.abs(yp.cpu().numpy() - ref)) / np.max(np.abs(ref)))
        assert errp <= 1e-12, f"planned SpMV differs from the oracle: {errp}"
        tp = timeit(lambda: D.spmv_planned(A, v), reps=20)
        e["steady_state_kept_plan"] = {"ms": round(tp * 1e3, 4), "GFLOP/s": round(2 * nnz / tp / 1e9, 1),
                                       "roofline": roofline(byts, tp, **committed_kernels_traffic([("spmv_plan_kernel", 1)], tp * 1e3)),
                                       "plan_build_ms": round(t_build * 1e3, 3), "parity_max_err_over_max_abs_vs_oracle": errp}
        res["spmv_cfg3"] = e

    section("spmv_cfg3", _spmv_cfg3)

    # ---- configs[2] gather of 200k random rows
    def _gather_cfg3():
        rows_host = synth.rows_with_replacement(200_000, m)
        rows = torch.from_numpy(rows_host).cuda()
        g = D.csr_gather_rows(A, rows)
        t = timeit(lambda: D.csr_gather_rows(A, rows))
        byts = 4 * 200_000 + 8 * 200_000 + 4 * 200_001 + 2 * 12 * g.nnz
        o = O.copy_csr_rows_numeric(p, j, x, rows_host)
        gp, gj, gx = g.to_host()
        assert np.array_equal(gp, o["indptr"]) and np.array_equal(gj, o["indices"]) and np.array_equal(gx, o["values"]), \
            "row gather differs from the oracle"
        e = {"ms": round(t * 1e3, 4), "nnz_out": g.nnz, "Mnnz/s": round(g.nnz / t / 1e6, 1),
             "roofline": roofline(byts, t, **committed_kernels_traffic([("gather_fused_kernel", 1)], t * 1e3)),
             "kernel": "gather_fused_kernel: lengths + look-back scan + copy in one launch, size read back once behind it",
             "parity": "bit-exact vs oracle (indptr, indices, values)"}
        t2 = timeit(lambda: D.csr_gather_rows(A, rows, one_launch=False))
        e["two_launch_form_ms"] = round(t2 * 1e3, 4)
        if want_cpu:
            t1 = cpu_time(lambda: O.copy_csr_rows_numeric(p, j, x, rows_host), 3)
            e["cpu_baseline"] = {"value": round(byts / t1 / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                                 "sample": "the whole cfg3 gather, copy_csr_rows restated (serial, as the reference), best of 3"}
        res["gather_cfg3"] = e
        del g, gp, gj, gx, o

    section("gather_cfg3", _gather_cfg3)

    # ---- configs[3] CSR (+) CSR at full size: 2M x 2M, 50 / row (nnz 1e8 each, ~50 % shared pattern)
    def _merges_cfg4():
        m4 = K4 = 2_000_000
        p1, j1, x1 = synth.device_csr_fixed(m4, K4, 50)
        p2, j2, x2 = synth.device_csr_overlapping(j1, m4, K4, 50)
        A1 = D.DeviceCSR(p1, j1, x1, m4, K4, int(j1.numel()))
        A2 = D.DeviceCSR(p2, j2, x2, m4, K4, int(j2.numel()))
        assert A1.rows_sorted() and A2.rows_sorted()
        rs = 200_000                                     # oracle sample: the first rs rows (1e7 entries per operand)
        hp1, hp2 = p1[: rs + 1].cpu().numpy(), p2[: rs + 1].cpu().numpy()
        hj1, hx1 = j1[: hp1[-1]].cpu().numpy(), x1[: hp1[-1]].cpu().numpy()
        hj2, hx2 = j2[: hp2[-1]].cpu().numpy(), x2[: hp2[-1]].cpu().numpy()
        for name, op, ofn in (("add", _lib.MX_OP_ADD, lambda: O.add_csr_elemwise(hp1, hp2, hj1, hj2, hx1, hx2, False)),
                              ("sub", _lib.MX_OP_SUB, lambda: O.add_csr_elemwise(hp1, hp2, hj1, hj2, hx1, hx2, True)),
                              ("mul", _lib.MX_OP_MUL, lambda: O.multiply_csr_elemwise(hp1, hp2, hj1, hj2, hx1, hx2))):
            R = D.csr_elemwise(op, A1, A2)
            t = timeit(lambda: D.csr_elemwise(op, A1, A2), reps=5)
            byts = 2 * 4 * (m4 + 1) + 12 * (A1.nnz + A2.nnz) + 12 * R.nnz + 4 * (m4 + 1)
            o = ofn()
            n_s = int(o["indptr"][-1])
            assert np.array_equal(R.indptr[: rs + 1].cpu().numpy(), o["indptr"]) and \
                np.array_equal(R.indices[:n_s].cpu().numpy(), o["indices"]) and \
                np.array_equal(R.values[:n_s].cpu().numpy(), o["values"]), f"CSR {name} CSR differs from the oracle"
            e = {"ms": round(t * 1e3, 4), "nnz_in": [A1.nnz, A2.nnz], "nnz_out": R.nnz,
                 "Gnnz_in/s": round((A1.nnz + A2.nnz) / t / 1e9, 2),
                 "roofline": roofline(byts, t, **committed_kernels_traffic(
                     [("merge_count_kernel<64, true>" if name == "mul" else "merge_count_kernel<64, false>", 1),
                      ("merge_fill_kernel<64, %d" % op, 1)], t * 1e3)),
                 "parity": f"bit-exact vs oracle on the first {rs} rows (indptr, indices, values)"}
            if want_cpu:
                t1 = cpu_time(ofn, 2)
                byts_s = 2 * 4 * (rs + 1) + 12 * (int(hp1[-1]) + int(hp2[-1])) + 12 * n_s + 4 * (rs + 1)
                e["cpu_baseline"] = {"value": round(byts_s / t1 / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                                     "Gnnz_in/s": round((int(hp1[-1]) + int(hp2[-1])) / t1 / 1e9, 4),
                                     "sample": f"first {rs} of {m4} rows, serial two-pointer merge restated "
                                               "(the reference is single-threaded here), best of 2"}
            res[f"csr_{name}_csr_cfg4"] = e
            del R, o
        # the sortedness check the R callers run before every merge (R/operators.R:58,64,748,754)
        A1._sorted = None
        t = timeit(lambda: (setattr(A1, "_sorted", None), A1.rows_sorted()), reps=10)
        byts = 4 * (m4 + 1) + 4 * A1.nnz
        res["rows_sorted_check_cfg4"] = {"ms": round(t * 1e3, 4),
                                         "roofline": roofline(byts, t, **committed_kernels_traffic([("rows_sorted_tile_kernel", 1)], t * 1e3))}
        # what remove_zeros runs after a subtraction has left explicit zeros behind (R/utils.R:263-330 -> misc.cpp:553-664):
        # the first operand with 30 % of its values zeroed
        gen = torch.Generator(device="cuda"); gen.manual_seed(11)
        xz = torch.where(torch.rand(x1.numel(), device="cuda", generator=gen) < 0.3, torch.zeros_like(x1), x1)
        Az = D.DeviceCSR(p1, j1, xz, m4, K4, A1.nnz)
        Rz = D.csr_drop_zeros(Az)
        t = timeit(lambda: D.csr_drop_zeros(Az), reps=5)
        byts = 2 * 4 * (m4 + 1) + 12 * Az.nnz + 12 * Rz.nnz
        oz = O.remove_zero_valued_csr_numeric(hp1, hj1, xz[: hp1[-1]].cpu().numpy(), False)
        n_s = int(oz["indptr"][-1])
        assert np.array_equal(Rz.indptr[: rs + 1].cpu().numpy(), oz["indptr"]) and \
            np.array_equal(Rz.indices[:n_s].cpu().numpy(), oz["indices"]) and \
            np.array_equal(Rz.values[:n_s].cpu().numpy(), oz["values"]), "remove_zero_valued_csr differs from the oracle"
        res["drop_zeros_cfg4"] = {"ms": round(t * 1e3, 4), "nnz_in": Az.nnz, "nnz_out": Rz.nnz,
                                  "roofline": roofline(byts, t, **committed_kernels_traffic(
                                      [("drop_count_kernel<32, 0, double>", 1), ("drop_fill_kernel<32, 0, double>", 1)], t * 1e3)),
                                  "kernels": "drop_count_kernel + scan + drop_fill_kernel (values read twice: the count's 8 B per "
                                             "entry are overhead, not algorithmic bytes)",
                                  "parity": f"bit-exact vs oracle on the first {rs} rows (indptr, indices, values)"}
        del Az, Rz, xz, oz
        del A1, A2, p1, j1, x1, p2, j2, x2
        torch.cuda.empty_cache()

    section("merges_cfg4", _merges_cfg4)

    # ---- end to end through the export-level C-ABI (host pointers in, host matrix out: what one .Call from R costs;
    # never the headline `value`)
    # The result comes from plain libc malloc, untouched, as R's allocVector hands it over (no huge-page advice: on a
    # THP=madvise machine that is 4-KiB pages unless the library asks — DESIGN §5.2); the operands are numpy arrays.
    def _export_call():
        import ctypes as C
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        libc.free.argtypes = [C.c_void_p]
        Yc = np.asfortranarray(B_host.T)
        lib = _lib.load()
        f64 = B_host.dtype == np.float64
        cfn = lib.mx_tcrossprod_csr_dense_numeric if f64 else lib.mx_tcrossprod_csr_dense_float32
        m_, n_, K_ = p.size - 1, Yc.shape[0], Yc.shape[1]
        c_bytes = m_ * n_ * B_host.dtype.itemsize

        def call():
            q = libc.malloc(c_bytes)
            t0 = time.perf_counter()
            _lib.check(cfn(C.c_void_p(p.ctypes.data), C.c_void_p(j.ctypes.data), C.c_void_p(x.ctypes.data), C.c_int(m_),
                           C.c_void_p(Yc.ctypes.data), C.c_int(n_), C.c_int(K_), C.c_int(1), C.c_void_p(q)))
            return time.perf_counter() - t0, q
        def phases():
            buf = C.create_string_buffer(512)
            lib.mx_last_call_phases(buf, C.c_size_t(512))
            out_ = {}
            for item in buf.value.decode().split(";")[1:]:
                k, _, v = item.partition("=")
                try:
                    out_[k] = float(v)
                except ValueError:
                    out_[k] = v
            return out_
        _, q = call()                                  # first call of the process: allocates the library's grow-only device scratch
        for _ in range(2):                             # two more untimed cold calls: the first 1-GB results of a process come from
            libc.free(q)                              # freshly mapped, not yet compacted memory (46 ms where later calls take 38)
            lib.mx_cache_invalidate(None)
            _, q = call()
        cold, cached, ph_cold, ph_cached, cold_rows = [], [], [], [], []
        for k in range(13):
            libc.free(q)                              # (freeing the previous 1 GB result is not part of the next call)
            if k < 8:
                lib.mx_cache_invalidate(None)         # CSR not on the device: upload + compute + download
            if 5 <= k < 8:                            # the other cold form (whole CSR up first, then column blocks), same process
                os.environ["MXGPU_EXPORT_COLD_COLS"] = "2"
            t, q = call()
            os.environ.pop("MXGPU_EXPORT_COLD_COLS", None)
            if k < 5:
                cold.append(t); ph_cold.append(phases())
            elif k < 8:
                cold_rows.append(t)
            else:
                cached.append(t); ph_cached.append(phases())
        out = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_double if f64 else C.c_float)), shape=(n_, m_)).T   # view; freed below
        n = out.shape[1]
        ref = np.zeros(2048 * n, dtype=B_host.dtype)
        O.gemm_csr_drm_as_drm(2048, n, p[:2049], j, x, B_host.reshape(-1), n, ref, n, threads, True)
        err = max(float(np.max(np.abs(out[:2048] - ref.reshape(2048, n))) / np.max(np.abs(ref))),
                  float(abs(out[-1].sum() - (x[p[-2]:p[-1]] @ B_host[j[p[-2]:p[-1]]]).sum()) / np.max(np.abs(ref))))
        assert err <= 1e-9, f"export-level SpMM differs from the oracle: {err}"
        del out
        libc.free(q)

        def med(v):
            return float(np.median(v))

        def phase_medians(ps):
            keys = [k for k in ps[0] if all(isinstance(q_.get(k), float) for q_ in ps)]
            return {k: round(med([q_[k] for q_ in ps]), 3) for k in keys}
        res["export_call_end_to_end"] = {
            "ms_cold": round(med(cold) * 1e3, 2), "ms_csr_cached": round(med(cached) * 1e3, 2),
            "ms_cold_all": [round(v * 1e3, 2) for v in cold], "ms_csr_cached_all": [round(v * 1e3, 2) for v in cached],
            "ms_cold_column_block_form": [round(v * 1e3, 2) for v in cold_rows],
            "phases_ms_cold": phase_medians(ph_cold), "phases_ms_csr_cached": phase_medians(ph_cached),
            "csr_state_cached": ph_cached[-1].get("csr"),
            "GFLOP/s_cold": round(2 * nnz * n / med(cold) / 1e9, 1), "GFLOP/s_csr_cached": round(2 * nnz * n / med(cached) / 1e9, 1),
            "parity_max_err_over_max_abs_vs_oracle": err,
            "note": "mx_tcrossprod_csr_dense_* on cfg2: ordinary (pageable) host vectors in, a freshly malloc'ed, untouched host "
                    "matrix out (as R allocates it); cold = CSR not on the device (upload, compute and download pipelined over "
                    "row blocks x column groups of the result, each group's pages touched and registered on their own; "
                    "ms_cold_column_block_form = the other cold form, forced: whole CSR up first, then column blocks), csr_cached = the same host vectors again (device-side CSR cache and the matrix's kept plan; "
                    "download-bound: 1 GB over PCIe); medians of 5 calls each, phases = medians of mx_last_call_phases "
                    "(setup = B upload queued + cache look-up; block 0 = first product queued; piece 0 = the first column group's pages exist and "
                    "are registered; queued = all blocks and downloads queued; fingerprint = cache key of a new "
                    "operand hashed while the queues drain; kernels = compute queue empty; D2H C = download queue empty)"}

    section("export_call", _export_call)

    # ---- the skewed variant of the headline matrix (log-normal row lengths, sigma = 1, same mean, same B)
    def _spmm_skewed():
        ps, js, xs = synth.csr_skewed_fast(m, K, 32, seed=synth.SEED_A, sigma=1.0)
        As = D.DeviceCSR.from_host(ps, js, xs, K)
        n_d = int(B.shape[1])
        Cs = torch.empty((n_d, m), dtype=B.dtype, device="cuda")
        lib.mxd_spmm_kernel_timing(0)
        D.spmm(As, B, out=Cs, colmajor=True)
        kname = lib.mxd_spmm_last_kernel().decode()
        rows_chk = 2048
        r0 = int(np.argmax(ps[1:] - ps[:-1]))                            # a block that contains the longest row
        r0 = max(0, min(m - rows_chk, r0 - rows_chk // 2))
        lo_, hi_ = int(ps[r0]), int(ps[r0 + rows_chk])
        ref = np.zeros(rows_chk * n_d, dtype=B_host.dtype)
        O.gemm_csr_drm_as_drm(rows_chk, n_d, (ps[r0:r0 + rows_chk + 1] - ps[r0]).astype(np.int32), js[lo_:hi_].copy(), xs[lo_:hi_].copy(),
                              B_host.reshape(-1), n_d, ref, n_d, threads, True)
        got = Cs[:, r0:r0 + rows_chk].t().cpu().numpy()
        err = float(np.max(np.abs(got - ref.reshape(rows_chk, n_d))) / np.max(np.abs(ref)))
        assert err <= 1e-10, f"skewed SpMM differs from the oracle: {err}"
        lib.mxd_spmm_kernel_timing(1)
        t = timeit(lambda: D.spmm(As, B, out=Cs, colmajor=True), reps=10)
        kt = (C.c_float * 64)()
        kc = C.c_int(0)
        _lib.check(lib.mxd_spmm_kernel_times(kt, 64, C.byref(kc)))
        lib.mxd_spmm_kernel_timing(0)
        k_s = float(np.mean(kt[2:kc.value])) / 1e3 if kc.value > 2 else t
        t_rw = timeit(lambda: D.spmm(As, B, out=Cs, colmajor=True, algo=1), reps=5)
        byts = synth.spmm_algorithmic_bytes(m, K, n_d, As.nnz, B_host.dtype.itemsize)
        res["spmm_cfg2_skewed"] = {
            "ms": round(t * 1e3, 4), "GFLOP/s": round(2.0 * As.nnz * n_d / t / 1e9, 1), "kernel": kname, "nnz": As.nnz,
            "row_lengths": {"mean": round(As.nnz / m, 2), "max": int((ps[1:] - ps[:-1]).max()), "empty_rows": int((ps[1:] == ps[:-1]).sum())},
            "plan": As.plan_info() if As._plan is not None and As._plan_ready else None,
            "roofline": roofline(byts, k_s, kernel_avg_ms=round(k_s * 1e3, 4), call_frac=round(byts / t / 1e9 / HBM_PEAK_GBS, 4)),
            "row_wave_kernel_ms": round(t_rw * 1e3, 4),
            "parity_max_err_over_max_abs_vs_oracle": err,
            "note": "cfg2's shape with log-normal row lengths (sigma 1, mean 32, synth.csr_skewed_fast): octets whose bundles are "
                    "uneven take the plan's dealt layout; plan kept on the DeviceCSR as for `value`; roofline.frac is kernel-level"}
        del As, Cs, ps, js, xs

    section("spmm_skewed", _spmm_skewed)

    # ---- the one workload the reference publishes a number for (vignette Rmd:247-251): dense 100 x 1e4 %*% CSC 1e4 x 1e4,
    # density 0.05 -> matmul_dense_csc_numeric (matmul.cpp:188-235: gemm_csr_drm_as_drm with the CSC read as CSR of its transpose)
    def _vignette_dense_csc():
        from matrixextra_amd import exports as G
        mv, Kv, nv = 10_000, 10_000, 100
        pv_, jv_, xv_ = synth.csr_fixed(mv, Kv, 500, seed=7)               # 5e6 entries: density 0.05 (columns of the CSC)
        Xd = np.asfortranarray(synth.dense_normal(nv, Kv, seed=8))         # Y_dense, column-major 100 x 1e4
        outv = G.matmul_dense_csc_numeric(Xd, pv_, jv_, xv_, 1)
        refv = O.matmul_dense_csc(Xd, pv_, jv_, xv_, threads, True)
        errv = float(np.max(np.abs(outv - refv)) / np.max(np.abs(refv)))
        assert errv <= 1e-12, f"dense x CSC differs from the oracle: {errv}"
        te = []
        for _ in range(7):
            t0 = time.perf_counter()
            G.matmul_dense_csc_numeric(Xd, pv_, jv_, xv_, 1)
            te.append(time.perf_counter() - t0)
        Av = D.DeviceCSR.from_host(pv_, jv_, xv_, Kv)
        Bv = torch.from_numpy(np.ascontiguousarray(Xd.T)).cuda()          # K x n row-major = X column-major
        algos = {}
        for name, kw in (("auto", dict(algo=0)), ("row_wave", dict(algo=1)), ("slab", dict(algo=2)), ("row_split", dict(algo=4)),
                         ("row_split_one_panel", dict(algo=4, npanels=1)), ("tile", dict(algo=5))):
            D.spmm(Av, Bv, colmajor=False, **kw)
            kn = lib.mxd_spmm_last_kernel().decode()
            # (the first leg follows seconds of CPU work — the oracle, the export timings —: 4 ms of launches do not bring an
            # idle GPU back to its clocks, so every leg is timed after 200 launches of itself, best of two rounds)
            timeit(lambda: D.spmm(Av, Bv, colmajor=False, **kw), reps=200)
            algos[name] = {"ms": round(min(timeit(lambda: D.spmm(Av, Bv, colmajor=False, **kw), reps=20) for _ in range(2)) * 1e3, 4),
                           "kernel": kn}
        tpl = timeit(lambda: D.spmm_planned(Av, Bv, colmajor=False), reps=20)
        algos["planned_kept_plan"] = {"ms": round(tpl * 1e3, 4), "kernel": "spmm_plan_kernel"}
        tdev = algos["auto"]["ms"] / 1e3
        bytv = synth.spmm_algorithmic_bytes(mv, Kv, nv, Av.nnz, 8)
        gbv = Av.nnz * nv * 8
        vig_traffic = {}
        on_chip = {}
        if algos["auto"]["kernel"] == "spmm_tile_kernel":            # AUTO = the LDS-tile kernel (round 5): one launch
            import ctypes as C
            ca, cb, ct, cp, ccpl = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_int()
            _lib.check(lib.mxd_spmm_auto_cost2(C.c_int(mv), C.c_int(nv), C.c_int(Kv), C.c_int64(Av.nnz), C.c_int(1), C.c_int(0), C.c_int(0), C.c_int(1),
                                               C.byref(ca), C.byref(cb), C.byref(ct), C.byref(cp), C.byref(ccpl)))
            vig_traffic = committed_kernels_traffic([("spmm_tile_kernel", 1)], tdev * 1e3)
            # what the kernel moves on chip: every entry reads one (padded) row of the slab from LDS; B leaves L2 once per
            # (row block, slab, K-tile): workgroups x tiles x 64 KB
            wslab = 32 * ccpl.value
            nsl = -(-nv // wslab)
            lds_bytes = Av.nnz * nsl * wslab * 8
            geo = {"slab_bytes": 256 * ccpl.value, "slabs": nsl}
            on_chip = {"lds_read": {"bytes_per_launch": int(lds_bytes), "achieved_GBps": round(lds_bytes / tdev / 1e9, 0),
                                    "guide_ceiling_GBps": 150000, "useful_bytes": int(gbv)},
                       "model_us": {"tile": round(ct.value, 1), "row_split": round(ca.value, 1), "planned": round(cb.value, 1)}, "geometry": geo}
        elif algos["auto"]["kernel"] == "spmm_rowsplit_kernel":      # AUTO's launches: the cursor kernel + one launch per column panel
            import ctypes as C
            ca, cb, cp = C.c_double(), C.c_double(), C.c_int()
            _lib.check(lib.mxd_spmm_auto_cost(C.c_int(mv), C.c_int(nv), C.c_int(Kv), C.c_int64(Av.nnz), C.c_int(0), C.c_int(0), C.byref(ca),
                                              C.byref(cb), C.byref(cp)))
            ks = [("spmm_rowsplit_kernel<double, 2, 64, false", cp.value)] + ([("rowsplit_cursors_kernel", 1)] if cp.value > 1 else [])
            vig_traffic = committed_kernels_traffic(ks, tdev * 1e3)
            vig_traffic["column_panels"] = cp.value
        ev = {"device_ms": algos["auto"]["ms"], "export_ms_median": round(float(np.median(te[2:])) * 1e3, 3),
              # what bounded it through round 4: every entry gathers one 800-byte row of B (8 MB, twice an XCD's L2) through the
              # CUs' L1s — the figure the row-split kernel would need (its time: kernels_ms.row_split)
              "l2_to_l1_gather": {"bytes_per_launch": int(gbv), "achieved_GBps": round(gbv / tdev / 1e9, 0),
                                  "guide_ceiling_GBps": [16000, 22000], "bare_gather_GBps_round2": 28000,
                                  "note": "nnz * n * 8: bytes a register-gather kernel pulls from L2; the tile kernel serves them from LDS"
                                  if algos["auto"]["kernel"] == "spmm_tile_kernel" else "nnz * n * 8"},
              "GFLOP/s_device": round(2.0 * Av.nnz * nv / tdev / 1e9, 1),
              "GFLOP/s_export": round(2.0 * Av.nnz * nv / float(np.median(te[2:])) / 1e9, 1),
              "kernels_ms": algos, "roofline": roofline(bytv, tdev, **vig_traffic), **on_chip,
              "parity_max_err_over_max_abs_vs_oracle": errv,
              "reference_published": {"ms": 72.74, "GFLOP/s": 13.7, "hardware": "unstated",
                                      "source": "inst/doc/Introducing_MatrixExtra.html:668 (vignette Rmd:247-251) — context only"}}
        if want_cpu:
            tc = cpu_time(lambda: O.matmul_dense_csc(Xd, pv_, jv_, xv_, threads, False), 3)
            tc1 = cpu_time(lambda: O.matmul_dense_csc(Xd, pv_, jv_, xv_, 1, False), 2)
            ev["cpu_baseline"] = {"value": round(2.0 * Av.nnz * nv / tc / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
                                  "ms": round(tc * 1e3, 2),
                                  "single_thread": {"value": round(2.0 * Av.nnz * nv / tc1 / 1e9, 3), "ms": round(tc1 * 1e3, 2), "cores": 1},
                                  "sample": "the whole product, matmul_dense_csc restated (gemm_csr_drm_as_drm, OpenMP dynamic), best of 3"}
        res["vignette_dense_csc"] = ev
        del Av, Bv

    section("vignette_dense_csc", _vignette_dense_csc)

    # ---- many short rows against a narrow B (sparse features x a small weight matrix): the row-split kernel's row-group form
    # (csrc/spmm_rowsplit.hip spmm_rowgroup_kernel) — gemm_csr_drm_as_drm (matmul.cpp:118-142), rows summed in storage order
    def _spmm_short_rows():
        ms_, Ks_, npr_, ns_ = 1_000_000, 10_000, 8, 16
        pq, jq, xq = synth.device_csr_fixed(ms_, Ks_, npr_, seed=31)
        Aq = D.DeviceCSR(pq, jq, xq, ms_, Ks_, int(jq.numel()))
        Bq_host = synth.dense_normal(Ks_, ns_, seed=32)
        Bq = torch.from_numpy(Bq_host).cuda()
        outq = torch.empty((ms_, ns_), dtype=torch.float64, device="cuda")
        legs = {}
        for name, kw in (("auto", dict(algo=0, keep_plan=False)), ("row_groups", dict(algo=4, npanels=1, wg_per_cu=-1)),
                         ("wave_per_row", dict(algo=4, npanels=1, wg_per_cu=1)), ("row_wave", dict(algo=1)), ("slab", dict(algo=2))):
            f = lambda: D.spmm(Aq, Bq, out=outq, colmajor=False, **kw)
            f()
            kn = lib.mxd_spmm_last_kernel().decode()
            timeit(f, reps=100)
            legs[name] = {"ms": round(min(timeit(f, reps=20) for _ in range(2)) * 1e3, 4), "kernel": kn}
        legs["planned_kept_plan"] = {"ms": round(timeit(lambda: D.spmm_planned(Aq, Bq, out=outq, colmajor=False), reps=20) * 1e3, 4),
                                     "kernel": "spmm_plan_kernel"}
        got = D.spmm(Aq, Bq, colmajor=False, keep_plan=False)
        rows = np.r_[0:256, ms_ - 256:ms_]
        ph = pq.cpu().numpy(); jh = jq.cpu().numpy(); xh = xq.cpu().numpy()
        sel = np.concatenate([np.arange(ph[r], ph[r + 1]) for r in rows])
        pp = np.concatenate([[0], np.cumsum(np.diff(ph)[rows])]).astype(np.int32)
        ref = O.tcrossprod_csr_dense(pp, jh[sel], xh[sel], np.asfortranarray(Bq_host.T), 1, True)
        bitwise = bool(np.array_equal(got[torch.from_numpy(rows).cuda()].cpu().numpy(), ref))
        assert bitwise, "row-group product differs from the storage-order FMA chain"
        t = legs["auto"]["ms"] / 1e3
        byts = synth.spmm_algorithmic_bytes(ms_, Ks_, ns_, Aq.nnz, 8)
        res["spmm_short_rows_narrow_B"] = {
            "workload": f"CSR {ms_}x{Ks_}, {npr_} entries/row, %*% dense {Ks_}x{ns_} f64, C row-major (device level, operands resident)",
            "ms": legs["auto"]["ms"], "GFLOP/s": round(2.0 * Aq.nnz * ns_ / t / 1e9, 1), "kernels_ms": legs,
            "roofline": roofline(byts, t), "l2_to_l1_gather": {"bytes_per_launch": int(Aq.nnz) * 128, "achieved_GBps": round(Aq.nnz * 128 / t / 1e9, 1)},
            "parity": "bit for bit the oracle's storage-order FMA chain on 512 sampled rows",
            "note": "AUTO (plan rebuilt per call, i.e. a one-shot product) = the row-split family's row-group form: 8 lanes own a row of A "
                    "and a 128-byte row of B, 8 rows per wavefront; before it AUTO ran the slab kernel here and one wavefront per "
                    "row for longer rows"}
        del Aq, Bq, outq
    section("spmm_short_rows", _spmm_short_rows)

    # ---- the vignette's usage loop through the export level (tools/vignette_loop.py; 60 iterations here, 200 in the GPU test)
    def _vignette_loop():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import vignette_loop
        res["vignette_lbfgs_loop"] = vignette_loop.run(iters=60)
    section("vignette_loop", _vignette_loop)

    # ---- what ONE export call costs for small operands (tools/small_calls.py has the full table): the
    # reference's own test size and a 1e5-entry matrix, p50 of the four hot-path exports beside the CPU restatement
    def _export_small_calls():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import small_calls
        pts = {"test_matmul_R_100x50": small_calls.point("test_matmul_R_100x50", 100, 50, 20, 20, 30),
               "nnz_1e5": small_calls.point("nnz_1e5", 5000, 10_000, 20, 32, 1500)}
        res["export_small_calls"] = {
            name: {leg: {"gpu_p50_us": v["gpu"]["p50_us"], "cpu_1_thread_p50_us": v["cpu_1_thread"]["p50_us"], "gpu_over_cpu": v["gpu_over_cpu"]}
                   for leg, v in pt.items() if leg != "shape"} | {"shape": pt["shape"]}
            for name, pt in pts.items()}
        res["export_small_calls"]["note"] = ("host arrays in, host arrays out through ctypes; operands + result within ~2 MiB (SpMM 6, merges 3: the measured crossovers against the regular path) take the small "
                                             "path (one pinned block up, same kernels, one block down, one sync); full table and crossovers: "
                                             "profiles/r04_small_calls.json")
    section("export_small_calls", _export_small_calls)

    # ---- AUTO's regime map is measured by tools/auto_map.py (4 minutes): its committed summary rides along
    def _auto_map_summary():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import prof_common as PC
        files = PC.newest_round("auto_map.json")
        if files:
            doc = json.load(open(files[-1]))
            res["auto_map"] = dict(doc["summary"], source=os.path.relpath(files[-1], ROOT), measured="offline, by tools/auto_map.py on an MI355X box")
    section("auto_map", _auto_map_summary)

    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1 or FORCE_DIST:
        if "MASTER_ADDR" not in os.environ:                 # forced one-rank run started without torchrun
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from matrixextra_amd import _lib, device as D, synth
    lib = _lib.load()                      # fails loudly if libmxgpu.so is missing

    if args.config == "cfg5-full":
        # BASELINE configs[4] WHOLE on this one GPU, through the export-level boundary (tools/cfg5_full.py: sharded over the
        # device listed 8 times, unsharded cold / cached, and one device-level launch; parity checks inside)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cfg5_full
        r = cfg5_full.run(verbose=True)
        stream = stream_copy_probe(torch, lib, _lib)       # (AFTER the run: with torch's allocator already active in the process the cold
                                                           # unsharded call measured 262-265 ms instead of 188-197, three runs each way)
        dl = r["device_level"]
        dl["roofline"]["frac_of_stream_copy"] = round(dl["roofline"]["achieved"] / stream["GBps"], 4)
        print(json.dumps({
            "metric": "CSR x dense SpMM GFLOP/s (fp32 dense / f64 CSR values, 8M x 200k, 64 nnz/row, k=256, whole matrix on one GPU) "
                      "+ achieved HBM BW% vs CPU ref",
            "value": dl["GFLOP/s"], "unit": "GFLOP/s", "n_gpus": 1, "steps": 5, "warmup": 2, "ms_per_step": dl["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dgRMatrix 8000000x200000 nnz/row=64 (CSR values f64) %*% dense 200000x256 f32 (BASELINE "
                                   "configs[4], the whole matrix on ONE MI355X); C col-major", "parallelism": "single"},
            "roofline": dl["roofline"], "stream_copy": stream, "cfg5_full": r, "device": _lib.device_name()}), flush=True)
        return
    cfg = dict(name=args.config, rows=args.rows, cols=args.cols, nnz_row=args.nnz_row, n=args.n, dtype=args.dtype,
               label=WORKLOADS[args.config]["label"] if not args.custom else "custom shape")
    stream = stream_copy_probe(torch, lib, _lib) if world == 1 or rank == 0 else None
    want_cpu = not args.no_cpu_baseline                      # (rank 0's host cores; a shorter sample beside N > 1 ranks)
    if world > 1 or FORCE_DIST:
        args.cpu_seconds = min(args.cpu_seconds, 8.0)
    r = spmm_leg(args, torch, dist, D, synth, lib, _lib, cfg, world, rank, args.steps, args.warmup, want_cpu, True)
    out = None
    if rank == 0:
        host = r.pop("_host")
        dist_line = world > 1 or FORCE_DIST
        what = {"cfg2": "fp64, 1M x 100k, 32 nnz/row, k=128", "cfg5": "fp32 dense / f64 CSR values, 1M x 200k per GPU, "
                "64 nnz/row, k=256"}[args.config] if not args.custom else "custom shape"
        if args.scaling == "strong" and dist_line:
            what = (what.replace("1M x 200k per GPU", "8M x 200k over all GPUs") + f", the whole matrix row-sharded over {world} GPU(s) + "
                    "RCCL all-gather of C") if not args.custom else "custom shape, rows over all GPUs"
        elif dist_line:
            what = what.replace("1M x 100k", "1M x 100k per GPU") if not args.custom else "custom shape, rows per GPU"
        out = {
            "metric": f"CSR x dense SpMM GFLOP/s ({what}; AUTO's plan kept per matrix — one_shot_* = plan rebuilt inside every step) "
                      f"+ achieved HBM BW% vs CPU ref",
            "value": r.pop("value"), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_calls": SETUP_CALLS,
            "ms_per_step": r.pop("ms_per_step"),
            "higher_is_better": True, "scaling": args.scaling if world > 1 or FORCE_DIST else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": dict(workload=r.pop("workload"), parallelism=f"rowshard{world}" if world > 1 else "single",
                           **r.pop("dims")),
            "roofline": r.pop("roofline"),
            "stream_copy": stream,
        }
        cpu = r.pop("cpu_baseline", None)
        one = (r.get("plan") or {}).get("rebuild_every_step")
        if one:                                             # the like-for-like figure of rounds 1-2 (`value` there), at top level
            out["one_shot_ms_per_step"] = one["ms_per_step"]
            out["one_shot_gflops"] = one["GFLOP/s"]
        out.update(r)
        out["device"] = _lib.device_name()
        if cpu:
            out["cpu_baseline"] = cpu
        if world == 1 and not FORCE_DIST and not args.no_extras and not args.custom and args.config == "cfg2" and args.algo == 0:
            out["extras"] = extras(args, torch, D, synth, _lib, host, want_cpu)
            del host
            torch.cuda.empty_cache()
            # configs[4]'s per-GPU shard on this one GPU (the N > 1 workload; its N = 1 reference point)
            c5 = dict(WORKLOADS["cfg5"], name="cfg5")
            try:
                r5 = spmm_leg(args, torch, dist, D, synth, lib, _lib, c5, 1, 0, max(5, args.steps // 2), 2, False, False)
                r5.pop("_host")
                out["extras"]["cfg5_shard"] = r5
            except Exception as exc:                         # noqa: BLE001 - the headline line still has to come out
                out["extras"].setdefault("errors", {})["cfg5_shard"] = repr(exc)[:400]
    if (world > 1 or FORCE_DIST) and not args.no_extras and not args.custom and args.config == "cfg2" and args.algo == 0 \
            and args.scaling == "strong":
        # BASELINE configs[4] on the same line: the 8M x 200k f32 product, strong-scaled over the same ranks (every rank takes
        # part: the leg holds collectives).  A failure here must not cost the headline line — and must not leave ranks
        # waiting for each other: every rank first learns whether all of them got through the set-up.
        c5 = dict(WORKLOADS["cfg5"], name="cfg5", rows=args.cfg5_strong_rows,
                  label="BASELINE configs[4], the whole matrix" if args.cfg5_strong_rows == 8_000_000 else "configs[4]'s shape, fewer rows")
        try:
            torch.cuda.empty_cache()
            # bytes this rank will hold: its block of A (+ AUTO's plan, ~1.6x), B, two gathered C buffers, generator scratch
            blk_nnz = c5["rows"] // world * c5["nnz_row"] * 1.05
            need = blk_nnz * 12 * 3.5 + c5["cols"] * c5["n"] * 4 + 2.0 * c5["rows"] * c5["n"] * 4 * 1.02 + (2 << 30)
            ok = torch.tensor([1.0 if torch.cuda.mem_get_info()[0] > need else 0.0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                raise RuntimeError(f"skipped: a rank has less than {need / 2**30:.1f} GiB of device memory free")
            r5 = spmm_leg(args, torch, dist, D, synth, lib, _lib, c5, world, rank, max(3, args.steps // 4), 1, False, False)
            if rank == 0:
                r5.pop("_host")
                r5["dtype"] = "f32"
                out.setdefault("extras", {})["cfg5_strong"] = r5
        except Exception as exc:                             # noqa: BLE001
            if rank == 0:
                out.setdefault("extras", {}).setdefault("errors", {})["cfg5_strong"] = repr(exc)[:400]
    if world > 1 or FORCE_DIST:
        # The JSON line must be the LAST line on stdout.  RCCL writes its version banner through C stdio, which — stdout
        # being a pipe — sits in the C buffer until the process exits, i.e. lands AFTER a line printed from Python.  So:
        # every rank empties its C buffers, the group is torn down, the other ranks leave, and only then rank 0 prints.
        import ctypes
        libc = ctypes.CDLL(None)
        libc.fflush(None)
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
        if rank == 0 and world > 1:
            time.sleep(1.0)                                  # the other ranks' exit-time output, if any, comes first
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
