"""bench_common.py — what bench.py (the headline line) and bench_extras.py (the neighbouring workloads) share: the roofline
record, the HBM-traffic figures quoted from the committed rocprofv3 summaries (profiles/), the STREAM-copy probe and the CPU
baseline of the SpMM (the oracle timed on the host's cores: a reported baseline, never the product path)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def bound_unit_of(kernel_name):
    """Which unit saturates in the dominant kernel, from the committed counter series (tools/units_probe.py ->
    profiles/rNN_cfg2_units.json: TA / TCP / SQ passes of the sweep and of the bare gather).  A statement about the kernel, not
    about this run: quoted with its file."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prof_common as PC
    if "spmm_plan_kernel" not in kernel_name:
        return None
    for f in PC.newest_round("*cfg2_units.json"):
        try:
            d = json.load(open(f))["kernels"]
            me, bare = d["spmm_plan_kernel_cfg2"]["derived"], d["bare_gather_B_4MB_16waves_x8"]["counters"]
            ta = next(v for k, v in me.items() if k.startswith("TA_busy_share_of_kernel_cycles"))
            per = next(v for k, v in me.items() if k.startswith("TA_cycles_per_vector_memory_instruction"))
            bare_per = bare["TA_TA_BUSY_sum"]["mean"] / 32.0e6
            return (f"texture addressers {100 * ta:.0f} % busy, {per:.0f} TA cycles per 1-KiB gather instruction (bare gather: "
                    f"{bare_per:.0f}) - {os.path.relpath(f, ROOT)}")
        except Exception:
            continue
    return None


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


# ----------------------------------------------------------------------------------------------- the printed line
LINE_LIMIT = 4096            # bytes: the driver keeps ~8 KB of stdout; round 5's 21.6 KB line was not parsed
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")
EXTRAS_FILE = os.path.join(ROOT, "gpurun_out", "bench_extras.json")


def _finite(o):
    """NaN / inf -> None, numpy scalars -> Python: the line is strict JSON (allow_nan=False)."""
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, np.integer, np.bool_)):
        o = o.item()
    if isinstance(o, float) and not np.isfinite(o):
        return None
    return o


def _scalars(d, keep=None):
    """the scalar fields of a record (strings cut to 160 characters); `keep` orders / restricts them"""
    out = {}
    for k in (keep or d):
        v = d.get(k)
        if isinstance(v, str):
            out[k] = v[:160]
        elif v is None or isinstance(v, (bool, int, float)):
            if k in d:
                out[k] = v
    return out


def _extra_ms_frac(e):
    """[ms, roofline frac] of one extras record, whatever its shape (None where it has none)"""
    if not isinstance(e, dict):
        return None
    ms = next((e[k] for k in ("ms", "ms_per_step", "device_ms", "ms_cold") if isinstance(e.get(k), (int, float))), None)
    r = e.get("roofline")
    fr = r.get("frac") if isinstance(r, dict) else None
    return [ms, fr] if ms is not None else None


SUMMARY_ORDER = ("cfg5_shard", "cfg2_rowmajor", "spmv_cfg3", "spmv_cfg4_shape", "gather_cfg3", "csr_add_csr_cfg4", "csr_mul_csr_cfg4",
                 "vignette_dense_csc", "spmm_cfg2_skewed", "spmm_cfg2_zipf", "cfg5_shard_skewed", "cfg5_strong",
                 "rows_sorted_check_cfg4", "export_call_end_to_end", "spmm_short_rows_narrow_B", "csr_sub_csr_cfg4")


def headline_line(out, extras_file=EXTRAS_FILE, limit=LINE_LIMIT):
    """`out` (bench.py's full result, extras and prose included) -> the ONE line printed last on stdout: <= `limit` bytes,
    strict JSON, self-sufficient (metric, value, config, roofline and cpu_baseline as scalars, a summary of <= 12 extras
    as {name: [ms, frac]}).  The full record — every extra with its notes — goes to `extras_file`."""
    out = _finite(out)
    extras = out.pop("extras", None)
    full = dict(out, extras=extras) if extras is not None else dict(out)
    wrote = None
    if extras_file:
        try:
            os.makedirs(os.path.dirname(extras_file), exist_ok=True)
            with open(extras_file, "w") as f:
                json.dump(full, f, allow_nan=False, indent=1)
            wrote = os.path.relpath(extras_file, ROOT)
        except OSError:
            wrote = None
    line = {k: out[k] for k in REQUIRED_KEYS if k in out and k not in ("config", "roofline")}
    cfg = out.get("config", {})
    line["config"] = dict(_scalars(cfg), workload=str(cfg.get("workload", ""))[:420])
    rf = out.get("roofline") or {}
    line["roofline"] = _scalars(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                     "frac_of_stream_copy", "traffic_source", "traffic_refused", "kernel", "kernel_avg_ms",
                                     "kernel_min_ms", "bound_unit", "scope"))
    g = rf.get("l2_to_l1_gather")
    if isinstance(g, dict):
        line["roofline"]["l2_to_l1_gather_GBps"] = g.get("achieved_GBps")
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _scalars(cb, ("value", "unit", "cores", "kind", "sample"))
        st = cb.get("single_thread")
        if isinstance(st, dict):
            line["cpu_baseline"]["single_thread_value"] = st.get("value")
    for k in ("setup_calls", "one_shot_ms_per_step", "one_shot_gflops", "rowmajor_ms_per_step", "rowmajor_gflops", "kernel_gflops",
              "parity_max_err_over_max_abs_vs_oracle", "spmm_call_avg_ms", "compute_only_gflops", "device"):
        if k in out:
            line[k] = out[k]
    sc = out.get("stream_copy")
    if isinstance(sc, dict):
        line["stream_copy_GBps"] = sc.get("GBps")
    if isinstance(out.get("distributed"), dict):
        d = out["distributed"]
        line["distributed"] = dict(_scalars(d), row_blocks=(d.get("row_blocks") or [])[:16])
    if isinstance(out.get("allgather"), dict):
        line["allgather"] = _scalars(out["allgather"])
    if isinstance(out.get("single_process"), dict):
        line["single_process"] = _scalars(out["single_process"])
    if isinstance(extras, dict):
        summ = {}
        names = [k for k in SUMMARY_ORDER if k in extras] + [k for k in extras if k not in SUMMARY_ORDER]
        for k in names:
            v = _extra_ms_frac(extras[k])
            if v is not None and len(summ) < 12:
                summ[k] = v
        line["extras_summary"] = summ
        if isinstance(extras.get("errors"), dict):
            line["extras_errors"] = sorted(extras["errors"])[:8]
    if wrote:
        line["extras_file"] = wrote
    # never above the limit: optional parts go first, the required keys never
    for drop in (None, "extras_errors", "allgather", "extras_summary", "distributed", "device", "single_process"):
        if drop:
            line.pop(drop, None)
        s = json.dumps(line, allow_nan=False, separators=(",", ":"))
        if len(s.encode()) <= limit:
            break
    else:
        line["config"]["workload"] = line["config"]["workload"][:120]
        line["metric"] = line["metric"][:120]
        s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert "\n" not in s
    return s
