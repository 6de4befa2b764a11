"""bench_extras.py — the `extras` object of bench.py's N = 1 line: BASELINE configs[2] (SpMV + gather of 200k rows), configs[3]
(CSR + CSR, CSR * CSR on 2M x 2M), one export-level call from host memory, cfg2 with skewed rows and with power-law columns,
the reference's published dense x CSC product, short rows against a narrow B, the vignette's usage loop and the small-call
latencies — each with its own roofline and CPU baseline.  Nothing here is part of `value`."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from bench_common import HBM_PEAK_GBS, ROOT, STREAM, committed_kernels_traffic, committed_traffic, roofline  # noqa: F401


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
        # X %*% v at this shape (v = 16 MB, more than an XCD's L2): the one-shot flat kernel and the kept plan, whose super-panels of
        # 2^18 columns keep the slice of v that all workgroups of a launch read inside L2 (round 6; matmul.cpp:381-419)
        try:
            v4 = torch.randn(K4, dtype=torch.float64, device="cuda")
            y_flat = D.spmv(A1, v4, algo=3)
            y_plan = D.spmv_planned(A1, v4)
            errw = float((y_flat - y_plan).abs().max() / y_flat.abs().max())
            assert errw <= 1e-12, f"planned SpMV (wide) differs from the flat kernel: {errw}"
            hv = v4.cpu().numpy()
            ref_rows = np.array([np.dot(hx1[hp1[r]:hp1[r + 1]], hv[hj1[hp1[r]:hp1[r + 1]]]) for r in range(0, rs, 997)])
            erro = float(np.max(np.abs(y_plan[:rs:997].cpu().numpy() - ref_rows)) / np.max(np.abs(ref_rows)))
            assert erro <= 1e-12, f"planned SpMV (wide) differs from the host dot products: {erro}"
            t_flat, t_plan = timeit(lambda: D.spmv(A1, v4, algo=3), reps=10), timeit(lambda: D.spmv_planned(A1, v4), reps=10)
            byts_v = 4 * (m4 + 1) + 12 * A1.nnz + 8 * K4 + 8 * m4
            res["spmv_cfg4_shape"] = {"ms": round(t_plan * 1e3, 4), "one_shot_flat_ms": round(t_flat * 1e3, 4),
                                      "roofline": roofline(byts_v, t_plan, scope="kept plan: one launch per super-panel of 2^18 columns"),
                                      "one_shot_flat_frac": round(byts_v / t_flat / 1e9 / 8000.0, 4),
                                      "parity": "1e-12 against the flat kernel (all rows) and against host dot products (sampled rows)",
                                      "shape": "2M x 2M, 50 per row, v f64 (16 MB)"}
            A1.drop_spmv_plan()
            del v4, y_flat, y_plan
        except Exception as exc:                              # noqa: BLE001 - the merges' legs below must still report
            res.setdefault("errors", {})["spmv_cfg4_shape"] = repr(exc)[:300]
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

    # ---- configs[3] / configs[2] with rows of uneven length (tools/cliff_hunt_ops.py): CSR + CSR and the
    # gather of 200k rows on log-normal row lengths (sigma 1) and with a few very long rows — the lane-group widths follow the
    # MEAN row length, what does not fit takes the blocked merge / the one-workgroup-per-pair kernels / the flat copy
    def _skewed_rows_ops():
        m4 = K4 = 2_000_000
        out_ = {}
        for tag, sigma, giants in (("lognormal_sigma1", 1.0, 0), ("four_rows_of_50000", 0.0, 4)):
            ops = []
            for seed in (synth.SEED_A, synth.SEED_A2):
                pz, jz, xz = synth.device_csr_zipf(m4, K4, 50, alpha=0.0, sigma=sigma, seed=seed)
                if giants:                                  # splice very long rows in: rebuild from per-row lengths
                    lens = (pz[1:] - pz[:-1]).to(torch.int64)
                    g = torch.Generator(device="cuda"); g.manual_seed(seed)
                    at = torch.randint(0, m4, (giants,), device="cuda", generator=g)
                    lens[at] = 50_000
                    rowid = torch.repeat_interleave(torch.arange(m4, dtype=torch.int64, device="cuda"), lens)
                    key = torch.unique(rowid * K4 + torch.randint(0, K4, (int(lens.sum().item()),), dtype=torch.int64, device="cuda", generator=g))
                    rowid = key // K4
                    jz = (key - rowid * K4).to(torch.int32).contiguous()
                    pz = torch.zeros(m4 + 1, dtype=torch.int64, device="cuda")
                    torch.cumsum(torch.bincount(rowid, minlength=m4), 0, out=pz[1:])
                    pz = pz.to(torch.int32)
                    xz = torch.rand(jz.numel(), dtype=torch.float64, device="cuda", generator=g) * 2.0 - 1.0
                    del lens, rowid, key
                ops.append(D.DeviceCSR(pz, jz, xz, m4, K4, int(jz.numel())))
            A1, A2 = ops
            R = D.csr_elemwise(_lib.MX_OP_ADD, A1, A2)
            lens1 = (A1.indptr[1:] - A1.indptr[:-1])
            long_rows = torch.nonzero(lens1 > 1024).flatten()[:8].cpu().numpy().tolist()
            for r_ in [0, 1, 12345] + long_rows:                     # sampled rows, the long ones among them, against the oracle
                sl = lambda A: (A.indptr[r_:r_ + 2].cpu().numpy() - int(A.indptr[r_]), A.indices[int(A.indptr[r_]):int(A.indptr[r_ + 1])].cpu().numpy(),
                                A.values[int(A.indptr[r_]):int(A.indptr[r_ + 1])].cpu().numpy())
                (q1, c1, v1), (q2, c2, v2), (q3, c3, v3) = sl(A1), sl(A2), sl(R)
                o = O.add_csr_elemwise(q1.astype(np.int32), q2.astype(np.int32), c1, c2, v1, v2, False)
                assert np.array_equal(o["indices"], c3) and np.array_equal(o["values"], v3), f"skewed CSR + CSR differs from the oracle in row {r_}"
            t_add = timeit(lambda: D.csr_elemwise(_lib.MX_OP_ADD, A1, A2), reps=5)
            t_mul = timeit(lambda: D.csr_elemwise(_lib.MX_OP_MUL, A1, A2), reps=5)
            g = torch.Generator(device="cuda"); g.manual_seed(3)
            rows_t = torch.randint(0, m4, (200_000,), dtype=torch.int32, device="cuda", generator=g)
            t_g = timeit(lambda: D.csr_gather_rows(A1, rows_t), reps=10)
            byts = 2 * 4 * (m4 + 1) + 12 * (A1.nnz + A2.nnz) + 12 * R.nnz + 4 * (m4 + 1)
            out_[tag] = {"nnz_in": [A1.nnz, A2.nnz], "longest_row": int(lens1.max().item()),
                         "csr_add_csr": {"ms": round(t_add * 1e3, 4), "roofline": roofline(byts, t_add, scope="whole call")},
                         "csr_mul_csr_ms": round(t_mul * 1e3, 4), "gather_200k_rows_ms": round(t_g * 1e3, 4),
                         "parity": "sampled rows (the longest among them) bit-exact vs the oracle"}
            del A1, A2, R, ops
            torch.cuda.empty_cache()
        res["skewed_rows_cfg4"] = out_

    section("skewed_rows_ops", _skewed_rows_ops)

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

    # ---- cfg2's shape with what real dgRMatrix data looks like: power-law columns AND log-normal rows
    # (synth.device_csr_zipf; the vignette's own application is LibSVM real-sim, Rmd:442-502).  AUTO reads the matrix's
    # profile (csrc/profile.hip): an XCD's L2 holds the hottest rows of B, i.e. most of the gather.
    def _spmm_zipf():
        pz, jz, xz = synth.device_csr_zipf(m, K, 40, alpha=1.0, sigma=1.0, seed=synth.SEED_A)
        Az = D.DeviceCSR(pz, jz, xz, m, K, int(jz.numel()))
        n_d = int(B.shape[1])
        Cz = torch.empty((n_d, m), dtype=B.dtype, device="cuda")
        prof = Az.profile()
        lib.mxd_spmm_kernel_timing(0)
        D.spmm(Az, B, out=Cz, colmajor=True)
        kname = lib.mxd_spmm_last_kernel().decode()
        rows_chk = 2048
        php = pz[:rows_chk + 1].cpu().numpy()
        e = int(php[-1])
        ref = np.zeros(rows_chk * n_d, dtype=B_host.dtype)
        O.gemm_csr_drm_as_drm(rows_chk, n_d, php.astype(np.int32), jz[:e].cpu().numpy(), xz[:e].cpu().numpy(), B_host.reshape(-1), n_d, ref, n_d,
                              threads, True)
        got = Cz[:, :rows_chk].t().cpu().numpy()
        err = float(np.max(np.abs(got - ref.reshape(rows_chk, n_d))) / np.max(np.abs(ref)))
        assert err <= 1e-10, f"Zipf SpMM differs from the oracle: {err}"
        lib.mxd_spmm_kernel_timing(1)
        t = timeit(lambda: D.spmm(Az, B, out=Cz, colmajor=True), reps=10)
        kt = (C.c_float * 64)()
        kc = C.c_int(0)
        _lib.check(lib.mxd_spmm_kernel_times(kt, 64, C.byref(kc)))
        lib.mxd_spmm_kernel_timing(0)
        k_s = float(np.mean(kt[2:kc.value])) / 1e3 if kc.value > 2 else t
        others = {"one_shot_auto": timeit(lambda: D.spmm(Az, B, out=Cz, colmajor=True, keep_plan=False), reps=5),
                  "row_split": timeit(lambda: D.spmm(Az, B, out=Cz, colmajor=True, algo=4), reps=5),
                  "row_wave": timeit(lambda: D.spmm(Az, B, out=Cz, colmajor=True, algo=1), reps=5)}
        byts = synth.spmm_algorithmic_bytes(m, K, n_d, Az.nnz, B_host.dtype.itemsize)
        l2_rows = (4 << 20) // (n_d * B_host.dtype.itemsize)
        import math
        li = int(math.log2(l2_rows))
        res["spmm_cfg2_zipf"] = {
            "ms": round(t * 1e3, 4), "GFLOP/s": round(2.0 * Az.nnz * n_d / t / 1e9, 1), "kernel": kname, "nnz": Az.nnz,
            "profile": {"mass_of_the_rows_of_B_one_L2_holds": round(float(prof[li]), 3), "byte_share_of_those_rows": round(l2_rows / K, 4),
                        "row_length_cv": round(float(prof[32]), 3), "longest_over_mean_row": round(float(prof[33]), 1),
                        "mean_row": round(float(prof[34]), 2)},
            "roofline": roofline(byts, k_s, kernel_avg_ms=round(k_s * 1e3, 4), call_frac=round(byts / t / 1e9 / HBM_PEAK_GBS, 4)),
            "other_kernels_ms": {k: round(v * 1e3, 4) for k, v in others.items()},
            "parity_max_err_over_max_abs_vs_oracle": err,
            "note": "cfg2's shape, columns ~ 1 / rank (dealt to ids by a permutation), row lengths log-normal (sigma 1); hot columns "
                    "collide inside a row, so ~33 entries per row remain of the 40 drawn; plan kept on the DeviceCSR as for `value`; "
                    "roofline.frac is kernel-level, `traffic` null (no committed counters for this workload)"}
        del Az, Cz, pz, jz, xz

    section("spmm_zipf", _spmm_zipf)

    # ---- configs[4]'s per-GPU shard (1M x 200k, 64 per row, f32, n = 256) with log-normal row lengths: the f32 sweep's folds
    # (spmm_plan.hip plan_flag_kernel; this shape took 22 ms before round 5's flag rule, equal rows take 3.6)
    def _cfg5_shard_skewed():
        m5, K5, n5 = 1_000_000, 200_000, 256
        p5, j5, x5 = synth.device_csr_zipf(m5, K5, 64, alpha=0.0, sigma=1.0, seed=synth.SEED_A)
        A5 = D.DeviceCSR(p5, j5, x5, m5, K5, int(j5.numel()))
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        B5 = torch.randn((K5, n5), dtype=torch.float32, device="cuda", generator=g)
        C5 = torch.empty((n5, m5), dtype=torch.float32, device="cuda")
        D.spmm(A5, B5, out=C5, colmajor=True)
        kname = lib.mxd_spmm_last_kernel().decode()
        rows_chk = 1024
        ph = p5[:rows_chk + 1].cpu().numpy()
        e = int(ph[-1])
        Bh = B5.cpu().numpy()
        ref = np.zeros(rows_chk * n5, dtype=np.float32)
        O.gemm_csr_drm_as_drm(rows_chk, n5, ph.astype(np.int32), j5[:e].cpu().numpy(), x5[:e].cpu().numpy(), Bh.reshape(-1), n5, ref, n5,
                              threads, True)
        got = C5[:, :rows_chk].t().cpu().numpy()
        err = float(np.max(np.abs(got - ref.reshape(rows_chk, n5))) / np.max(np.abs(ref)))
        assert err <= 1e-5, f"skewed cfg5 shard differs from the oracle: {err}"
        t = timeit(lambda: D.spmm(A5, B5, out=C5, colmajor=True), reps=8)
        byts = synth.spmm_algorithmic_bytes(m5, K5, n5, A5.nnz, 4)
        res["cfg5_shard_skewed"] = {
            "ms": round(t * 1e3, 4), "GFLOP/s": round(2.0 * A5.nnz * n5 / t / 1e9, 1), "kernel": kname, "nnz": A5.nnz, "dtype": "f32",
            "plan": A5.plan_info() if A5._plan is not None and A5._plan_ready else None,
            "roofline": roofline(byts, t, scope="whole call (repack of B + sweep)"),
            "parity_max_err_over_max_abs_vs_oracle": err,
            "note": "configs[4]'s per-GPU shard shape, row lengths log-normal (sigma 1, mean 64), f32: dealt octets, plain LDS "
                    "read-modify-write folds except where two lane groups can fold one row in one step; plan kept"}
        del A5, B5, C5, p5, j5, x5

    section("cfg5_shard_skewed", _cfg5_shard_skewed)

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
