#!/usr/bin/env python3
"""units_probe.py — which unit saturates in spmm_plan_kernel.

The planned sweep moves nnz*n*s = 32.8 GB of B lines from L2 into the CUs at ~19 TB/s on cfg2; the bare register gather of
the same lines (tools/microbench/spmm_gather_ceiling, B shrunk to 4 MB) reaches 28 TB/s.  This takes the TA / TCP / SQ
counter series on both, one rocprofv3 --pmc pass per group (never with a trace domain; the profiled program sits right
after `--`), and writes gpurun_out/units/r06_cfg2_units.json: per counter the mean over the kernel's launches, plus the
ratios the reading needs (TA busy, share of wave cycles waiting, instructions per wave, L1 requests per line ...).

Run on the GPU box from the repo root:  python3 tools/units_probe.py [out_dir]
This script itself never touches the GPU (it only starts rocprofv3 children)."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/units")

PASSES = [
    "GRBM_GUI_ACTIVE GRBM_COUNT",
    "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum",
    "TA_BUSY_avr TA_TA_BUSY_sum TA_BUSY_max TA_BUSY_min",
    "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum",
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum",
    "TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum",
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum",
    "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum",
    "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum",
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum",
    "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum",

    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS",
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA",
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT",
]

PROGRAMS = {
    # name: (argv after `--`, kernel-name filter, which launches to keep)
    # (the sweep alone — tools/sweep_time.py, cfg2 drawn on the device, 3 + 6 products: a counter pass of the whole bench.py
    # command took 7 minutes, most of it the profiler around the thousands of small launches of its input generation and checks)
    "spmm_plan_kernel_cfg2": (["python3", os.path.join(ROOT, "tools", "sweep_time.py"), "cfg2"], "spmm_plan_kernel", "all"),
    # (UNITS_CFG5=1: the same series on configs[4]'s per-GPU shard — f32, n = 256, 64 per row — instead of the bare gather)
    **({"spmm_plan_kernel_cfg5_shard": (["python3", os.path.join(ROOT, "tools", "sweep_time.py"), "cfg5"], "spmm_plan_kernel", "all")}
       if os.environ.get("UNITS_CFG5") else {}),
    "bare_gather_B_4MB_16waves_x8": ([os.path.join(ROOT, "tools/microbench/build/spmm_gather_ceiling")], "gather<8, 1024>", "second_half"),
    "bare_gather_B_102MB_16waves_x8": ([os.path.join(ROOT, "tools/microbench/build/spmm_gather_ceiling")], "gather<8, 1024>", "first_half"),
}


def avail():
    f = os.path.join(OUT, "avail.txt")
    if not os.path.exists(f):
        r = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        open(f, "w").write(r.stdout + r.stderr)
    return open(f).read()


def run_pass(tag, i, counters, argv):
    d = os.path.join(OUT, f"{tag}_pmc_{i}")
    if not glob.glob(d + "/**/*counter_collection.csv", recursive=True) and not os.environ.get("UNITS_NO_RUN"):
        with open(d + ".log", "w") as log:
            try:
                subprocess.run(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", *argv],
                               stdout=log, stderr=subprocess.STDOUT, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp", SWEEP_SETUP="3", SWEEP_STEPS="6"), timeout=240)
            except subprocess.TimeoutExpired:
                pass
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    os.makedirs(OUT, exist_ok=True)
    av = avail()
    done = {}
    for tag, (argv, filt, keep) in PROGRAMS.items():
        if tag.startswith("bare_gather_B_102MB"):
            continue                                            # same passes as the 4 MB run: split below
        if os.environ.get("UNITS_CFG5") and not tag.endswith("cfg5_shard"):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(dict))        # counter -> dispatch -> value
        for i, p in enumerate(PASSES):
            cs = [c for c in p.split() if re.search(r'\b%s\b' % c, av)]
            if not cs:
                continue
            # (under these three counters rocprofv3 aborts inside torch's first random-number kernel and then hangs until it is
            # killed — twice, 39 and 10 minutes: they are taken on the torch-free microbenchmark only)
            if "STALLED_BY" in p and argv[0].startswith("python"):
                continue
            for r in run_pass(tag.split("_B_")[0] if "_B_" in tag else tag, i, cs, argv):
                if filt in r["Kernel_Name"]:
                    acc[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        done[tag] = acc
    res = {"what": __doc__.split("\n\n")[0], "passes": PASSES, "kernels": {}}
    for tag, (argv, filt, keep) in PROGRAMS.items():
        if tag not in done and "bare_gather_B_4MB_16waves_x8" not in done:
            continue
        acc = done[tag if tag in done else "bare_gather_B_4MB_16waves_x8"]
        k = {}
        for c, by in sorted(acc.items()):
            ids = sorted(by)
            if keep == "first_half":
                ids = ids[: len(ids) // 2]
            elif keep == "second_half":
                ids = ids[len(ids) // 2:]
            v = [by[i] for i in ids]
            if v:
                k[c] = {"launches": len(v), "mean": sum(v) / len(v)}
        res["kernels"][tag] = {"filter": filt, "counters": k, "derived": derive(k)}
    json.dump(res, open(os.path.join(OUT, "r06_cfg5_units.json" if os.environ.get("UNITS_CFG5") else "r06_cfg2_units.json"), "w"), indent=1)
    for tag, k in res["kernels"].items():
        print(tag)
        for n, v in k["derived"].items():
            print(f"   {n:48s} {v}")


def derive(k):
    g = lambda n: k.get(n, {}).get("mean")                      # noqa: E731
    d = {}

    def ratio(name, a, b, scale=1.0):
        if g(a) is not None and g(b):
            d[name] = round(scale * g(a) / g(b), 4)
    ratio("wave_cycles_waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    ratio("wave_cycles_issue_stalled (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
    ratio("wave_cycles_issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
    ratio("vmem_issue_share (SQ_ACTIVE_INST_VMEM / SQ_WAVE_CYCLES)", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES")
    ratio("lds_issue_share (SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES)", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES")
    ratio("valu_issue_share (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")
    ratio("valu_per_vmem_rd", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD")
    ratio("salu_per_vmem_rd", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD")
    ratio("lds_per_vmem_rd", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")
    ratio("lds_bank_conflict_share", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
    ratio("TA_busy_percent (TA_BUSY_avr)", "TA_BUSY_avr", "TA_BUSY_avr", 0)     # placeholder, replaced below
    if g("TA_BUSY_avr") is not None:
        d["TA_busy_percent (TA_BUSY_avr)"] = round(g("TA_BUSY_avr"), 2)
    # GRBM_GUI_ACTIVE adds up the 8 XCDs; TA_TA_BUSY_sum the 256 active TAs (TA_BUSY_avr divides by the 288 physical ones)
    ratio("TA_busy_share_of_kernel_cycles (TA_TA_BUSY_sum / 256 over GRBM_GUI_ACTIVE / 8)", "TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE", 8.0 / 256.0)
    ratio("TA_cycles_per_vector_memory_instruction (TA_TA_BUSY_sum / SQ_INSTS_VMEM)", "TA_TA_BUSY_sum", "SQ_INSTS_VMEM_RD")
    ratio("busiest_TA_share_of_kernel_cycles (TA_BUSY_max over GRBM_GUI_ACTIVE / 8)", "TA_BUSY_max", "GRBM_GUI_ACTIVE", 8.0)
    ratio("TA_addr_stalled_by_TC / TA_busy", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum")
    ratio("TA_data_stalled_by_TC / TA_busy", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum")
    ratio("TA_addr_stalled_by_TD / TA_busy", "TA_ADDR_STALLED_BY_TD_CYCLES_sum", "TA_TA_BUSY_sum")
    ratio("TCP_pending_stall / TCP_gate_en2", "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN2_sum")
    ratio("TCP_tagconflict_stall / TCP_gate_en2", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_GATE_EN2_sum")
    ratio("TCP_TD_stall / TCP_gate_en2", "TCP_TD_TCP_STALL_CYCLES_sum", "TCP_GATE_EN2_sum")
    ratio("TCP_TCR_stall / TCP_gate_en2", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_GATE_EN2_sum")
    ratio("TCP_TA_data_stall / TCP_gate_en2", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN2_sum")
    ratio("TCP_L2_read_requests_per_L1_access", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum")
    ratio("TCP_L2_read_latency_cycles_per_request", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum")
    ratio("TD_busy / TA_busy", "TD_TD_BUSY_sum", "TA_TA_BUSY_sum")
    ratio("L2_hit_rate", "TCC_HIT_sum", "TCC_REQ_sum")
    return d


if __name__ == "__main__":
    main()
