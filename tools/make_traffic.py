#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of one bench configuration into an entry of profiles/traffic.json.

    python3 tools/make_traffic.py <pmc_dir> <key> [--kernel SUBSTR] [--copy-to profiles/<name>]

<pmc_dir> is what tools/pmc.sh (or tools/profile_r02.sh) wrote under gpurun_out/: one sub-directory per
``--pmc`` pass holding ``*_counter_collection.csv``, plus (optionally) ``trace/*kernel_stats.csv``.
Per-launch means over every dispatch of the kernel whose name contains SUBSTR (default ``brats_march``).

HBM bytes per launch, exactly as /opt/skills/guides/MI355X_MICROARCH.md ("HBM") prescribes:
    FETCH_SIZE [KB] * 1024 * 2   (gfx950 tallies the 128-B requests of 16-B-per-lane loads at 64 B)
  + WRITE_SIZE [KB] * 1024       (exact for 16-B-per-lane stores)
FETCH_SIZE and WRITE_SIZE come from separate passes (the TCC block has 4 counter slots).  Cross-check printed:
TCC_MISS_sum * 128 B.  The entry records the digest of the kernel sources (tools/srchash.py) so that bench.py
can tell a stale measurement from a current one.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import srchash  # noqa: E402

CUS, SIMDS, XCDS = 256, 1024, 8


def collect(pmc_dir: str, kernel_substr: str):
    agg = collections.defaultdict(list)
    names = set()
    for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel_substr in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    names.add(r["Kernel_Name"])
    stats_ms = None
    for f in glob.glob(os.path.join(pmc_dir, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel_substr in r["Name"]:
                    stats_ms = float(r["AverageNs"]) * 1e-6
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}, sorted(names), stats_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("key")
    ap.add_argument("--kernel", default="brats_march")
    ap.add_argument("--copy-to", default=None, help="copy the CSVs of the passes to this directory under profiles/")
    ap.add_argument("--source-label", default=None, help="value of the entry's 'source' field (where the CSVs are committed)")
    ap.add_argument("--json", default=os.path.join(srchash.ROOT, "profiles", "traffic.json"))
    a = ap.parse_args()
    mean, count, names, stats_ms = collect(a.pmc_dir, a.kernel)
    if "FETCH_SIZE" not in mean or "WRITE_SIZE" not in mean:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE rows for a kernel matching {a.kernel!r} under {a.pmc_dir}: {sorted(mean)}")
    entry = {
        "source_digest": srchash.source_digest(),
        "kernel": names[0] if len(names) == 1 else names,
        "launches_averaged": count,
        "fetch_size_kb": round(mean["FETCH_SIZE"], 1),
        "write_size_kb": round(mean["WRITE_SIZE"], 1),
        "hbm_bytes_per_launch": int(mean["FETCH_SIZE"] * 1024 * 2 + mean["WRITE_SIZE"] * 1024),
    }
    if "TCC_MISS_sum" in mean:
        entry["tcc_miss"] = round(mean["TCC_MISS_sum"])
        entry["tcc_miss_x128_bytes"] = int(mean["TCC_MISS_sum"] * 128)
        if "TCC_HIT_sum" in mean:
            entry["l2_hit_rate"] = round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4)
    if stats_ms is not None:
        entry["kernel_ms_under_profiler"] = round(stats_ms, 4)
    on_chip = {}
    cyc = mean.get("GRBM_GUI_ACTIVE")
    if cyc:
        cyc /= XCDS                                        # rocprofv3 sums the 8 XCDs
        entry["gui_active_cycles_per_xcd"] = round(cyc)
        if stats_ms:
            on_chip["effective_clock_ghz"] = round(cyc / (stats_ms * 1e-3) / 1e9, 3)
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in mean:
            entry["tcp_cache_line_accesses"] = round(mean["TCP_TOTAL_CACHE_ACCESSES_sum"])
            # line look-ups per clock per CU (NOT a roof by itself: the micro-benchmark reaches 0.93-0.99 with one line per quad or
            # with bank conflicts, 1.67-1.76 with conflict-free multi-line quads)
            on_chip["l1_tag_accesses_per_clk_per_cu"] = round(mean["TCP_TOTAL_CACHE_ACCESSES_sum"] / cyc / CUS, 4)
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in mean and "TCP_TCC_READ_REQ_sum" in mean:
            # vector-L1 hit rate: line look-ups that did not go on to L2 (read requests to the TCC per look-up)
            entry["tcp_tcc_read_requests"] = round(mean["TCP_TCC_READ_REQ_sum"])
            on_chip["l1_hit_rate"] = round(1.0 - mean["TCP_TCC_READ_REQ_sum"] / mean["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
        if "SQ_INSTS_VALU" in mean:
            entry["valu_instructions"] = round(mean["SQ_INSTS_VALU"])
            # a SIMD-32 retires one wave64 VALU instruction per 2 clocks at best (guide: v_fma_f32 2 cyc; fp64 and
            # transcendental instructions take longer, so this is a lower bound of the VALU pipe's busy share)
            on_chip["valu_issue_fraction_of_peak"] = round(mean["SQ_INSTS_VALU"] * 2.0 / (SIMDS * cyc), 4)
        if "SQ_INSTS_VMEM_RD" in mean:
            entry["vmem_read_instructions"] = round(mean["SQ_INSTS_VMEM_RD"])
            # what the texture path (TA -> vector L1 -> TD) spends per wave-level gather; tools/micro/tcp_tag_rate.hip measured the
            # floor for all-hit dwordx4 gathers at 16.2-17.6 clk (one line per quad), 19.2 (two lines, different 16-B slots),
            # 32 (two lines, same slot), 36 / 64 (four lines): profiles/r03_tcp_tag_rate.txt
            on_chip["clk_per_wave_gather_per_cu"] = round(cyc * CUS / mean["SQ_INSTS_VMEM_RD"], 2)
        if "TA_TA_BUSY_sum" in mean:
            on_chip["ta_busy_fraction"] = round(mean["TA_TA_BUSY_sum"] / (cyc * CUS), 4)
        if "TD_TD_BUSY_sum" in mean:
            on_chip["td_busy_fraction"] = round(mean["TD_TD_BUSY_sum"] / (cyc * CUS), 4)
        if "SQ_WAVE_CYCLES" in mean and "SQ_WAIT_ANY" in mean and "SQ_WAIT_INST_ANY" in mean:
            wc = mean["SQ_WAVE_CYCLES"]
            on_chip["wave_time"] = {"waiting_on_waitcnt": round(mean["SQ_WAIT_ANY"] / wc, 4),
                                    "issue_stalled": round(mean["SQ_WAIT_INST_ANY"] / wc, 4),
                                    "issuing": round(mean.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)}
        if "SQ_INSTS_LDS" in mean:
            entry["lds_instructions"] = round(mean["SQ_INSTS_LDS"])
        if "SQ_BUSY_CYCLES" in mean and "SQ_ACTIVE_INST_VALU" in mean:
            on_chip["sq_active_inst_valu_over_busy"] = round(mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_BUSY_CYCLES"], 4)
    if on_chip:
        entry["on_chip"] = on_chip
    if a.copy_to:
        dst = a.copy_to if os.path.isabs(a.copy_to) else os.path.join(srchash.ROOT, a.copy_to)
        os.makedirs(dst, exist_ok=True)
        for f in glob.glob(os.path.join(a.pmc_dir, "**", "*.csv"), recursive=True):
            tag = os.path.relpath(f, a.pmc_dir).split(os.sep)[0]
            base = os.path.basename(f)
            kind = "kernel_stats" if "kernel_stats" in base else "counter_collection" if "counter_collection" in base else None
            if kind:
                shutil.copy(f, os.path.join(dst, f"{tag}_{kind}.csv"))
        entry["source"] = os.path.relpath(dst, srchash.ROOT) + "/*.csv"
    if a.source_label:
        entry["source"] = a.source_label
    try:
        with open(a.json) as fh:
            db = json.load(fh)
    except (OSError, ValueError):
        db = {}
    db["_comment"] = ("Per-launch hardware counters of the dominant kernel from rocprofv3 PMC passes (one --pmc run per counter "
                      "group; tools/profile_r02.sh + tools/make_traffic.py). hbm_bytes = FETCH_SIZE[KB]*1024*2 (gfx950 tallies "
                      "128-B requests at 64 B, MI355X_MICROARCH.md HBM section; cross-check tcc_miss_x128_bytes) + "
                      "WRITE_SIZE[KB]*1024. source_digest = tools/srchash.py of the kernel sources measured; bench.py reports an "
                      "entry only while the tree still has that digest.")
    db[a.key] = entry
    with open(a.json, "w") as fh:
        json.dump(db, fh, indent=2)
        fh.write("\n")
    print(json.dumps({a.key: entry}, indent=2))


if __name__ == "__main__":
    main()
