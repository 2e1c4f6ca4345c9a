#!/usr/bin/env python3
"""Collate the raw rocprofv3 output of tools/profile_r02.sh / profile_r02_c45.sh (under gpurun_out/r02/)
into the tracked summaries under profiles/ (newest run of every pass).  python3 tools/collect_profiles.py"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "r02", "prof")
RAW45 = os.path.join(ROOT, "gpurun_out", "r02", "prof45")
OUT = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def counters(d, kernel_substr):
    """{counter: (sum over dispatches of the kernel, dispatches)}, plus the dispatch's launch info."""
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    agg, info = collections.defaultdict(lambda: [0.0, set()]), {}
    for row in csv.DictReader(open(f)):
        if kernel_substr in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
            info = {"kernel": row["Kernel_Name"], "grid": row.get("Grid_Size"), "wg": row.get("Workgroup_Size"),
                    "lds": row.get("LDS_Block_Size"), "vgpr": row.get("VGPR_Count"), "agpr": row.get("Accum_VGPR_Count"),
                    "sgpr": row.get("SGPR_Count")}
    return {k: (v[0], len(v[1])) for k, v in agg.items()}, info


def per_dispatch(c, name):
    v, n = c[name]
    return v / n


def main():
    for tag, name in (("c2", "wave"), ("c3", "c3_loocv"), ("c4", "c4"), ("c5", "c5")):
        ks = newest(os.path.join(RAW, f"stats_{tag}", "**", "*kernel_stats.csv"))
        shutil.copy(ks, os.path.join(OUT, f"r02_{name}_kernel_stats.csv"))
        with open(os.path.join(OUT, f"r02_{name}_bench.json"), "w") as f:
            f.write(json.dumps(last_json_line(os.path.join(RAW, f"stats_{tag}.json"))) + "\n")
    shutil.copy(newest(os.path.join(RAW, "stats_wide", "**", "*kernel_stats.csv")), os.path.join(OUT, "r02_wide_kernel_stats.csv"))
    with open(os.path.join(OUT, "r02_bench_default.json"), "w") as f:
        f.write(json.dumps(last_json_line(os.path.join(RAW, "bench_default.json"))) + "\n")

    # HBM-side traffic of the headline kernel
    K = "fused_wave_kernel<float, 32, 30, 1, 40"
    fetch, info = counters(os.path.join(RAW, "pmc_fetch"), K)
    write, _ = counters(os.path.join(RAW, "pmc_write"), K)
    tcc, _ = counters(os.path.join(RAW, "pmc_tcc"), K)
    plain, pinfo = counters(os.path.join(RAW, "pmc_fetch_plain"), K)
    b = 1_000_000
    alg = last_json_line(os.path.join(RAW, "stats_c2.json"))["roofline"]["algorithmic_bytes_per_launch"]
    fk, wk = per_dispatch(fetch, "FETCH_SIZE"), per_dispatch(write, "WRITE_SIZE")
    hbm = fk * 1024 * 2 + wk * 1024
    phbm = per_dispatch(plain, "FETCH_SIZE") * 1024 * 2 + wk * 1024
    traffic = {
        "command": "rocprofv3 --pmc <counter(s)> --output-format csv -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1   "
                   "(one pass per counter group: FETCH_SIZE; WRITE_SIZE; TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum; tools/profile_r02.sh)",
        "kernel": info["kernel"], "dispatch": info, "neighbourhoods_per_launch": b,
        "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
        "TCC_EA0_RDREQ_sum": per_dispatch(tcc, "TCC_EA0_RDREQ_sum"), "TCC_HIT_sum": per_dispatch(tcc, "TCC_HIT_sum"),
        "TCC_MISS_sum": per_dispatch(tcc, "TCC_MISS_sum"),
        "correction": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-byte requests at 64 B -> doubled "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
        "hbm_bytes_per_launch_corrected": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
        "plain_tables": {"kernel": pinfo["kernel"], "FETCH_SIZE_KiB": per_dispatch(plain, "FETCH_SIZE"),
                         "hbm_bytes_per_launch_corrected": phbm, "traffic_over_algorithmic": phbm / alg},
        "reading": "prepared tables: two 128-byte lines per gathered row (31 M rows -> 62 M read requests); the plain tables add "
                   "one line per neighbour response.  A 160-byte row cannot take fewer than two lines, so 31 x 256 B + indices = "
                   "1.53 x the algorithmic bytes is the floor of this row size; L2 hits bring the measured figure slightly below.",
    }
    with open(os.path.join(OUT, "r02_wave_pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
        f.write("\n")

    # HBM-side traffic of the config-4 / config-5 kernels (same correction)
    other = {}
    for c, kern in ((4, "fused_wave_kernel<double, 64, 50, 1, 8"), (5, "fused_rhs_kernel<float, 16>")):
        if not os.path.isdir(os.path.join(RAW45, f"fetch_c{c}")):
            continue
        f_, inf = counters(os.path.join(RAW45, f"fetch_c{c}"), kern)
        w_, _ = counters(os.path.join(RAW45, f"write_c{c}"), kern)
        bj = last_json_line(os.path.join(RAW45, f"fetch_c{c}.json"))
        hb = per_dispatch(f_, "FETCH_SIZE") * 1024 * 2 + per_dispatch(w_, "WRITE_SIZE") * 1024
        algc = bj["roofline"]["algorithmic_bytes_per_launch"]
        other[str(c)] = {"kernel": inf["kernel"], "neighbourhoods_per_launch": bj["config"]["batch_per_gpu"],
                         "FETCH_SIZE_KiB": per_dispatch(f_, "FETCH_SIZE"), "WRITE_SIZE_KiB": per_dispatch(w_, "WRITE_SIZE"),
                         "hbm_bytes_per_launch_corrected": hb, "algorithmic_bytes_per_launch": algc,
                         "traffic_over_algorithmic": hb / algc}
    if other:
        with open(os.path.join(OUT, "r02_c45_pmc_traffic.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --cpu-sample 0 "
                                  "--config N --steps 3 --warmup 1 (tools/profile_r02_c45.sh); gfx950 FETCH_SIZE x 2 correction",
                       "configs": other}, f, indent=1)
            f.write("\n")

    # SQ counters: headline kernel (per two-neighbourhood task), config 4 / 5 kernels (per neighbourhood)
    def sq_table(dirs, kernel, units, unit_name):
        rows, info = {}, {}
        for d in dirs:
            c, i = counters(d, kernel)
            info = i or info
            for k in c:
                rows[k] = per_dispatch(c, k)
        lines = [f"kernel `{info.get('kernel')}`, grid {info.get('grid')} threads, {info.get('vgpr')} VGPRs (allocation granules as "
                 f"rocprofv3 reports them); values per launch and per {unit_name} ({units} per launch)", "",
                 f"| counter | per launch | per {unit_name} |", "|---|---|---|"]
        for k in sorted(rows):
            lines.append(f"| {k} | {rows[k]:.4g} | {rows[k] / units:.1f} |")
        return lines, rows

    md = ["# SQ counters of the round-2 kernels (final build)", "",
          "`rocprofv3 --pmc <8 SQ counters> -- python3 bench.py --cpu-sample 0 [--config N] --steps 3 --warmup 1`, two passes per kernel "
          "(+ one GRBM pass for the headline kernel): `tools/profile_r02.sh`, `tools/profile_r02_c45.sh`; collated by "
          "`tools/collect_profiles.py`.  SQ_* wave counters are in quad-cycles (MI355X_MICROARCH.md).", ""]
    lines, rows = sq_table([os.path.join(RAW, "pmc_sq1"), os.path.join(RAW, "pmc_sq2"), os.path.join(RAW, "pmc_grbm")], K, 500_000,
                           "two-neighbourhood task")
    md += ["## headline (config 2)", ""] + lines + [""]
    for c, kern, units in ((5, "fused_rhs_kernel<float, 16>", 500_000), (4, "fused_wave_kernel<double, 64, 50, 1, 8", 2_000_000)):
        if os.path.isdir(os.path.join(RAW45, f"sq1_c{c}")):
            lines, _ = sq_table([os.path.join(RAW45, f"sq1_c{c}"), os.path.join(RAW45, f"sq2_c{c}")], kern, units, "neighbourhood")
            md += [f"## config {c}", ""] + lines + [""]
    # readings of the headline table
    t = {k: v / 5e5 for k, v in rows.items()}
    clk_cycles = rows["GRBM_GUI_ACTIVE"] / 8  # per XCD = shader cycles of one launch
    simd_cycles_per_task = clk_cycles * 1024 / 5e5
    wave_life = t["SQ_WAVE_CYCLES"]
    md += ["## Readings (headline kernel)", "",
           f"* {t['SQ_INSTS_VALU']:.0f} VALU + {t['SQ_INSTS_SALU']:.0f} SALU + {t['SQ_INSTS_LDS']:.0f} LDS + {t['SQ_INSTS_VMEM']:.0f} VMEM "
           "wave-instructions per task (round 1: 1 533 VALU; round 2 before the per-lane exchange slots were hoisted out of the task "
           "loop: 1 467); no MFMA.",
           f"* shader cycles per launch: GRBM_GUI_ACTIVE / 8 XCDs = {clk_cycles / 1e6:.2f} M; x 1024 SIMDs / 500 000 tasks = "
           f"{simd_cycles_per_task:.0f} SIMD cycles per task.",
           f"* a wave spends {100 * t['SQ_WAIT_ANY'] / wave_life:.0f} % of its life in `s_waitcnt` (SQ_WAIT_ANY), "
           f"{100 * t['SQ_WAIT_INST_ANY'] / wave_life:.0f} % ready but not issuing (SQ_WAIT_INST_ANY), the rest issuing.",
           f"* LDS: SQ_LDS_IDX_ACTIVE = {t['SQ_LDS_IDX_ACTIVE']:.0f} LDS-array cycles per task per CU = "
           f"{100 * t['SQ_LDS_IDX_ACTIVE'] / (simd_cycles_per_task / 4):.0f} % of the CU's cycles per task ({simd_cycles_per_task / 4:.0f}); "
           f"bank conflicts {t['SQ_LDS_BANK_CONFLICT']:.0f} cycles per task.",
           "* VALU issue model (tools/ubench, profiles/r02_ubench.jsonl): ~990 packed ops x 4.4 cycles + 64 transcendental x 8.2 + the "
           f"remaining plain ops x 2.3 = ~{990 * 4.4 + 64 * 8.2 + (t['SQ_INSTS_VALU'] - 1054) * 2.3:.0f} SIMD cycles per task of "
           f"{simd_cycles_per_task:.0f}: the VALU is ~{100 * (990 * 4.4 + 64 * 8.2 + (t['SQ_INSTS_VALU'] - 1054) * 2.3) / simd_cycles_per_task:.0f} % "
           f"occupied, the LDS ~{100 * t['SQ_LDS_IDX_ACTIVE'] / (simd_cycles_per_task / 4):.0f} %, the HBM side "
           f"~{100 * 1.11 / (clk_cycles / 2.03e6):.0f} % (1.11 ms gather floor of the prepared tables); the three overlap only across "
           "the 3 waves of a SIMD -- since the exchange and elimination phases issue at a higher priority than the distance phase "
           "(s_setprio), better than round-robin arbitration let them (the same stream took 8 500 cycles per task before).", ""]
    with open(os.path.join(OUT, "r02_wave_pmc_sq.md"), "w") as f:
        f.write("\n".join(md) + "\n")
    print("headline: traffic/algorithmic = %.3f, VALU %.0f, LDS %.0f, SALU %.0f per task" % (
        hbm / alg, rows["SQ_INSTS_VALU"] / 5e5, rows["SQ_INSTS_LDS"] / 5e5, rows.get("SQ_INSTS_SALU", 0) / 5e5))


if __name__ == "__main__":
    main()
