#!/usr/bin/env python3
"""Collate the raw rocprofv3 output of tools/profile_r06.sh (gpurun_out/r06/prof) into the tracked summaries
under profiles/r06_*.  python3 tools/collect_profiles_r06.py"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "r06", "prof")
OUT = os.path.join(ROOT, "profiles")
KERNELS = {2: "fused_wave_kernel<float, 32, 30, 1, 40", 4: "fused_wave_kernel<double, 64, 50, 1, 8", 5: "fused_rhs_mf_kernel<16"}
SHAPES = {2: {"k": 30, "d": 40, "dtype": "f32"}, 4: {"k": 50, "d": 8, "dtype": "f64"}, 5: {"k": 64, "d": 40, "dtype": "f32"}}
CORRECTION = ("gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-byte requests at 64 B -> doubled "
              "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def resources(kernel_name):
    """VGPRs / spills / LDS of an instantiation from the build's own record (muygpys_amd/lib/kernel_resources.json,
    -Rpass-analysis=kernel-resource-usage): rocprofv3's VGPR_Count / LDS_Block_Size columns read 84 / 0 for every
    kernel of this library on this image."""
    import subprocess

    path = os.path.join(ROOT, "muygpys_amd", "lib", "kernel_resources.json")
    if not os.path.exists(path):
        return None
    res = json.load(open(path))
    names = sorted(res)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    want = kernel_name.replace(" ", "").replace("void", "")
    for mangled, plain in zip(names, dem):
        if plain.replace(" ", "").replace("void", "").split("(")[0] == want.split("(")[0]:
            r = res[mangled]
            return {"VGPRs": r.get("VGPRs"), "AGPRs": r.get("AGPRs"), "VGPRs_spilled": r.get("VGPRs Spill"),
                    "SGPRs_spilled": r.get("SGPRs Spill"), "scratch_bytes_per_lane": r.get("ScratchSize [bytes/lane]"),
                    "waves_per_SIMD": r.get("Occupancy [waves/SIMD]"), "LDS_static_bytes": r.get("LDS Size [bytes/block]"),
                    "source": "muygpys_amd/lib/kernel_resources.json (compiler remarks of the build)"}
    return None


def counters(d, kernel_substr):
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))  # (gpurun merges: older runs' files stay)
    agg, info = collections.defaultdict(lambda: [0.0, set()]), {}
    rows = list(csv.DictReader(open(f)))
    for row in rows:
        if kernel_substr in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
            info = {"kernel": row["Kernel_Name"], "grid": row.get("Grid_Size"), "wg": row.get("Workgroup_Size")}
    return {k: v[0] / len(v[1]) for k, v in agg.items()}, info


def main():
    for tag, name in (("c2", "wave"), ("c3", "c3_loocv"), ("c4", "c4"), ("c5", "c5"), ("dropin", "dropin")):
        ks = newest(os.path.join(RAW, f"stats_{tag}", "**", "*kernel_stats.csv"))
        shutil.copy(ks, os.path.join(OUT, f"r06_{name}_kernel_stats.csv"))
        with open(os.path.join(OUT, f"r06_{name}_bench.json"), "w") as f:
            f.write(json.dumps(last_json_line(os.path.join(RAW, f"stats_{tag}.json"))) + "\n")
    for name in ("bench_default", "bench_8M", "bench_4M"):
        with open(os.path.join(OUT, f"r06_{name}.json"), "w") as f:
            f.write(json.dumps(last_json_line(os.path.join(RAW, name + ".json"))) + "\n")

    traffic = {}
    for c in (2, 4, 5):
        fetch, info = counters(os.path.join(RAW, f"fetch_c{c}"), KERNELS[c])
        write, _ = counters(os.path.join(RAW, f"write_c{c}"), KERNELS[c])
        bj = last_json_line(os.path.join(RAW, f"fetch_c{c}.json"))
        hbm = fetch["FETCH_SIZE"] * 1024 * 2 + write["WRITE_SIZE"] * 1024
        alg = bj["roofline"]["algorithmic_bytes_per_launch"]
        traffic[str(c)] = {"shape": SHAPES[c], "kernel": info["kernel"], "dispatch": info, "resources": resources(info["kernel"]),
                           "neighbourhoods_per_launch": bj["config"]["batch_per_gpu"],
                           "FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"],
                           "hbm_bytes_per_launch_corrected": hbm, "algorithmic_bytes_per_launch": alg,
                           "traffic_over_algorithmic": hbm / alg}
    tcc, _ = counters(os.path.join(RAW, "tcc_c2"), KERNELS[2])
    traffic["2"].update({k: tcc[k] for k in ("TCC_EA0_RDREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum")})
    cmd = ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum (one pass per group) --output-format "
           "csv -- python3 bench.py --cpu-sample 0 --no-secondary [--config N] --steps 3 --warmup 1   (tools/profile_r06.sh)")
    with open(os.path.join(OUT, "r06_wave_pmc_traffic.json"), "w") as f:
        json.dump({"command": cmd, "correction": CORRECTION, "configs": {"2": traffic["2"]}}, f, indent=1)
        f.write("\n")
    with open(os.path.join(OUT, "r06_c45_pmc_traffic.json"), "w") as f:
        json.dump({"command": cmd, "correction": CORRECTION, "configs": {"4": traffic["4"], "5": traffic["5"]}}, f, indent=1)
        f.write("\n")

    md = ["# SQ counters of the round-6 build (final)", "",
          "`rocprofv3 --pmc <8 SQ counters> -- python3 bench.py --cpu-sample 0 --no-secondary [--config N] --steps 3 --warmup 1`, two passes "
          "per kernel (+ one GRBM pass for the headline kernel): `tools/profile_r06.sh`, collated by `tools/collect_profiles_r06.py`.  "
          "SQ_* wave counters are in quad-cycles (MI355X_MICROARCH.md).", ""]
    rows_by_c = {}
    batch = {c: last_json_line(os.path.join(RAW, f"fetch_c{c}.json"))["config"]["batch_per_gpu"] for c in (2, 4, 5)}
    for c, units, unit_name in ((2, batch[2] // 2, "two-neighbourhood task"), (4, batch[4], "neighbourhood"), (5, batch[5], "neighbourhood")):
        rows, info = {}, {}
        dirs = [os.path.join(RAW, f"sq1_c{c}"), os.path.join(RAW, f"sq2_c{c}")] + ([os.path.join(RAW, "grbm_c2")] if c == 2 else [])
        for d in dirs:
            cc, i = counters(d, KERNELS[c])
            info = i or info
            rows.update(cc)
        rows_by_c[c] = rows
        md += [f"## config {c}" + (" (headline)" if c == 2 else ""), "",
               f"kernel `{info.get('kernel')}`, grid {info.get('grid')} threads; registers (build record): {resources(info.get('kernel', ''))}; "
               f"values per launch and per {unit_name} ({units} per launch)", "",
               f"| counter | per launch | per {unit_name} |", "|---|---|---|"]
        md += [f"| {k} | {rows[k]:.4g} | {rows[k] / units:.1f} |" for k in sorted(rows)] + [""]
    rows = rows_by_c[2]
    t = {k: v / 5e5 for k, v in rows.items()}
    clk = rows["GRBM_GUI_ACTIVE"] / 8
    simd = clk * 1024 / 5e5
    life = t["SQ_WAVE_CYCLES"]
    md += ["## Readings (headline kernel)", "",
           f"* {t['SQ_INSTS_VALU']:.0f} VALU + {t['SQ_INSTS_SALU']:.0f} SALU + {t['SQ_INSTS_LDS']:.0f} LDS + {t['SQ_INSTS_VMEM']:.0f} VMEM "
           "wave-instructions per task (round 4: 1 067 VALU + 253 LDS; round 3: 1 027 VALU + 126 SALU + 256 LDS).",
           f"* shader cycles per launch: GRBM_GUI_ACTIVE / 8 XCDs = {clk / 1e6:.2f} M; x 1024 SIMDs / 500 000 tasks = {simd:.0f} SIMD cycles "
           "per task (round 3: 6 279).",
           f"* a wave spends {100 * t['SQ_WAIT_ANY'] / life:.0f} % of its life in `s_waitcnt` (SQ_WAIT_ANY), "
           f"{100 * t['SQ_WAIT_INST_ANY'] / life:.0f} % ready but not issuing (SQ_WAIT_INST_ANY), the rest issuing.",
           f"* LDS: SQ_LDS_IDX_ACTIVE = {t['SQ_LDS_IDX_ACTIVE']:.0f} LDS-array cycles per task per CU = "
           f"{100 * t['SQ_LDS_IDX_ACTIVE'] / (simd / 4):.0f} % of the CU's cycles per task ({simd / 4:.0f}); bank conflicts "
           f"{t['SQ_LDS_BANK_CONFLICT']:.0f} cycles per task.",
           f"* VALU: SQ_ACTIVE_INST_VALU = {t['SQ_ACTIVE_INST_VALU']:.0f} quad-cycles per task = "
           f"{100 * 4 * t['SQ_ACTIVE_INST_VALU'] / simd:.0f} % of the SIMD's cycles.", ""]
    for c, units in ((4, float(batch[4])), (5, float(batch[5]))):
        r = {k: v / units for k, v in rows_by_c[c].items()}
        md += [f"## Readings (config {c})", "",
               f"* {r['SQ_INSTS_VALU']:.0f} VALU + {r['SQ_INSTS_SALU']:.0f} SALU + {r['SQ_INSTS_LDS']:.0f} LDS wave-instructions per "
               f"neighbourhood; bank conflicts {r['SQ_LDS_BANK_CONFLICT']:.0f} of {r['SQ_LDS_IDX_ACTIVE']:.0f} LDS-array cycles; "
               f"{100 * r['SQ_WAIT_ANY'] / r['SQ_WAVE_CYCLES']:.0f} % of a wave's life in `s_waitcnt`.", ""]
    with open(os.path.join(OUT, "r06_wave_pmc_sq.md"), "w") as f:
        f.write("\n".join(md) + "\n")
    # round-5 extras: the full-size config-4 flow, the k-NN scan at its shape, fp64 general-nu timings, two-rank lines,
    # the small-launch split, shape sweeps
    for name in ("c4_pipeline_2M_bayes.json", "c4_pipeline_2M_bayes-log.json", "c4_pipeline_2M_lbfgs.json", "c4_pipeline_2M_lbfgs-analytic.json",
                 "c4_pipeline_10M_bayes-log.json", "gen64.json", "bench_2rank_strong.json", "bench_2rank_weak.json"):
        src = os.path.join(RAW, name)
        if os.path.exists(src) and os.path.getsize(src) > 2:
            try:
                obj = json.load(open(src)) if not name.startswith("bench_") and name != "gen64.json" else last_json_line(src)
            except Exception:
                continue
            with open(os.path.join(OUT, "r06_" + name), "w") as f:
                json.dump(obj, f, indent=1)
                f.write("\n")
    ks = newest(os.path.join(RAW, "stats_c3_shard8", "**", "*kernel_stats.csv"))
    if ks:
        shutil.copy(ks, os.path.join(OUT, "r06_c3_shard8_kernel_stats.csv"))
        with open(os.path.join(OUT, "r06_c3_shard8_bench.json"), "w") as f:
            f.write(json.dumps(last_json_line(os.path.join(RAW, "stats_c3_shard8.json"))) + "\n")
    ks = newest(os.path.join(RAW, "stats_gradbench_c4", "**", "*kernel_stats.csv"))
    if ks:
        shutil.copy(ks, os.path.join(OUT, "r06_backward_c4_kernel_stats.csv"))
    for name in ("c3_shard8_split.txt", "shape_sweep_f64.md", "shape_sweep_f32.md", "gradbench_c4.txt", "gradbench_c2.txt", "gradbench_c2_aniso.txt",
                 "bayesbench.txt"):
        src = os.path.join(RAW, name)
        if os.path.exists(src) and os.path.getsize(src) > 2:
            shutil.copy(src, os.path.join(OUT, "r06_" + name))
    ks = newest(os.path.join(RAW, "stats_knn", "**", "*kernel_stats.csv"))
    if ks:
        shutil.copy(ks, os.path.join(OUT, "r06_knn_kernel_stats.csv"))
    try:
        knn, info = counters(os.path.join(RAW, "pmc_knn"), "knn_scan")
        with open(os.path.join(OUT, "r06_knn_pmc.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES "
                                  "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES -- python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 400000",
                       "kernel": info.get("kernel"), "per_dispatch_average": knn,
                       "mfma_busy_share_of_simd_time": knn["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * knn["SQ_WAVE_CYCLES"]) if knn.get("SQ_WAVE_CYCLES") else None,
                       "note": "MFMA busy cycles are per SIMD; SQ_WAVE_CYCLES are quad-cycles summed over waves"}, f, indent=1)
            f.write("\n")
    except Exception as exc:  # the k-NN passes are optional evidence
        print("knn pmc:", exc)
    print(json.dumps({c: round(traffic[str(c)]["traffic_over_algorithmic"], 3) for c in (2, 4, 5)}),
          "VALU/task %.0f LDS %.0f SALU %.0f; simd cycles/task %.0f" % (t["SQ_INSTS_VALU"], t["SQ_INSTS_LDS"], t["SQ_INSTS_SALU"], simd))


if __name__ == "__main__":
    main()
