#!/usr/bin/env python3
"""Collate what tools/profile_r06_backward.sh wrote under gpurun_out/r06/bwd into profiles/r06_backward_pmc.json and
profiles/r06_backward_kernel_stats.csv (run here, after the GPU call)."""
import collections
import csv
import glob
import json
import shutil

O = "gpurun_out/r06/bwd"
SUB = "true, false, false, true, false, true>"  # the BWD instantiation of the headline shape


def counters(d):
    f = glob.glob(f"{O}/{d}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if SUB in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    per = collections.defaultdict(list)
    for (c, _), v in acc.items():
        per[c].append(sum(v))
    return {c: sum(v) / len(v) for c, v in per.items()}


def main():
    fetch, write, sq1, sq2 = counters("fetch"), counters("write"), counters("sq1"), counters("sq2")
    b, k, d, es = 1_000_000, 30, 40, 4
    stats = glob.glob(f"{O}/stats/*/*_kernel_stats.csv")[0]
    avg = [float(r["AverageNs"]) for r in csv.DictReader(open(stats)) if SUB in r["Name"]][0]
    hbm = fetch["FETCH_SIZE"] * 1024 * 2 + write["WRITE_SIZE"] * 1024
    alg = 5336 + 2 * (k + 1) * d * es + k * es + es  # the forward's bytes + the gathered rows' gradient read and written + partials
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (one pass per group) -- python3 tools/bwdbench.py --outs x,ls,noise,y --rounds 3  (tools/profile_r06_backward.sh)",
        "correction": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM/rocprofv3 section); WRITE_SIZE as reported",
        "kernel": "mgp::fused_wave_kernel<float,32,30,1,40,true,false,false,true,false,backward> (BWD instantiation, every cotangent)",
        "neighbourhoods_per_launch": b, "kernel_ms_avg": avg / 1e6,
        "FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"],
        "traffic_bytes_per_launch": hbm, "fetch_bytes": fetch["FETCH_SIZE"] * 2048, "write_bytes": write["WRITE_SIZE"] * 1024,
        "traffic_rate_TBps": hbm / (avg * 1e-9) / 1e12,
        "algorithmic_bytes_per_neighbourhood": alg, "traffic_over_algorithmic": hbm / (alg * b),
        "sq_per_task_of_two_neighbourhoods": {c: v / (b / 2) for c, v in {**sq1, **sq2}.items()},
    }
    json.dump(out, open("profiles/r06_backward_pmc.json", "w"), indent=1)
    shutil.copy(stats, "profiles/r06_backward_kernel_stats.csv")
    print({k_: out[k_] for k_ in ("kernel_ms_avg", "fetch_bytes", "write_bytes", "traffic_over_algorithmic")})


if __name__ == "__main__":
    main()
