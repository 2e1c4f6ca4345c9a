# SQ counters of the config-4 / config-5 kernels (two passes each).  GPU box: bash tools/profile_r02_c45.sh
set -x
O=gpurun_out/r02/prof45
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${CONFIGS:-5 4}; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/sq1_c$c -- python3 bench.py --cpu-sample 0 --config $c --steps 3 --warmup 1 > $O/sq1_c$c.json 2> $O/sq1_c$c.err
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/sq2_c$c -- python3 bench.py --cpu-sample 0 --config $c --steps 3 --warmup 1 > $O/sq2_c$c.json 2> $O/sq2_c$c.err
done
for c in ${CONFIGS:-5 4}; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c$c -- python3 bench.py --cpu-sample 0 --config $c --steps 3 --warmup 1 > $O/fetch_c$c.json 2> $O/fetch_c$c.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c$c -- python3 bench.py --cpu-sample 0 --config $c --steps 3 --warmup 1 > $O/write_c$c.json 2> $O/write_c$c.err
done
python3 - <<'PY'
import csv, glob, collections
for c in (5, 4, 2):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/r02/prof45/sq*_c{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "fused" in row["Kernel_Name"]:
                a = agg[(row["Kernel_Name"][:60], row["Counter_Name"])]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    # counter rows come per dispatch (summed over dimensions); report per dispatch
    disp = collections.Counter()
    for (k, cn), (v, n) in sorted(agg.items()):
        print(c, k, cn, v, n)
PY
find $O -name "*.csv" -size +3000k -delete
