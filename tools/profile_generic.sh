set -x
O=gpurun_out/r02/profgen
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/sq1 -- python3 tools/kbench.py --k 100 --d 40 --b 100000 --rounds 2 --paths generic > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $O/sq2 -- python3 tools/kbench.py --k 100 --d 40 --b 100000 --rounds 2 --paths generic > $O/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, set()])
info = {}
for f in glob.glob("gpurun_out/r02/profgen/sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fused_generic" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
            info = (row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["VGPR_Count"])
print(info)
for k, (v, d) in sorted(agg.items()):
    print(k, v / len(d) / 1e5)
PY
