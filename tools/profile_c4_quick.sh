# quick SQ counter passes for one bench config (GPU box): bash tools/profile_c4_quick.sh <config> <tag>
C=${1:-4}; TAG=${2:-c4}
O=gpurun_out/r04/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-sample 0 --no-secondary --config $C --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/sq1 -- $B > $O/sq1.json 2> $O/sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/sq2 -- $B > $O/sq2.json 2> $O/sq2.err
python3 - <<PY
import csv,glob,collections
for p in ("sq1","sq2"):
    f=sorted(glob.glob("$O/%s/**/*counter_collection.csv"%p,recursive=True))[-1]
    agg=collections.defaultdict(lambda:[0.0,set()])
    for r in csv.DictReader(open(f)):
        if "fused_" in r["Kernel_Name"] and "partials" not in r["Kernel_Name"]:
            a=agg[(r["Kernel_Name"][:70],r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for (k,c),v in sorted(agg.items()): print(k,c,v[0]/len(v[1]))
PY
find $O -name "*.csv" -size +3000k -delete
