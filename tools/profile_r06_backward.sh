# Round-6 evidence for the full backward at the headline shape (GPU box: bash tools/profile_r06_backward.sh):
# kernel stats and the memory-side traffic of `tools/bwdbench.py --outs x,ls,noise,y` (separate PMC passes).
set -x
O=gpurun_out/r06/bwd
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 tools/bwdbench.py --outs x,ls,noise,y --rounds 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
find $O -name "*kernel_trace.csv" -size +2000k -delete
find $O -name "*.csv" -size +3000k -delete
find $O -type f | head -40
