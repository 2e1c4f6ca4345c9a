python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
O=gpurun_out/r03/prof2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-sample 0 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- $B --config 5 --steps 5 > $O/stats_c5.json 2> $O/stats_c5.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -- $B --config 5 --steps 3 --warmup 1 > $O/fetch_c5.json 2> $O/fetch_c5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -- $B --config 5 --steps 3 --warmup 1 > $O/write_c5.json 2> $O/write_c5.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/sq1_c5 -- $B --config 5 --steps 3 --warmup 1 > $O/sq1_c5.json 2> $O/sq1_c5.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/sq2_c5 -- $B --config 5 --steps 3 --warmup 1 > $O/sq2_c5.json 2> $O/sq2_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd -- python3 tools/bwdbench.py --outs "x,ls,noise,y" > $O/stats_bwd.log 2> $O/stats_bwd.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_aniso -- python3 tools/bwdbench.py --aniso 1 --outs "x,ls,noise,y" > $O/stats_bwd_aniso.log 2> $O/stats_bwd_aniso.err
find $O -name "*kernel_trace.csv" -size +2000k -delete
find $O -name "*.csv" -size +3000k -delete
tail -2 $O/stats_bwd.log $O/stats_bwd_aniso.log; tail -c 300 $O/stats_c5.json
