# Round-2 evidence for profiles/: kernel stats of the default bench line, HBM-side traffic of the
# headline kernel (separate PMC passes, as the microarchitecture guide prescribes), SQ counters, and
# kernel stats of the config-3/4/5 lines.  Run on the GPU box: bash tools/profile_r02.sh
set -x
O=gpurun_out/r02/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --cpu-sample 0 > $O/stats_c2.json 2> $O/stats_c2.err
for c in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$c -- python3 bench.py --cpu-sample 0 --config $c --steps 5 > $O/stats_c$c.json 2> $O/stats_c$c.err
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 > $O/pmc_tcc.json 2> $O/pmc_tcc.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_plain -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 --no-prepared-tables > $O/pmc_fetch_plain.json 2> $O/pmc_fetch_plain.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq1 -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 > $O/pmc_sq1.json 2> $O/pmc_sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $O/pmc_sq2 -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 > $O/pmc_sq2.json 2> $O/pmc_sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 > $O/pmc_grbm.json 2> $O/pmc_grbm.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
find $O -name "*.csv" -size +3000k -delete
ls -la $O $O/*/* | head -80
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wide -- python3 tools/kbench.py --k 100 --d 40 --b 200000 --paths auto,generic --rounds 3 > $O/stats_wide.log 2> $O/stats_wide.err
find $O -name "*kernel_trace.csv" -size +2000k -delete
