# Round-6 evidence for profiles/ (GPU box: bash tools/profile_r06.sh; then python3 tools/collect_profiles_r06.py here).
# Kernel stats of the bench lines, HBM-side traffic (separate PMC passes, gfx950 correction applied by the
# collector), SQ counters of the headline / config-4 / config-5 kernels, the un-profiled default line, and
# the headline at 8 M training points (table beyond the Infinity Cache).
set -x
O=gpurun_out/r06/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-sample 0 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- $B > $O/stats_c2.json 2> $O/stats_c2.err
for c in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$c -- $B --config $c --steps 5 > $O/stats_c$c.json 2> $O/stats_c$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dropin -- $B --route dropin --steps 10 > $O/stats_dropin.json 2> $O/stats_dropin.err
for c in 2 4 5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c$c -- $B --config $c --steps 3 --warmup 1 > $O/fetch_c$c.json 2> $O/fetch_c$c.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c$c -- $B --config $c --steps 3 --warmup 1 > $O/write_c$c.json 2> $O/write_c$c.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/sq1_c$c -- $B --config $c --steps 3 --warmup 1 > $O/sq1_c$c.json 2> $O/sq1_c$c.err
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/sq2_c$c -- $B --config $c --steps 3 --warmup 1 > $O/sq2_c$c.json 2> $O/sq2_c$c.err
done
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_c2 -- $B --steps 3 --warmup 1 > $O/tcc_c2.json 2> $O/tcc_c2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm_c2 -- $B --steps 3 --warmup 1 > $O/grbm_c2.json 2> $O/grbm_c2.err
$B --points 8000000 > $O/bench_8M.json 2> $O/bench_8M.err
$B --points 4000000 > $O/bench_4M.json 2> $O/bench_4M.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
find $O -name "*kernel_trace.csv" -size +2000k -delete
find $O -name "*.csv" -size +3000k -delete
ls $O | head -50
# round 5-6 extras: the full-size config-4 flow (k-NN + 26-trial Bayes + prediction), the k-NN scan at that shape, the
# general-smoothness Matern in fp64, a two-rank run of bench.py over gloo on this one device (weak and strong scaling)
for o in bayes-log lbfgs-analytic; do
  python3 examples/anisotropic_bayes_pipeline.py --points 10000000 --batch 2000000 --optimizer $o --out $O/c4_pipeline_2M_$o.json > $O/c4_pipeline_2M_$o.log 2>&1
done
python3 examples/anisotropic_bayes_pipeline.py --points 10000000 --batch 10000000 --optimizer bayes-log --out $O/c4_pipeline_10M_bayes-log.json > $O/c4_pipeline_10M.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3_shard8 -- $B --config 3 --batch 125000 --steps 400 --warmup 100 > $O/stats_c3_shard8.json 2> $O/stats_c3_shard8.err
python3 tools/gradbench.py > $O/gradbench_c4.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gradbench_c4 -- python3 tools/gradbench.py > $O/stats_gradbench_c4.log 2>&1
python3 tools/bayesbench.py > $O/bayesbench.txt 2>&1
python3 tools/gradbench.py --b 1000000 --n 1000000 --k 30 --d 40 --dtype f32 --aniso 0 > $O/gradbench_c2.txt 2>&1
python3 tools/gradbench.py --b 1000000 --n 1000000 --k 30 --d 40 --dtype f32 --aniso 1 > $O/gradbench_c2_aniso.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_knn -- python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 400000 > $O/stats_knn.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_knn -- python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 400000 > $O/pmc_knn.log 2>&1
python3 bench.py --gpus 2 --one-device --backend gloo --cpu-sample 0 --no-secondary --config 3 --scaling strong > $O/bench_2rank_strong.json 2> $O/bench_2rank_strong.err
python3 bench.py --gpus 2 --one-device --backend gloo --cpu-sample 0 --no-secondary --config 3 --scaling weak > $O/bench_2rank_weak.json 2> $O/bench_2rank_weak.err
python3 tools/c3bench.py --b 125000 > $O/c3_shard8_split.txt 2>&1
find $O -name "*kernel_trace.csv" -size +2000k -delete
find $O -name "*.csv" -size +3000k -delete
ls $O | head -80
