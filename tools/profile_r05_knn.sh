set -x
cd /root/repo
O=gpurun_out/r05/knn; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_neighbors.py -q -m gpu -x 2>&1 | tail -3
{
python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 400000
python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 2000000
python3 tools/knnbench.py --n 1000000 --d 8 --k 50 --queries 1000000
python3 tools/knnbench.py --n 1000000 --d 16 --k 30 --queries 1000000
python3 tools/knnbench.py --n 1000000 --d 40 --k 30 --queries 1000000
python3 tools/knnbench.py --n 1000000 --d 64 --k 30 --queries 500000
} 2>&1 | grep -v amdgpu.ids > $O/knnbench.txt
cat $O/knnbench.txt
cd /tmp && export TMPDIR=/tmp && cd /root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_knn -- python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 400000 > $O/stats_knn.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES --kernel-include-regex knn_scan --output-format csv -d $O/pmc_knn -- python3 tools/knnbench.py --n 10000000 --d 8 --k 50 --queries 400000 > $O/pmc_knn.log 2>&1
python3 examples/anisotropic_bayes_pipeline.py --points 10000000 --batch 2000000 --optimizer lbfgs-analytic --out $O/c4_pipeline_2M_lbfgs-analytic.json > $O/pipe.log 2>&1
tail -3 $O/pipe.log
find $O -name "*kernel_trace.csv" -size +2000k -delete
find $O -name "*.csv" -size +3000k -delete
ls -R $O | head -30
