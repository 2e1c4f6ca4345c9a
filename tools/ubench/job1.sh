set -x
mkdir -p gpurun_out/r02
./tools/ubench/ubench > gpurun_out/r02/ubench.jsonl 2> gpurun_out/r02/ubench.err
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/r02/pmc_sq1 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES -d gpurun_out/r02/pmc_sq2 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d gpurun_out/r02/pmc_grbm --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02/pmc_grbm.log 2>&1
python3 bench.py --cpu-sample 0 > gpurun_out/r02/bench0.json 2> gpurun_out/r02/bench0.err
