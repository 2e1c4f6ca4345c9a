mkdir -p gpurun_out/r02
python3 tools/opbench.py > gpurun_out/r02/opbench.md 2>&1; cat gpurun_out/r02/opbench.md
python3 -m pytest tests -m gpu -q 2>&1 | tail -6
