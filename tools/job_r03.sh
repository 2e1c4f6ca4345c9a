python3 tools/scratch_bayes.py 2>&1 | grep -v Warn | tail -8
