for st in 1 2 3; do echo stage $st; MUYGPYS_HIP_LIB=$PWD/muygpys_amd/lib/variants/libstage$st.so python3 tools/kbench.py --k 100 --d 40 --b 200000 --paths auto --rounds 3 2>&1 | tail -1; done
echo full; python3 tools/kbench.py --k 100 --d 40 --b 200000 --paths auto --rounds 3 2>&1 | tail -1
