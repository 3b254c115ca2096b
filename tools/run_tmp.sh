IRF_CHECK_EVERY_REP=1 timeout -k 10 300 python tools/count_bench.py 50000000 10 2>&1 | grep -v amdgpu.ids | grep -v "296676.536555 max 1182.4 coverage sum 72496.439197"
bash tools/count_variants.sh 50000000 prof f1 bu2 noat 2>&1 | grep -v amdgpu.ids | tail -n +2
