set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_f_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r03_f_tests.log; [ $rc = 0 ] || exit 1
echo "== bench"; timeout -k 10 300 python -u bench.py --no-block-calib --no-cpu-baseline > gpurun_out/r03_f_bench.json 2> gpurun_out/r03_f_bench.err || { tail -20 gpurun_out/r03_f_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_f_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','frac_hbm_peak')})
for k,v in d['families'].items(): print(k, v['frac_hbm_peak'], v['ms_per_step'])
for k,v in d['families_c3_c4'].items(): print(k, v['frac_hbm_peak'], v['us_per_launch'])
"
