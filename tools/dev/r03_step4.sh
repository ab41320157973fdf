set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== tests"; timeout -k 10 400 python -u -m pytest tests/test_gpu_act_fresh.py tests/test_gpu_act_fused.py tests/test_gpu_act.py tests/test_gpu_graph_replay.py tests/test_gpu_norm_quant.py tests/test_gpu_dp.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
echo "== bench (kernels only)"; timeout -k 10 300 python -u bench.py --no-block-calib --no-cpu-baseline --no-families > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err || { tail -20 gpurun_out/r03_b_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_b_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','frac_hbm_peak')})
for k,v in d['families'].items(): print(k, v['frac_hbm_peak'], v['ms_per_step'])
"
echo "== rehearsal --gpus 2 with block calib"; MIQ_BENCH_ONE_GPU=1 timeout -k 10 500 python -u bench.py --gpus 2 --steps 3 --no-families --block-calib > gpurun_out/r03_a_g2b.json; echo rc=$?
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_a_g2b.json').read().strip().splitlines()[-1]); print(d.get('block_calib'))"
