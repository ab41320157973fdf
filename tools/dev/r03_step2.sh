set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== capture guard test"; timeout -k 10 120 python -u -m pytest tests/test_gpu_graph_replay.py tests/test_gpu_graph_step.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
echo "== graph step outcome"; timeout -k 10 120 python -u tools/dev/graph_step_outcome.py 2>&1 | tail -3 || exit 1
echo "== bench default"; timeout -k 10 400 python -u bench.py > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err || { tail -20 gpurun_out/r03_a_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_a_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','frac_hbm_peak')}); print(d['roofline'])
for k,v in d['families'].items(): print(k, v['frac_hbm_peak'], v['ms_per_step'])
for k,v in d['families_c3_c4'].items(): print(k, v)
for k,v in d['block_calib'].items(): print(k, {a:b for a,b in v.items() if a in ('step','graph_steps','steps','notes','s_per_step','s_layer_this_run')} if isinstance(v, dict) else v)
"
echo "== rehearsal --gpus 2 (c2)"; MIQ_BENCH_ONE_GPU=1 timeout -k 10 300 python -u bench.py --gpus 2 --steps 3 --no-families > gpurun_out/r03_a_g2.json 2> gpurun_out/r03_a_g2.err || { tail -20 gpurun_out/r03_a_g2.err; exit 1; }
cut -c1-400 gpurun_out/r03_a_g2.json
echo "== rehearsal --gpus 2 (c4)"; MIQ_BENCH_ONE_GPU=1 timeout -k 10 300 python -u bench.py --gpus 2 --steps 3 --workload c4 > gpurun_out/r03_a_g2c4.json 2> gpurun_out/r03_a_g2c4.err || { tail -20 gpurun_out/r03_a_g2c4.err; exit 1; }
cut -c1-400 gpurun_out/r03_a_g2c4.json
echo "== rehearsal --gpus 2 with block calib"; MIQ_BENCH_ONE_GPU=1 timeout -k 10 400 python -u bench.py --gpus 2 --steps 3 --no-families --block-calib > gpurun_out/r03_a_g2b.json 2> gpurun_out/r03_a_g2b.err || { tail -20 gpurun_out/r03_a_g2b.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_a_g2b.json').read().strip().splitlines()[-1]); print(d.get('block_calib'))"
echo "== cold capture"; timeout -k 10 400 python -u tools/dev/cold_capture_check.py 2>&1 | tee gpurun_out/r03_a_cold_capture.txt
