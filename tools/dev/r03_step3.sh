set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== tests"; timeout -k 10 300 python -u -m pytest tests/test_gpu_uniform.py tests/test_gpu_block_llama.py tests/test_gpu_graph_step.py tests/test_gpu_graph_replay.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
echo "== cold capture"; timeout -k 10 400 python -u tools/dev/cold_capture_check.py 2>&1 | tee gpurun_out/r03_a_cold_capture.txt
echo "== rehearsal --gpus 2 with block calib"; MIQ_BENCH_ONE_GPU=1 timeout -k 10 400 python -u bench.py --gpus 2 --steps 3 --no-families --block-calib > gpurun_out/r03_a_g2b.json 2> gpurun_out/r03_a_g2b.err; echo rc=$?
grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp" gpurun_out/r03_a_g2b.err | tail -40
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_a_g2b.json').read().strip().splitlines()[-1]); print(d.get('block_calib'))"
