set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== rehearsal with a 1 s deadline"; MIQ_BENCH_BLOCK_DEADLINE=1 MIQ_BENCH_ONE_GPU=1 timeout -k 10 170 python -u bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-families --block-calib > gpurun_out/r03_i_g2d.json 2> gpurun_out/r03_i_g2d.err; rc=$?
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_i_g2d.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['block_calib'])"
echo rc=$rc
