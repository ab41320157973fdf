#!/bin/bash
# One-GPU rehearsal of `bench.py --gpus 2` (gloo, both ranks on GPU 0) incl. the data-parallel block calibration, once with
# the default deadline and once with a deadline of one second (the result line must then carry an error object in place of
# the block calibration and the exit code must be 0).  Run from the repository root on a GPU box.
set -o pipefail
mkdir -p gpurun_out
show() { python3 -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], {k: (v if not isinstance(v, dict) else {a: v[a] for a in ('ranks','s_per_step','steps') if a in v}) for k, v in d['block_calib'].items()})" "$1"; }
for deadline in 420 1; do
  echo "== deadline $deadline s"
  MIQ_BENCH_BLOCK_DEADLINE=$deadline MIQ_BENCH_ONE_GPU=1 timeout -k 10 170 python -u bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-families --block-calib > gpurun_out/rehearse_$deadline.json 2> gpurun_out/rehearse_$deadline.err
  rc=$?; show gpurun_out/rehearse_$deadline.json; echo rc=$rc; [ $rc = 0 ] || exit 1
done
