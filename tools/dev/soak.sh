#!/bin/bash
# randomized shape / dtype / option cases of tests/test_gpu_random_shapes.py under other seeds (run on a GPU box)
set -o pipefail
for off in "$@"; do
  echo "== seed offset $off"
  MIQ_TEST_SEED_OFFSET=$off timeout -k 10 300 python -u -m pytest tests/test_gpu_random_shapes.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 | cut -c1-300 || exit 1
done
