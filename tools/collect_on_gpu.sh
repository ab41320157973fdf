#!/bin/bash
# Collects one measurement directory on the GPU box (run from the repository root):
#   bash tools/collect_on_gpu.sh gpurun_out/r01_h
# bench line, kernel trace of the bench, the two PMC passes (own runs, no tracing), operator families, the larger
# block shapes, and a kernel trace of the Llama-2-7B fp16 calibration step.  Every GPU step has its own time limit
# and the script stops at the first failure.  tools/collect_profiles.py installs the directory into profiles/.
set -e -o pipefail
D=${1:?output directory}
R=$(pwd)
mkdir -p "$D"
D=$(cd "$D" && pwd)
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, 'llm-quantization_amd'); from miq import _lib; print(_lib.source_digest())" > "$D/csrc.digest"
echo "[collect] bench"; timeout -k 10 600 python3 -u bench.py > "$D/bench.json.log" 2> "$D/bench.err"
cd /tmp
echo "[collect] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-block-calib --no-families > "$D/trace.log" 2>&1
echo "[collect] pmc fetch"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/fetch" -- python3 "$R/bench.py" --eager --steps 2 --warmup 1 --no-cpu-baseline --no-block-calib --no-families > "$D/fetch.log" 2>&1
echo "[collect] pmc write"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/write" -- python3 "$R/bench.py" --eager --steps 2 --warmup 1 --no-cpu-baseline --no-block-calib --no-families > "$D/write.log" 2>&1
cd "$R"
echo "[collect] families"; timeout -k 10 600 python3 -u tools/bench_families.py > "$D/families.json" 2> "$D/families.err"
echo "[collect] blocks"
timeout -k 10 300 python3 -u tools/bench_block.py --model 13b --amp 2>/dev/null | grep '^{' > "$D/blocks_extra.jsonl"
timeout -k 10 300 python3 -u tools/bench_block.py --model 70b --abits 8 2>/dev/null | grep '^{' >> "$D/blocks_extra.jsonl"
timeout -k 10 300 python3 -u tools/bench_block.py --model 70b --abits 8 --amp 2>/dev/null | grep '^{' >> "$D/blocks_extra.jsonl"
echo "[collect] fused-operator A/B"; timeout -k 10 300 python3 -u tools/ab_block.py --model 7b --amp --reps 5 2>/dev/null | grep '^{' > "$D/ab_block.json"
cd /tmp
echo "[collect] block step trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/block7b" -- python3 "$R/tools/bench_block.py" --model 7b --amp > "$D/block7b.log" 2>&1
echo "[collect] SQ counters (row quantizers, bench kernels)"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d "$D/sq_rows" -- python3 "$R/tools/row_pmc.py" > "$D/sq_rows.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d "$D/sq_bench" -- python3 "$R/bench.py" --eager --steps 2 --warmup 1 --no-cpu-baseline --no-block-calib --no-families > "$D/sq_bench.log" 2>&1
cd "$R"
python3 tools/row_pmc_summary.py "$D"/sq_rows/*/*counter_collection.csv "$D/sq_rows.json" > /dev/null
python3 tools/row_pmc_summary.py "$D"/sq_bench/*/*counter_collection.csv "$D/sq_bench.json" > /dev/null
echo "[collect] row-sharded workload, one GPU"; timeout -k 10 300 python3 -u bench.py --workload c4 --steps 10 --warmup 3 > "$D/bench_c4.json.log" 2> "$D/bench_c4.err"
echo "[collect] norm + quantizer, separate against fused"; timeout -k 10 120 python3 -u tools/dev/norm_quant_time.py > "$D/norm_quant_time.txt" 2>&1
echo "[collect] LET kernels, attention-score pass"; timeout -k 10 200 python3 -u tools/bench_let.py > "$D/bench_let.txt" 2>&1
timeout -k 10 120 python3 -u tools/dev/softmax_time.py > "$D/softmax_time.txt" 2>&1
echo "[collect] micro-benchmarks"
for b in actbench readbw statbench; do
    if [ -x tools/ubench/$b ]; then timeout -k 10 120 tools/ubench/$b > "$D/ubench_$b.txt" 2>&1; fi
done
echo "[collect] done"
