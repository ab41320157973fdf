set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== act_time default"; timeout -k 10 120 python -u tools/dev/act_time.py 2>&1 | grep -v Warn | tail -8 || exit 1
for rows in 32 64; do for depth in 4 8; do
echo "== act_time MIQ_STAT_ROWS=$rows MIQ_STAT_DEPTH=$depth"; MIQ_STAT_ROWS=$rows MIQ_STAT_DEPTH=$depth timeout -k 10 120 python -u tools/dev/act_time.py 2>&1 | grep -v Warn | tail -8 || exit 1
done; done
echo "== act_time MIQ_STAT_ROWS=16 MIQ_STAT_DEPTH=4"; MIQ_STAT_ROWS=16 MIQ_STAT_DEPTH=4 timeout -k 10 120 python -u tools/dev/act_time.py 2>&1 | grep -v Warn | tail -8 || exit 1
R=$(pwd); cd /tmp
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_b_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-block-calib --no-families > $R/gpurun_out/r03_b_trace.log 2>&1 || { tail -5 $R/gpurun_out/r03_b_trace.log; exit 1; }
cd $R; python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03_b_trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
