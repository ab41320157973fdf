set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "" "MIQ_BWD_DEEP=1" "MIQ_BWD_CHUNKS_X=2" "MIQ_BWD_DEEP=1 MIQ_BWD_CHUNKS_X=2"; do
echo "== act_time $v"; env $v timeout -k 10 120 python -u tools/dev/act_time.py 2>&1 | grep -v Warn | tail -6 || exit 1
done
