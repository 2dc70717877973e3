cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02j_small -- python3 $GRAFT_REPO_ROOT/tools/small_batch_trace.py 1024 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r02j_small/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
