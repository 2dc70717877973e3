cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walk', round(d['config']['stage_ms']['walk_sets'],4), 'step', round(d['ms_per_step'],3), round(d['value']/1e6,2), d['config']['distinct_lp_rows_last_step'])"
python - <<'PY'
import os, sys, time
os.environ["SUBGACC_QUIET"]="1"; sys.path.insert(0, os.getcwd())
import torch, surel_plus_amd as sp, bench
from surel_plus_amd.graphs import preset_graph
csr = preset_graph("cit2")
print(bench.batch_size_and_graph(sp, csr, 200, 4, "philox", 10))
PY
