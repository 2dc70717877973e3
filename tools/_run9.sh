cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "parity or fuzz" 2>&1 | tail -5
bash tools/walk_phases.sh 2>&1 | grep -v amdgpu.ids
for W in cit2 ppa; do echo -n "$W: "; python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walk', round(d['config']['stage_ms']['walk_sets'],4), 'step', round(d['ms_per_step'],3), round(d['value']/1e6,2))"; done
