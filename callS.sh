cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -v "^W2026\|^E2026" gpurun_out/pytest_gpu.log | tail -5
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json
python - <<'PY'
import json; d=json.loads(open('gpurun_out/bench_default.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for k in ('float64_leg','cfg5_leg'):
    l=d.get(k) or d['config'].get(k)
    print(k, json.dumps(l)[:600])
PY
