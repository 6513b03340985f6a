cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -3
PINN_HIP_LIB=pinns-tf2.0_amd/pinn_native/libpinn_hip_stamps.so timeout 120 python profiles/stamps.py f64 2>&1 | grep -v "^W2026\|^E2026" | head -8
timeout 300 python bench.py --no-cfg5-leg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_new.json; python - <<'PY'
import json; d=json.loads(open('gpurun_out/bench_new.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
l=d.get('float64_leg') or d['config'].get('float64_leg'); print(l['value'], l['ms_per_step'], l['roofline']['avg_launch_ms'])
PY
