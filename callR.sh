cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$v"
HIP_FORCE_DEV_KERNARG=$v PINN_HIP_LIB=pinns-tf2.0_amd/pinn_native/libpinn_hip_stamps.so timeout 120 python profiles/stamps.py 2>&1 | grep "workgroup duration\|prologue\|stage weights\|fwd dense 1 "
HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-f64-leg --no-cfg5-leg 2>&1 | tail -1 > gpurun_out/bench_new.json; python - <<'PY'
import json; d=json.loads(open('gpurun_out/bench_new.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
PY
done
