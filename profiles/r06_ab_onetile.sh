#!/bin/bash
# same-box A/B: headline leg only, old vs new library, twice each
B="--gpus 1 --steps 20 --warmup 5 --no-cfg34-legs --no-f64-leg --no-cpu-baseline --no-script-leg --no-final-error --no-cfg5-leg"
for i in 1 2; do
for v in old new; do
  if [ $v = new ]; then unset PINN_HIP_LIB; else export PINN_HIP_LIB=$PWD/pinns-tf2.0_amd/pinn_native/abl/libpinn_hip_$v.so; fi
  python bench.py $B 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done; done
unset PINN_HIP_LIB
for v in old new; do
  echo "== stamps $v"; PINN_HIP_LIB=$PWD/pinns-tf2.0_amd/pinn_native/abl/libpinn_hip_stamps_$v.so python profiles/stamps.py f64 10000 2>&1 | tail -24
done
