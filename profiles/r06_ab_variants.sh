#!/bin/bash
# same-box A/B of library variants on the headline leg (and cfg 5 with CFG5=1): bench.py's kernel time by HIP events
#   bash profiles/r06_ab_variants.sh <variant> [<variant> ...]     variant = a name under pinn_native/abl/ or "tree" (the in-tree library)
B="--gpus 1 --steps 20 --warmup 5 --no-cfg34-legs --no-f64-leg --no-cpu-baseline --no-script-leg --no-final-error"
[ -z "$CFG5" ] && B="$B --no-cfg5-leg"
for i in 1 2; do
for v in "$@"; do
  if [ $v = tree ]; then unset PINN_HIP_LIB; else export PINN_HIP_LIB=$PWD/pinns-tf2.0_amd/pinn_native/abl/libpinn_hip_$v.so; fi
  timeout 120 python bench.py $B 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);c=j.get('cfg5_leg')
print('%-14s step %.2f us  kernel %.2f us' % ('$v', 1e3*j['ms_per_step'], 1e3*j['roofline']['avg_launch_ms']) + ('   cfg5 step %.1f us kernel %.1f us' % (1e3*c['ms_per_step'], 1e3*c['roofline']['avg_launch_ms']) if c else ''))"
done; done
