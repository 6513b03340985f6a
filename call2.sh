cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "f64 or fuzz or ragged" 2>&1 | tail -25 > gpurun_out/pytest_call2.log
tail -8 gpurun_out/pytest_call2.log
timeout 300 python profiles/time_f64.py > gpurun_out/time_f64.txt 2>&1; cat gpurun_out/time_f64.txt
PINN_HIP_LIB=pinns-tf2.0_amd/pinn_native/libpinn_hip_stamps.so timeout 120 python profiles/stamps.py f64 10000 > gpurun_out/stamps_f64_20d.txt 2>&1; cat gpurun_out/stamps_f64_20d.txt
