# Aero-L P channel with and without the detector-window jump (make the partner: hipcc ... -DAEROL_WINDOW_JUMP=0 -o gpurun_tmp/libjaero_hip_nowin.so), equal step counts
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in product nowin; do
  L=$GRAFT_REPO_ROOT/jaero_amd/libjaero_hip.so; [ $v = nowin ] && L=$GRAFT_REPO_ROOT/gpurun_tmp/libjaero_hip_nowin.so
  JAERO_HIP_LIB=$L python bench.py --workload aerol --steps 24 --warmup 12 --no-cpu-baseline --as-written 0 --no-state 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('$v',d['value'],d['ms_per_step'],d.get('step_ms'),c.get('kernel_ms_per_step'))"
done; done
