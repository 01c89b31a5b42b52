cd $GRAFT_REPO_ROOT
for v in nowin flush128; do
  echo "== $v"; JAERO_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_tmp/libjaero_hip_$v.so timeout 600 python -m pytest tests/test_recording.py -m gpu -q --tb=line 2>&1 | tail -3
done
