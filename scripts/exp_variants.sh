cd $GRAFT_REPO_ROOT
cp jaero_amd/libjaero_hip.so /tmp/orig.so
for v in "$@"; do
  cp jaero_amd/_variants/$v.so jaero_amd/libjaero_hip.so
  echo "== $v"
  timeout 120 python bench.py --no-cpu-baseline --steps 12 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_total'])"
done
cp /tmp/orig.so jaero_amd/libjaero_hip.so
