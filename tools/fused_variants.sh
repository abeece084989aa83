for v in "" "GOSS_GPU_LIB=$PWD/gossamer_amd/libgossgpu_g1.so" "GOSS_GPU_NO_FUSED=1"; do
  echo "== $v"
  env $v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['device_ms_per_step'])"
done
