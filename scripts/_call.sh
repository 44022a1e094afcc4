cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/c11.txt
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_train_step.py tests/test_gpu_stress.py -x -q 2>&1 | tail -3 >> gpurun_out/c11.txt
for v in presort default presort default; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  echo "== $v" >> gpurun_out/c11.txt
  BRUSH_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['fwd_ms'], d['fwd_bwd_ms'])" >> gpurun_out/c11.txt
done
cat gpurun_out/c11.txt
