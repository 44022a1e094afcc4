cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/c19.txt
timeout 600 python -m pytest tests/test_gpu_depth_sort.py tests/test_gpu_render.py -x -q 2>&1 | tail -3 >> gpurun_out/c19.txt
for v in presort default presort default; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  echo "== $v" >> gpurun_out/c19.txt
  BRUSH_HIP_LIB=$lib WORKLOAD=1m_1080p STEPS=30 timeout 120 python scripts/stage_times.py 2>/dev/null | cut -c1-150 >> gpurun_out/c19.txt
done
cat gpurun_out/c19.txt
