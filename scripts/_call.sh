cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_camera_models.py tests/test_gpu_tile_partition.py -x -q 2>&1 | tail -3 > gpurun_out/c6_tests.txt
for v in old default old default; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  for wl in 1m_1080p 1m_1080p_lowopac; do
    echo "== $v $wl" >> gpurun_out/c6_ab.txt
    BRUSH_HIP_LIB=$lib WORKLOAD=$wl STEPS=30 timeout 120 python scripts/stage_times.py 2>/dev/null >> gpurun_out/c6_ab.txt
  done
done
cat gpurun_out/c6_tests.txt gpurun_out/c6_ab.txt
