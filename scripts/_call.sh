cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/c17.txt
timeout 900 python -m pytest tests/test_gpu_loss_optim.py tests/test_gpu_tile_partition.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -3 >> gpurun_out/c17.txt
for v in tw16 default tw16 default; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  echo "== $v" >> gpurun_out/c17.txt
  BRUSH_HIP_LIB=$lib WORKLOAD=1m_1080p STEPS=30 timeout 120 python scripts/stage_times.py 2>/dev/null | sed 's/.*Rasterize /Rasterize /' >> gpurun_out/c17.txt
done
cat gpurun_out/c17.txt
