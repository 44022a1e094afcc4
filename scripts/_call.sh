cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/c14.txt
for v in default lv1 lv2 lv3 default; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  echo "== $v" >> gpurun_out/c14.txt
  BRUSH_HIP_LIB=$lib WORKLOAD=1m_1080p STEPS=20 timeout 120 python scripts/stage_times.py 2>/dev/null | sed 's/.*Rasterize /Rasterize /' >> gpurun_out/c14.txt
done
cat gpurun_out/c14.txt
