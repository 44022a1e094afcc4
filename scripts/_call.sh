cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/c20.txt
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_render.py -x -q 2>&1 | tail -3 >> gpurun_out/c20.txt
for v in k5 k16 k5 k16; do
  echo "== clear in $v" >> gpurun_out/c20.txt
  if [ "$v" = k5 ]; then export BH_VC_CLEAR_IN_K5=1; else unset BH_VC_CLEAR_IN_K5; fi
  WORKLOAD=1m_1080p STEPS=30 timeout 120 python scripts/stage_times.py 2>/dev/null | cut -c1-260 >> gpurun_out/c20.txt
done
cat gpurun_out/c20.txt
