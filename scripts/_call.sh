cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_full_size.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -5 > gpurun_out/c5_tests.txt
STEPS=30 timeout 600 bash scripts/ab_k.sh old ac default old ac default > gpurun_out/c5_ab.txt 2>&1
cat gpurun_out/c5_tests.txt gpurun_out/c5_ab.txt
