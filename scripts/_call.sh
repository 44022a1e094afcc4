cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_camera_models.py -x -q 2>&1 | tail -5 > gpurun_out/c12_tests.txt
STEPS=30 timeout 600 bash scripts/ab_k.sh presort default presort default > gpurun_out/c12_ab.txt 2>&1
cat gpurun_out/c12_tests.txt gpurun_out/c12_ab.txt
