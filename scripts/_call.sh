cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ply.py tests/test_gpu_cpp_host.py -x -q 2>&1 | tail -15 > gpurun_out/c2_ply.txt
cat gpurun_out/c2_ply.txt
