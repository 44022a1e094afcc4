cd /root/repo
mkdir -p gpurun_out
STEPS=30 timeout 600 bash scripts/ab_k.sh default bw6 al64 al256 default bw6 al64 al256 > gpurun_out/c10.txt 2>&1
cat gpurun_out/c10.txt
