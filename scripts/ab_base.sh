cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "primitives or render or train_step or loss or tile_partition or refine or dp" 2>&1 | tail -5
for i in 1 2 3; do
  a=$(BRUSH_HIP_LIB=brush_amd/variants/libbrush_hip_base.so python bench.py --no-cpu-baseline --steps 100 | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  b=$(python bench.py --no-cpu-baseline --steps 100 | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  echo "base $a   new $b"
done
