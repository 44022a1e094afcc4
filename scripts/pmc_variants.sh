#!/bin/bash
# Developer tool (runs on the GPU box): SQ counter passes over a few train steps for several library variants
# (brush_amd/variants/libbrush_hip_<v>.so, "default" = the in-tree build); per-kernel averages for kernels matching $PAT.
#   PAT=rasterize_backward scripts/pmc_variants.sh default oldk17
PAT=${PAT:-rasterize_backward}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then lib=""; else lib="$GRAFT_REPO_ROOT/brush_amd/variants/libbrush_hip_$v.so"; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_var_$v; rm -rf $OUT; mkdir -p $OUT
  BRUSH_HIP_LIB=$lib WORKLOAD=${WORKLOAD:-1m_1080p} STEPS=3 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o sq -- python $GRAFT_REPO_ROOT/scripts/stage_times.py > $OUT/run.log 2>&1
  BRUSH_HIP_LIB=$lib WORKLOAD=${WORKLOAD:-1m_1080p} STEPS=3 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq2 -- python $GRAFT_REPO_ROOT/scripts/stage_times.py >> $OUT/run.log 2>&1
  echo "=== $v"
  python - "$PAT" $OUT <<'PY'
import csv, glob, re, sys, collections
pat, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if re.search(pat, k):
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s avg %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  find $OUT -name "*.csv" -size +5M -delete
done
