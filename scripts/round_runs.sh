#!/bin/bash
# Runs on the GPU box (via gpurun): the round's measured artefacts.   scripts/round_runs.sh <tag>   -> gpurun_out/<tag>_*
# Every step under its own timeout: a hung step must not eat the box.
TAG=${1:-r2}
cd $GRAFT_REPO_ROOT; O=gpurun_out
# the driver's own command
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench.err; echo "bench rc $?"
timeout 200 python bench.py --steps 20 --warmup 5 --lists exact --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_exact_lists_n1.json 2>> $O/${TAG}_bench.err; echo "exact lists rc $?"
timeout 200 python bench.py --steps 20 --warmup 5 --sh-degree 3 --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_sh3_n1.json 2>> $O/${TAG}_bench.err; echo "sh3 rc $?"
timeout 300 python bench.py --workload 6m_4k --sh-degree 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_6m_4k_sh3_n1.json 2>> $O/${TAG}_bench.err; echo "6m4k rc $?"
timeout 200 python bench.py --steps 20 --warmup 5 --feed loader --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_feed_loader_n1.json 2>> $O/${TAG}_bench.err; echo "feed rc $?"
timeout 200 python bench.py --steps 20 --warmup 5 --no-noise --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_no_noise_n1.json 2>> $O/${TAG}_bench.err; echo "no-noise rc $?"
timeout 200 python bench.py --steps 20 --warmup 5 --no-view-ids --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_no_view_ids_n1.json 2>> $O/${TAG}_bench.err; echo "no-view-ids rc $?"
# exchange-path overhead on one GPU: a 1-rank RCCL group (collectives degenerate), every comm / exchange combination
for comm in native torch; do for ex in sparse dense; do
  BH_FORCE_PG=1 MASTER_PORT=$((29600 + RANDOM % 200)) timeout 200 python bench.py --steps 20 --warmup 5 --comm $comm --exchange $ex --no-cpu-baseline --no-extra --no-pmc > $O/${TAG}_bench_pg1_${comm}_${ex}.json 2>> $O/${TAG}_bench.err; echo "pg1 $comm $ex rc $?"
done; done
timeout 600 bash scripts/profile_bench.sh $TAG > $O/${TAG}_profile.log 2>&1; echo "profile rc $?"
# the converging run's late phase under rocprofv3 (kernel trace only): per-kernel table of its last 400 steps, both list modes
for mode in exact_lists cuts_view_ids; do
  (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_${TAG}_loop_$mode -o loop -- python $GRAFT_REPO_ROOT/bench.py --loop-only $mode --loop-steps 3000 > $GRAFT_REPO_ROOT/$O/${TAG}_loop_only_$mode.json 2>> $GRAFT_REPO_ROOT/$O/${TAG}_bench.err)
  f=$(find $O/prof_${TAG}_loop_$mode -name "*kernel_trace.csv" | head -1)
  python scripts/late_phase_stats.py "$f" $O/${TAG}_late_phase_kstats_$mode.csv 400; echo "late phase $mode rc $?"
  rm -rf $O/prof_${TAG}_loop_$mode
done
# the GPU suite's log of the same tree
timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -1 $O/${TAG}_gpu_tests.log
