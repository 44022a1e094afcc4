cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload 6m_4k --sh-degree 3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s4_6m4k.json 2> gpurun_out/s4_6m4k.err; echo "6m4k rc $?"
timeout 200 python bench.py --feed loader --no-cpu-baseline > gpurun_out/s4_feed.json 2> gpurun_out/s4_feed.err; echo "feed rc $?"
timeout 200 python bench.py --sh-degree 3 --no-cpu-baseline > gpurun_out/s4_sh3.json 2> gpurun_out/s4_sh3.err; echo "sh3 rc $?"
