set -x
TAG=${1:-r04}
bash scripts/profile_bench.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1
python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace_${TAG} -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scale-modes > gpurun_out/trace_${TAG}.log 2>&1
python scripts/step_timeline.py gpurun_out/prof_trace_${TAG} 3 > gpurun_out/${TAG}_step_timeline.txt
python scripts/bench_system.py > gpurun_out/${TAG}_system.json 2> gpurun_out/${TAG}_system.err
python scripts/bench_multirhs.py --nrhs 1,8,16,64,512 --reps 7 > gpurun_out/${TAG}_multirhs.json 2> gpurun_out/${TAG}_multirhs.err
python bench.py --mode rhs --no-cpu-baseline > gpurun_out/${TAG}_bench_rhs.json 2> gpurun_out/${TAG}_bench_rhs.err
python bench.py --mode problems --no-cpu-baseline > gpurun_out/${TAG}_bench_problems.json 2> gpurun_out/${TAG}_bench_problems.err
bash scripts/hop_tables.sh ${TAG}
tail -c 400 gpurun_out/${TAG}_bench_rhs.json; tail -c 300 gpurun_out/${TAG}_bench_problems.json
