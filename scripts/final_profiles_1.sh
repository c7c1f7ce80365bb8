set -x
bash scripts/profile_bench.sh r03 > gpurun_out/prof_r03.log 2>&1
python bench.py > gpurun_out/bench_r03_default.json 2> gpurun_out/bench_r03_default.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace_r03 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/trace_r03.log 2>&1
python scripts/step_timeline.py gpurun_out/prof_trace_r03 3 > gpurun_out/r03_step_timeline.txt
python scripts/bench_system.py > gpurun_out/r03_system.json 2> gpurun_out/r03_system.err
python scripts/bench_multirhs.py --nrhs 1,8,16,64,512 --reps 7 > gpurun_out/r03_multirhs.json 2> gpurun_out/r03_multirhs.err
python bench.py --mode rhs --no-cpu-baseline > gpurun_out/r03_bench_rhs.json 2> gpurun_out/r03_bench_rhs.err
python bench.py --mode problems --no-cpu-baseline > gpurun_out/r03_bench_problems.json 2> gpurun_out/r03_bench_problems.err
tail -c 400 gpurun_out/r03_bench_rhs.json; tail -c 300 gpurun_out/r03_bench_problems.json
