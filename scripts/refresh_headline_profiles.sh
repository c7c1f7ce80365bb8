set -x
# The headline workload's profiles after a change that leaves the kernels alone (host-side symbolic / ordering): kernel stats and the
# three counter passes (scripts/profile_bench.sh), the traffic summary installed on the box, then the default bench line measured
# against it and the step timeline.  scripts/final_profiles_1.sh / _2.sh remain the full set.
TAG=${1:-r04}
bash scripts/profile_bench.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1 || exit 1
python scripts/summarize_profile.py gpurun_out/profile_${TAG} profiles ${TAG} > /dev/null || exit 1
python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace_${TAG} -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scale-modes > gpurun_out/trace_${TAG}.log 2>&1 || exit 1
python scripts/step_timeline.py gpurun_out/prof_trace_${TAG} 3 > gpurun_out/${TAG}_step_timeline.txt
python scripts/bench_system.py > gpurun_out/${TAG}_system.json 2> gpurun_out/${TAG}_system.err
cut -c1-400 gpurun_out/bench_${TAG}_default.json
