set -x
TAG=${1:-r04}
bash scripts/profile_bench.sh ${TAG}_cfg3 cfg3 > gpurun_out/prof_${TAG}_cfg3.log 2>&1
bash scripts/profile_bench.sh ${TAG}_cfg5 cfg5 > gpurun_out/prof_${TAG}_cfg5.log 2>&1
python bench.py --configs 1,2,3,4,4b,5 --steps 10 --cfg-cpu-units 2 > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err
cut -c1-300 gpurun_out/${TAG}_configs.jsonl
