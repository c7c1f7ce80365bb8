#!/bin/bash
# usage: sweep_env.sh "VAR=a VAR2=b" "VAR=c" ...   -- one bench.py run per argument
for e in "$@"; do
  env $e python bench.py > gpurun_out/b.json 2>gpurun_out/b.err && python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$e', round(d['value'],1), round(d['ms_per_step'],3), round(d['phases']['factor']['avg_ms'],3), round(d['phases']['trisolve']['avg_ms'],3), d['config']['levels'], d['config']['nsuper'], round(d['config']['setup_s'],2))" || tail -3 gpurun_out/b.err
done
