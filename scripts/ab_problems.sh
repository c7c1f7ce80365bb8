#!/bin/bash
# A/B of environment settings over arbitrary generated problems: scripts/ab_problems.sh "<ENV=..>|-" "<ENV=..>|-" -- "<problems.maker(...)>" ...
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
for pb in "$@"; do
  for e in "${envs[@]}"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    echo "== $pb [$e]"
    env $ee timeout -k 10 300 python scripts/try_problem.py "$pb" --no-oracle 2>&1 | grep -E "^setup|^factor|^unit|rror" || exit 1
  done
done
