#!/bin/bash
# Overlap mode on/off over problems whose top fronts grow: factorisation time per problem (scripts/try_problem.py)
for pb in "config2(n=20000, long_range_frac=0.01)" "config2(n=40000, long_range_frac=0.01)" "config2(n=70000, long_range_frac=0.01)" \
          "config2(long_range_frac=0.001)" "config2(long_range_frac=0.003)" \
          "config3(nblocks=25, blk=2000)" "config3(nblocks=12, blk=4000)" "config5(n=5000, npsd=50, psd_dim=40)" "config5(n=5000, npsd=20, psd_dim=64)"; do
  for e in "HIPKKT_FACTOR_OVERLAP=1" "HIPKKT_FACTOR_OVERLAP=0"; do
    echo "== $pb $e"
    env $e HIPKKT_VERBOSE=${VERB:-0} timeout -k 10 200 python scripts/try_problem.py "$pb" --no-oracle 2>&1 | grep -E "setup|factor|largest launch|tiles" || exit 1
  done
done
