#!/bin/bash
# Everything profiles/ is refreshed from, in two halves (a GPU call is limited to 20 minutes).
#   scripts/profile_all.sh <tag> a|b        then   python scripts/install_profiles.py <tag> <prefix> && python scripts/c5_install.py <tag> <prefix>
set -u
tag=$1
half=$2
if [ "$half" = a ]; then
  scripts/profile_round.sh $tag && scripts/bench_configs.sh $tag
else
  scripts/pmc.sh $tag && scripts/pmc.sh ${tag}_meas --scene nee15_measured --iterations 40 --steps 1 --warmup 0 &&
  scripts/c5_all.sh $tag
fi
