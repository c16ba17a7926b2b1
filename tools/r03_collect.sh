#!/bin/bash
# Counter collection + bench line on the current build (tag = $1).
set -u
TAG=${1:-r03k}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python tools/collect_profiles.py $TAG headline c3 c5 > $O/${TAG}_collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/${TAG}_collect.log
cp $O/profiles_$TAG/latest_pmc.json profiles/latest_pmc.json; cp $O/profiles_$TAG/latest_other_configs.json profiles/latest_other_configs.json
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-300 $O/${TAG}_bench.json
