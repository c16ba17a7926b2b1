#!/bin/bash
# Closing sequence on the shipped build (second half of round 3): GPU tests, smoke, fuzz slices with the new knobs, counter collection
# (stamps profiles/latest_*.json with the build id), bench line.  usage: bash tools/r03_final2.sh <tag> [fuzz seconds]
set -u
TAG=${1:-r03u}; FZ=${2:-120}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/${TAG}_pytest.log 2>&1; rc=$?; grep -E "passed|failed|error" $O/${TAG}_pytest.log | tail -2; [ $rc -ne 0 ] && { tail -30 $O/${TAG}_pytest.log; exit $rc; }
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1
timeout -k 10 $((FZ + 120)) python tests/fuzz_gpu.py $FZ 40000000 2>&1 | grep -v amdgpu | tail -2 | tee $O/${TAG}_fuzz.log
timeout -k 10 $((FZ + 180)) python tests/fuzz_gpu.py $FZ 41000000 big 2>&1 | grep -v amdgpu | tail -2 | tee -a $O/${TAG}_fuzz.log
timeout -k 10 1000 python tools/collect_profiles.py $TAG headline c3 c5 > $O/${TAG}_collect.log 2>&1; echo "collect rc=$?"
cp $O/profiles_$TAG/latest_pmc.json profiles/latest_pmc.json; cp $O/profiles_$TAG/latest_other_configs.json profiles/latest_other_configs.json
SECONDS=0; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$? in ${SECONDS} s"; cut -c1-200 $O/${TAG}_bench.json
