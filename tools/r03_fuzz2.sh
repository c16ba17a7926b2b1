#!/bin/bash
# campaign on the library as shipped: small cases, then big ones; output straight into files (progress line every 30 s)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
python -c "import importlib,sys; sys.path.insert(0,'.'); print('build', importlib.import_module('squigly-trace_amd').build_id())" | tee $O/${1}_fuzz_small.log
timeout -k 10 $(( ${2:-500} + 120 )) python tests/fuzz_gpu.py ${2:-500} ${4:-44000000} >> $O/${1}_fuzz_small.log 2>&1; tail -1 $O/${1}_fuzz_small.log
timeout -k 10 $(( ${3:-500} + 220 )) python tests/fuzz_gpu.py ${3:-500} ${5:-45000000} big > $O/${1}_fuzz_big.log 2>&1; tail -1 $O/${1}_fuzz_big.log
