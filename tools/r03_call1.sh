#!/bin/bash
# Round 3, GPU call 1: the GPU test suite (with the new full-size C3/C5 tests), the bench line, the op-rate microbenchmark,
# streaming-form A/B variants, the co-residency diagnostic and this round's first counter collection.  Steps are chained
# with && only inside a group that must not continue after a GPU fault; output goes to gpurun_out/r03a_*.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== pytest -m gpu"; ( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r03a_pytest.log 2>&1; rc=$?; tail -3 $O/r03a_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== bench"; timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/r03a_bench.json 2> $O/r03a_bench.err; rc=$?; echo "rc=$rc"; cut -c1-400 $O/r03a_bench.json; [ $rc -ne 0 ] && exit $rc
echo "== op_rate"; timeout -k 10 120 tools/ubench/op_rate > $O/r03a_op_rate.txt 2>&1; echo "rc=$?"; tail -5 $O/r03a_op_rate.txt
echo "== streaming variants"
for scene in blob6 hf708; do for i in 1 2; do
  spp=64
  echo "-- $scene product"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=$spp | tail -1
  echo "-- $scene product lds12"; timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=$spp lds_node_kb=12 | tail -1
  echo "-- $scene cull16"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libcull16.so timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=$spp | tail -1
  echo "-- $scene w6 (3 WG/CU, 12 KB nodes)"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libw6.so timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=$spp trace_blocks_per_cu=3 lds_node_kb=12 | tail -1
  echo "-- $scene w6 (2 WG/CU)"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libw6.so timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=$spp | tail -1
  echo "-- $scene w6c16 (3 WG/CU, 12 KB nodes)"; SQ_LIB_PATH=$PWD/squigly-trace_amd/libw6c16.so timeout -k 10 300 python tools/gpu_frames.py scene=$scene frames=3 spp=$spp trace_blocks_per_cu=3 lds_node_kb=12 | tail -1
done; done > $O/r03a_stream_ab.txt 2>&1
cat $O/r03a_stream_ab.txt
echo "== overlap / co-residency"; timeout -k 10 300 python tools/gpu_overlap.py quick > $O/r03a_overlap.txt 2>&1; echo "rc=$?"; cat $O/r03a_overlap.txt
echo "== counters"; timeout -k 10 1100 python tools/collect_profiles.py r03a headline c3 c5 overlap > $O/r03a_collect.log 2>&1; echo "rc=$?"; tail -5 $O/r03a_collect.log
