#!/bin/bash
# Smoke run of the measurement tools on a GPU box (each under its own timeout; stops at the first one that has to be killed).
# Usage: bash tests/run_tools_smoke.sh  -> gpurun_out/tools_smoke/<tool>.log, gpurun_out/tools_smoke/summary.txt
out=gpurun_out/tools_smoke
mkdir -p $out
: > $out/summary.txt
run() {
  name=$1; shift
  timeout -k 10 170 "$@" > $out/$name.log 2>&1
  rc=$?
  echo "$name rc=$rc" >> $out/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping" >> $out/summary.txt; cat $out/summary.txt; exit 1; fi
}
run ab_ragged python tools/ab_ragged.py
run pcie_timeline python tools/pcie_timeline.py --repeats 1
run bench_api_mixed python tools/bench_api_mixed.py 24 72
run probe_tail python tools/probe_tail.py
run probe_cheap_content python tools/probe_cheap_content.py
run single_image_trace python tools/single_image_trace.py
run cheap_step python tools/cheap_step.py xauto15 2
run probe_graph python tools/probe_graph.py
run run_agent_demo python tools/run_agent_demo.py
run agent_ranks_demo python tools/agent_ranks_demo.py $out/agent_ranks_demo.json 3 auto
run bench_table python tools/bench_table.py
run bench_lift python tools/bench_lift.py
run bench_cnn python tools/bench_cnn.py
run probe_scaling python tools/probe_scaling.py
run ab_side_levels python tools/ab_side_levels.py
run ab_tiles python tools/ab_tiles.py
run cnn_gap_probe python tools/cnn_gap_probe.py
run sweep_v4 python tools/sweep_v4.py
cat $out/summary.txt
