#!/bin/bash
# timing-only variants of the band CNN that bound what an all-heads-per-workgroup, weight-stationary form could buy
# (band_cnn.hpp: CNN_EXP_STAGE_EVERY, CNN_EXP_NO_WFRAG; results of the variants are WRONG by construction).
#   here (CPU):  tools/cnn_ws_experiment.sh build      -> build/cnn_ws/lib_<variant>.so
#   GPU box:     tools/cnn_ws_experiment.sh run > gpurun_out/cnn_ws.log
set -e
DIR=build/cnn_ws
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value"
if [ "$1" = "build" ]; then
  mkdir -p $DIR
  /opt/rocm/bin/hipcc $FLAGS -o $DIR/lib_base.so llicti_amd/csrc/llicti_hip.hip &
  /opt/rocm/bin/hipcc $FLAGS -DCNN_EXP_STAGE_EVERY=4 -o $DIR/lib_stage4.so llicti_amd/csrc/llicti_hip.hip &
  /opt/rocm/bin/hipcc $FLAGS -DCNN_EXP_NO_WFRAG=1 -o $DIR/lib_nowfrag.so llicti_amd/csrc/llicti_hip.hip &
  wait
  /opt/rocm/bin/hipcc $FLAGS -DCNN_EXP_STAGE_EVERY=4 -DCNN_EXP_NO_WFRAG=1 -o $DIR/lib_stage4_nowfrag.so llicti_amd/csrc/llicti_hip.hip &
  /opt/rocm/bin/hipcc $FLAGS -DCNN_EXP_STAGE_EVERY=1000000 -DCNN_EXP_NO_WFRAG=1 -o $DIR/lib_nostage_nowfrag.so llicti_amd/csrc/llicti_hip.hip &
  wait
  ls -la $DIR
else
  for rep in 1 2; do
    for V in base stage4 nowfrag stage4_nowfrag nostage_nowfrag; do
      echo "== $V (run $rep)"
      LLICTI_HIP_SO=$PWD/$DIR/lib_$V.so timeout -k 10 120 python tools/bench_cnn.py 2>/dev/null | grep -E "lvl 0|total"
    done
  done
fi
