#!/bin/bash
# PMC counters of the bench command, one rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a
# pass on gfx950; --pmc is never combined with other trace domains than --kernel-trace).  Run on the GPU box:
#   tools/collect_pmc.sh gpurun_out/pmc_r1 ;  python tools/pmc_traffic.py gpurun_out/pmc_r1 profiles/r1
set -e
#   tools/collect_pmc.sh gpurun_out/pmc_tab table   -> the same passes over the full-table CDF kernel alone (configs[3])
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-ac-leg --no-pcie-legs"
if [ "$2" = "table" ]; then CMD="python3 tools/bench_table.py"; fi
mkdir -p "$OUT"
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  D=$OUT/$(echo $G | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o run -- $CMD > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
  echo "pass [$G] done"
done
