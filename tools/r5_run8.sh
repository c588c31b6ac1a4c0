set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/probe_tail_variants.py gen /tmp/tailv > gpurun_out/r5_tailv_gen.log 2>&1 || { tail -20 gpurun_out/r5_tailv_gen.log; exit 1; }
for v in $VARIANTS; do
  LLICTI_HIP_SO=$PWD/build/tailv/lib_$v.so timeout -k 10 120 python tools/probe_tail_variants.py run /tmp/tailv $v 2>gpurun_out/r5_tailv_$v.err | tee -a gpurun_out/r5_tailv.jsonl
done
