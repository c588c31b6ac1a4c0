set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for V in $VARIANTS; do
  LLICTI_HIP_SO=$PWD/build/abv/lib_$V.so timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-pcie-legs --no-ac-leg 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
done
