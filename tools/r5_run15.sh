set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in old new old new; do
  echo "== $V"
  LLICTI_HIP_SO=$PWD/build/abv/lib_$V.so timeout -k 10 200 python tools/bench_api_mixed.py 24 500 2>/dev/null | head -2
done
