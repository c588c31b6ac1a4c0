#!/bin/bash
# A/B several builds of the HIP library in one GPU session: tools/ab_variants.sh DIR name1 name2 ...
DIR=$1; shift
for V in "$@"; do
  LLICTI_HIP_SO=$PWD/$DIR/lib_$V.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['value'], d['ms_per_step'], 'enc', d['enc_mpix_s'], 'dec', d['dec_mpix_s'], 'cnn TF/s', d['roofline']['achieved'], 'cnn ms', d['roofline']['kernel_ms_per_step'])"
done
