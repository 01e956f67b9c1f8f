#!/bin/bash
# A/B of library variants (scripts/build_variant.sh): scripts/ab_bench.sh "<bench args>" name1 name2 ...   ("base" = the product library)
ARGS=$1; shift
for n in "$@"; do
  if [ "$n" = base ]; then unset KU_LIB; else export KU_LIB=$PWD/krakenuniq_amd/variants/libku_$n.so; fi
  python bench.py --cpu-sample 0 --no-extras $ARGS 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$n', 'value', j['value'], 'ms/step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
done
