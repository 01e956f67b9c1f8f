#!/bin/bash
# stage ablation of the fused kernel (variant "abl" = -DKU_ABLATION): KU_ABLATE bits 1 probe, 2 HLL, 4 n_kmers, 8 taxa store, 32 resolve, 64 anchor/locus
export KU_LIB=$PWD/krakenuniq_amd/variants/libku_abl.so
for a in 0 1 2 4 8 32 64 3 15 47 111; do
  KU_ABLATE=$a python bench.py --cpu-sample 0 --no-extras --steps 6 $* 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('ablate $a kernel_ms', j['roofline']['kernel_ms'])"
done
