#!/bin/bash
# Same-box A/B of the B = 64 sampler step: alternating runs of bench.py under two environments.
#   [PREC=f16f8] tools/ab_bench.sh "<env A>" "<env B>" [pairs] [extra bench args]     (e.g. "SR3_NO_F8C=1" "")
# PREC = arithmetic mode measured (default f16f8, the fastest; bench.py itself defaults to the reference's f32).
# Prints ms_per_step and the per-family split of every run; run on the GPU box (gpurun).
A="$1"; B="$2"; N="${3:-3}"; shift 3 || true
for i in $(seq 1 "$N"); do
  for tag in A B; do
    if [ "$tag" = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --no-alt --no-cpu-baseline --no-full-loop --steps 30 --warmup 5 --precision "${PREC:-f16f8}" "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['family_ms_per_step']; print('$tag [%s]' % '$E', 'ms_per_step %.3f' % d['ms_per_step'], 'conv %.3f gn %.3f attn %.3f other %.3f' % (f['conv_igemm'], f['groupnorm'], f['attention'], f['embed']+f['update_layout']), 'conv TF %.1f' % d['roofline']['achieved'])"
  done
done
