#!/bin/bash
# GPU box: same-box A/B of the twist-one specialisation of the sumcheck kernel (GM_SC_TW1=0: the general kernel for every prover)
O=${1:-gpurun_out/r6_sumcheck_tw1_ab.txt}
echo "# $(date -u +%F) library $(sha256sum gemini_amd/libgemini_hip.so | cut -c1-12): sumcheck_roofline of bench.py's snark -i 24 (two 2^24 sumchecks: the first with twist alpha, the second with twist ONE), psnark -i 22 third-sumcheck span" > $O
for v in 0 1 0 1; do
  GM_SC_TW1=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tables --psnark-logn 0 --strong-msm-logn 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['time_prover']['sumcheck_roofline']
print('GM_SC_TW1=$v', 'snark24', d['time_prover']['value'], 's', 'sumcheck kernel_ms', s.get('kernel_ms'), 'frac', s.get('frac'), {k: s[k] for k in s if 'round' in k or 'launch' in k})" >> $O
  GM_SC_TW1=$v python tools/run_psnark.py -i 22 --repeat 4 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('GM_SC_TW1=$v', 'psnark22', d['time_prover_s'], 'second', sorted(r['Second sumcheck'] for r in d['runs']), 'third', sorted(r['Third sumcheck'] for r in d['runs']))" >> $O
done
cat $O
