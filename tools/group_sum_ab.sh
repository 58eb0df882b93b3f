#!/bin/bash
# GPU box: same-box A/B of the bucket-reduction variants the library carries as knobs (review r5: "run the k_group_sum variants as a real same-box
# A/B", kill criterion < 2 % of the one-call 2^20 MSM): the number of index fields (GM_MSM_FIELDS=2: 7 + 8 bits, two levels, instead of 5 + 5 + 5,
# three), where the odd bit goes (GM_MSM_ODD_BIT), the threads per launch (GM_MSM_LPO_LOG).  Interleaved runs, the headline loop only.
O=${1:-gpurun_out/r6_group_sum_ab.txt}
echo "# $(date -u +%F) library $(sha256sum gemini_amd/libgemini_hip.so | cut -c1-12): bench.py --headline-only --steps 60 --warmup 10; Mscalar/s, ms per step, stage_ms reduce (k_group_sum x 3), merge" > $O
run() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py --headline-only --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['stage_ms']
print('%-34s' % '$label', d['value'], d['ms_per_step'], 'reduce', s.get('reduce'), 'merge', s.get('merge'), 'acc0', s.get('acc0'), 'clock', d['roofline'].get('shader_clock_mhz_in_kernel'))" >> $O
}
for rep in 1 2 3; do
  run "default(3 fields 5+5+5, lpo 17)" GM_NOOP=1
  run "GM_MSM_FIELDS=2 (7+8, two levels)" GM_MSM_FIELDS=2
  run "GM_MSM_LPO_LOG=16" GM_MSM_LPO_LOG=16
  run "GM_MSM_LPO_LOG=18" GM_MSM_LPO_LOG=18
  run "GM_MSM_ODD_BIT=high" GM_MSM_ODD_BIT=high
done
cat $O
