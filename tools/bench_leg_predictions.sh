#!/bin/bash
# GPU box: what the strong-scaling legs of `bench.py --gpus N` should show on a node, predicted from one GPU: one rank's share of `psnark -i 24`
# (world 1 at -i 23 / 22 / 21 through gm_psnark_new_time_sharded) and of ONE MSM of 2^26 pairs (a plain one-call MSM of 2^25 / 2^24 / 2^23 pairs).
O=${1:-gpurun_out/r6_bench_leg_predictions.txt}
echo "# $(date -u +%F) library $(sha256sum gemini_amd/libgemini_hip.so | cut -c1-12): predictions for bench.py --gpus 2 / 4 / 8 (before collectives)" > $O
p() { python tools/run_psnark.py "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['time_prover_s'])"; }
b=$(p -i 24 --repeat 3)
echo "psnark -i 24, one GPU: $b s" >> $O
for g in 2 4 8; do
  lg=$(python -c "import math; print(24 - int(math.log2($g)))")
  s=$(p -i $lg --repeat 4 --block-sharded --transport shm)
  echo "  g=$g: one rank's share (world 1 at -i $lg) $s s -> $(python -c "print(round($b / $s, 2))") x" >> $O
done
python - >> $O <<'PY'
import time, numpy as np, ctypes as C
import gemini_amd as gm
gm.capi.init(0)
lib = gm.capi.load()
gm.capi.check(lib.gm_set_auto_tables(C.c_int(0), C.c_size_t(0)))
import torch
rng = np.random.default_rng(5)
gx = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
gy = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
mont = lambda v: [(((v << 384) % q) >> (64 * i)) & (2**64 - 1) for i in range(6)]
g_aff = np.array(mont(gx) + mont(gy), dtype=np.uint64)
def rnd(n):
    v = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); v[:, 3] &= np.uint64(2**62 - 1); return v
base = None
for lg in (26, 25, 24, 23):
    n = 1 << lg
    bases = gm.G1Bases.fixed_base(g_aff, rnd(n))
    d = torch.from_numpy(rnd(n).view(np.int64)).cuda(); torch.cuda.synchronize()
    bases.msm_device(d.data_ptr(), n, mont=False)
    t0 = time.perf_counter()
    for _ in range(4): bases.msm_device(d.data_ptr(), n, mont=False)
    dt = (time.perf_counter() - t0) / 4
    base = base or dt
    print(f"one MSM of 2^{lg} pairs: {dt * 1e3:.2f} ms" + ("" if lg == 26 else f" -> {base / dt:.2f} x for g = {1 << (26 - lg)}"))
    bases.free(); del d
PY
cat $O
