"""Time-bounded randomised differential soak of the G1 MSM against the CPU restatement (oracle/gemini_oracle.c): NOT collected by
default (the file name), run on the GPU box as

    SOAK_SECONDS=600 python -m pytest tests/soak_msm.py -q -s            # writes gpurun_out/soak_msm.json

What it is after: the generated group law's rare paths under realistic mixes (gen_madd30.py: identity accumulators, doublings,
cancellations inside the asm statement; the XYZZ + XYZZ statement of k_merge / k_group_sum), the run / partial bookkeeping of
k_acc0 with ragged chunks, both bucket paths (plain, fixed-base tables incl. the shared bucket set), offset / reversed walks.
Every case is one MSM compared bit for bit (after normalisation) with the CPU Pippenger on the same pairs; the seed of a failing
case is printed, so it can be replayed with SOAK_SEED / SOAK_CASE."""
import ctypes as C
import json
import os
import time

import numpy as np
import pytest

from tests.util import jac_to_affine_ints, rand_bases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _neg_points(oracle, pts):
    """-P for affine Montgomery points (n, 12)"""
    from oracle import pyref

    out = []
    for p in pts:
        a = oracle.affine_to_ints(p)  # one point -> (x, y) or None for the identity
        out.append(np.zeros(12, dtype=np.uint64) if a is None else np.asarray(oracle.ints_to_affine((a[0], (pyref.Q_MOD - a[1]) % pyref.Q_MOD))).reshape(12))
    return np.stack(out)


def _scalars(oracle, pyref, rng, kind, n, seed):
    sc = oracle.random_fr(seed, n)
    if kind == "uniform":
        return sc
    if kind == "all_equal":
        return np.tile(sc[0], (n, 1))
    if kind == "sparse":
        m = rng.integers(0, 4, size=n)
        sc[m == 0] = 0
        sc[m == 1] = 0
        sc[m == 1, 0] = 1
        small = m == 2
        sc[small, 1:] = 0
        sc[small, 0] &= np.uint64(0xFFFFFF)
        return sc
    if kind == "extreme":  # r - 1, r - 2, 2^k, 2^k - 1: every digit at its sign boundary somewhere
        vals = [pyref.R_MOD - 1, pyref.R_MOD - 2, 1, 2]
        vals += [1 << int(k) for k in rng.integers(1, 254, size=6)]
        vals += [(1 << int(k)) - 1 for k in rng.integers(2, 254, size=6)]
        pick = rng.integers(0, len(vals), size=n)
        return oracle.ints_to_limbs([vals[i] for i in pick], 4)
    if kind == "few":
        k = int(min(n, rng.integers(2, 6)))
        return sc[rng.integers(0, k, size=n)]
    raise AssertionError(kind)


def test_soak_msm(gm, oracle, pyref):
    budget = float(os.environ.get("SOAK_SECONDS", "30"))
    seed0 = int(os.environ.get("SOAK_SEED", "20240929"))
    only_case = os.environ.get("SOAK_CASE")
    lib = gm.capi.load()
    gm.capi.check(lib.gm_set_auto_tables(C.c_int(0), C.c_size_t(0)))
    gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1)))
    t_end = time.time() + budget
    max_log = float(os.environ.get("SOAK_MAX_LOG", "17.2"))  # SOAK_MAX_LOG=20.3: up to 2^20 pairs per case (the CPU side takes ~1 s there)
    NB = 1 << int(max_log)
    pool = rand_bases(oracle, 4242, NB)
    stats = {"cases": 0, "pairs": 0, "by_key": {}, "by_scalars": {}, "by_path": {}, "max_n": 0, "failures": []}
    case = 0
    try:
        while time.time() < t_end:
            if only_case is not None and case != int(only_case):
                case += 1
                if case > int(only_case):
                    break
                continue
            rng = np.random.default_rng(seed0 + case)
            n = int(min(NB, max(1, round(2 ** rng.uniform(0 if max_log < 18 else 12, max_log)))))
            key_kind = ["random", "random", "few_points", "plus_minus", "with_identities"][int(rng.integers(0, 5))]
            nb = int(min(NB, n + int(rng.integers(0, 300))))
            if key_kind == "random":
                start = int(rng.integers(0, NB - nb + 1))
                host = pool[start:start + nb].copy()
            elif key_kind == "few_points":  # every bucket run is doublings: the cold block of the mixed addition
                k = int(rng.integers(1, 4))
                host = pool[rng.integers(0, k, size=nb)].copy()
            elif key_kind == "plus_minus":  # P and -P from a small pool: cancellations to the identity inside runs
                k = int(rng.integers(1, 5))
                both = np.concatenate([pool[:k], _neg_points(oracle, pool[:k])])
                host = both[rng.integers(0, 2 * k, size=nb)].copy()
            else:
                start = int(rng.integers(0, NB - nb + 1))
                host = pool[start:start + nb].copy()
                host[rng.integers(0, nb, size=max(1, nb // 7))] = 0
            sc_kind = ["uniform", "uniform", "all_equal", "sparse", "extreme", "few"][int(rng.integers(0, 6))]
            sc = _scalars(oracle, pyref, rng, sc_kind, n, seed0 + 7 * case + 1)
            reversed_ = bool(rng.integers(0, 2))
            off = int(rng.integers(n - 1, nb)) if reversed_ else int(rng.integers(0, nb - n + 1))
            idx = (off - np.arange(n)) if reversed_ else (off + np.arange(n))
            path = ["plain", "tables20", "tables_small_c"][int(rng.integers(0, 3))]
            reg = gm.G1Bases.register(host)
            try:
                if path == "tables20":
                    reg.precompute(20)
                elif path == "tables_small_c":
                    reg.precompute(int(rng.integers(8, 17)))
                want = oracle.msm_pippenger(host[idx], sc)
                got = reg.msm_bigint(sc, offset=off, reversed_=reversed_)
                ok = jac_to_affine_ints(oracle, got) == jac_to_affine_ints(oracle, want)
            finally:
                reg.free()
            stats["cases"] += 1
            stats["pairs"] += n
            stats["max_n"] = max(stats["max_n"], n)
            for k_, v_ in (("by_key", key_kind), ("by_scalars", sc_kind), ("by_path", path)):
                stats[k_][v_] = stats[k_].get(v_, 0) + 1
            if not ok:
                stats["failures"].append({"case": case, "seed": seed0, "n": n, "key": key_kind, "scalars": sc_kind, "path": path, "reversed": reversed_, "offset": off})
                print("SOAK FAILURE", stats["failures"][-1], flush=True)
            case += 1
    finally:
        gm.capi.check(lib.gm_set_msm_table_min(C.c_size_t(1 << 17)))
        gm.capi.check(lib.gm_set_auto_tables(C.c_int(1), C.c_size_t(0)))
    stats["seconds"] = budget
    stats["seed"] = seed0
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/soak_msm.json", "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps({k: v for k, v in stats.items() if k != "failures"}), flush=True)
    assert not stats["failures"], stats["failures"][:5]
