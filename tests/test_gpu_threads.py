"""Thread-safety of the C ABI (include/gemini_hip.h "Threading"): the reference calls next_message on
distinct provers from different rayon threads (src/subprotocols/sumcheck/proof.rs:85, `Prover: Send + Sync`
prover.rs:30).  N host threads drive their own sumcheck provers, MSMs (host scalars, resident vectors) and
the vector entry points concurrently through ctypes (which releases the GIL during a call); every result
must equal the serial run bit for bit, whatever the interleaving."""
import threading

import numpy as np
import pytest

from tests.util import rand_bases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gemini_amd

    gemini_amd.capi.init()
    return gemini_amd


def _prover_transcript(gm, f, g, tw, ch):
    P = gm.TimeProver(f, g, tw)
    out = []
    try:
        vm = None
        k = 0
        while True:
            m = P.next_message(vm)
            if m is None:
                break
            out.append(np.concatenate([m[0], m[1]]))
            vm = ch[k]
            k += 1
        ff = P.final_foldings()
        out.append(np.concatenate([ff[0], ff[1]]))
    finally:
        P.free()
    return np.stack(out)


def _space_transcript(gm, f, g, tw, ch):
    P = gm.SpaceProver(f[::-1].copy(), g[::-1].copy(), tw)
    out = []
    try:
        vm = None
        k = 0
        while True:
            m = P.next_message(vm)
            if m is None:
                break
            out.append(np.concatenate([m[0], m[1]]))
            vm = ch[k]
            k += 1
    finally:
        P.free()
    return np.stack(out)


def test_concurrent_provers_msms_and_vector_ops(gm, oracle):
    from gemini_amd.fr import FrVec, evaluate_le, fold_polynomial, ip, tensor

    T = 6
    n = 1 << 12
    bases_host = rand_bases(oracle, 71, n)
    bases = gm.G1Bases.register(bases_host)
    jobs = []
    for t in range(T):
        f = oracle.fr_to_mont(oracle.random_fr(1000 + t, n - 3 * t))
        g = oracle.fr_to_mont(oracle.random_fr(2000 + t, n - 5 * t))
        tw = oracle.fr_to_mont(oracle.random_fr(3000 + t, 1))[0]
        ch = oracle.fr_to_mont(oracle.random_fr(4000 + t, 16))
        sc = oracle.random_fr(5000 + t, n - 7 * t)  # canonical scalars for gm_g1_msm_h
        jobs.append((f, g, tw, ch, sc))

    def work(t):
        f, g, tw, ch, sc = jobs[t]
        res = {}
        res["time"] = _prover_transcript(gm, f, g, tw, ch)
        res["msm_h"] = bases.msm_bigint(sc)
        res["space"] = _space_transcript(gm, f[:257], g[:257], tw, ch)
        v = FrVec.from_host(f)
        w = FrVec.from_host(g[: len(f)] if len(g) >= len(f) else np.concatenate([g, f[len(g):]]))
        res["msm_v"] = bases.msm_vec(v)
        res["ip"] = ip(v, w)
        res["eval"] = evaluate_le(v, ch[:3])
        fo = fold_polynomial(v, ch[1])
        res["fold"] = fo.to_host()
        tv = tensor(ch[: 5 + (t % 3)])
        res["tensor"] = tv.to_host()
        for x in (v, w, fo, tv):
            x.free()
        return res

    serial = [work(t) for t in range(T)]
    for rep in range(3):
        got = [None] * T
        errs = []

        def run(t):
            try:
                got[t] = work(t)
            except Exception as e:  # noqa: BLE001 -- reported below with the thread index
                errs.append((t, repr(e)))

        threads = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errs, errs
        for t in range(T):
            for key, exp in serial[t].items():
                assert np.array_equal(np.asarray(got[t][key]), np.asarray(exp)), f"thread {t}, {key}, repetition {rep}"
    bases.free()


def test_concurrent_batch_commit_and_one_call_msm(gm, oracle):
    """gm_g1_msm_v_batch (two full-size lanes + four small lanes) while other threads issue one-call MSMs on
    the same registered bases: the library lock keeps the workspaces single-flight."""
    from gemini_amd.fr import FrVec

    n = 1 << 13
    bases = gm.G1Bases.register(rand_bases(oracle, 72, n))
    vecs = [FrVec.from_host(oracle.fr_to_mont(oracle.random_fr(6000 + j, n >> (j % 5)))) for j in range(10)]
    exp_batch = bases.msm_vec_batch(vecs, [len(v) for v in vecs])
    exp_single = [bases.msm_vec(v) for v in vecs]
    for j in range(len(vecs)):
        assert np.array_equal(exp_batch[j], exp_single[j])
    out = {}

    def batch():
        out["b"] = [bases.msm_vec_batch(vecs, [len(v) for v in vecs]) for _ in range(3)]

    def singles(k):
        out[k] = [bases.msm_vec(vecs[(k + i) % len(vecs)]) for i in range(12)]

    ths = [threading.Thread(target=batch)] + [threading.Thread(target=singles, args=(k,)) for k in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for b in out["b"]:
        for j in range(len(vecs)):
            assert np.array_equal(b[j], exp_batch[j])
    for k in range(3):
        for i in range(12):
            assert np.array_equal(out[k][i], exp_single[(k + i) % len(vecs)])
    for v in vecs:
        v.free()
    bases.free()
