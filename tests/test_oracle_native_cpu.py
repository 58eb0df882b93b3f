"""The TIMED build of the CPU restatement (oracle/libgemini_oracle_native.so: x86-64-v3 + ADX, the Fq product in mulx / adcx / adox
asm, branch-free Fq additions) against the portable checker build of the same source.  bench.py's cpu_baseline legs time the native
build (the reference runs ark-ff with its `asm` feature, Cargo.toml:77-82); nothing is CHECKED against it -- the checker of every
parity test stays the portable library, and this file holds the two equal."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.skipif(not orc.native_available(), reason="host without BMI2 / ADX / AVX2")


def _both(fn):
    a = fn()
    with orc.native():
        assert orc._which == "native"
        b = fn()
    assert orc._which == "portable"
    return a, b


def test_native_build_is_a_second_library():
    p = orc.lib()
    with orc.native():
        n = orc.lib()
    assert p is not n and orc.lib() is p


@pytest.mark.parametrize("n,seed", [(1, 3), (2, 4), (33, 5), (1 << 12, 6)])
def test_msm_same_point(n, seed):
    ks = orc.random_fr(seed, n)
    bases = orc.g1_fixed_base_mul(orc.g1_generator(), ks)
    sc = orc.random_fr(seed + 100, n)
    r_minus_1 = np.array(orc.ints_to_limbs([0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001 - 1], 4)[0])
    sc[0] = r_minus_1  # boundary scalars: r - 1, 0, 1, 2^k
    if n > 3:
        sc[1] = 0
        sc[2] = orc.ints_to_limbs([1], 4)[0]
        sc[3] = orc.ints_to_limbs([1 << 200], 4)[0]
    for threads in (1, 0):
        a, b = _both(lambda: orc.msm_pippenger(bases, sc, threads=threads))
        assert orc.affine_to_ints(orc.g1_to_affine(a)) == orc.affine_to_ints(orc.g1_to_affine(b))
        assert (a == b).all()  # the same sequence of group operations: the same Jacobian representative


def test_fixed_base_and_degenerate_inputs_same_points():
    ks = orc.random_fr(11, 257)
    a, b = _both(lambda: orc.g1_fixed_base_mul(orc.g1_generator(), ks))
    assert (a == b).all()
    # every base the same point: each bucket run is a chain of doublings (examples/snark.rs:59-63); P and -P cancel
    g = np.tile(np.asarray(orc.g1_generator(), dtype=np.uint64), (64, 1))
    sc = orc.random_fr(12, 64)
    x, y = _both(lambda: orc.msm_pippenger(g, sc))
    assert (x == y).all()


def test_sumcheck_and_a_whole_proof_same_bytes():
    n = 1 << 10
    f, g = orc.fr_to_mont(orc.random_fr(21, n)), orc.fr_to_mont(orc.random_fr(22, n))
    tw = orc.fr_to_mont(orc.random_fr(23, 1))[0]

    def run():
        P = orc.TimeProver(f, g, tw)
        out, vm, ch = [], None, orc.fr_to_mont(orc.random_fr(24, 16))
        k = 0
        while True:
            m = P.next_message(vm)
            if m is None:
                break
            out.append(np.concatenate(m))
            vm = ch[k]
            k += 1
        return np.stack(out)

    a, b = _both(run)
    assert (a == b).all()
    from oracle import snark_c, wire_ref

    inst = snark_c.dummy_instance(123456789, 1 << 6) if hasattr(snark_c, "dummy_instance") else None
    if inst is not None:
        pa, pb = _both(lambda: wire_ref.snark_proof(snark_c.new_time(*inst), True))
        assert pa == pb
