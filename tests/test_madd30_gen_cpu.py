"""CPU: the generated asm of the hot kernel's mixed addition (gemini_amd/csrc/gen_madd30.py -> g1_madd30_gen.inc).

The generator interprets the instruction list it emits (integer semantics of every opcode, 64-bit column accumulators
and 32-bit limb sums asserted not to wrap) against big-integer XYZZ arithmetic: EFD madd-2008-s on residues mod q --
what `Projective::add_assign(&Affine)` computes inside VariableBaseMSM::msm_bigint (src/kzg/msm/variable_base.rs:125-139)
up to the projective representative.  Covered: loose representatives at the invariant bounds, an identity accumulator, a
negated base, p == 0 handled inside the statement (doubling when r == 0 as well, the identity otherwise).  The committed .inc must be what the
generator prints, so the code that ships is the code that was checked."""
import io
import os
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gemini_amd", "csrc")


def _gen():
    sys.path.insert(0, CSRC)
    try:
        import gen_madd30
    finally:
        sys.path.pop(0)
    return gen_madd30


def test_madd30_interpreted_against_big_integers():
    _gen().selftest(160)


def test_add30_interpreted_against_big_integers():
    """the XYZZ + XYZZ statement of k_merge / k_group_sum (add-2008-s): identity operands, p == 0 flagged with both
    operands intact, loose representatives at the bounds"""
    _gen().selftest_add(120)


def test_committed_inc_is_what_the_generator_emits():
    with open(os.path.join(CSRC, "g1_madd30_gen.inc")) as f:
        assert f.read() == _gen().render()


def test_bounds_leave_headroom():
    g = _gen()
    gen = g.Gen()
    gen.madd()
    # every value stays far below the 2^386 limit of the loose representation (39.6 q)
    assert max(gen.bounds.values()) < 9
