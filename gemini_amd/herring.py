"""Host mirror of herring's sumcheck over a bilinear module (src/herring): the TimeProver of
src/herring/time_prover.rs:42-137 for the two module instances that are in scope:
FModule (F x F -> F) and G1Module (G1 x F -> G1).  G2/GT/pairing modules are out of scope."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .sumcheck import TimeProver as _FieldTimeProver


class FModuleTimeProver(_FieldTimeProver):
    """TimeProver<FModule>: messages a = <f_e, g_e>, b = <f_e, g_o> + <f_o, g_e>; the twist only enters fold"""

    def __init__(self, f, g, twist_mont):
        super().__init__(f, g, twist_mont)
        capi.check(capi.load().gm_sc_set_herring(C.c_uint64(self.handle), C.c_int(1)))


class G1ModuleTimeProver:
    """TimeProver<G1Module>: f = G1 points ((n, 12) affine Montgomery, or (n, 13) Rust records), g = Fr"""

    def __init__(self, f_points, g, twist_mont):
        capi.ensure_init()
        fp = capi.u64(f_points)
        gm_ = capi.u64(g).reshape(-1, 4)
        h = C.c_uint64()
        capi.check(capi.load().gm_hg1_new(capi.ptr(fp), C.c_size_t(fp.shape[1] * 8), C.c_size_t(len(fp)), capi.ptr(gm_), C.c_size_t(len(gm_)),
                                          capi.ptr(capi.u64(twist_mont).reshape(4)), C.byref(h)))
        self.handle = h.value

    def next_message(self, verifier_message=None):
        a = np.empty(18, dtype=np.uint64)
        b = np.empty(18, dtype=np.uint64)
        has = C.c_int()
        ch = None if verifier_message is None else capi.ptr(capi.u64(verifier_message).reshape(4))
        capi.check(capi.load().gm_hg1_round(C.c_uint64(self.handle), ch, capi.ptr(a), capi.ptr(b), C.byref(has)))
        return (a, b) if has.value else None

    def fold(self, challenge):
        capi.check(capi.load().gm_hg1_fold(C.c_uint64(self.handle), capi.ptr(capi.u64(challenge).reshape(4))))

    def rounds(self) -> int:
        t = C.c_size_t()
        capi.check(capi.load().gm_hg1_rounds(C.c_uint64(self.handle), C.byref(t), None))
        return t.value

    def round(self) -> int:
        r = C.c_size_t()
        capi.check(capi.load().gm_hg1_rounds(C.c_uint64(self.handle), None, C.byref(r)))
        return r.value

    def final_foldings(self):
        f0 = np.empty(18, dtype=np.uint64)
        g0 = np.empty(4, dtype=np.uint64)
        has = C.c_int()
        capi.check(capi.load().gm_hg1_final(C.c_uint64(self.handle), capi.ptr(f0), capi.ptr(g0), C.byref(has)))
        return (f0, g0) if has.value else None

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_hg1_free(C.c_uint64(self.handle)))
            self.handle = 0
