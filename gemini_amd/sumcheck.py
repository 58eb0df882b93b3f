"""Host mirror of src/subprotocols/sumcheck: `trait Prover` (prover.rs:30-45), `TimeProver`
(time_prover.rs:42-137) and the `Sumcheck::prove` round loop (proof.rs:36-66).  Each
next_message is one fused fold+message kernel launch in libgemini_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .fr import FrVec


class TimeProver:
    def __init__(self, f, g, twist_mont, borrow=False):
        """Witness::new + TimeProver::new: copies f and g (time_prover.rs:26-32, 57-67).  borrow=True (device vectors only):
        no copy, the prover reads f and g in place until its first fold -- the caller leaves them alone until then."""
        capi.ensure_init()
        h = C.c_uint64()
        tw = capi.u64(twist_mont).reshape(4)
        if isinstance(f, FrVec) and isinstance(g, FrVec):
            new = capi.load().gm_sc_new_borrow if borrow else capi.load().gm_sc_new_v
            capi.check(new(C.c_uint64(f.handle), C.c_uint64(g.handle), capi.ptr(tw), C.byref(h)))
            self._keep = (f, g) if borrow else None
        else:
            if borrow:
                raise ValueError("borrow=True needs device vectors (FrVec)")
            fm = capi.u64(f).reshape(-1, 4)
            gm = capi.u64(g).reshape(-1, 4)
            capi.check(capi.load().gm_sc_new(capi.ptr(fm), C.c_size_t(len(fm)), capi.ptr(gm), C.c_size_t(len(gm)), capi.ptr(tw), C.byref(h)))
        self.handle = h.value

    def next_message(self, verifier_message=None):
        a = np.empty(4, dtype=np.uint64)
        b = np.empty(4, dtype=np.uint64)
        has = C.c_int()
        ch = None if verifier_message is None else capi.ptr(capi.u64(verifier_message).reshape(4))
        capi.check(capi.load().gm_sc_round(C.c_uint64(self.handle), ch, capi.ptr(a), capi.ptr(b), C.byref(has)))
        return (a, b) if has.value else None

    def fold(self, challenge):
        capi.check(capi.load().gm_sc_fold(C.c_uint64(self.handle), capi.ptr(capi.u64(challenge).reshape(4))))

    def rounds(self) -> int:
        t = C.c_size_t()
        capi.check(capi.load().gm_sc_rounds(C.c_uint64(self.handle), C.byref(t), None))
        return t.value

    def round(self) -> int:
        r = C.c_size_t()
        capi.check(capi.load().gm_sc_rounds(C.c_uint64(self.handle), None, C.byref(r)))
        return r.value

    def final_foldings(self):
        f0 = np.empty(4, dtype=np.uint64)
        g0 = np.empty(4, dtype=np.uint64)
        has = C.c_int()
        capi.check(capi.load().gm_sc_final(C.c_uint64(self.handle), capi.ptr(f0), capi.ptr(g0), C.byref(has)))
        return (f0, g0) if has.value else None

    def state(self):
        """(f, g, twist) as they stand after the folds so far (TimeProver's pub fields)"""
        nf, ng = C.c_size_t(), C.c_size_t()
        tw = np.empty(4, dtype=np.uint64)
        capi.check(capi.load().gm_sc_lens(C.c_uint64(self.handle), C.byref(nf), C.byref(ng), capi.ptr(tw)))
        f = np.empty((nf.value, 4), dtype=np.uint64)
        g = np.empty((ng.value, 4), dtype=np.uint64)
        capi.check(capi.load().gm_sc_download(C.c_uint64(self.handle), capi.ptr(f), capi.ptr(g)))
        return f, g, tw

    def set_shard(self, pair_offset: int):
        capi.check(capi.load().gm_sc_set_shard(C.c_uint64(self.handle), C.c_uint64(pair_offset)))

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_sc_free(C.c_uint64(self.handle)))
            self.handle = 0


SPACE_TIME_THRESHOLD = 22  # src/lib.rs:76


class SpaceProver:
    """src/subprotocols/sumcheck/space_prover.rs: keeps the big-endian streams and the challenges only;
    every message is recomputed from the streams on the device (gm_sp_*)."""

    def __init__(self, f_stream, g_stream, twist_mont, borrow=False):
        """borrow=True (device vectors only): the prover reads the streams in place for its whole life"""
        capi.ensure_init()
        h = C.c_uint64()
        tw = capi.u64(twist_mont).reshape(4)
        if isinstance(f_stream, FrVec) and isinstance(g_stream, FrVec):
            new = capi.load().gm_sp_new_borrow if borrow else capi.load().gm_sp_new_v
            capi.check(new(C.c_uint64(f_stream.handle), C.c_uint64(g_stream.handle), capi.ptr(tw), C.byref(h)))
            self._keep = (f_stream, g_stream) if borrow else None
        else:
            if borrow:
                raise ValueError("borrow=True needs device vectors (FrVec)")
            fm = capi.u64(f_stream).reshape(-1, 4)
            gm_ = capi.u64(g_stream).reshape(-1, 4)
            capi.check(capi.load().gm_sp_new(capi.ptr(fm), C.c_size_t(len(fm)), capi.ptr(gm_), C.c_size_t(len(gm_)), capi.ptr(tw), C.byref(h)))
        self.handle = h.value

    def next_message(self, verifier_message=None):
        a = np.empty(4, dtype=np.uint64)
        b = np.empty(4, dtype=np.uint64)
        has = C.c_int()
        ch = None if verifier_message is None else capi.ptr(capi.u64(verifier_message).reshape(4))
        capi.check(capi.load().gm_sp_round(C.c_uint64(self.handle), ch, capi.ptr(a), capi.ptr(b), C.byref(has)))
        return (a, b) if has.value else None

    def fold(self, challenge):
        capi.check(capi.load().gm_sp_fold(C.c_uint64(self.handle), capi.ptr(capi.u64(challenge).reshape(4))))

    def rounds(self) -> int:
        t = C.c_size_t()
        capi.check(capi.load().gm_sp_rounds(C.c_uint64(self.handle), C.byref(t), None))
        return t.value

    def round(self) -> int:
        r = C.c_size_t()
        capi.check(capi.load().gm_sp_rounds(C.c_uint64(self.handle), None, C.byref(r)))
        return r.value

    def final_foldings(self):
        f0 = np.empty(4, dtype=np.uint64)
        g0 = np.empty(4, dtype=np.uint64)
        has = C.c_int()
        capi.check(capi.load().gm_sp_final(C.c_uint64(self.handle), capi.ptr(f0), capi.ptr(g0), C.byref(has)))
        return (f0, g0) if has.value else None

    def to_time_prover(self) -> "TimeProver":
        """TimeProver::from(&SpaceProver) (space_prover.rs:269-307)"""
        h = C.c_uint64()
        capi.check(capi.load().gm_sp_to_time(C.c_uint64(self.handle), C.byref(h)))
        tp = TimeProver.__new__(TimeProver)
        tp.handle = h.value
        return tp

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_sp_free(C.c_uint64(self.handle)))
            self.handle = 0


class ElasticProver:
    """src/subprotocols/sumcheck/elastic_prover.rs: a SpaceProver that turns into a TimeProver when
    fewer than SPACE_TIME_THRESHOLD rounds remain."""

    def __init__(self, f_stream, g_stream, twist_mont):
        self.space = SpaceProver(f_stream, g_stream, twist_mont)
        self.time = None

    def _cur(self):
        return self.time if self.time is not None else self.space

    def next_message(self, verifier_message=None):
        # `Prover::next_message` of each variant folds first; the elastic switch lives in fold (:44-57)
        if verifier_message is not None and self.time is None:
            self.fold(verifier_message)
            verifier_message = None
        return self._cur().next_message(verifier_message)

    def fold(self, challenge):
        if self.time is None:
            p = self.space
            if p.rounds() - p.round() < SPACE_TIME_THRESHOLD:
                self.time = p.to_time_prover()
                self.time.fold(challenge)
                p.free()
            else:
                p.fold(challenge)
        else:
            self.time.fold(challenge)

    def rounds(self) -> int:
        return self._cur().rounds()

    def round(self) -> int:
        return self._cur().round()

    def final_foldings(self):
        return self._cur().final_foldings()

    def free(self):
        if self.time is not None:
            self.time.free()
        if self.space.handle:
            self.space.free()


class Sumcheck:
    """src/subprotocols/sumcheck/proof.rs:19-31"""

    def __init__(self, messages, challenges, rounds, final_foldings):
        self.messages = messages
        self.challenges = challenges
        self.rounds = rounds
        self.final_foldings = final_foldings

    @staticmethod
    def prove(transcript, prover) -> "Sumcheck":
        """proof.rs:36-66.  `transcript` offers append_round_msg / append_fr / get_challenge."""
        messages, challenges = [], []
        verifier_message = None
        while True:
            message = prover.next_message(verifier_message)
            if message is None:
                break
            transcript.append_round_msg(b"evaluations", message[0], message[1])
            challenge = transcript.get_challenge(b"challenge")
            verifier_message = challenge
            messages.append(message)
            challenges.append(challenge)
        rounds = prover.rounds()
        ff = prover.final_foldings()
        transcript.append_fr(b"final-folding", ff[0])
        transcript.append_fr(b"final-folding", ff[1])
        return Sumcheck(messages, challenges, rounds, [ff])

    @staticmethod
    def prove_native(transcript, prover: "TimeProver") -> "Sumcheck":
        """the same round loop run inside the library (gm_sumcheck_prove): no Python per round"""
        cap = prover.rounds() + 1
        msgs = np.zeros((cap, 8), dtype=np.uint64)
        chs = np.zeros((cap, 4), dtype=np.uint64)
        ff = np.zeros(8, dtype=np.uint64)
        k = C.c_size_t()
        capi.check(capi.load().gm_sumcheck_prove(C.c_uint64(transcript.handle), C.c_uint64(prover.handle), capi.ptr(msgs), capi.ptr(chs),
                                                 C.c_size_t(cap), capi.ptr(ff), C.byref(k)))
        n = k.value
        return Sumcheck([(msgs[i, :4].copy(), msgs[i, 4:].copy()) for i in range(n)], [chs[i].copy() for i in range(n)], prover.rounds(),
                        [(ff[:4].copy(), ff[4:].copy())])

    @staticmethod
    def new_space(transcript, f_stream, g_stream, twist_mont) -> "Sumcheck":
        """proof.rs:133-142"""
        prover = SpaceProver(f_stream, g_stream, twist_mont)
        try:
            return Sumcheck.prove(transcript, prover)
        finally:
            prover.free()

    @staticmethod
    def new_elastic(transcript, f_stream, g_stream, twist_mont) -> "Sumcheck":
        """proof.rs:145-154"""
        prover = ElasticProver(f_stream, g_stream, twist_mont)
        try:
            return Sumcheck.prove(transcript, prover)
        finally:
            prover.free()

    @staticmethod
    def prove_batch(transcript, provers) -> "Sumcheck":
        """proof.rs:69-122 (run inside the library: gm_sumcheck_prove_batch)"""
        k = len(provers)
        cap = max(p.rounds() for p in provers) + 1
        msgs = np.zeros((cap, 8), dtype=np.uint64)
        chs = np.zeros((cap, 4), dtype=np.uint64)
        ff = np.zeros((k, 8), dtype=np.uint64)
        handles = np.array([p.handle for p in provers], dtype=np.uint64)
        r = C.c_size_t()
        capi.check(capi.load().gm_sumcheck_prove_batch(C.c_uint64(transcript.handle), capi.ptr(handles), C.c_size_t(k), capi.ptr(msgs), capi.ptr(chs),
                                                       C.c_size_t(cap), capi.ptr(ff), C.byref(r)))
        n = r.value
        return Sumcheck([(msgs[i, :4].copy(), msgs[i, 4:].copy()) for i in range(n)], [chs[i].copy() for i in range(n)], n,
                        [(ff[j, :4].copy(), ff[j, 4:].copy()) for j in range(k)])

    @staticmethod
    def prove_batch_generic(transcript, provers) -> "Sumcheck":
        """proof.rs:69-122 over any `Prover` objects (space / elastic / time), round loop on the host"""
        from .fr import R_MOD, fr_from_int, fr_to_int

        rounds = max(p.rounds() for p in provers) + 1
        coefficients = [fr_to_int(transcript.get_challenge(b"batch-sumcheck")) for _ in provers]
        messages, challenges = [], []
        vm = None
        for _ in range(rounds):
            a = b = 0
            for p, c in zip(provers, coefficients):
                m = p.next_message(vm)
                if m is None:
                    ff = p.final_foldings()
                    ma, mb = fr_to_int(ff[0]) * fr_to_int(ff[1]) % R_MOD, 0
                else:
                    ma, mb = fr_to_int(m[0]), fr_to_int(m[1])
                a = (a + ma * c) % R_MOD
                b = (b + mb * c) % R_MOD
            msg = (fr_from_int(a), fr_from_int(b))
            transcript.append_round_msg(b"evaluations", msg[0], msg[1])
            vm = transcript.get_challenge(b"challenge")
            messages.append(msg)
            challenges.append(vm)
        finals = []
        for p in provers:
            ff = p.final_foldings()
            transcript.append_fr(b"final-folding-lhs", ff[0])
            transcript.append_fr(b"final-folding-rhs", ff[1])
            finals.append(ff)
        return Sumcheck(messages, challenges, rounds, finals)

    @staticmethod
    def new_time(transcript, f, g, twist_mont, native: bool = True) -> "Sumcheck":
        """proof.rs:125-130"""
        prover = TimeProver(f, g, twist_mont)
        try:
            return Sumcheck.prove_native(transcript, prover) if native else Sumcheck.prove(transcript, prover)
        finally:
            prover.free()
