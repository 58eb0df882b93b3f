"""Host mirror of `GeminiTranscript` over merlin (src/transcript.rs), bound to the C++ implementation
inside libgemini_hip.so (gemini_amd/csrc/transcript.cpp).  Works without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

PROTOCOL_NAME = b"GEMINI-v0"  # src/lib.rs:74

# The ark-serialize framing of group elements is a property of the CURVE CRATE the reference is built against
# (gemini_amd/wire.py): 0 = ark-ec's default (ark-test-curves: the reference's examples and tests), 1 = the zcash
# framing of ark-bls12-381 (the reference's benches).  Process-wide default for new transcripts and for the G2 powers
# the preprocessing prover absorbs.
_default_group_encoding = 0


def set_curve_crate(name: str):
    global _default_group_encoding
    _default_group_encoding = {"ark-test-curves": 0, "ark-bls12-381": 1}[name]


def default_group_encoding() -> int:
    return _default_group_encoding


def _b(x: bytes):
    return (C.c_uint8 * len(x)).from_buffer_copy(x) if x else None


class Transcript:
    def __init__(self, label: bytes = PROTOCOL_NAME):
        h = C.c_uint64()
        capi.check(capi.load().gm_transcript_new(_b(label), C.c_size_t(len(label)), C.byref(h)))
        self.handle = h.value
        if _default_group_encoding:
            self.set_g1_encoding(_default_group_encoding)

    def append_message(self, label: bytes, message: bytes):
        capi.check(capi.load().gm_transcript_append_message(C.c_uint64(self.handle), _b(label), C.c_size_t(len(label)), _b(message), C.c_size_t(len(message))))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        out = (C.c_uint8 * n)()
        capi.check(capi.load().gm_transcript_challenge_bytes(C.c_uint64(self.handle), _b(label), C.c_size_t(len(label)), out, C.c_size_t(n)))
        return bytes(out)

    # GeminiTranscript::append_serializable for the value kinds the prover absorbs
    def append_fr(self, label: bytes, a_mont):
        a = capi.u64(a_mont).reshape(-1, 4)
        capi.check(capi.load().gm_transcript_append_fr(C.c_uint64(self.handle), _b(label), C.c_size_t(len(label)), capi.ptr(a), C.c_size_t(len(a))))

    def append_round_msg(self, label: bytes, a_mont, b_mont):
        self.append_fr(label, np.stack([capi.u64(a_mont).reshape(4), capi.u64(b_mont).reshape(4)]))

    def append_g1(self, label: bytes, jac, with_len: bool = False):
        j = capi.u64(jac).reshape(-1, 18)
        capi.check(capi.load().gm_transcript_append_g1(C.c_uint64(self.handle), _b(label), C.c_size_t(len(label)), capi.ptr(j), C.c_size_t(len(j)), C.c_int(int(with_len))))

    def set_g1_encoding(self, encoding: int):
        """0: ark-ec default framing (ark-test-curves); 1: zcash framing (ark-bls12-381).  See gemini_amd/wire.py."""
        capi.check(capi.load().gm_transcript_set_g1_encoding(C.c_uint64(self.handle), C.c_int(int(encoding))))

    def get_challenge(self, label: bytes) -> np.ndarray:
        out = np.empty(4, dtype=np.uint64)
        capi.check(capi.load().gm_transcript_challenge_fr(C.c_uint64(self.handle), _b(label), C.c_size_t(len(label)), capi.ptr(out)))
        return out

    def free(self):
        if self.handle:
            capi.check(capi.load().gm_transcript_free(C.c_uint64(self.handle)))
            self.handle = 0
