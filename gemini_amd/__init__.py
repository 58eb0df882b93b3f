"""gemini_amd: MI355X-native implementation of the Gemini (arkworks-rs/gemini) prover hot path.

The product is `libgemini_hip.so` (hand-written HIP for gfx950 behind the extern "C" ABI of
include/gemini_hip.h).  This package is the host-side mirror of the reference's interface for that
path -- VariableBaseMSM / ChunkedPippenger / HashMapPippenger (ark-ec), CommitterKey (src/kzg),
TimeProver / Sumcheck (src/subprotocols/sumcheck), the misc.rs vector helpers and the
Merlin-based GeminiTranscript -- written over ctypes so the parity tests read like the
reference's own tests.  Nothing here computes on the CPU what the reference computes in its hot
path: every MSM, sumcheck round and vector pass is a call into the library.
"""
from . import capi  # noqa: F401
from .msm import VariableBaseMSM, ChunkedPippenger, HashMapPippenger, G1Bases, msm_chunks  # noqa: F401
from .fr import FrVec, fold_polynomial, powers, tensor, hadamard, ip, evaluate_le, linear_combination  # noqa: F401
from .sumcheck import TimeProver, SpaceProver, ElasticProver, Sumcheck  # noqa: F401
from .transcript import Transcript, PROTOCOL_NAME  # noqa: F401
