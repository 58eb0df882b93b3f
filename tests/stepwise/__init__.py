"""TEST INFRASTRUCTURE: the step-wise Python statements of the provers (one FFI call per step), kept as the byte-for-byte cross-check of the provers
compiled into the library -- and as what runs on the Python-level sharded keys of tests/stepwise/dist.py.  The product package has ONE orchestration
per prover (the native one); importing this package registers the step-wise ones with gemini_amd.snark / gemini_amd.psnark, which is what
`native=False` then reaches."""
from gemini_amd import psnark as _psnark
from gemini_amd import snark as _snark
from tests.stepwise import psnark_steps, snark_steps

_snark.register_stepwise("new_time", snark_steps.new_time)
_snark.register_stepwise("new_elastic", snark_steps.new_elastic)
_psnark.register_stepwise("new_time", psnark_steps.new_time)
_psnark.register_stepwise("new_elastic", psnark_steps.new_elastic)
