#!/usr/bin/env python3
"""examples/psnark.rs --time-prover -i <logn> on the device path (examples/psnark.rs:70-81):
dummy_r1cs(2^logn), CommitterKey::new(num_constraints + num_variables, 5), Proof::index, Proof::new_time.
Prints the spans the reference prints with print-trace."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--instance-logsize", type=int, default=16)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--elastic", action="store_true", help="Proof::new_elastic over device-resident streams, max_msm_buffer = 2^20 "
                    "(examples/psnark.rs elastic_snark_main) instead of --time-prover")
    args = ap.parse_args()
    import gemini_amd as gm
    from gemini_amd.circuit import dummy_r1cs
    from gemini_amd.kzg import CommitterKey
    from gemini_amd.psnark import Proof

    gm.capi.init()
    n = 1 << args.instance_logsize
    rng = np.random.default_rng(2022420)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % gm.fr.R_MOD
    r1cs = dummy_r1cs(rnd(), n)
    t0 = time.perf_counter()
    tau = np.array([(rnd() >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    # examples/psnark.rs: time main num_constraints + num_variables powers (:76), elastic main 3 * instance_size + 1 (:62)
    ck = CommitterKey.new(3 * n if args.elastic else 2 * n, 5, tau)
    t_srs = time.perf_counter() - t0
    t0 = time.perf_counter()
    index = Proof.index(ck, r1cs)
    t_index = time.perf_counter() - t0
    out = {"logn": args.instance_logsize, "srs_s": round(t_srs, 3), "index_s": round(t_index, 3), "runs": []}
    for _ in range(args.repeat):
        if args.elastic:
            from gemini_amd.circuit import R1csStream
            from gemini_amd.kzg import CommitterKeyStream

            stream = R1csStream(r1cs)
            proof = Proof.new_elastic(CommitterKeyStream.from_committer_key(ck), stream, index, 1 << 20)
            stream.free()
        else:
            proof = Proof.new_time(ck, r1cs, index)
        out["runs"].append({k: round(v, 4) for k, v in proof.spans.items()})
        out["proof_size_B"] = proof.compressed_size()
    key = "ark_gemini::psnark::elastic_prover" if args.elastic else "ark_gemini::psnark::time_prover"
    out["elastic_prover_s" if args.elastic else "time_prover_s"] = min(r[key] for r in out["runs"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
