#!/usr/bin/env python3
"""Promised (gm_snark_footprint / gm_psnark_footprint) against used (gm_mem_stats high-water mark) for the provers compiled into the
library, one fresh process per row so that the MSM workspaces start empty.  GPU box: python tools/footprint_table.py > profiles/r5_footprint.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROW = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import gemini_amd as gm
from gemini_amd import snark
from gemini_amd.circuit import R1csStream, dummy_r1cs
from gemini_amd.kzg import CommitterKey, CommitterKeyStream
from gemini_amd.psnark import Proof as PProof
kind, logn = sys.argv[1], int(sys.argv[2])
gm.capi.init()
n = 1 << logn
tau = np.array([123456789, 987654321, 55, 7], dtype=np.uint64)
r1cs = dummy_r1cs(31337, n)
if kind.startswith("snark"):
    ck = CommitterKey.new(n, 3, tau)
    el = kind.endswith("elastic")
    stream = R1csStream(r1cs) if el else None
    fp = gm.capi.snark_footprint(ck.powers_of_g.handle, n, el)
    run = (lambda: snark.new_elastic(stream, CommitterKeyStream.from_committer_key(ck), 1 << 20, native=True)) if el else (lambda: snark.Proof.new_time(r1cs, ck, native=True))
else:
    ck = CommitterKey.new(3 * n, 5, tau)
    index = PProof.index(ck, r1cs)
    mode = {"psnark_time": 0, "psnark_elastic": 1, "psnark_literal": 2}[kind]
    stream = R1csStream(r1cs) if mode else None
    fp = gm.capi.psnark_footprint(ck.powers_of_g.handle, n, n, mode)
    cks = CommitterKeyStream.from_committer_key(ck, min_device_chunk=1 if mode == 2 else None) if mode else None
    run = (lambda: PProof.new_elastic(cks, stream, index, 1 << 20, native=True)) if mode else (lambda: PProof.new_time(ck, r1cs, index, native=True))
before = gm.capi.mem_stats()
gm.capi.mem_reset_peak()
run()
after = gm.capi.mem_stats()
print(json.dumps({"kind": kind, "logn": logn, "promised_vectors": fp["vectors"], "promised_workspaces": fp["workspaces_to_grow"], "promised": fp["needed"],
                  "used": after["in_use_peak"] - before["in_use"], "workspaces_grown": after["msm_workspaces"] - before["msm_workspaces"],
                  "resident_before": before["in_use"], "tables": after["tables"], "keys": after["keys"]}))
''' % ROOT


def main():
    rows = [("snark_time", 20), ("snark_time", 24), ("snark_time", 26), ("snark_elastic", 24), ("snark_elastic", 28), ("psnark_time", 20), ("psnark_time", 24),
            ("psnark_time", 26), ("psnark_elastic", 24), ("psnark_elastic", 26), ("psnark_literal", 22)]
    print("# promised = gm_*_footprint before the proof; used = high-water mark of the library's bytes in use during it, minus what was resident before")
    print("# (key, tables, instance, streams).  GB = 1e9 bytes.  One fresh process per row: the MSM workspaces start empty.")
    print("%-16s %5s %12s %12s %10s %10s %8s %14s" % ("prover", "logn", "promised GB", "of it ws GB", "used GB", "ws grown", "ratio", "resident GB"))
    for kind, logn in rows:
        out = subprocess.run([sys.executable, "-c", ROW, kind, str(logn)], capture_output=True, text=True)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:  # noqa: BLE001
            print("%-16s %5d FAILED %s" % (kind, logn, out.stderr.strip().splitlines()[-1:] if out.stderr else ""))
            continue
        g = lambda v: v / 1e9
        print("%-16s %5d %12.2f %12.2f %10.2f %10.2f %8.2f %14.2f" % (kind, logn, g(d["promised"]), g(d["promised_workspaces"]), g(d["used"]), g(d["workspaces_grown"]),
                                                                    d["promised"] / max(d["used"], 1), g(d["resident_before"])))


if __name__ == "__main__":
    main()
