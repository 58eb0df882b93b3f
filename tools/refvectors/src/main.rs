//! Dumps reference outputs of arkworks-rs/gemini for the parity tests of the MI355X implementation.
//!
//! For every instance size it records the inputs the other side needs to rebuild the same instance -- the field
//! element `e` of `dummy_r1cs`, the trapdoor `tau` and the generators `g`, `g2` that `CommitterKey::new` draws --
//! and the reference's outputs: the serialised proofs (compressed and uncompressed; every commitment, every sumcheck
//! message and -- through them -- every Fiat-Shamir challenge is in there) plus two stand-alone transcript checks
//! that isolate the only two conventions the other side had to recall: how `append_serializable` frames a G1
//! element and how `get_challenge` maps 64 bytes to a scalar.
//!
//! `tau`, `g`, `g2` are private to `CommitterKey::new`; they are recovered by drawing from a second, identically
//! seeded `test_rng()` in the same order (checked below: commit([1]) == g, commit([0, 1]) == tau * g).
//!
//! Usage: cargo run --release -- <out-dir> [max-logn]      (writes <out-dir>/ref_snark_<curve>.json etc.)
use ark_ec::pairing::Pairing;
use ark_ec::{AffineRepr, CurveGroup, Group};
use ark_ff::{Field, One, PrimeField, Zero};
use ark_gemini::circuit::dummy_r1cs;
use ark_gemini::iterable::dummy::{dummy_r1cs_stream, DummyStreamer};
use ark_gemini::kzg::{Commitment, CommitterKey, CommitterKeyStream};
use ark_serialize::CanonicalSerialize;
use ark_std::{test_rng, UniformRand};
use std::fmt::Write as _;

fn hex(bytes: &[u8]) -> String {
    let mut s = String::with_capacity(2 * bytes.len());
    for b in bytes {
        write!(s, "{:02x}", b).unwrap();
    }
    s
}
fn ser<S: CanonicalSerialize>(v: &S, compressed: bool) -> String {
    let mut out = Vec::new();
    if compressed {
        v.serialize_compressed(&mut out).unwrap();
    } else {
        v.serialize_uncompressed(&mut out).unwrap();
    }
    hex(&out)
}
/// canonical integer, big-endian hex
fn fp_hex<F: PrimeField>(x: &F) -> String {
    let mut le = Vec::new();
    x.serialize_uncompressed(&mut le).unwrap();
    le.reverse();
    format!("0x{}", hex(&le))
}

/// src/transcript.rs:16-34, restated on merlin (the module is private in the reference)
fn append_serializable<S: CanonicalSerialize>(t: &mut merlin::Transcript, label: &'static [u8], v: &S) {
    let mut message = Vec::new();
    v.serialize_uncompressed(&mut message).unwrap();
    t.append_message(label, &message)
}
fn get_challenge<F: Field>(t: &mut merlin::Transcript, label: &'static [u8]) -> F {
    loop {
        let mut bytes = [0u8; 64];
        t.challenge_bytes(label, &mut bytes);
        if let Some(e) = F::from_random_bytes(&bytes) {
            return e;
        }
    }
}

fn run<E: Pairing>(curve: &str, out_dir: &str, max_logn: usize)
where
    E::G1Affine: AffineRepr<BaseField = E::BaseField>,
    E::BaseField: PrimeField,
{
    type SnarkProof<E> = ark_gemini::snark::Proof<E>;
    type PsnarkProof<E> = ark_gemini::psnark::Proof<E>;
    let g1_xy = |p: &E::G1Affine| -> String {
        match p.xy() {
            Some((x, y)) => format!("[\"{}\", \"{}\"]", fp_hex(x), fp_hex(y)),
            None => "null".to_string(),
        }
    };
    let mut cases = Vec::new();
    for logn in 3..=max_logn {
        let n = 1usize << logn;
        // ---- time prover, examples/snark.rs:69-79 -------------------------------------------------------
        let rng = &mut test_rng();
        let shadow = &mut test_rng();
        let r1cs = dummy_r1cs::<E::ScalarField>(rng, n);
        let e = E::ScalarField::rand(shadow);
        assert_eq!(r1cs.z[0], e);
        let ck = CommitterKey::<E>::new(2 * n, 5, rng);
        let tau = E::ScalarField::rand(shadow);
        let g = E::G1::rand(shadow);
        let g2 = E::G2::rand(shadow).into_affine();
        assert_eq!(ck.commit(&[E::ScalarField::one()]).0, g);
        assert_eq!(ck.commit(&[E::ScalarField::zero(), E::ScalarField::one()]).0, g * tau);
        let proof = SnarkProof::<E>::new_time(&r1cs, &ck);
        let snark_verifies = proof.verify(&r1cs, &ark_gemini::kzg::VerifierKey::from(&ck)).is_ok();  // src/snark/tests.rs:71
        let witness = ck.commit(&r1cs.w);
        // ---- stand-alone transcript checks ---------------------------------------------------------------
        let mut t = merlin::Transcript::new(ark_gemini::PROTOCOL_NAME);
        append_serializable(&mut t, b"witness", &witness);
        let alpha: E::ScalarField = get_challenge(&mut t, b"alpha");
        let mut raw = [0u8; 64];
        merlin::Transcript::new(ark_gemini::PROTOCOL_NAME).challenge_bytes(b"raw", &mut raw);
        // ---- preprocessing prover, examples/psnark.rs:70-81 (small sizes: its index is O(n) commitments) ----
        let psnark = if logn <= 6 {
            let rng = &mut test_rng();
            let r1cs = dummy_r1cs::<E::ScalarField>(rng, n);
            let ck = CommitterKey::<E>::new(2 * n, 5, rng);
            let index = PsnarkProof::<E>::index(&ck, &r1cs);
            let p = PsnarkProof::<E>::new_time(&ck, &r1cs, &index);
            // the reference's own verdict on its example configuration (examples/psnark.rs:76: 2n + 1 powers) and on a key
            // with one more power.  The other side's restated verifier predicts false / true: the accumulated products of
            // the sorted vectors have 2n + 2 coefficients and msm_unchecked drops the top one (src/kzg/time.rs:82).
            // dummy_r1cs: A = B = C diagonal, so the joint matrix has n non-zero entries.
            let verifies_example_key = p.verify(&r1cs, &ark_gemini::kzg::VerifierKey::from(&ck), &index, n).is_ok();
            let verifies_long_key = {
                let rng = &mut test_rng();
                let r1cs = dummy_r1cs::<E::ScalarField>(rng, n);
                let ck = CommitterKey::<E>::new(2 * n + 1, 5, rng);
                let index = PsnarkProof::<E>::index(&ck, &r1cs);
                let p = PsnarkProof::<E>::new_time(&ck, &r1cs, &index);
                p.verify(&r1cs, &ark_gemini::kzg::VerifierKey::from(&ck), &index, n).is_ok()
            };
            format!(
                "{{\"index\": {}, \"proof_compressed\": \"{}\", \"proof_uncompressed\": \"{}\", \"powers_of_g2_uncompressed\": \"{}\", \"verifies_example_key\": {}, \"verifies_key_with_one_more_power\": {}}}",
                format!("[{}]", index.iter().map(|c| format!("\"{}\"", ser(c, false))).collect::<Vec<_>>().join(", ")),
                ser(&p, true),
                ser(&p, false),
                ser(&ark_gemini::kzg::VerifierKey::from(&ck).powers_of_g2, false),
                verifies_example_key,
                verifies_long_key
            )
        } else {
            "null".to_string()
        };
        // ---- elastic prover on the generator-copies key, examples/snark.rs:54-66 ----------------------------
        let rng = &mut test_rng();
        let g1_gen = E::G1Affine::generator();
        let r1cs_stream = dummy_r1cs_stream::<E::ScalarField, _>(rng, n);
        let cks = CommitterKeyStream::<E, _> { powers_of_g: DummyStreamer::new(g1_gen, n + 1), powers_of_g2: vec![E::G2Affine::generator(); 4] };
        let elastic = SnarkProof::<E>::new_elastic(r1cs_stream, cks, 1 << 20);
        cases.push(format!(
            concat!(
                "{{\"logn\": {}, \"e\": \"{}\", \"tau\": \"{}\", \"g\": {}, \"g2_uncompressed\": \"{}\",\n",
                "  \"witness_commitment_uncompressed\": \"{}\", \"witness_commitment_compressed\": \"{}\",\n",
                "  \"alpha_after_witness\": \"{}\", \"raw_challenge_bytes\": \"{}\", \"raw_challenge_as_fr\": \"{}\",\n",
                "  \"proof_compressed\": \"{}\",\n  \"proof_uncompressed\": \"{}\", \"verifies\": {},\n",
                "  \"elastic_generator_key\": {{\"proof_compressed\": \"{}\", \"proof_uncompressed\": \"{}\"}},\n",
                "  \"psnark\": {}}}"
            ),
            logn,
            fp_hex(&e),
            fp_hex(&tau),
            g1_xy(&g.into_affine()),
            ser(&g2, false),
            ser(&witness, false),
            ser(&witness, true),
            fp_hex(&alpha),
            hex(&raw),
            fp_hex(&{
                // what from_random_bytes makes of the first accepted draw after `raw`
                let mut t = merlin::Transcript::new(ark_gemini::PROTOCOL_NAME);
                get_challenge::<E::ScalarField>(&mut t, b"raw")
            }),
            ser(&proof, true),
            ser(&proof, false),
            snark_verifies,
            ser(&elastic, true),
            ser(&elastic, false),
            psnark
        ));
        eprintln!("{curve}: logn {logn} done");
    }
    let body = format!(
        "{{\"generator\": \"tools/refvectors (arkworks-rs/gemini + ark-* @ algebra#df51425, merlin 3.0.0)\", \"curve_crate\": \"{}\",\n \"g1_generator_uncompressed\": \"{}\", \"g1_generator_compressed\": \"{}\",\n \"cases\": [\n{}\n]}}\n",
        curve,
        ser(&Commitment::<E>(E::G1::generator()), false),
        ser(&Commitment::<E>(E::G1::generator()), true),
        cases.join(",\n")
    );
    let path = format!("{}/ref_{}.json", out_dir, curve.replace('-', "_"));
    std::fs::write(&path, body).unwrap();
    eprintln!("wrote {path}");
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let out_dir = args.get(1).cloned().unwrap_or_else(|| "../../tests/golden".to_string());
    let max_logn: usize = args.get(2).map(|s| s.parse().unwrap()).unwrap_or(9);
    // the curve crate of the reference's examples / tests (ark-ec default point framing) ...
    run::<ark_test_curves::bls12_381::Bls12_381>("ark-test-curves", &out_dir, max_logn);
    // ... and of its benches (zcash point framing, benches/proofs_bench.rs)
    run::<ark_bls12_381::Bls12_381>("ark-bls12-381", &out_dir, max_logn);
}
