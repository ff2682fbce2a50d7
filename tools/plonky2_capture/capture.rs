//! Capture hook for the reference (zama-ai/verifiable-fhe-paper, tag 2024_08_07; plonky2 = "=0.2.0"): drop this file at
//! `src/vtfhe/capture.rs` and apply `reference.patch` (two lines: `pub mod capture;` in src/vtfhe/mod.rs and the `use` of `prove` in
//! src/vtfhe/ivc_based_vpbs.rs).  Every `prove::<F, C, D>(..)` of `verified_pbs` (ivc_based_vpbs.rs:302, :333, :364) then goes through
//! `prove_and_capture`, which has plonky2's `prove` signature and behaviour and, when `VPBS_CAPTURE_DIR` is set, writes per step
//!
//!   $VPBS_CAPTURE_DIR/step_{k:03}/   every file of tests/golden/PLONKY2_FIXTURE_FORMAT.md that the PUBLIC plonky2 API can produce
//!                                    (witness wires, constants/sigmas values, caps, openings, FriProof words, proof bytes, public inputs,
//!                                    circuit digest, meta.json with the gate ids / selector groups / FRI parameters)
//!   $VPBS_CAPTURE_DIR/circuit/       once: the built cyclic circuit itself -- gate ids, selector and constant columns, the copy-constraint
//!                                    forest (representative_map) -- which tools/plonky2_capture/to_fixture.py turns into the STEPCIRC
//!                                    file the MI355X prover loads (verifiable-fhe-paper_amd/circuit_file.py): the REAL step circuit,
//!                                    recursive-verifier rows included.
//!
//! Run:  VPBS_CAPTURE_DIR=/tmp/vpbs_capture cargo test --release test_ivc_blind_rot -- --nocapture      (N = 8: src/ntt/mod.rs selects params_8)
//! Only public items of plonky2 0.2.0 are used: iop::generator::generate_partial_witness, iop::witness::{PartialWitness, Witness},
//! plonk::prover::prove, the pub fields of CommonCircuitData / ProverOnlyCircuitData / Proof / OpeningSet / FriProof.
//! The Z / partial-product values and the quotient chunks live only inside `prove`; the fixture omits them and the consumer recomputes both
//! from the wires (that is the stronger check anyway: it exercises the gate constraints).
use std::fs;
use std::path::PathBuf;
use std::sync::atomic::{AtomicUsize, Ordering};

use anyhow::Result;
use plonky2::field::extension::{Extendable, FieldExtension};
use plonky2::field::types::{Field, PrimeField64};
use plonky2::hash::hash_types::RichField;
use plonky2::iop::generator::generate_partial_witness;
use plonky2::iop::target::Target;
use plonky2::iop::witness::{PartialWitness, Witness};
use plonky2::plonk::circuit_data::{CommonCircuitData, ProverOnlyCircuitData};
use plonky2::plonk::config::{GenericConfig, GenericHashOut};
use plonky2::plonk::proof::ProofWithPublicInputs;
use plonky2::util::timing::TimingTree;

static STEP: AtomicUsize = AtomicUsize::new(0);

fn write_u64(path: PathBuf, xs: impl IntoIterator<Item = u64>) {
    let bytes: Vec<u8> = xs.into_iter().flat_map(|x| x.to_le_bytes()).collect();
    fs::write(path, bytes).expect("capture: write failed");
}

fn c<F: PrimeField64>(x: &F) -> u64 {
    x.to_canonical_u64()
}

fn ext<F: RichField + Extendable<D>, const D: usize>(x: &F::Extension) -> Vec<u64> {
    x.to_basefield_array().iter().map(|b| b.to_canonical_u64()).collect()
}

pub fn prove_and_capture<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    prover_data: &ProverOnlyCircuitData<F, C, D>,
    common_data: &CommonCircuitData<F, D>,
    inputs: PartialWitness<F>,
    timing: &mut TimingTree,
) -> Result<ProofWithPublicInputs<F, C, D>> {
    let dir = match std::env::var("VPBS_CAPTURE_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => return plonky2::plonk::prover::prove::<F, C, D>(prover_data, common_data, inputs, timing),
    };
    let step = STEP.fetch_add(1, Ordering::SeqCst);
    let n = common_data.degree();
    let num_wires = common_data.config.num_wires;
    let num_routed = common_data.config.num_routed_wires;

    // the witness `prove` is about to compute, from a clone of the PartialWitness: generate_partial_witness + full_witness semantics
    // (an unset wire is zero)
    let partition = generate_partial_witness(inputs.clone(), prover_data, common_data);
    let wires: Vec<u64> = (0..num_wires)
        .flat_map(|col| (0..n).map(move |row| (row, col)))
        .map(|(row, col)| partition.try_get_target(Target::wire(row, col)).map(|v| v.to_canonical_u64()).unwrap_or(0))
        .collect();
    drop(partition);

    let proof_with_pis = plonky2::plonk::prover::prove::<F, C, D>(prover_data, common_data, inputs, timing)?;

    let out = dir.join(format!("step_{:03}", step));
    fs::create_dir_all(&out).expect("capture: mkdir failed");
    write_u64(out.join("witness_wires.u64"), wires);

    // constants (selectors first) then sigmas, as values on H
    let cs = &prover_data.constants_sigmas_commitment;
    let cs_values: Vec<u64> = cs.polynomials.iter().flat_map(|p| p.clone().fft().values.iter().map(c).collect::<Vec<_>>()).collect();
    write_u64(out.join("constants_sigmas_values.u64"), cs_values.iter().copied());
    write_u64(out.join("constants_sigmas_cap.u64"), cs.merkle_tree.cap.0.iter().flat_map(|h| h.to_vec().iter().map(c).collect::<Vec<_>>()));
    write_u64(out.join("circuit_digest.u64"), prover_data.circuit_digest.to_vec().iter().map(c));
    write_u64(out.join("public_inputs.u64"), proof_with_pis.public_inputs.iter().map(c));

    let proof = &proof_with_pis.proof;
    let caps = [&proof.wires_cap, &proof.plonk_zs_partial_products_cap, &proof.quotient_polys_cap];
    write_u64(out.join("caps.u64"), caps.iter().flat_map(|cap| cap.0.iter().flat_map(|h| h.to_vec().iter().map(c).collect::<Vec<_>>()).collect::<Vec<_>>()));

    // openings in the order of include/vpbs_prover.h: constants, plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys (at zeta),
    // then plonk_zs_next (at g * zeta); every value as [c0, c1]
    let os = &proof.openings;
    let opening_words: Vec<u64> = os.constants.iter().chain(&os.plonk_sigmas).chain(&os.wires).chain(&os.plonk_zs).chain(&os.partial_products)
        .chain(&os.quotient_polys).chain(&os.plonk_zs_next).flat_map(|e| ext::<F, D>(e)).collect();
    write_u64(out.join("openings.u64"), opening_words);

    // FriProof as flat words: commit-phase caps; per query { per oracle: leaf, siblings; per round: evals, siblings }; final poly; pow witness
    let fri = &proof.opening_proof;
    let mut w: Vec<u64> = Vec::new();
    for cap in &fri.commit_phase_merkle_caps {
        for h in &cap.0 {
            w.extend(h.to_vec().iter().map(c));
        }
    }
    for q in &fri.query_round_proofs {
        for (leaf, merkle_proof) in &q.initial_trees_proof.evals_proofs {
            w.extend(leaf.iter().map(c));
            for s in &merkle_proof.siblings {
                w.extend(s.to_vec().iter().map(c));
            }
        }
        for st in &q.steps {
            for e in &st.evals {
                w.extend(ext::<F, D>(e));
            }
            for s in &st.merkle_proof.siblings {
                w.extend(s.to_vec().iter().map(c));
            }
        }
    }
    for e in &fri.final_poly.coeffs {
        w.extend(ext::<F, D>(e));
    }
    w.push(c(&fri.pow_witness));
    write_u64(out.join("fri.u64"), w);
    fs::write(out.join("proof_bytes.bin"), proof_with_pis.to_bytes()).expect("capture: write failed");

    // meta.json (hand-written JSON: no serde dependency in the reference crate)
    let gate_ids: Vec<String> = common_data.gates.iter().map(|g| format!("\"{}\"", g.0.id().replace('\\', "\\\\").replace('"', "\\\""))).collect();
    let sel = &common_data.selectors_info;
    let groups: Vec<String> = sel.groups.iter().map(|r| format!("[{}, {}]", r.start, r.end)).collect();
    let fp = &common_data.fri_params;
    let meta = format!(
        "{{\"log_n\": {}, \"n_wires\": {}, \"n_routed\": {}, \"num_challenges\": {}, \"n_constants\": {}, \"n_public_inputs\": {}, \
          \"quotient_degree_factor\": {}, \"num_partial_products\": {}, \"gate_ids\": [{}], \"selector_indices\": {:?}, \"selector_groups\": [{}], \
          \"num_gate_constraints\": {}, \"fri\": {{\"rate_bits\": {}, \"cap_height\": {}, \"proof_of_work_bits\": {}, \"num_query_rounds\": {}, \
          \"reduction_arity_bits\": {:?}}}, \"step\": {}}}",
        common_data.degree_bits(), num_wires, num_routed, common_data.config.num_challenges, common_data.num_constants,
        common_data.num_public_inputs, common_data.quotient_degree_factor, common_data.num_partial_products, gate_ids.join(", "),
        sel.selector_indices, groups.join(", "), common_data.num_gate_constraints, fp.config.rate_bits, fp.config.cap_height,
        fp.config.proof_of_work_bits, fp.config.num_query_rounds, fp.reduction_arity_bits, step);
    fs::write(out.join("meta.json"), &meta).expect("capture: write failed");

    if step == 0 {
        // the circuit itself: what CircuitData keeps of it after build()
        let cdir = dir.join("circuit");
        fs::create_dir_all(&cdir).expect("capture: mkdir failed");
        fs::write(cdir.join("meta.json"), &meta).expect("capture: write failed");
        write_u64(cdir.join("constants_sigmas_values.u64"), cs_values.iter().copied());
        // forest of copy constraints over target indices: wire (row, column) -> row * num_wires + column; virtual targets follow
        write_u64(cdir.join("representative_map.u64"), prover_data.representative_map.iter().map(|&r| r as u64));
        // the wire positions of the public inputs (all of them wires of PublicInput-hash rows or virtual targets routed there)
        write_u64(cdir.join("public_input_targets.u64"), prover_data.public_inputs.iter().map(|t| match t {
            Target::Wire(wi) => (wi.row * num_wires + wi.column) as u64,
            Target::VirtualTarget { index } => (n * num_wires + index) as u64,
        }));
        write_u64(cdir.join("k_is.u64"), common_data.k_is.iter().map(c));
        eprintln!("[vpbs capture] circuit: degree 2^{}, {} wires ({} routed), {} gate types, {} public inputs -> {}", common_data.degree_bits(),
                  num_wires, num_routed, common_data.gates.len(), common_data.num_public_inputs, cdir.display());
    }
    // one line per captured prove(): tools/plonky2_capture/README.md shows what a complete run prints
    eprintln!("[vpbs capture] step {:03}: witness + proof ({} bytes) -> {}", step, proof_with_pis.to_bytes().len(), out.display());
    Ok(proof_with_pis)
}
