"""The verifiers of this build against the reference's OWN verifier text.

tests/golden/verifier_trace.json holds what the contract rendered from
/root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol computed when its Yul was executed on proofs of the CPU
oracle (oracle/tools/gen_verifier_trace.py, run in the build container): the accept / reject answer and the contract's named
memory words -- every Fiat-Shamir challenge, x^n, l_last, l_blind, l_0, the instance evaluation, the quotient evaluation
and commitment, the SHPLONK scalar and both pairing inputs.  Here oracle/plonk.py:verify (CPU) and shp_verify (the
library's native verifier) must reproduce them word for word, and the oracle must still produce the very proof bytes the
contract accepted."""
import importlib
import json
import random

import pytest

from conftest import golden_path

GOLD = json.load(open(golden_path("verifier_trace.json")))
WORDS = GOLD["trace_words"]


def rebuild(oracle, case):
    PL = importlib.import_module("oracle.plonk")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    plan_bytes, lay = getattr(C, case["builder"])(**case["kwargs"])
    plan = PL.Plan(plan_bytes)
    if case["srs"] == "real":
        srs = oracle.srs_from_halo2_raw(open(golden_path("ppot_0080_11_raw"), "rb").read())
    else:
        srs = oracle.srs_synthetic(plan.k, int(case["tau"], 16))
    return PL, C, plan_bytes, plan, lay, srs


def witness_of(C, case, lay, hash_fn):
    if case["builder"] == "build_phased_plan":
        return C.phased_witness_inputs(lay, random.Random(case["witness_seed"]))
    return C.withdraw_witness_inputs(lay, hash_fn, random.Random(case["witness_seed"]), case["balance"], case["value"])


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: c["builder"] + "_k%d" % c["kwargs"]["k"])
def test_oracle_verifier_reproduces_the_contract(oracle, case):
    PL, C, plan_bytes, plan, lay, srs = rebuild(oracle, case)
    opk = PL.keygen(plan, srs)
    assert opk.vk_digest.hex() == case["vk_digest"]
    g2, s_g2 = bytes.fromhex(case["g2"]), bytes.fromhex(case["s_g2"])
    # the proof the contract accepted is the proof the oracle produces today
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = witness_of(C, case, lay, H)
    proof, inst = PL.prove(opk, srs, wit, bytes.fromhex(case["blinding_seed"]))
    assert proof.hex() == case["vectors"][0]["proof"] and ["0x%064x" % v for v in inst] == case["vectors"][0]["instances"]
    for vec in case["vectors"]:
        tr = {}
        ok = PL.verify(opk, bytes.fromhex(vec["proof"]), [int(v, 16) for v in vec["instances"]], g2=g2, s_g2=s_g2, trace=tr)
        assert ok == vec["accepted"], vec["label"]
        if vec["trace"] is not None:
            for name in WORDS:
                assert "0x%064x" % tr[name] == vec["trace"][name], f'{vec["label"]}: {name}'
            # a circuit with several phases: the challenges the contract squeezed between the advice commitments
            assert ["0x%064x" % v for v in tr.get("USER_CHALLENGES", [])] == vec["trace"].get("USER_CHALLENGES", []), vec["label"]
    assert case["vectors"][0]["accepted"] and not any(v["accepted"] for v in case["vectors"][1:])
    assert sum(v["trace"] is not None for v in case["vectors"]) >= 3        # full runs that end in a failed pairing are covered


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: c["builder"] + "_k%d" % c["kwargs"]["k"])
def test_native_verifier_reproduces_the_contract(shp, oracle, gpu_ctx, case):
    """shp_verify / shp_verify_trace on the same vectors; the HIP prover's proof for the same inputs is the accepted one"""
    PL, C, plan_bytes, plan, lay, srs = rebuild(oracle, case)
    if case["srs"] == "real":
        params = shp.Params.read_raw(gpu_ctx, open(golden_path("ppot_0080_11_raw"), "rb").read())
    else:
        params = shp.Params.synthetic(gpu_ctx, plan.k, int(case["tau"], 16))
    assert params.g2() == (bytes.fromhex(case["g2"]), bytes.fromhex(case["s_g2"]))
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    assert pk.vk_digest().hex() == case["vk_digest"]
    wit = witness_of(C, case, lay, shp.poseidon2_hash_host)
    pr = shp.Prover(pk)
    proof, inst = pr.prove(wit, bytes.fromhex(case["blinding_seed"]))
    assert proof.hex() == case["vectors"][0]["proof"], "the HIP prover's proof is not the one the reference's contract accepted"
    for vec in case["vectors"]:
        p, ins = bytes.fromhex(vec["proof"]), [int(v, 16) for v in vec["instances"]]
        assert pk.verify(p, ins) == vec["accepted"], vec["label"]
        ok, words, chal = pk.verify_trace(p, ins, with_challenges=True)
        assert ok == vec["accepted"]
        if vec["trace"] is not None:
            assert ["0x%064x" % c for c in chal] == vec["trace"].get("USER_CHALLENGES", []), vec["label"]
        if vec["trace"] is not None:
            for name, w in zip(WORDS, words):
                assert "0x%064x" % w == vec["trace"][name], f'{vec["label"]}: {name}'
    pr.close(); pk.close(); params.close()
