"""GPU: whole proofs from the HIP prover, byte for byte against the CPU oracle (same plan, witness,
SRS and blinding seed), and accepted by the restated verifier equations."""
import importlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
TAU = 0x1234567890ABCDEF1234567890ABCDEF55


def setup(shp, oracle, gpu_ctx, small):
    from oracle import plonk as PL
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    if small:
        plan_bytes, lay = C.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
    else:
        plan_bytes, lay = C.build_withdraw_plan()
    plan = PL.Plan(plan_bytes)
    osrs = oracle.srs_synthetic(plan.k, TAU)
    opk = PL.keygen(plan, osrs)
    params = shp.Params.from_parts(gpu_ctx, plan.k, osrs.g, osrs.g_lagrange)
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    return PL, C, plan, lay, osrs, opk, params, pk


@pytest.mark.parametrize("small", [True, False])
def test_keygen_matches_oracle(shp, oracle, gpu_ctx, small):
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, small)
    f, s = pk.commitments(plan.num_fixed, len(plan.perm_cols))
    for i in range(plan.num_fixed):
        assert np.array_equal(f[i], opk.fixed_commitments[i]), f"fixed commitment {i}"
    for i in range(len(plan.perm_cols)):
        assert np.array_equal(s[i], opk.sigma_commitments[i]), f"sigma commitment {i}"
    assert pk.vk_digest() == opk.vk_digest
    assert pk.proof_len == PL.proof_layout(plan)[2]
    pk.close(); params.close()


@pytest.mark.parametrize("small", [True, False])
def test_proof_bytes_match_oracle_and_verify(shp, oracle, gpu_ctx, small):
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, small)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    prover = shp.Prover(pk)
    for trial in range(2):
        rnd = random.Random(1000 + trial)
        bal, val = (10**18, 10**17) if not small else (2**31 + 5, 2**20 + 3 + trial)
        wit = C.withdraw_witness_inputs(lay, H, rnd, bal, val)
        seed = bytes([trial + 1] * 32)
        proof, inst = prover.prove(wit, seed)
        oproof, oinst = PL.prove(opk, osrs, wit, seed)
        assert inst == oinst
        assert len(proof) == len(oproof)
        if proof != oproof:
            first = next(i for i in range(len(proof)) if proof[i] != oproof[i])
            raise AssertionError(f"proof differs from the oracle at byte {first} (of {len(proof)})")
        assert PL.verify(opk, proof, inst, tau=TAU)
        bad = bytearray(proof); bad[70] ^= 1
        assert not PL.verify(opk, bytes(bad), inst, tau=TAU)
    # library host hash == oracle hash (used by bench.py's witness generator)
    assert shp.poseidon2_hash_host([1, 2, 3, 4, 5, 6, 7]) == H([1, 2, 3, 4, 5, 6, 7])
    prover.close(); pk.close(); params.close()


def test_error_paths(shp, oracle, gpu_ctx):
    """non-canonical witness element -> SHP_ERR_FORMAT; an unsatisfying witness gives a proof the
    verifier rejects; a lookup input outside the table -> SHP_ERR_UNSATISFIED"""
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, True)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(5), 2**31, 2**20)
    prover = shp.Prover(pk)
    with pytest.raises(shp.ShpError) as e:
        prover.prove_bytes(b"".join((R if i == lay.ID else x).to_bytes(32, "little") for i, x in enumerate(wit)), bytes(32))
    assert e.value.code == 4
    bad = list(wit)
    bad[lay.path(0, 0)] ^= 1 if bad[lay.path(0, 0)] != H(wit[:1]) else 2     # breaks the Merkle root / membership
    bad[lay.VALUE] = 2**40                                                     # and the range decomposition
    proof, inst = prover.prove(bad, bytes(32))
    assert not PL.verify(opk, proof, inst, tau=TAU)
    prover.close(); pk.close(); params.close()

    # tiny plan with a bare lookup: advice cell must be in {0..15}
    B = importlib.import_module("zkos-monorepo_amd.circuits.builder")
    b = B.PlanBuilder(6)
    a0 = b.advice_column()
    tab = b.fixed_column()
    for r in range(16):
        b.fix(tab, r, r)
    b.lookup([B.Adv(a0, 0)], [B.Fix(tab, 0)])
    b.enable_equality(B.COL_ADVICE, a0)
    b.n_winputs = 2
    b.w(B.W_INPUT, (a0, 0), imm=0)
    b.w(B.W_INPUT, (a0, 1), imm=1)
    b.copy((B.COL_ADVICE, a0, 0), (B.COL_ADVICE, a0, 1))
    tiny = b.build()
    tplan = PL.Plan(tiny)
    tsrs = oracle.srs_synthetic(6, TAU)
    tparams = shp.Params.from_parts(gpu_ctx, 6, tsrs.g, tsrs.g_lagrange)
    tpk = shp.ProvingKey(gpu_ctx, tparams, tiny)
    tpr = shp.Prover(tpk)
    topk = PL.keygen(tplan, tsrs)
    proof, inst = tpr.prove([7, 7], bytes(32))
    oproof, _ = PL.prove(topk, tsrs, [7, 7], bytes(32))
    assert proof == oproof and PL.verify(topk, proof, inst, tau=TAU)
    with pytest.raises(shp.ShpError) as e:
        tpr.prove([99, 99], bytes(32))
    assert e.value.code == 6
    # the refused proof left transforms in flight on the prover's second stream: the next proof is unaffected
    proof2, inst2 = tpr.prove([7, 7], bytes(32))
    assert proof2 == oproof and inst2 == inst
    tpr.close(); tpk.close(); tparams.close()


def test_pool_batch_equals_single_lane(shp, oracle, gpu_ctx):
    """shp_pool / shp_prove_batch (the "whole node" serve path): a batch sharded over lanes gives,
    proof for proof, the bytes of a single lane and of the oracle; submit/wait and the error path"""
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, True)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    rnd = random.Random(77)
    count = 7
    wits = [C.withdraw_witness_inputs(lay, H, rnd, 2**31 + i, 2**20 + i) for i in range(count)]
    blobs = [b"".join(int(x).to_bytes(32, "little") for x in w) for w in wits]
    seeds = [bytes([i + 1]) * 32 for i in range(count)]
    pool = shp.Pool(pk, 3)
    got = pool.prove_batch(b"".join(blobs), b"".join(seeds), count)
    single = shp.Prover(pk)
    for i in range(count):
        p1, i1 = single.prove_bytes(blobs[i], seeds[i])
        assert got[i] == (p1, i1), f"batch proof {i} differs from the single-lane proof"
    oproof, oinst = PL.prove(opk, osrs, wits[3], seeds[3])
    assert got[3] == (oproof, oinst)
    assert PL.verify(opk, got[5][0], got[5][1], tau=TAU)
    # asynchronous tickets, collected out of order
    t = [pool.submit(blobs[i], seeds[i]) for i in range(4)]
    for i in (2, 0, 3, 1):
        assert pool.wait(t[i]) == got[i]
    # a non-canonical witness fails its own ticket with SHP_ERR_FORMAT and leaves the pool usable
    bad = bytearray(blobs[0]); bad[0:32] = R.to_bytes(32, "little")
    tb = pool.submit(bytes(bad), seeds[0])
    with pytest.raises(shp.ShpError) as e:
        pool.wait(tb)
    assert e.value.code == 4
    with pytest.raises(shp.ShpError):
        pool.prove_batch(bytes(bad) + blobs[1], seeds[0] + seeds[1], 2)
    assert pool.wait(pool.submit(blobs[1], seeds[1])) == got[1]
    done, queued = pool.stats()
    assert done >= count + 8 and queued == 0
    pool.close(); single.close(); pk.close(); params.close()


def test_config1_real_srs_k11_proof_and_pairing(shp, oracle, gpu_ctx):
    """BASELINE config 1 through the HIP path: the reference's own k = 11 SRS file, a withdraw-shaped proof,
    byte-identical to the oracle's and accepted by the restated verifier WITH the real pairing
    (templates/Halo2Verifier.sol:553-559); calldata as the reference encodes it"""
    from conftest import golden_path
    PL = importlib.import_module("oracle.plonk")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    raw = open(golden_path("ppot_0080_11_raw"), "rb").read()
    plan_bytes, lay = C.build_deposit_plan(k=11, height=3, range_bits=8, range_chunks=10)   # config 1 names the deposit circuit
    params = shp.Params.read_raw(gpu_ctx, raw)
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    osrs = oracle.srs_from_halo2_raw(raw)
    opk = PL.keygen(PL.Plan(plan_bytes), osrs)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(3), 2**31 + 5, 2**20 + 3)
    prover = shp.Prover(pk)
    proof, inst = prover.prove(wit, bytes(range(32)))
    oproof, oinst = PL.prove(opk, osrs, wit, bytes(range(32)))
    assert (proof, inst) == (oproof, oinst)
    assert PL.verify(opk, proof, inst, g2=osrs.g2, s_g2=osrs.s_g2)
    bad = bytearray(proof); bad[-70] ^= 4
    assert not PL.verify(opk, bytes(bad), inst, g2=osrs.g2, s_g2=osrs.s_g2)
    assert params.g2() == (osrs.g2, osrs.s_g2)
    cd = shp.encode_calldata(proof, inst)
    assert cd[:4].hex() == "1e8e1e13" and len(cd) == 4 + 96 + len(proof) + 32 + 32 * 8
    prover.close(); pk.close(); params.close()


def test_generate_keys_with_min_k_downsizes_real_srs(shp, oracle, gpu_ctx):
    """generate_keys_with_min_k(circuit, full_params) (crates/shielder_bindings/build.rs:22-31): the reference's k = 11
    SRS downsized to a k = 9 plan; the proof equals the oracle's over the oracle-downsized SRS and passes both
    pairing checks (oracle + shp_verify)"""
    from conftest import golden_path
    PL = importlib.import_module("oracle.plonk")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    B = importlib.import_module("oracle.binding")
    raw = open(golden_path("ppot_0080_11_raw"), "rb").read()
    plan_bytes, lay = C.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
    full = shp.Params.read_raw(gpu_ctx, raw)
    params, k, pk = shp.ProvingKey.generate_keys_with_min_k(gpu_ctx, full, plan_bytes)
    assert (k, params.k, pk.k) == (9, 9, 9)
    ofull = oracle.srs_from_halo2_raw(raw)
    osrs = B.Srs(9, ofull.g[:512].copy(), oracle.g1_lagrange_from_monomial(ofull.g[:512], 9), ofull.g2, ofull.s_g2)
    opk = PL.keygen(PL.Plan(plan_bytes), osrs)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(21), 2**31 + 1, 2**20 + 2)
    prover = shp.Prover(pk)
    proof, inst = prover.prove(wit, bytes([3]) * 32)
    assert (proof, inst) == PL.prove(opk, osrs, wit, bytes([3]) * 32)
    assert PL.verify(opk, proof, inst, g2=osrs.g2, s_g2=osrs.s_g2)
    assert pk.verify(proof, inst)
    bad = bytearray(proof); bad[100] ^= 1
    assert not pk.verify(bytes(bad), inst)
    with pytest.raises(shp.ShpError):
        shp.ProvingKey.generate_keys_with_min_k(gpu_ctx, full, b"nope" + plan_bytes[4:])
    prover.close(); pk.close(); params.close(); full.close()


def test_pk_save_load_roundtrip(shp, oracle, gpu_ctx):
    """marshall_pk / unmarshall_pk: a key loaded from its saved form proves the same bytes; a saved key
    whose parts were altered does not load"""
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, True)
    blob = pk.save()
    assert blob[:8] == b"SHPK0003"
    pk2 = shp.ProvingKey.load(gpu_ctx, params, blob)
    assert pk2.vk_digest() == pk.vk_digest() and (pk2.k, pk2.n_inputs, pk2.proof_len) == (pk.k, pk.n_inputs, pk.proof_len)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(11), 2**31 + 9, 2**20 + 1)
    a, b = shp.Prover(pk), shp.Prover(pk2)
    assert a.prove(wit, bytes([9]) * 32) == b.prove(wit, bytes([9]) * 32)
    assert pk2.save() == blob
    for pos in (3, 40, len(blob) - 100, len(blob) - 1):
        bad = bytearray(blob); bad[pos] ^= 1
        with pytest.raises(shp.ShpError):
            shp.ProvingKey.load(gpu_ctx, params, bytes(bad))
    with pytest.raises(shp.ShpError):
        shp.ProvingKey.load(gpu_ctx, params, blob[:-5])
    a.close(); b.close(); pk2.close(); pk.close(); params.close()


def test_serve_end_to_end(shp, oracle, gpu_ctx):
    """TEE wire protocol (serve.py) over TCP on top of the native proving pool: the proof that comes back in
    `EncryptedProof` verifies and its pub_inputs JSON are the circuit's 8 public inputs in order"""
    import asyncio, base64, json
    S = importlib.import_module("zkos-monorepo_amd.serve")
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, True)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(21), 2**31 + 1, 2**20 + 2)
    le = lambda x: list(int(x).to_bytes(32, "little"))
    h = lay.height
    inputs = {"id": le(wit[lay.ID]), "nullifier_old": le(wit[lay.NULLIFIER_OLD]), "account_balance_old": le(wit[lay.BALANCE_OLD]),
              "token_address": le(wit[lay.TOKEN]), "path": [b for x in wit[4:4 + 7 * h] for b in le(x)], "value": le(wit[lay.VALUE]),
              "nullifier_new": le(wit[lay.NULLIFIER_NEW]), "commitment": le(wit[lay.COMMITMENT]), "mac_salt": le(wit[lay.MAC_SALT])}
    payload = {"circuit_type": "Withdraw", "user_public_key": [], "circuit_inputs": list(json.dumps(inputs).encode())}
    req = {"GenerateProof": {"payload": list(json.dumps(payload).encode())}}
    pool = shp.Pool(pk, 2)

    async def run():
        srv = await S.Server(S.PoolBackend(pool, h)).serve(port=0)
        port = srv.sockets[0].getsockname()[1]
        async with srv:
            assert await S.request("127.0.0.1", port, "Ping") == "Pong"
            return await asyncio.gather(*[S.request("127.0.0.1", port, req) for _ in range(3)])

    resps = asyncio.run(run())
    _, oinst = PL.synthesize(plan, wit)

    def check(resps):
        for r in resps:
            proof = base64.b64decode(r["EncryptedProof"]["proof"])
            pub = json.loads(base64.b64decode(r["EncryptedProof"]["pub_inputs"]))
            inst = [int.from_bytes(bytes(pub[f]), "little") for f in S.WITHDRAW_PUB_FIELDS]
            assert inst == list(oinst)
            assert PL.verify(opk, proof, inst, tau=TAU)
        assert len({r["EncryptedProof"]["proof"] for r in resps}) == len(resps)       # fresh randomness per request

    check(resps)
    # the same requests against the library's own TCP front (shp_server): no interpreter in the request path
    native = shp.NativeServer({"Withdraw": pool}, h)

    async def run_native():
        async def session():                                 # one connection, several requests in sequence
            reader, writer = await asyncio.open_connection("127.0.0.1", native.port)
            out = []
            for r in ("Ping", req, req):
                writer.write(S.encode_frame(r))
                await writer.drain()
                out.append(await S.read_frame(reader))
            writer.close()
            return out
        return await asyncio.gather(session(), session())

    sessions = asyncio.run(run_native())
    assert all(s[0] == "Pong" for s in sessions)
    check([r for s in sessions for r in s[1:]])
    assert native.stats() == (4, 0, 2)
    # a field element that is not canonical (vec_to_f panics on it in the reference): the connection ends, the server
    # keeps serving
    bad_inputs = dict(inputs, value=[255] * 32)
    bad_payload = {"circuit_type": "Withdraw", "user_public_key": [], "circuit_inputs": list(json.dumps(bad_inputs).encode())}

    async def run_bad():
        reader, writer = await asyncio.open_connection("127.0.0.1", native.port)
        writer.write(S.encode_frame({"GenerateProof": {"payload": list(json.dumps(bad_payload).encode())}}))
        await writer.drain()
        gone = await reader.read(4)
        writer.close()
        return gone, await S.request("127.0.0.1", native.port, "Ping")

    gone, pong = asyncio.run(run_bad())
    assert gone == b"" and pong == "Pong"
    native.close()
    pool.close(); pk.close(); params.close()


def test_deposit_shaped_plan_proof_bytes(shp, oracle, gpu_ctx):
    """the Deposit-shaped plan (DepositProverKnowledge argument / instance order) through the HIP prover"""
    from oracle import plonk as PL
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    plan_bytes, lay = C.build_deposit_plan(k=9, height=1, range_bits=4, range_chunks=8)
    osrs = oracle.srs_synthetic(9, TAU)
    opk = PL.keygen(PL.Plan(plan_bytes), osrs)
    params = shp.Params.from_parts(gpu_ctx, 9, osrs.g, osrs.g_lagrange)
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(4), 2**31 + 5, 2**20 + 3)
    pr = shp.Prover(pk)
    proof, inst = pr.prove(wit, bytes([3]) * 32)
    assert (proof, inst) == PL.prove(opk, osrs, wit, bytes([3]) * 32)
    assert inst[3:6] == [wit[lay.VALUE], wit[lay.COMMITMENT], wit[lay.TOKEN]]
    assert PL.verify(opk, proof, inst, tau=TAU)
    pr.close(); pk.close(); params.close()


def test_new_account_shaped_plan_proof_and_serve(shp, oracle, gpu_ctx):
    """the NewAccount-shaped plan (NewAccountProverKnowledge argument order, the contract's 13 public inputs;
    the ElGamal ciphertext replaced by documented Poseidon2 placeholders): proof bytes equal the oracle's, the public
    inputs mean what the layout says, and a NewAccount request in the TEE wire format is served end to end"""
    import asyncio, base64, json
    from oracle import plonk as PL
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    S = importlib.import_module("zkos-monorepo_amd.serve")
    plan_bytes, lay = C.build_withdraw_plan(9, 0, 4, 8, kind="new_account")
    osrs = oracle.srs_synthetic(9, TAU)
    opk = PL.keygen(PL.Plan(plan_bytes), osrs)
    params = shp.Params.synthetic(gpu_ctx, 9, TAU)
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    assert (pk.n_inputs, pk.n_instances) == (9, 13)
    assert pk.compiled_level() == 2, "the NewAccount-shaped plan is shipped with its generated quotient kernel"
    wit = C.new_account_witness_inputs(lay, random.Random(6), 2**31 + 9)
    pr = shp.Prover(pk)
    proof, inst = pr.prove(wit, bytes([8]) * 32)
    assert (proof, inst) == PL.prove(opk, osrs, wit, bytes([8]) * 32)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    idv, nul, dep, com, tok, salt, msalt, arx, ary = wit
    c1x, c1y = H([salt, arx]), H([salt, ary])
    assert inst == [H([idv, nul, dep, tok]), H([idv]), dep, com, tok, arx, ary, c1x, c1y, H([idv, c1x]), H([idv, c1y]), msalt,
                    H([msalt, idv])]
    assert PL.verify(opk, proof, inst, tau=TAU) and pk.verify(proof, inst)
    # over the wire
    le = lambda x: list(int(x).to_bytes(32, "little"))
    inputs = {f: le(v) for f, v in zip(S.NEW_ACCOUNT_INPUT_FIELDS, wit)}
    payload = {"circuit_type": "NewAccount", "user_public_key": [], "circuit_inputs": list(json.dumps(inputs).encode())}
    req = {"GenerateProof": {"payload": list(json.dumps(payload).encode())}}
    pool = shp.Pool(pk, 2)

    async def run():
        srv = await S.Server({"NewAccount": S.PoolBackend(pool, 0)}).serve(port=0)
        port = srv.sockets[0].getsockname()[1]
        async with srv:
            return await S.request("127.0.0.1", port, req)

    r = asyncio.run(run())
    proof2 = base64.b64decode(r["EncryptedProof"]["proof"])
    pub = json.loads(base64.b64decode(r["EncryptedProof"]["pub_inputs"]))
    assert list(pub) == list(S.NEW_ACCOUNT_PUB_FIELDS)
    assert [int.from_bytes(bytes(pub[f]), "little") for f in S.NEW_ACCOUNT_PUB_FIELDS] == inst
    assert pk.verify(proof2, inst)
    pool.close(); pr.close(); pk.close(); params.close()


@pytest.mark.parametrize("small", [True, False])
def test_witness_schedule_threads_give_identical_proofs(shp, oracle, gpu_ctx, small):
    """the witness program cut into parallel tasks (1, 2, 4, 7 host threads) yields the same proof bytes as
    the sequential program and as the oracle"""
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, small)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    bal, val = (2**31 + 5, 2**20 + 3) if small else (10**18, 10**17)
    wit = C.withdraw_witness_inputs(lay, H, random.Random(8), bal, val)
    pr = shp.Prover(pk)
    want = PL.prove(opk, osrs, wit, bytes([5]) * 32)
    for threads in (1, 2, 4, 7, 4):
        pr.witness_threads(threads)
        for _ in range(3):
            assert pr.prove(wit, bytes([5]) * 32) == want, f"{threads} witness threads"
    for throughput in (True, False):                     # shp_prover_mode: scheduling only, same bytes
        pr.mode(throughput)
        assert pr.prove(wit, bytes([5]) * 32) == want, f"mode throughput={throughput}"
    pr.close(); pk.close(); params.close()


def test_pool_with_msm_combiner_same_bytes(shp, oracle, gpu_ctx, monkeypatch):
    """SHP_POOL_COMBINE=1: the commitments of all lanes waiting at an MSM phase are served by one multi-job launch
    (msm_run_jobs through MsmCombiner); proofs stay byte-identical to a lone lane and to the oracle"""
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, True)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    rnd = random.Random(99)
    count = 24
    wits = [C.withdraw_witness_inputs(lay, H, rnd, 2**31 + i, 2**20 + i) for i in range(count)]
    blobs = [b"".join(int(x).to_bytes(32, "little") for x in w) for w in wits]
    seeds = [bytes([i + 7]) * 32 for i in range(count)]
    monkeypatch.setenv("SHP_POOL_COMBINE", "1")
    pool = shp.Pool(pk, 6)
    monkeypatch.delenv("SHP_POOL_COMBINE")
    got = pool.prove_batch(b"".join(blobs), b"".join(seeds), count)
    single = shp.Prover(pk)
    for i in range(count):
        assert got[i] == single.prove_bytes(blobs[i], seeds[i]), f"combined-MSM proof {i} differs"
    assert got[5] == PL.prove(opk, osrs, wits[5], seeds[5])
    pool.close(); single.close(); pk.close(); params.close()


def test_pool_queue_bound_and_argument_errors(shp, oracle, gpu_ctx):
    """bounded queue (the reference's task pool answers 504 when it is full): SHP_ERR_BUSY, then the pool drains;
    wrong witness length -> SHP_ERR_INVALID_ARG"""
    PL, C, plan, lay, osrs, opk, params, pk = setup(shp, oracle, gpu_ctx, True)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(2), 2**31 + 7, 2**20 + 9)
    blob = b"".join(int(x).to_bytes(32, "little") for x in wit)
    pool = shp.Pool(pk, 1, queue_cap=1)
    tickets, busy = [], 0
    for i in range(6):
        try:
            tickets.append(pool.submit(blob, bytes([i + 1]) * 32))
        except shp.ShpError as e:
            assert e.code == 7
            busy += 1
    assert busy >= 1 and len(tickets) >= 1               # how many fit depends on when the lane picks the first up
    for t in tickets:
        proof, inst = pool.wait(t)
        assert PL.verify(opk, proof, inst, tau=TAU)
    with pytest.raises(shp.ShpError) as e:
        pool.submit(blob[:-32], bytes(32))
    assert e.value.code == 1
    with pytest.raises(shp.ShpError):
        pool.wait(10**9)                            # unknown ticket
    single = shp.Prover(pk)
    with pytest.raises(shp.ShpError) as e:
        single.prove_bytes(blob + bytes(32), bytes(32))
    assert e.value.code == 1
    single.close(); pool.close(); pk.close(); params.close()


def test_native_verifier_matches_oracle(shp, oracle, gpu_ctx):
    """shp_verify (Circuit::verify, crates/shielder_bindings/src/circuits/mod.rs:113-124): accepts what the oracle's
    restated verifier accepts and rejects what it rejects -- valid proof, every kind of tampering, wrong public
    input, wrong length -- over a synthetic SRS (G2 half derived from tau) and over the reference's k = 11 SRS"""
    from conftest import golden_path
    PL = importlib.import_module("oracle.plonk")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    # synthetic SRS through the library (shp_srs_synthetic computes s_g2 = tau g2)
    plan_bytes, lay = C.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
    params = shp.Params.synthetic(gpu_ctx, 9, TAU)
    osrs = oracle.srs_synthetic(9, TAU)
    from oracle import pairing as PA
    g2, s_g2 = params.g2()
    assert PA.g2_from_bytes(s_g2) == PA.g2_mul(PA.g2_from_bytes(g2), TAU)
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    opk = PL.keygen(PL.Plan(plan_bytes), osrs)
    wit = C.withdraw_witness_inputs(lay, H, random.Random(31), 2**31 + 5, 2**20 + 3)
    pr = shp.Prover(pk)
    proof, inst = pr.prove(wit, bytes([7]) * 32)
    assert pk.verify(proof, inst) and PL.verify(opk, proof, inst, tau=TAU) and PL.verify(opk, proof, inst, g2=g2, s_g2=s_g2)
    assert pk.public_inputs(wit) == inst                     # *_pub_inputs without proving
    rnd = random.Random(1)
    positions = [0, 63, 64 * 10 + 5, 64 * 15, len(proof) - 129, len(proof) - 128, len(proof) - 1] + [rnd.randrange(len(proof)) for _ in range(8)]
    for pos in positions:
        bad = bytearray(proof); bad[pos] ^= 1 << rnd.randrange(8)
        assert pk.verify(bytes(bad), inst) == PL.verify(opk, bytes(bad), inst, tau=TAU) == False, f"tampered byte {pos}"
    for i in range(len(inst)):
        wrong = list(inst); wrong[i] = (wrong[i] + 1) % R
        assert not pk.verify(proof, wrong)
    assert not pk.verify(proof[:-1], inst) and not pk.verify(proof + b"\x00", inst) and not pk.verify(proof, inst[:-1])
    pr.close(); pk.close(); params.close()
    # the reference's own SRS (deposit-shaped plan, config 1)
    raw = open(golden_path("ppot_0080_11_raw"), "rb").read()
    plan_bytes, lay = C.build_deposit_plan(k=11, height=3, range_bits=8, range_chunks=10)
    params = shp.Params.read_raw(gpu_ctx, raw)
    pk = shp.ProvingKey(gpu_ctx, params, plan_bytes)
    wit = C.withdraw_witness_inputs(lay, H, random.Random(3), 2**31 + 5, 2**20 + 3)
    pr = shp.Prover(pk)
    proof, inst = pr.prove(wit, bytes(range(32)))
    assert pk.verify(proof, inst)
    bad = bytearray(proof); bad[200] ^= 2
    assert not pk.verify(bytes(bad), inst)
    pr.close(); pk.close(); params.close()
