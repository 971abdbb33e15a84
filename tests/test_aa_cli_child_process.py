"""GPU: examples/shp_cli.cpp -- the C ABI driven by a plain C++ host program with no Python and no torch in it -- run as a
child process, its proofs compared byte for byte with the CPU oracle's.

This file sorts first on purpose: the child must be started by a process that has not initialised the GPU itself (a
process that has may not start another program on this pool), so the test runs before any other GPU test creates a
context, and skips itself if one exists already."""
import importlib
import json
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAU = 0x1234567890ABCDEF1234567890ABCDEF55


def test_cpp_host_over_the_c_abi_matches_oracle(oracle, tmp_path):
    b = sys.modules.get("zkos-monorepo_amd.binding")
    if b is not None and b.Context.created:
        pytest.skip("this process has already initialised the GPU: it must not start another program")
    cli = os.path.join(ROOT, "zkos-monorepo_amd", "shp_cli")
    assert os.path.exists(cli), "zkos-monorepo_amd/shp_cli is not built (make -C zkos-monorepo_amd/csrc)"
    PL = importlib.import_module("oracle.plonk")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    plan_bytes, lay = C.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    rnd = random.Random(99)
    count = 5
    wits = [C.withdraw_witness_inputs(lay, H, rnd, 2**31 + i, 2**20 + i) for i in range(count)]
    (tmp_path / "plan.bin").write_bytes(plan_bytes)
    (tmp_path / "witnesses.bin").write_bytes(b"".join(int(x).to_bytes(32, "little") for w in wits for x in w))
    out = subprocess.run([cli, str(tmp_path / "plan.bin"), str(tmp_path / "witnesses.bin"), "9", "%x" % TAU, "3",
                          str(tmp_path / "proofs.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["proofs"] == count and j["accepted"] == count and j["tampered_rejected"] is True
    plan = PL.Plan(plan_bytes)
    srs = oracle.srs_synthetic(9, TAU)
    opk = PL.keygen(plan, srs)
    got = (tmp_path / "proofs.bin").read_bytes()
    plen = PL.proof_layout(plan)[2]
    assert len(got) == plen * count and j["proof_len"] == plen
    for i, w in enumerate(wits):
        want, inst = PL.prove(opk, srs, w, bytes([i + 1]) * 32)        # the CLI seeds proof i with byte i + 1
        assert got[i * plen:(i + 1) * plen] == want, f"proof {i} from the C++ host differs from the oracle"
        assert PL.verify(opk, want, inst, tau=TAU)


def test_cpp_prover_server_answers_a_withdraw_request(oracle, tmp_path):
    """examples/shp_serve.cpp -- SRS, key, pool and the library's request loop (shp_server_start) in a C++ process with no
    Python in it -- started as a child: a Withdraw request in the TEE wire format over TCP comes back as an EncryptedProof
    whose proof the restated verifier accepts for the circuit's public inputs; the server reports what it served"""
    import base64
    import socket
    import struct
    b = sys.modules.get("zkos-monorepo_amd.binding")
    if b is not None and b.Context.created:
        pytest.skip("this process has already initialised the GPU: it must not start another program")
    exe = os.path.join(ROOT, "zkos-monorepo_amd", "shp_serve")
    assert os.path.exists(exe), "zkos-monorepo_amd/shp_serve is not built (make -C zkos-monorepo_amd/csrc)"
    PL = importlib.import_module("oracle.plonk")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    S = importlib.import_module("zkos-monorepo_amd.serve")
    plan_bytes, lay = C.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    wit = C.withdraw_witness_inputs(lay, H, random.Random(31), 2**31 + 7, 2**20 + 9)
    (tmp_path / "plan.bin").write_bytes(plan_bytes)
    le = lambda x: list(int(x).to_bytes(32, "little"))
    h = lay.height
    inputs = {"id": le(wit[lay.ID]), "nullifier_old": le(wit[lay.NULLIFIER_OLD]), "account_balance_old": le(wit[lay.BALANCE_OLD]),
              "token_address": le(wit[lay.TOKEN]), "path": [b for x in wit[4:4 + 7 * h] for b in le(x)], "value": le(wit[lay.VALUE]),
              "nullifier_new": le(wit[lay.NULLIFIER_NEW]), "commitment": le(wit[lay.COMMITMENT]), "mac_salt": le(wit[lay.MAC_SALT])}
    payload = {"circuit_type": "Withdraw", "user_public_key": [], "circuit_inputs": list(json.dumps(inputs).encode())}
    frame = S.encode_frame({"GenerateProof": {"payload": list(json.dumps(payload).encode())}})
    proc = subprocess.Popen([exe, "9", "%x" % TAU, "2", "0", str(h), "Withdraw=" + str(tmp_path / "plan.bin")],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        hello = json.loads(proc.stdout.readline())
        port = hello["listening"]
        assert port > 0 and hello["circuits"] == 1

        def exchange(sock, body_frame):
            sock.sendall(body_frame)
            hdr = b""
            while len(hdr) < 4:
                chunk = sock.recv(4 - len(hdr))
                assert chunk, "the server closed the connection"
                hdr += chunk
            (n,) = struct.unpack(">I", hdr)
            body = b""
            while len(body) < n:
                body += sock.recv(n - len(body))
            return json.loads(body)

        with socket.create_connection(("127.0.0.1", port), timeout=60) as s:
            assert exchange(s, S.encode_frame("Ping")) == "Pong"
            resps = [exchange(s, frame) for _ in range(2)]
    finally:
        proc.stdin.close()                                         # the server runs until its stdin closes
        out, err = proc.stdout.read(), proc.stderr.read()
        proc.wait(timeout=60)
    assert proc.returncode == 0, err[-2000:]
    assert json.loads(out.strip().splitlines()[-1]) == {"served": 2, "failed": 0, "connections": 1}
    plan = PL.Plan(plan_bytes)
    opk = PL.keygen(plan, oracle.srs_synthetic(9, TAU))
    _, oinst = PL.synthesize(plan, wit)
    for r in resps:
        proof = base64.b64decode(r["EncryptedProof"]["proof"])
        pub = json.loads(base64.b64decode(r["EncryptedProof"]["pub_inputs"]))
        inst = [int.from_bytes(bytes(pub[f]), "little") for f in S.WITHDRAW_PUB_FIELDS]
        assert inst == list(oinst)
        assert PL.verify(opk, proof, inst, tau=TAU)
    assert resps[0]["EncryptedProof"]["proof"] != resps[1]["EncryptedProof"]["proof"]      # fresh randomness per request


@pytest.mark.parametrize("batch", [0, 64])
def test_bench_two_ranks_with_real_pools(batch):
    """bench.py's multi-rank path with REAL proving pools: two ranks (one process each, gloo barrier, no data-path collective),
    both on the one GPU of a test box -- the launch the driver makes on an 8-GPU node, at world size 2.  Weak scaling (every
    rank proves lanes x proofs-per-lane per step) and BASELINE config 3's shape (--batch: one fixed batch cut into contiguous
    shards).  One JSON line from rank 0, the proofs of every rank's last step verified and counted over gloo."""
    b = sys.modules.get("zkos-monorepo_amd.binding")
    if b is not None and b.Context.created:
        pytest.skip("this process has already initialised the GPU: it must not start another program")
    env = dict(os.environ, SHP_MSM_TABLE_GB="20")                   # two replicas of the tables on one GPU
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lanes", "8",
           "--proofs-per-lane", "4", "--latency-proofs", "8", "--verify-proofs", "6", "--cpu-proofs", "0", "--micro"]
    if batch:
        cmd += ["--batch", str(batch)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["unit"] == "proofs/s" and j["value"] > 0
    if batch:
        assert j["scaling"] == "strong" and j["config"]["proofs_per_step"] == batch and j["config"]["proofs_per_step_rank0"] == batch // 2
    else:
        assert j["scaling"] == "weak" and j["config"]["proofs_per_step"] == 2 * 8 * 4
    assert j["verified"] == 12                                      # 6 on each rank, summed over gloo
    assert abs(j["value"] - j["config"]["proofs_per_step"] * 2 / (2 * j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert j["config"]["compiled_level"] == 2 and "micro" not in j and "cpu_baseline" not in j
    assert j["setup_s"]["srs_and_tables"] > 0 and j["p50_latency_ms"] > 0


def test_ntt_in_29_bit_lazy_arithmetic_matches_oracle():
    """SHP_NTT29=1 selects the transform's other tile kernel (nine 29-bit limbs, lazy reduction, decimation in time; the
    switch is read once per process, hence the child): the NTT, coset-extension and no-allocation parity tests and a whole
    proof run again under it -- every result must stay bit-identical to the oracle's"""
    b = sys.modules.get("zkos-monorepo_amd.binding")
    if b is not None and b.Context.created:
        pytest.skip("this process has already initialised the GPU: it must not start another program")
    env = dict(os.environ, SHP_NTT29="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_gpu_primitives.py"), os.path.join(ROOT, "tests", "test_zz_no_alloc.py"),
                          os.path.join(ROOT, "tests", "test_gpu_prover.py"),
                          "-k", "ntt_matches_oracle or three_pass or coset or noalloc or (proof_bytes and True)"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
