#!/usr/bin/env python3
"""Soak: N proofs through a 16-lane pool, EVERY proof checked by the native verifier (pairing included) and the public
inputs against shp_public_inputs.   python tools/soak.py [--proofs 20000] [--lanes 16] [--serve | --lone]
--serve: the proofs are requested over TCP in the TEE wire format from the library's own request loop (shp_server), one
connection per lane, and checked the same way.
--lone: the proofs come one after the other from ONE latency-mode prover (its two streams, the lookups ahead of the witness,
the transforms beside the commitments), with a malformed witness every 997th proof (the prover must refuse it and go on),
checked the same way on helper threads."""
import argparse, importlib, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shp = importlib.import_module("zkos-monorepo_amd")
shp.set_hw_queues(int(os.environ.get("SHP_BENCH_HW_QUEUES", "22")))
circuits = importlib.import_module("zkos-monorepo_amd.circuits")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=20000)
    ap.add_argument("--lanes", type=int, default=16)
    ap.add_argument("--serve", action="store_true")
    ap.add_argument("--lone", action="store_true")
    a = ap.parse_args()
    shp.prefer_sleeping_waits(a.lanes)
    ctx = shp.Context(0)
    plan, lay = circuits.build_withdraw_plan()
    params = shp.Params.synthetic(ctx, 13, 0xDEADBEEF12345678)
    pk = shp.ProvingKey(ctx, params, plan)
    pool = None if a.lone else shp.Pool(pk, a.lanes, queue_cap=4 * a.lanes)
    rnd = random.Random(9)
    wits = [circuits.withdraw_witness_inputs(lay, shp.poseidon2_hash_host, rnd) for _ in range(16)]
    blobs = [b"".join(int(x).to_bytes(32, "little") for x in w) for w in wits]
    pubs = [pk.public_inputs(w) for w in wits]
    bad = [0]

    def one(i):
        t = None
        while t is None:
            try:
                t = pool.submit(blobs[i % 16], i.to_bytes(32, "little"))
            except shp.ShpError as e:
                if e.code != 7:
                    raise
                time.sleep(0.001)
        proof, inst = pool.wait(t)
        if inst != pubs[i % 16] or not pk.verify(proof, inst):
            bad[0] += 1

    if a.lone:
        prover = shp.Prover(pk)
        broken_blob = blobs[0][:32 * lay.VALUE] + b"\xff" * 32 + blobs[0][32 * lay.VALUE + 32:]   # not a canonical field element
        refused = 0

        def check(job):
            i, proof, inst = job
            if inst != pubs[i % 16] or not pk.verify(proof, inst):
                bad[0] += 1

        t0 = time.time()
        with ThreadPoolExecutor(max_workers=12) as ex:
            pending = []
            for i in range(a.proofs):
                if i % 997 == 996:
                    try:
                        prover.prove_bytes(broken_blob, i.to_bytes(32, "little"))
                        bad[0] += 1                           # must not come back with a proof
                    except shp.ShpError:
                        refused += 1
                proof, inst = prover.prove_bytes(blobs[i % 16], i.to_bytes(32, "little"))
                pending.append(ex.submit(check, (i, proof, inst)))
                if len(pending) > 4096:
                    for f in pending[:2048]:
                        f.result()
                    del pending[:2048]
            for f in pending:
                f.result()
        el = time.time() - t0
        print(f"soak of one latency-mode prover: {a.proofs} proofs in {el:.1f} s ({a.proofs / el:.0f}/s incl. verification), "
              f"{refused} broken witnesses refused in between, {bad[0]} bad")
        prover.close(); pk.close(); params.close()
        sys.exit(1 if bad[0] else 0)
    if a.serve:
        import base64, json, socket, struct
        S = importlib.import_module("zkos-monorepo_amd.serve")
        le = lambda x: list(int(x).to_bytes(32, "little"))
        h = lay.height
        frames = []
        for wit in wits:
            inputs = {"id": le(wit[lay.ID]), "nullifier_old": le(wit[lay.NULLIFIER_OLD]), "account_balance_old": le(wit[lay.BALANCE_OLD]),
                      "token_address": le(wit[lay.TOKEN]), "path": [b for x in wit[4:4 + 7 * h] for b in le(x)], "value": le(wit[lay.VALUE]),
                      "nullifier_new": le(wit[lay.NULLIFIER_NEW]), "commitment": le(wit[lay.COMMITMENT]), "mac_salt": le(wit[lay.MAC_SALT])}
            payload = {"circuit_type": "Withdraw", "user_public_key": [], "circuit_inputs": list(json.dumps(inputs).encode())}
            frames.append(S.encode_frame({"GenerateProof": {"payload": list(json.dumps(payload).encode())}}))
        server = shp.NativeServer({"Withdraw": pool}, h)

        def connection(c):                                   # requests c, c + lanes, c + 2 lanes, ... on one connection
            sock = socket.create_connection(("127.0.0.1", server.port), timeout=120)
            for i in range(c, a.proofs, a.lanes):
                sock.sendall(frames[i % 16])
                hdr = b""
                while len(hdr) < 4:
                    hdr += sock.recv(4 - len(hdr))
                (n,) = struct.unpack(">I", hdr)
                body = bytearray()
                while len(body) < n:
                    body += sock.recv(n - len(body))
                r = json.loads(bytes(body))["EncryptedProof"]
                pub = json.loads(base64.b64decode(r["pub_inputs"]))
                inst = [int.from_bytes(bytes(pub[f]), "little") for f in S.WITHDRAW_PUB_FIELDS]
                if inst != pubs[i % 16] or not pk.verify(base64.b64decode(r["proof"]), inst):
                    bad[0] += 1
            sock.close()

        t0 = time.time()
        with ThreadPoolExecutor(max_workers=a.lanes) as ex:
            list(ex.map(connection, range(a.lanes)))
        el = time.time() - t0
        served, failed, conns = server.stats()
        server.close()
        if (served, failed, conns) != (a.proofs, 0, a.lanes):
            bad[0] += 1
        print(f"soak over TCP (shp_server): {a.proofs} requests on {a.lanes} connections in {el:.1f} s "
              f"({a.proofs / el:.0f}/s incl. verification), server counted {served} served / {failed} failed, {bad[0]} bad")
        pool.close(); pk.close(); params.close()
        sys.exit(1 if bad[0] else 0)
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=3 * a.lanes) as ex:
        list(ex.map(one, range(a.proofs)))
    el = time.time() - t0
    print(f"soak: {a.proofs} proofs in {el:.1f} s ({a.proofs / el:.0f}/s incl. verification), {bad[0]} bad")
    pool.close(); pk.close(); params.close()
    sys.exit(1 if bad[0] else 0)


if __name__ == "__main__":
    main()
