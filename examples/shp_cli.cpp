// A host program with no Python and no torch in it: the C ABI of include/shielder_hip.h used the way
// the reference's Rust front-ends would use it -- load a circuit plan, build the SRS
// and the key, prove a batch through the proving pool, verify every proof, print one JSON line.
//
//   shp_cli <plan.bin> <witnesses.bin> <k> <tau hex> <lanes> [proofs_out.bin]
// witnesses.bin = N witnesses back to back (n_inputs x 32-byte little-endian canonical each).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../include/shielder_hip.h"

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int32_t rc_ = (call);                                                        \
        if (rc_ != SHP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, shp_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s plan.bin witnesses.bin k tau_hex lanes [proofs_out.bin]\n", argv[0]); return 2; }
    std::vector<uint8_t> plan = slurp(argv[1]), wits = slurp(argv[2]);
    const uint32_t k = (uint32_t)atoi(argv[3]);
    uint8_t tau[32] = {0};
    {
        const char* h = argv[4];
        size_t n = strlen(h);
        for (size_t i = 0; i < n && i < 64; i++) {                // hex, most significant digit first
            char c = h[n - 1 - i];
            uint8_t v = c >= 'a' ? c - 'a' + 10 : c >= 'A' ? c - 'A' + 10 : c - '0';
            tau[i / 2] |= (uint8_t)(v << (4 * (i & 1)));
        }
    }
    const uint32_t lanes = (uint32_t)atoi(argv[5]);
    shp_ctx* ctx = nullptr;
    shp_srs* srs = nullptr;
    shp_pk* pk = nullptr;
    shp_pool* pool = nullptr;
    CHECK(shp_ctx_create(0, &ctx));
    CHECK(shp_srs_synthetic(ctx, k, tau, &srs));
    CHECK(shp_keygen(ctx, srs, plan.data(), plan.size(), &pk));
    uint32_t kk, n_inputs, n_inst, proof_len;
    CHECK(shp_pk_info(pk, &kk, &n_inputs, &n_inst, &proof_len));
    if (wits.size() % ((size_t)n_inputs * 32)) { fprintf(stderr, "witness file is not a multiple of %u inputs\n", n_inputs); return 2; }
    const size_t count = wits.size() / ((size_t)n_inputs * 32);
    CHECK(shp_pool_create(pk, lanes, 0, &pool));
    std::vector<uint8_t> seeds(32 * count), proofs((size_t)proof_len * count), insts((size_t)32 * n_inst * count + 1);
    for (size_t i = 0; i < count; i++) memset(seeds.data() + 32 * i, (int)(i + 1), 32);
    std::vector<size_t> lens(count);
    auto t0 = std::chrono::steady_clock::now();
    CHECK(shp_prove_batch(pool, wits.data(), n_inputs, count, seeds.data(), proofs.data(), proof_len, lens.data(), insts.data(),
                          (size_t)32 * n_inst));
    const double prove_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t accepted = 0;
    for (size_t i = 0; i < count; i++) {
        int32_t ok = 0;
        CHECK(shp_verify(pk, proofs.data() + i * proof_len, lens[i], insts.data() + i * 32 * n_inst, n_inst, &ok));
        accepted += ok ? 1 : 0;
    }
    // a flipped bit must be rejected
    int32_t ok = 1;
    proofs[70] ^= 1;
    CHECK(shp_verify(pk, proofs.data(), lens[0], insts.data(), n_inst, &ok));
    proofs[70] ^= 1;
    size_t cd_len = 0;
    CHECK(shp_encode_calldata(proofs.data(), lens[0], insts.data(), n_inst, nullptr, 0, &cd_len));
    if (argc > 6) {
        std::ofstream o(argv[6], std::ios::binary);
        o.write((const char*)proofs.data(), (std::streamsize)proofs.size());
    }
    printf("{\"proofs\": %zu, \"accepted\": %zu, \"tampered_rejected\": %s, \"proof_len\": %u, \"calldata_len\": %zu, "
           "\"prove_seconds\": %.6f, \"lanes\": %u, \"version\": \"%s\"}\n",
           count, accepted, ok ? "false" : "true", proof_len, cd_len, prove_s, lanes, shp_version());
    CHECK(shp_pool_destroy(pool));
    CHECK(shp_pk_destroy(pk));
    CHECK(shp_srs_destroy(srs));
    CHECK(shp_ctx_destroy(ctx));
    return accepted == count && !ok ? 0 : 1;
}
