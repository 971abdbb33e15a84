// A prover server with no Python in it: the reference's enclave server (tee/crates/shielder-prover-tee/src/main.rs,
// server.rs) reduced to what the C ABI of include/shielder_hip.h provides -- build the SRS, generate a key per circuit
// plan, give each key a proving pool, hand the pools to the library's request loop (shp_server_start) and wait.
//
//   shp_serve <k> <tau hex | ptau file> <lanes> <port> <merkle height> <CircuitType>=<plan.bin> [<CircuitType>=<plan.bin> ...]
// CircuitType = NewAccount | Deposit | Withdraw.  The second argument is a snarkjs .ptau file (a ceremony SRS) when a
// file of that name exists, else the trapdoor of a synthetic SRS (tests and load generation only: whoever knows it can
// forge proofs).  Prints one JSON line with the port it listens on, then serves until stdin closes or SIGTERM arrives.
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../include/shielder_hip.h"

static std::vector<uint8_t> slurp(const char* path, bool* ok = nullptr) {
    std::ifstream f(path, std::ios::binary);
    if (ok) *ok = (bool)f;
    if (!f) return {};
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int32_t rc_ = (call);                                                        \
        if (rc_ != SHP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, shp_last_error()); return 1; } \
    } while (0)

static volatile sig_atomic_t g_stop = 0;
static void on_signal(int) { g_stop = 1; }

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s k tau_hex|file.ptau lanes port merkle_height CircuitType=plan.bin ...\n", argv[0]);
        return 2;
    }
    const uint32_t k = (uint32_t)atoi(argv[1]), lanes = (uint32_t)atoi(argv[3]), height = (uint32_t)atoi(argv[5]);
    const uint16_t port = (uint16_t)atoi(argv[4]);
    if (lanes > 16) setenv("SHP_BLOCKING_SYNC", "1", 0);          // more lanes than a GPU's usual share of host cores: waits sleep
    shp_ctx* ctx = nullptr;
    shp_srs* srs = nullptr;
    CHECK(shp_ctx_create(0, &ctx));
    bool is_file = false;
    std::vector<uint8_t> ptau = slurp(argv[2], &is_file);
    if (is_file) {
        CHECK(shp_srs_from_ptau(ctx, ptau.data(), ptau.size(), &srs));
    } else {
        uint8_t tau[32] = {0};
        const char* h = argv[2];
        const size_t n = strlen(h);
        for (size_t i = 0; i < n && i < 64; i++) {                // hex, most significant digit first
            const char c = h[n - 1 - i];
            const uint8_t v = c >= 'a' ? c - 'a' + 10 : c >= 'A' ? c - 'A' + 10 : c - '0';
            tau[i / 2] |= (uint8_t)(v << (4 * (i & 1)));
        }
        CHECK(shp_srs_synthetic(ctx, k, tau, &srs));
    }
    static const char* const NAMES[3] = {"NewAccount", "Deposit", "Withdraw"};   // CircuitType (protocol.rs:62-68)
    std::vector<shp_pk*> pks;
    std::vector<shp_pool*> pools;
    std::vector<int32_t> kinds;
    for (int a = 6; a < argc; a++) {
        const char* eq = strchr(argv[a], '=');
        int32_t kind = -1;
        for (int t = 0; t < 3 && eq; t++)
            if (strlen(NAMES[t]) == (size_t)(eq - argv[a]) && strncmp(argv[a], NAMES[t], eq - argv[a]) == 0) kind = t;
        if (kind < 0) { fprintf(stderr, "bad argument %s\n", argv[a]); return 2; }
        bool ok = false;
        const std::vector<uint8_t> plan = slurp(eq + 1, &ok);
        if (!ok) { fprintf(stderr, "cannot open %s\n", eq + 1); return 2; }
        shp_pk* pk = nullptr;
        shp_pool* pool = nullptr;
        CHECK(shp_keygen(ctx, srs, plan.data(), plan.size(), &pk));
        CHECK(shp_pool_create(pk, lanes, 0, &pool));
        pks.push_back(pk);
        pools.push_back(pool);
        kinds.push_back(kind);
    }
    shp_server* server = nullptr;
    CHECK(shp_server_start(pools.data(), kinds.data(), (uint32_t)pools.size(), height, "127.0.0.1", port, &server));
    uint16_t bound = 0;
    CHECK(shp_server_port(server, &bound));
    printf("{\"listening\": %u, \"lanes\": %u, \"circuits\": %zu, \"version\": \"%s\"}\n", bound, lanes, pools.size(), shp_version());
    fflush(stdout);
    signal(SIGTERM, on_signal);
    signal(SIGINT, on_signal);
    char c;
    while (!g_stop && read(0, &c, 1) > 0) {}                      // until stdin closes (the parent went away) or a signal
    uint64_t served = 0, failed = 0, conns = 0;
    CHECK(shp_server_stats(server, &served, &failed, &conns));
    CHECK(shp_server_stop(server));
    for (shp_pool* p : pools) CHECK(shp_pool_destroy(p));
    for (shp_pk* p : pks) CHECK(shp_pk_destroy(p));
    CHECK(shp_srs_destroy(srs));
    CHECK(shp_ctx_destroy(ctx));
    printf("{\"served\": %llu, \"failed\": %llu, \"connections\": %llu}\n", (unsigned long long)served,
           (unsigned long long)failed, (unsigned long long)conns);
    return 0;
}
