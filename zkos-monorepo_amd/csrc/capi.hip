// C-ABI entry points of libshielder_hip that are not owned by a kernel file: library/device
// management, the field front door, the NTT entry points and the SRS ("Params") object.
#include <string.h>

#include <stdlib.h>
#include "engine.hpp"
#include "hostcrypto.hpp"
#include "verify.hpp"

namespace shp {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

// ------------------------------------------------------------------ group FFT (SRS ingest)
// g_lagrange = n^-1 * FFT_{w^-1}(g): ParamsKZG::from_parts, reached from
// /root/reference/crates/powers-of-tau/lib.rs:71.  Radix-2 decimation in time over XYZZ points,
// one launch per stage; every butterfly is one 254-bit double-and-add by the stage twiddle.
__device__ G1X g1_mul_raw(const G1X& p, const Fr& k) {
    G1X acc = G1X::inf();
    bool started = false;
    for (int i = 253; i >= 0; i--) {
        if (started) acc = g1_dbl(acc);
        if ((k.l[i >> 5] >> (i & 31)) & 1) {
            acc = started ? g1_add(acc, p) : p;
            started = true;
        }
    }
    return acc;
}

__global__ void g1fft_load_kernel(const G1Affine* g, G1X* a, uint32_t k) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << k)) return;
    uint32_t j = k ? (__brev(i) >> (32 - k)) : 0;
    a[j] = G1X::from_affine(g[i]);
}

__global__ void g1fft_stage_kernel(G1X* a, uint32_t k, uint32_t s, const Fr* tw_raw) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 1u << k;
    if (b >= n / 2) return;
    uint32_t half = 1u << (s - 1);
    uint32_t grp = b / half, j = b % half;
    uint32_t lo = grp * 2 * half + j, hi = lo + half;
    G1X u = a[lo];
    G1X t = a[hi];
    if (j) t = g1_mul_raw(t, tw_raw[(size_t)j * (n >> s)]);
    a[lo] = g1_add(u, t);
    a[hi] = g1_add(u, t.negated());
}

__global__ void g1fft_finish_kernel(const G1X* a, G1Affine* out, uint32_t n, Fr n_inv_raw) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = g1_to_affine(g1_mul_raw(a[i], n_inv_raw));
}

static void g1_lagrange_from_monomial(shp_ctx* ctx, const G1Affine* d_g, G1Affine* d_out, uint32_t k) {
    const uint32_t n = 1u << k;
    Fr w = fr_root_of_unity(k);
    Fr wi = Fr::inv(w);
    std::vector<Fr> tw(n / 2 ? n / 2 : 1);
    Fr acc = Fr::one();
    for (uint32_t i = 0; i < n / 2; i++) {
        tw[i] = acc.to_raw();
        acc = acc * wi;
    }
    Fr* d_tw = nullptr;
    G1X* d_a = nullptr;
    SHP_HIP(hipMalloc(&d_tw, tw.size() * sizeof(Fr)));
    SHP_HIP(hipMalloc(&d_a, (size_t)n * sizeof(G1X)));
    SHP_HIP(hipMemcpyAsync(d_tw, tw.data(), tw.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(g1fft_load_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, ctx->stream, d_g, d_a, k);
    for (uint32_t s = 1; s <= k; s++)
        hipLaunchKernelGGL(g1fft_stage_kernel, dim3(ceil_div(n / 2, 64)), dim3(64), 0, ctx->stream, d_a, k, s, d_tw);
    Fr n_inv = Fr::inv(Fr::from_u64((uint64_t)n)).to_raw();
    hipLaunchKernelGGL(g1fft_finish_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, d_a, d_out, n, n_inv);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tw);
    (void)hipFree(d_a);
    SHP_HIP(e);
    SHP_HIP(e2);
}

// synthetic SRS: g[i] = tau^i * G, one lane per power (254-bit double-and-add on the generator)
__global__ void srs_powers_kernel(const Fr* scalars_raw, G1Affine* out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1X g;
    g.x = Fq::from_u64(1); g.y = Fq::from_u64(2); g.zz = Fq::one(); g.zzz = Fq::one();
    out[i] = g1_to_affine(g1_mul_raw(g, scalars_raw[i]));
}

static bool fq_canonical(const Fq& v) { return !v.geq_p(); }
static bool on_curve(const G1Affine& p) {
    if (p.is_inf()) return true;
    Fq lhs = Fq::sqr(p.y);
    Fq rhs = Fq::sqr(p.x) * p.x + Fq::from_u64(3);
    return lhs == rhs;
}

}  // namespace shp

using namespace shp;

// `share` (optional): an SRS over the same two bases whose MSM plans this one takes part in instead of building its own
static void srs_finish(shp_srs* s, const shp_srs* share = nullptr) {
    {
        std::vector<uint8_t> m(4 + 3 * sizeof(G1Affine) + 256, 0);
        memcpy(m.data(), &s->k, 4);
        SHP_HIP(hipMemcpy(m.data() + 4, s->d_g, (s->n >= 2 ? 2 : 1) * sizeof(G1Affine), hipMemcpyDeviceToHost));
        SHP_HIP(hipMemcpy(m.data() + 4 + 2 * sizeof(G1Affine), s->d_gl, sizeof(G1Affine), hipMemcpyDeviceToHost));
        if (s->has_g2) {
            memcpy(m.data() + 4 + 3 * sizeof(G1Affine), s->g2, 128);
            memcpy(m.data() + 4 + 3 * sizeof(G1Affine) + 128, s->s_g2, 128);
        }
        keccak256(m.data(), m.size(), s->fingerprint);
    }
    if (share) {
        // same k, same points: the multiples tables (up to 228 GB at k = 13) exist once, whichever of the two is destroyed first
        s->msm_gl = share->msm_gl;
        s->msm_g = share->msm_g;
        s->msm_gl->refs.fetch_add(1);
        s->msm_g->refs.fetch_add(1);
        return;
    }
    // the Lagrange basis first: it carries most of a proof's commitments and gets the larger share of the HBM
    msm_create_for_srs(s->ctx, s->d_gl, s->n, true, &s->msm_gl);
    msm_create_for_srs(s->ctx, s->d_g, s->n, false, &s->msm_g);
}

static void srs_release(shp_srs* s) {
    if (!s) return;
    if (s->msm_g) shp_msm_destroy(s->msm_g);
    if (s->msm_gl) shp_msm_destroy(s->msm_gl);
    if (s->d_g) (void)hipFree(s->d_g);
    if (s->d_gl) (void)hipFree(s->d_gl);
    delete s;
}

// canonical little-endian Fq2 coordinates -> the Montgomery form the files hold
static void g2_to_montgomery(const uint8_t* src_canon, uint8_t* dst) {
    for (int c = 0; c < 4; c++) {
        Fq raw;
        memcpy(raw.l, src_canon + 32 * c, 32);
        Fq m = Fq::from_raw(raw);
        memcpy(dst + 32 * c, m.l, 32);
    }
}

// stored Montgomery Fq2 coordinates -> canonical little-endian bytes
static void g2_to_canonical(const uint8_t* src_mont, uint8_t* dst) {
    for (int c = 0; c < 4; c++) {
        Fq v;
        memcpy(v.l, src_mont + 32 * c, 32);
        SHP_REQUIRE(fq_canonical(v), SHP_ERR_FORMAT, "srs: G2 coordinate not reduced");
        Fq raw = v.to_raw();
        memcpy(dst + 32 * c, raw.l, 32);
    }
}

static void read_points(const uint8_t* p, size_t n, std::vector<G1Affine>& out) {
    out.resize(n);
    for (size_t i = 0; i < n; i++) {
        memcpy(out[i].x.l, p + 64 * i, 32);
        memcpy(out[i].y.l, p + 64 * i + 32, 32);
        SHP_REQUIRE(fq_canonical(out[i].x) && fq_canonical(out[i].y), SHP_ERR_FORMAT, "srs: G1 coordinate not reduced");
        SHP_REQUIRE(on_curve(out[i]), SHP_ERR_FORMAT, "srs: G1 point not on curve");
    }
}

// Proving lanes are HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (HIP's default is 4) and the
// kernels of streams that share a queue run one after the other: 16 lanes on 4 queues prove 1149 proofs/s, on 16 queues
// 1400.  The setting is process-wide and belongs to the operator, so the library only fills in a default: when the
// variable is UNSET at load time (before this library's first HIP call) it becomes 16.  An exported value is never
// overwritten -- a smaller one earns one line on stderr (SHP_KEEP_HW_QUEUES=1 silences it) -- unless the operator asks
// for it: SHP_HW_QUEUES=N in the environment makes the library set GPU_MAX_HW_QUEUES=N.  None of this has an effect if
// the host process initialised HIP earlier.
namespace {
struct HwQueueDefault {
    HwQueueDefault() {
        if (const char* want = getenv("SHP_HW_QUEUES")) {
            const int n = atoi(want);
            if (n >= 1 && n <= 32) setenv("GPU_MAX_HW_QUEUES", want, 1);
            return;
        }
        const char* cur = getenv("GPU_MAX_HW_QUEUES");
        if (!cur || !*cur) {
            setenv("GPU_MAX_HW_QUEUES", "16", 1);
            return;
        }
        const char* keep = getenv("SHP_KEEP_HW_QUEUES");
        if (atoi(cur) < 16 && !(keep && atoi(keep) == 1))
            fprintf(stderr, "[shielder_hip] GPU_MAX_HW_QUEUES=%s: proving lanes beyond that many share hardware queues and run one after "
                            "the other; SHP_HW_QUEUES=16..22 raises it for this process, SHP_KEEP_HW_QUEUES=1 silences this line\n", cur);
    }
} g_hw_queue_default;
}  // namespace

extern "C" {

const char* shp_last_error(void) { return g_last_error.c_str(); }
const char* shp_version(void) { return "shielder_hip 0.1 (gfx950)"; }

int32_t shp_device_count(int32_t* out) {
    return guarded([&] {
        SHP_REQUIRE(out, SHP_ERR_INVALID_ARG, "device_count: null out");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *out = (e == hipSuccess) ? n : 0;
    });
}

int32_t shp_ctx_create(int32_t device, shp_ctx** out) {
    return guarded([&] {
        SHP_REQUIRE(out, SHP_ERR_INVALID_ARG, "ctx_create: null out");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        SHP_REQUIRE(e == hipSuccess && n > 0, SHP_ERR_NO_DEVICE,
                    "no HIP device available: libshielder_hip has no CPU fallback");
        SHP_REQUIRE(device >= 0 && device < n, SHP_ERR_INVALID_ARG, "ctx_create: device index out of range");
        SHP_HIP(hipSetDevice(device));
        // hosts with fewer cores than proving lanes: stream waits sleep instead of spinning (on the 16-core share of
        // one MI355X: 24 lanes 870 -> 1090 proofs/s; 16 lanes are best either way)
        if (const char* e = getenv("SHP_BLOCKING_SYNC"))
            if (*e == '1') (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        hipDeviceProp_t prop;
        SHP_HIP(hipGetDeviceProperties(&prop, device));
        std::unique_ptr<shp_ctx> c(new shp_ctx);
        c->device = device;
        c->num_cus = prop.multiProcessorCount;
        SHP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        try {
            ntt_static_setup(c.get());
        } catch (...) {
            ntt_static_teardown(c.get());
            for (int d = 0; d < 2; d++) {
                if (c->ntt_small[d]) (void)hipFree(c->ntt_small[d]);
                if (c->ntt_small29_a[d]) (void)hipFree(c->ntt_small29_a[d]);
                if (c->ntt_small29_b[d]) (void)hipFree(c->ntt_small29_b[d]);
                if (c->ntt_small29_c[d]) (void)hipFree(c->ntt_small29_c[d]);
            }
            (void)hipStreamDestroy(c->stream);
            throw;
        }
        *out = c.release();
    });
}

int32_t shp_ctx_destroy(shp_ctx* ctx) {
    return guarded([&] {
        if (!ctx) return;
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        for (int d = 0; d < 2; d++) {
            if (ctx->ntt_small[d]) (void)hipFree(ctx->ntt_small[d]);
            if (ctx->ntt_small29_a[d]) (void)hipFree(ctx->ntt_small29_a[d]);
            if (ctx->ntt_small29_b[d]) (void)hipFree(ctx->ntt_small29_b[d]);
            if (ctx->ntt_small29_c[d]) (void)hipFree(ctx->ntt_small29_c[d]);
        }
        for (auto& kv : ctx->ntt_tables)
            for (int d = 0; d < 2; d++) {
                (void)hipFree(kv.second.lo[d]);
                (void)hipFree(kv.second.hi[d]);
                for (int q = 0; q < 2; q++)
                    if (kv.second.pass[d][q]) (void)hipFree(kv.second.pass[d][q]);
                if (kv.second.dit[d]) (void)hipFree(kv.second.dit[d]);
            }
        ctx->ntt_tmp.release();
        ntt_static_teardown(ctx);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
    });
}

int32_t shp_ctx_set_hbm_limit(shp_ctx* ctx, uint64_t bytes) {
    return guarded([&] {
        SHP_REQUIRE(ctx, SHP_ERR_INVALID_ARG, "ctx_set_hbm_limit: null context");
        ctx->hbm_limit.store(bytes);
    });
}

int32_t shp_srs_msm_info(const shp_srs* srs, int32_t basis, int32_t* direct, uint32_t* window_bits, uint64_t* table_bytes) {
    return guarded([&] {
        SHP_REQUIRE(srs && (basis == SHP_BASIS_MONOMIAL || basis == SHP_BASIS_LAGRANGE), SHP_ERR_INVALID_ARG,
                    "srs_msm_info: null SRS or unknown basis");
        const shp_msm* m = basis == SHP_BASIS_LAGRANGE ? srs->msm_gl : srs->msm_g;
        SHP_REQUIRE(m, SHP_ERR_INVALID_ARG, "srs_msm_info: the SRS has no tables for this basis");
        int32_t rc = shp_msm_info(m, direct, window_bits, table_bytes);
        SHP_REQUIRE(rc == SHP_OK, rc, shp_last_error());
    });
}

int32_t shp_ctx_sync(shp_ctx* ctx) {
    return guarded([&] {
        SHP_REQUIRE(ctx, SHP_ERR_INVALID_ARG, "ctx_sync: null ctx");
        SHP_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int32_t shp_ctx_stream(shp_ctx* ctx, void** stream_out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && stream_out, SHP_ERR_INVALID_ARG, "ctx_stream: null argument");
        *stream_out = (void*)ctx->stream;
    });
}

int32_t shp_malloc(shp_ctx* ctx, size_t bytes, void** d_out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_out, SHP_ERR_INVALID_ARG, "malloc: null argument");
        SHP_HIP(hipSetDevice(ctx->device));
        SHP_HIP(hipMalloc(d_out, bytes ? bytes : 1));
    });
}
int32_t shp_free(shp_ctx* ctx, void* d_ptr) {
    return guarded([&] {
        SHP_REQUIRE(ctx, SHP_ERR_INVALID_ARG, "free: null ctx");
        if (d_ptr) SHP_HIP(hipFree(d_ptr));
    });
}
int32_t shp_upload(shp_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_dst && h_src, SHP_ERR_INVALID_ARG, "upload: null argument");
        SHP_HIP(hipSetDevice(ctx->device));
        SHP_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        SHP_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int32_t shp_download(shp_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    return guarded([&] {
        SHP_REQUIRE(ctx && h_dst && d_src, SHP_ERR_INVALID_ARG, "download: null argument");
        SHP_HIP(hipSetDevice(ctx->device));
        SHP_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        SHP_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int32_t shp_fr_from_le32(const uint8_t in[32], uint64_t out[4]) {
    return guarded([&] {
        SHP_REQUIRE(in && out, SHP_ERR_INVALID_ARG, "fr_from_le32: null argument");
        Fr raw;
        memcpy(raw.l, in, 32);
        SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "fr_from_le32: value is not a canonical field element");
        Fr m = Fr::from_raw(raw);
        memcpy(out, m.l, 32);
    });
}
int32_t shp_fr_to_le32(const uint64_t in[4], uint8_t out[32]) {
    return guarded([&] {
        SHP_REQUIRE(in && out, SHP_ERR_INVALID_ARG, "fr_to_le32: null argument");
        Fr m;
        memcpy(m.l, in, 32);
        Fr raw = m.to_raw();
        memcpy(out, raw.l, 32);
    });
}

// Keccak(label-separated big-endian fields) mod r  (shielder-account/src/secrets.rs:27-56)
static void derive_secret(const uint8_t first_le32[32], const char* label, const uint8_t* tail, size_t tail_len, uint8_t out_le32[32]) {
    std::vector<uint8_t> msg;
    for (int i = 31; i >= 0; i--) msg.push_back(first_le32[i]);            // U256::to_be_bytes
    msg.insert(msg.end(), label, label + strlen(label));
    msg.insert(msg.end(), tail, tail + tail_len);
    uint8_t h[32];
    keccak256(msg.data(), msg.size(), h);
    Fr raw;                                                                // U256::from_be_slice(hash)
    for (int i = 0; i < 8; i++)
        raw.l[i] = ((uint32_t)h[31 - 4 * i]) | ((uint32_t)h[30 - 4 * i] << 8) | ((uint32_t)h[29 - 4 * i] << 16) | ((uint32_t)h[28 - 4 * i] << 24);
    Fr red = Fr::from_raw(raw).to_raw();                                   // reduce_mod(FIELD_MODULUS)
    memcpy(out_le32, red.l, 32);
}

int32_t shp_derive_nullifier(const uint8_t id_le32[32], uint32_t nonce, uint8_t out_le32[32]) {
    return guarded([&] {
        SHP_REQUIRE(id_le32 && out_le32, SHP_ERR_INVALID_ARG, "derive_nullifier: null argument");
        const uint8_t tail[4] = {(uint8_t)(nonce >> 24), (uint8_t)(nonce >> 16), (uint8_t)(nonce >> 8), (uint8_t)nonce};
        derive_secret(id_le32, "nullifier", tail, 4, out_le32);
    });
}

int32_t shp_derive_id(const uint8_t private_key_le32[32], uint64_t chain_id, uint32_t account_nonce, uint8_t out_le32[32]) {
    return guarded([&] {
        SHP_REQUIRE(private_key_le32 && out_le32, SHP_ERR_INVALID_ARG, "derive_id: null argument");
        uint8_t tail[12];
        for (int i = 0; i < 8; i++) tail[i] = (uint8_t)(chain_id >> (8 * (7 - i)));
        for (int i = 0; i < 4; i++) tail[8 + i] = (uint8_t)(account_nonce >> (8 * (3 - i)));
        derive_secret(private_key_le32, "id", tail, 12, out_le32);
    });
}

int32_t shp_encode_calldata(const uint8_t* proof, size_t proof_len, const uint8_t* instances_le32, size_t n_instances,
                            uint8_t* out, size_t out_cap, size_t* out_len) {
    return guarded([&] {
        SHP_REQUIRE(proof && out_len && (instances_le32 || n_instances == 0), SHP_ERR_INVALID_ARG, "encode_calldata: null argument");
        const size_t padded = (proof_len + 31) / 32 * 32;
        const size_t total = 4 + 64 + 32 + padded + 32 + 32 * n_instances;
        *out_len = total;
        if (!out) return;                                  // size query
        SHP_REQUIRE(out_cap >= total, SHP_ERR_INVALID_ARG, "encode_calldata: buffer too small");
        memset(out, 0, total);
        static const char sig[] = "verifyProof(bytes,uint256[])";
        uint8_t h[32];
        keccak256(reinterpret_cast<const uint8_t*>(sig), sizeof(sig) - 1, h);
        memcpy(out, h, 4);
        auto put_u256 = [&](size_t off, uint64_t v) {
            for (int i = 0; i < 8; i++) out[off + 31 - i] = (uint8_t)(v >> (8 * i));
        };
        uint8_t* body = out + 4;
        put_u256(4 + 0, 64);                               // offset of `proof`
        put_u256(4 + 32, 64 + 32 + padded);                // offset of `instances`
        put_u256(4 + 64, proof_len);
        memcpy(body + 96, proof, proof_len);
        const size_t io = 4 + 96 + padded;
        put_u256(io, n_instances);
        for (size_t i = 0; i < n_instances; i++) {
            Fr raw;
            memcpy(raw.l, instances_le32 + 32 * i, 32);
            SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "encode_calldata: public input is not a canonical field element");
            for (int b = 0; b < 32; b++) out[io + 32 + 32 * i + b] = instances_le32[32 * i + 31 - b];   // uint256 big-endian
        }
    });
}

int32_t shp_ntt_fr(shp_ctx* ctx, void* d_data, uint32_t log_n, uint32_t batch, int32_t inverse, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_data, SHP_ERR_INVALID_ARG, "ntt: null argument");
        SHP_HIP(hipSetDevice(ctx->device));
        NttOpts o;
        o.inverse = inverse != 0;
        ntt_launch(ctx, (const Fr*)d_data, (Fr*)d_data, log_n, batch, o, ctx->pick(stream));
    });
}

int32_t shp_coeff_to_extended(shp_ctx* ctx, const void* d_in, void* d_out, uint32_t k, uint32_t ext_k, uint32_t batch,
                              void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_in && d_out, SHP_ERR_INVALID_ARG, "coeff_to_extended: null argument");
        SHP_REQUIRE(ext_k >= k, SHP_ERR_INVALID_ARG, "coeff_to_extended: ext_k < k");
        SHP_REQUIRE(d_in != d_out, SHP_ERR_INVALID_ARG, "coeff_to_extended: must be out of place");
        SHP_HIP(hipSetDevice(ctx->device));
        NttOpts o;
        o.in_len = 1u << k;
        o.in_stride = (size_t)1 << k;
        o.pre_pattern = 1;
        ntt_launch(ctx, (const Fr*)d_in, (Fr*)d_out, ext_k, batch, o, ctx->pick(stream));
    });
}

int32_t shp_extended_to_coeff(shp_ctx* ctx, void* d_data, uint32_t ext_k, uint32_t batch, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_data, SHP_ERR_INVALID_ARG, "extended_to_coeff: null argument");
        SHP_HIP(hipSetDevice(ctx->device));
        NttOpts o;
        o.inverse = true;
        o.post_pattern = 1;
        ntt_launch(ctx, (const Fr*)d_data, (Fr*)d_data, ext_k, batch, o, ctx->pick(stream));
    });
}

// ---------------------------------------------------------------- SRS
int32_t shp_srs_from_parts(shp_ctx* ctx, uint32_t k, const void* h_g, const void* h_g_lagrange, const uint8_t* g2_128,
                           const uint8_t* s_g2_128, shp_srs** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && h_g && out, SHP_ERR_INVALID_ARG, "srs_from_parts: null argument");
        SHP_REQUIRE(k >= 1 && k <= 24, SHP_ERR_INVALID_ARG, "srs_from_parts: k out of range");
        SHP_HIP(hipSetDevice(ctx->device));
        shp_srs* s = new shp_srs;
        try {
            s->ctx = ctx;
            s->k = k;
            s->n = (size_t)1 << k;
            SHP_HIP(hipMalloc(&s->d_g, s->n * sizeof(G1Affine)));
            SHP_HIP(hipMalloc(&s->d_gl, s->n * sizeof(G1Affine)));
            SHP_HIP(hipMemcpy(s->d_g, h_g, s->n * sizeof(G1Affine), hipMemcpyHostToDevice));
            if (h_g_lagrange) SHP_HIP(hipMemcpy(s->d_gl, h_g_lagrange, s->n * sizeof(G1Affine), hipMemcpyHostToDevice));
            else g1_lagrange_from_monomial(ctx, s->d_g, s->d_gl, k);
            if (g2_128 && s_g2_128) {
                memcpy(s->g2, g2_128, 128);
                memcpy(s->s_g2, s_g2_128, 128);
                s->has_g2 = true;
            }
            srs_finish(s);
        } catch (...) {
            srs_release(s);
            throw;
        }
        *out = s;
    });
}

// ParamsKZG::downsize (halo2 v0.3.0, called at /root/reference/crates/shielder-cli/src/shielder_ops/pk.rs:110 and inside
// generate_keys_with_min_k): the first 2^k powers, g_lagrange re-derived for the smaller domain, same G2 part
int32_t shp_srs_downsize(const shp_srs* srs, uint32_t k, shp_srs** out) {
    return guarded([&] {
        SHP_REQUIRE(srs && out, SHP_ERR_INVALID_ARG, "srs_downsize: null argument");
        SHP_REQUIRE(k >= 1 && k <= srs->k, SHP_ERR_INVALID_ARG, "srs_downsize: k must not exceed the SRS's k");
        SHP_HIP(hipSetDevice(srs->ctx->device));
        shp_srs* s = new shp_srs;
        try {
            s->ctx = srs->ctx;
            s->k = k;
            s->n = (size_t)1 << k;
            SHP_HIP(hipMalloc(&s->d_g, s->n * sizeof(G1Affine)));
            SHP_HIP(hipMalloc(&s->d_gl, s->n * sizeof(G1Affine)));
            SHP_HIP(hipMemcpy(s->d_g, srs->d_g, s->n * sizeof(G1Affine), hipMemcpyDeviceToDevice));
            if (k == srs->k) SHP_HIP(hipMemcpy(s->d_gl, srs->d_gl, s->n * sizeof(G1Affine), hipMemcpyDeviceToDevice));
            else g1_lagrange_from_monomial(s->ctx, s->d_g, s->d_gl, k);
            memcpy(s->g2, srs->g2, 128);
            memcpy(s->s_g2, srs->s_g2, 128);
            s->has_g2 = srs->has_g2;
            srs_finish(s, k == srs->k ? srs : nullptr);
        } catch (...) {
            srs_release(s);
            throw;
        }
        *out = s;
    });
}

int32_t shp_srs_synthetic(shp_ctx* ctx, uint32_t k, const uint8_t tau_le32[32], shp_srs** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && tau_le32 && out, SHP_ERR_INVALID_ARG, "srs_synthetic: null argument");
        SHP_REQUIRE(k >= 1 && k <= 20, SHP_ERR_INVALID_ARG, "srs_synthetic: k out of range");
        SHP_HIP(hipSetDevice(ctx->device));
        Fr raw;
        memcpy(raw.l, tau_le32, 32);
        SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "srs_synthetic: tau is not a canonical field element");
        const Fr tau = Fr::from_raw(raw);
        const uint32_t n = 1u << k;
        std::vector<Fr> pw(n);
        Fr acc = Fr::one();
        for (uint32_t i = 0; i < n; i++) { pw[i] = acc.to_raw(); acc = acc * tau; }
        Fr* d_s = nullptr;
        G1Affine* d_g = nullptr;
        SHP_HIP(hipMalloc(&d_s, (size_t)n * sizeof(Fr)));
        SHP_HIP(hipMalloc(&d_g, (size_t)n * sizeof(G1Affine)));
        std::vector<G1Affine> g(n);
        hipError_t e = hipMemcpy(d_s, pw.data(), (size_t)n * sizeof(Fr), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(srs_powers_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, d_s, d_g, n);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpy(g.data(), d_g, (size_t)n * sizeof(G1Affine), hipMemcpyDeviceToHost);
        (void)hipFree(d_s);
        (void)hipFree(d_g);
        SHP_HIP(e);
        // the G2 half of the same SRS (g2 = the standard generator, s_g2 = tau g2), so that proofs over a
        // synthetic SRS verify through the same pairing as proofs over a ceremony SRS
        uint8_t g2b[128], sg2b[128];
        shp_synthetic_g2(tau, g2b, sg2b);
        int32_t rc = shp_srs_from_parts(ctx, k, g.data(), nullptr, g2b, sg2b, out);
        SHP_REQUIRE(rc == SHP_OK, rc, shp_last_error());
    });
}

int32_t shp_srs_from_ptau(shp_ctx* ctx, const uint8_t* b, size_t len, shp_srs** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && b && out, SHP_ERR_INVALID_ARG, "srs_from_ptau: null argument");
        SHP_REQUIRE(len >= 64, SHP_ERR_FORMAT, "ptau: file too short");
        uint64_t hdr;
        memcpy(&hdr, b + 16, 8);                                  // lib.rs:76-79
        SHP_REQUIRE(hdr >= 8 && 24 + hdr <= len, SHP_ERR_FORMAT, "ptau: bad header size");
        uint32_t k;
        memcpy(&k, b + 24 + hdr - 8, 4);                          // lib.rs:81-85
        SHP_REQUIRE(k >= 1 && k <= 24, SHP_ERR_FORMAT, "ptau: power out of range");
        size_t n = (size_t)1 << k;
        size_t g1_off = 24 + hdr + 12;                            // lib.rs:121-123
        size_t g2_off = g1_off + 64 * (2 * n - 1) + 12;           // lib.rs:111-119
        SHP_REQUIRE(g2_off + 256 <= len, SHP_ERR_FORMAT, "ptau: truncated sections");
        std::vector<G1Affine> g;
        read_points(b + g1_off, n, g);
        uint8_t g2[128], s_g2[128];
        g2_to_canonical(b + g2_off, g2);
        g2_to_canonical(b + g2_off + 128, s_g2);
        int32_t rc = shp_srs_from_parts(ctx, k, g.data(), nullptr, g2, s_g2, out);
        SHP_REQUIRE(rc == SHP_OK, rc, shp_last_error());
    });
}

int32_t shp_srs_from_halo2_raw(shp_ctx* ctx, const uint8_t* b, size_t len, shp_srs** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && b && out, SHP_ERR_INVALID_ARG, "srs_from_halo2_raw: null argument");
        SHP_REQUIRE(len >= 4, SHP_ERR_FORMAT, "raw params: file too short");
        uint32_t k;
        memcpy(&k, b, 4);
        SHP_REQUIRE(k >= 1 && k <= 24, SHP_ERR_FORMAT, "raw params: k out of range");
        size_t n = (size_t)1 << k;
        SHP_REQUIRE(len == 4 + 2 * n * 64 + 256, SHP_ERR_FORMAT, "raw params: unexpected length");
        std::vector<G1Affine> g, gl;
        read_points(b + 4, n, g);
        read_points(b + 4 + n * 64, n, gl);
        uint8_t g2[128], s_g2[128];
        g2_to_canonical(b + 4 + 2 * n * 64, g2);
        g2_to_canonical(b + 4 + 2 * n * 64 + 128, s_g2);
        int32_t rc = shp_srs_from_parts(ctx, k, g.data(), gl.data(), g2, s_g2, out);
        SHP_REQUIRE(rc == SHP_OK, rc, shp_last_error());
    });
}

// marshall_params (/root/reference/crates/shielder_bindings/build.rs:24-29): halo2's RawBytes form of ParamsKZG,
// k | g | g_lagrange | g2 | s_g2 with every coordinate in little-endian Montgomery form -- the layout of the
// reference's resources/ppot_0080_11_raw, which shp_srs_from_halo2_raw reads back
int32_t shp_srs_save(const shp_srs* srs, uint8_t* out, size_t out_cap, size_t* out_len) {
    return guarded([&] {
        SHP_REQUIRE(srs && out_len, SHP_ERR_INVALID_ARG, "srs_save: null argument");
        SHP_REQUIRE(srs->has_g2, SHP_ERR_INVALID_ARG, "srs_save: this SRS carries no G2 elements");
        const size_t tab = srs->n * sizeof(G1Affine), total = 4 + 2 * tab + 256;
        *out_len = total;
        if (!out) return;
        SHP_REQUIRE(out_cap >= total, SHP_ERR_INVALID_ARG, "srs_save: buffer too small");
        memcpy(out, &srs->k, 4);
        SHP_HIP(hipMemcpy(out + 4, srs->d_g, tab, hipMemcpyDeviceToHost));
        SHP_HIP(hipMemcpy(out + 4 + tab, srs->d_gl, tab, hipMemcpyDeviceToHost));
        g2_to_montgomery(srs->g2, out + 4 + 2 * tab);
        g2_to_montgomery(srs->s_g2, out + 4 + 2 * tab + 128);
    });
}

int32_t shp_srs_destroy(shp_srs* srs) {
    return guarded([&] { srs_release(srs); });
}

int32_t shp_srs_k(const shp_srs* srs, uint32_t* k) {
    return guarded([&] {
        SHP_REQUIRE(srs && k, SHP_ERR_INVALID_ARG, "srs_k: null argument");
        *k = srs->k;
    });
}

int32_t shp_srs_get_g(const shp_srs* srs, int32_t basis, void* h_out, size_t n_points) {
    return guarded([&] {
        SHP_REQUIRE(srs && h_out, SHP_ERR_INVALID_ARG, "srs_get_g: null argument");
        SHP_REQUIRE(n_points <= srs->n, SHP_ERR_INVALID_ARG, "srs_get_g: too many points requested");
        const G1Affine* src = basis == SHP_BASIS_LAGRANGE ? srs->d_gl : srs->d_g;
        SHP_HIP(hipMemcpy(h_out, src, n_points * sizeof(G1Affine), hipMemcpyDeviceToHost));
    });
}

int32_t shp_srs_get_g2(const shp_srs* srs, uint8_t g2_128[128], uint8_t s_g2_128[128]) {
    return guarded([&] {
        SHP_REQUIRE(srs && g2_128 && s_g2_128, SHP_ERR_INVALID_ARG, "srs_get_g2: null argument");
        SHP_REQUIRE(srs->has_g2, SHP_ERR_INVALID_ARG, "srs_get_g2: this SRS carries no G2 elements");
        memcpy(g2_128, srs->g2, 128);
        memcpy(s_g2_128, srs->s_g2, 128);
    });
}

int32_t shp_srs_commit(shp_srs* srs, int32_t basis, const void* d_polys, uint32_t batch, void* h_out_affine,
                       void* stream) {
    return guarded([&] {
        SHP_REQUIRE(srs && d_polys && h_out_affine, SHP_ERR_INVALID_ARG, "srs_commit: null argument");
        SHP_HIP(hipSetDevice(srs->ctx->device));
        shp_msm* m = basis == SHP_BASIS_LAGRANGE ? srs->msm_gl : srs->msm_g;
        msm_run(m, (const Fr*)d_polys, srs->n, srs->n, batch, (G1Affine*)h_out_affine, srs->ctx->pick(stream));
    });
}

}  // extern "C"
