// gfx950: is the "VALU writes SGPR/VCC -> VALU reads it" hazard (LLVM: 2 wait states on gfx940+) real for
// v_mad_u64_u32 carry-out -> v_addc_co_u32, what do the wait states cost, and does rotating the carry
// through three SGPR pairs (consumer two instructions behind its producer) remove the cost?
// Each variant sums 96-bit column accumulators of the same products; results are checked on the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define REP4(x) x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a = t * 2654435761u + 12345u, b = (t ^ 0x9e3779b9u) * 40503u + 977u;
    a |= 0xf0000000u; b |= 0xf0000000u;       // big operands: carries out of bit 63 are frequent
    uint64_t lo = 0;
    uint32_t hi = 0;
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) {          // back to back (no wait states): is it even correct?
            REP4(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 1) {   // s_nop 1 between producer and consumer
            REP4(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 2) {   // three rotating SGPR pairs, consumer two instructions behind
            uint64_t c0, c1, c2;
            REP4(asm volatile("v_mad_u64_u32 %0, %4, %2, %3, %0\n"
                              "v_mad_u64_u32 %0, %5, %2, %3, %0\n"
                              "v_mad_u64_u32 %0, %6, %2, %3, %0\n"
                              "v_addc_co_u32 %1, %4, 0, %1, %4\n"
                              "v_addc_co_u32 %1, %5, 0, %1, %5\n"
                              "v_addc_co_u32 %1, %6, 0, %1, %6\n"
                              : "+v"(lo), "+v"(hi) : "v"(a), "v"(b), "s"(c0), "s"(c1), "s"(c2));)
        } else if (KIND == 3) {   // s_nop 0 only (one wait state)
            REP4(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");)
        }
        a += 0x01000193u;
    }
    out[3 * t] = (uint32_t)lo;
    out[3 * t + 1] = (uint32_t)(lo >> 32);
    out[3 * t + 2] = hi;
}

// KIND 2 as written above passes c0..c2 as *inputs* ("s") although the asm writes them: they are dead
// afterwards and never read before being written, but to stay within the rules use a separate kernel
// with proper early-clobber outputs.
__global__ __launch_bounds__(256) void k2ok(uint32_t* out, uint32_t iters) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a = t * 2654435761u + 12345u, b = (t ^ 0x9e3779b9u) * 40503u + 977u;
    a |= 0xf0000000u; b |= 0xf0000000u;
    uint64_t lo = 0;
    uint32_t hi = 0;
    for (uint32_t i = 0; i < iters; i++) {
        uint64_t c0, c1, c2;
        REP4(asm volatile("v_mad_u64_u32 %0, %2, %5, %6, %0\n"
                          "v_mad_u64_u32 %0, %3, %5, %6, %0\n"
                          "v_mad_u64_u32 %0, %4, %5, %6, %0\n"
                          "v_addc_co_u32 %1, %2, 0, %1, %2\n"
                          "v_addc_co_u32 %1, %3, 0, %1, %3\n"
                          "v_addc_co_u32 %1, %4, 0, %1, %4\n"
                          : "+v"(lo), "+v"(hi), "=&s"(c0), "=&s"(c1), "=&s"(c2) : "v"(a), "v"(b));)
        a += 0x01000193u;
    }
    out[3 * t] = (uint32_t)lo;
    out[3 * t + 1] = (uint32_t)(lo >> 32);
    out[3 * t + 2] = hi;
}

// 8-limb add carry chain with / without wait states
template <int NOPS>
__global__ __launch_bounds__(256) void kadd(uint32_t* out, uint32_t iters) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t r[8], b[8];
    for (int i = 0; i < 8; i++) { r[i] = t * (0x9e3779b9u + i) + i; b[i] = 0xfffffff0u ^ (t * (31 + i)); }
    for (uint32_t i = 0; i < iters; i++) {
        if (NOPS == 0) {
            REP4(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_addc_co_u32 %2, vcc, %2, %10, vcc\n"
                              "v_addc_co_u32 %3, vcc, %3, %11, vcc\n v_addc_co_u32 %4, vcc, %4, %12, vcc\n v_addc_co_u32 %5, vcc, %5, %13, vcc\n"
                              "v_addc_co_u32 %6, vcc, %6, %14, vcc\n v_addc_co_u32 %7, vcc, %7, %15, vcc\n"
                              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                              : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");)
        } else {
            REP4(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n s_nop 1\n v_addc_co_u32 %2, vcc, %2, %10, vcc\n s_nop 1\n"
                              "v_addc_co_u32 %3, vcc, %3, %11, vcc\n s_nop 1\n v_addc_co_u32 %4, vcc, %4, %12, vcc\n s_nop 1\n v_addc_co_u32 %5, vcc, %5, %13, vcc\n s_nop 1\n"
                              "v_addc_co_u32 %6, vcc, %6, %14, vcc\n s_nop 1\n v_addc_co_u32 %7, vcc, %7, %15, vcc\n"
                              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                              : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");)
        }
    }
    for (int i = 0; i < 8; i++) out[8 * t + i] = r[i];
}

static void host_ref(uint32_t t, uint32_t iters, uint32_t* o) {
    uint32_t a = t * 2654435761u + 12345u, b = (t ^ 0x9e3779b9u) * 40503u + 977u;
    a |= 0xf0000000u; b |= 0xf0000000u;
    unsigned __int128 acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        for (int j = 0; j < 12; j++) acc += (unsigned __int128)((uint64_t)a * b);
        a += 0x01000193u;
    }
    o[0] = (uint32_t)acc; o[1] = (uint32_t)(acc >> 32); o[2] = (uint32_t)(acc >> 64);
}
static void host_add_ref(uint32_t t, uint32_t iters, uint32_t* o) {
    uint32_t r[8], b[8];
    for (int i = 0; i < 8; i++) { r[i] = t * (0x9e3779b9u + i) + i; b[i] = 0xfffffff0u ^ (t * (31 + i)); }
    for (uint32_t i = 0; i < iters * 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 8; j++) { c += (uint64_t)r[j] + b[j]; r[j] = (uint32_t)c; c >>= 32; }
    }
    for (int i = 0; i < 8; i++) o[i] = r[i];
}

template <class F>
static double timed(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    f();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    const uint32_t iters = 1000;
    size_t maxthreads = (size_t)cus * 4 * 256;
    uint32_t* d;
    (void)hipMalloc(&d, maxthreads * 8 * 4);
    std::vector<uint32_t> h(maxthreads * 8);
    const char* names[] = {"mad->addc back to back", "mad, s_nop 1, addc", "3 rotating sgpr carries (\"s\" in)", "mad, s_nop 0, addc"};
    for (int w : {1, 2, 4}) {
        int blocks = cus * w;
        for (int kind = 0; kind < 5; kind++) {
            if (kind == 2) continue;   // carries as inputs: the allocator may merge them; kept for reference only
            double ms = timed([&] {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                    default: hipLaunchKernelGGL(k2ok, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                }
            });
            (void)hipMemcpy(h.data(), d, (size_t)blocks * 256 * 3 * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (uint32_t t = 0; t < (uint32_t)blocks * 256; t += 37) {
                uint32_t o[3];
                host_ref(t, iters, o);
                if (o[0] != h[3 * t] || o[1] != h[3 * t + 1] || o[2] != h[3 * t + 2]) bad++;
            }
            double cyc = ms * 1e-3 * 2.4e9 / (iters * 12.0 * w);     // cycles per (mad+addc) pair per wave slot
            printf("waves/SIMD=%d  %-36s %.3f ms  %.2f cycles per mad+addc pair per SIMD  mismatches %d\n", w,
                   kind < 4 ? names[kind] : "3 rotating sgpr carries (outputs)", ms, cyc, bad);
        }
        for (int nops = 0; nops < 2; nops++) {
            double ms = timed([&] {
                if (nops) hipLaunchKernelGGL(kadd<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
                else hipLaunchKernelGGL(kadd<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            });
            (void)hipMemcpy(h.data(), d, (size_t)blocks * 256 * 8 * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (uint32_t t = 0; t < (uint32_t)blocks * 256; t += 37) {
                uint32_t o[8];
                host_add_ref(t, iters, o);
                for (int i = 0; i < 8; i++) if (o[i] != h[8 * t + i]) { bad++; break; }
            }
            double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0 * w);
            printf("waves/SIMD=%d  8-limb add chain %-19s %.3f ms  %.1f cycles per 8-limb add per SIMD  mismatches %d\n", w,
                   nops ? "with s_nop 1" : "back to back", ms, cyc, bad);
        }
    }
    return 0;
}
