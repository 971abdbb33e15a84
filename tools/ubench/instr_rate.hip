// Instruction-rate microbenchmark for gfx950: how many lane-issue slots the integer instructions a
// 254-bit Montgomery product is built from really cost.  Prints ops/clk/CU for each.
// Build: hipcc -O3 --offload-arch=gfx950 -o instr_rate instr_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a = t * 2654435761u + seed, b = t ^ 0x9e3779b9u;
    uint64_t acc0 = t, acc1 = t + 1, acc2 = t + 2, acc3 = t + 3;
    uint32_t x0 = t, x1 = t + 5, x2 = t + 7, x3 = t + 11;
    double d0 = t, d1 = t + 1.0, d2 = t + 2.0, d3 = t + 3.0, da = 1.0000001, db = 0.5;
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) {   // v_mad_u64_u32, 4 independent accumulators
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                              "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 1) {   // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
        } else if (KIND == 2) {   // v_mul_hi_u32
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
        } else if (KIND == 3) {   // v_lshl_add_u64
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));)
        } else if (KIND == 4) {   // v_add_co_u32 + v_addc_co_u32 pair
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");)
        } else if (KIND == 5) {   // v_mov_b32 (rotating)
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if (KIND == 6) {   // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));)
        } else if (KIND == 7) {   // v_mad_u32_u24
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
        } else if (KIND == 8) {   // v_add3_u32
            REP8(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
        } else if (KIND == 9) {   // v_mad_u64_u32, ONE dependent accumulator (latency)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                              "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                              : "+v"(acc0) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 10) {  // v_mad_u64_u32 with SGPR carry-out destination + v_addc consuming it
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n"
                              "v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(x0), "v"(x1) : "vcc");)
        } else if (KIND == 11) {  // v_alignbit_b32 (funnel shift)
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 29\n v_alignbit_b32 %1, %1, %2, 29\n v_alignbit_b32 %2, %2, %3, 29\n v_alignbit_b32 %3, %3, %0, 29\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if (KIND == 12) {  // v_lshrrev_b64
            REP8(asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));)
        } else if (KIND == 14) {  // v_mad_u64_u32 with an SGPR factor and its carry into an SGPR pair (what the compiler emits for q * p_i)
            REP8(asm volatile("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n"
                              "v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "s"(seed) : "s20", "s21");)
        } else if (KIND == 15) {  // v_and_b32 with a literal
            REP8(asm volatile("v_and_b32 %0, 0x1fffffff, %1\n v_and_b32 %1, 0x1fffffff, %2\n v_and_b32 %2, 0x1fffffff, %3\n v_and_b32 %3, 0x1fffffff, %0\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if (KIND == 16) {  // the carry step of a 29-bit normalisation: v_lshrrev_b32 + v_add_u32
            REP8(asm volatile("v_lshrrev_b32 %0, 29, %1\n v_add_u32 %1, %1, %0\n v_lshrrev_b32 %2, 29, %3\n v_add_u32 %3, %3, %2\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if (KIND == 13) {  // v_mad_u64_u32 whose addend is built from two other registers (mov pair cost)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %1\n v_mad_u64_u32 %1, vcc, %4, %5, %2\n"
                              "v_mad_u64_u32 %2, vcc, %4, %5, %3\n v_mad_u64_u32 %3, vcc, %4, %5, %0\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");)
        }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = (uint32_t)(acc0 + acc1 + acc2 + acc3) + x0 + x1 + x2 + x3 + (uint32_t)(d0 + d1 + d2 + d3);
}

// Each figure is a SLOPE: the kernel is timed at `iters` and at 2 x `iters` loop iterations and the difference is
// divided by the extra instructions, so launch overhead, the clock ramp after idle and the kernel's prologue cancel.
// A 50 ms untimed run of the same kernel brings the clocks up first.
template <int KIND>
static void run(const char* name, int waves_per_simd, uint32_t* d_out, int cus, double clk_ghz) {
    // blocks of 256 threads = 1 wave per SIMD each
    int blocks = cus * waves_per_simd;
    const uint32_t iters = 100000;                              // 3.2 M instructions per wave: 5 ms and up per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](uint32_t it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, it, 1u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    };
    double warm = 0;
    while (warm < 50.0) warm += timed(iters);                   // warm-up: at least 50 ms of this very kernel
    double t1 = 1e30, t2 = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        t1 = std::min(t1, timed(iters));
        t2 = std::min(t2, timed(2 * iters));
    }
    const double instr_per_wave = (double)iters * 32.0;         // 8 reps x 4 instrs per iteration
    auto cycles = [&](double ms, double n_iters_scale) {
        double per_simd_per_s = instr_per_wave * n_iters_scale * waves_per_simd / (ms * 1e-3);   // wave-instructions per SIMD per second
        return clk_ghz * 1e9 / per_simd_per_s;
    };
    printf("%-28s waves/SIMD=%d  %.3f ms @%u  %.3f ms @%u  -> slope %.2f cycles per wave64-instruction per SIMD (ratio %.2f; %.2f GHz)\n",
           name, waves_per_simd, t1, iters, t2, 2 * iters, cycles(t2 - t1, 1.0), cycles(t2, 2.0), clk_ghz);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    double clk = p.clockRate * 1e-6;
    printf("device %s CUs %d clock %.2f GHz\n", p.name, cus, clk);
    uint32_t* d;
    hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mad_u64_u32 x4 indep", w, d, cus, clk);
        run<9>("v_mad_u64_u32 dependent", w, d, cus, clk);
        run<10>("v_mad_u64_u32 + v_addc", w, d, cus, clk);
        run<1>("v_mul_lo_u32", w, d, cus, clk);
        run<3>("v_lshl_add_u64", w, d, cus, clk);
        run<4>("v_add_co/addc_co", w, d, cus, clk);
        run<5>("v_mov_b32", w, d, cus, clk);
        run<6>("v_fma_f64", w, d, cus, clk);
        run<8>("v_add3_u32", w, d, cus, clk);
        run<11>("v_alignbit_b32", w, d, cus, clk);
        run<12>("v_lshrrev_b64", w, d, cus, clk);
        run<14>("v_mad_u64_u32 sgpr", w, d, cus, clk);
        run<15>("v_and_b32 literal", w, d, cus, clk);
        run<16>("v_lshrrev_b32 + v_add_u32", w, d, cus, clk);
    }
    return 0;
}
