// How many kernel launches per second does one process sustain from T host threads, each on a stream of its own?
// The proving pool issues ~80 HIP calls per proof from 16 threads; this measures the ceiling of that path alone:
// empty kernels (one wave), a stream synchronisation every `chunk` launches, T = 1, 2, 4, 8, 16, 32 threads.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/bin/launch_rate tools/ubench/launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

__global__ void empty_kernel(unsigned* p) {
    if (p && threadIdx.x == 1000) *p = 1;
}

static void check(hipError_t e, const char* what) {
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); exit(1); }
}

int main(int argc, char** argv) {
    const int chunk = argc > 1 ? atoi(argv[1]) : 8;
    const int per_thread = argc > 2 ? atoi(argv[2]) : 20000;
    check(hipSetDevice(0), "hipSetDevice");
    printf("GPU_MAX_HW_QUEUES=%s chunk=%d launches/thread=%d\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset", chunk, per_thread);
    for (int T : {1, 2, 4, 8, 16, 32}) {
        std::vector<hipStream_t> st(T);
        for (auto& s : st) check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
        for (auto& s : st) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (unsigned*)nullptr); check(hipStreamSynchronize(s), "warm"); }
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                check(hipSetDevice(0), "hipSetDevice");
                for (int i = 0; i < per_thread; i++) {
                    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st[t], (unsigned*)nullptr);
                    if ((i + 1) % chunk == 0) check(hipStreamSynchronize(st[t]), "sync");
                }
                check(hipStreamSynchronize(st[t]), "sync");
            });
        for (auto& x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %2d: %8.0f launches/s in all, %6.2f us per launch per thread\n", T, T * (double)per_thread / s, s / per_thread * 1e6);
        for (auto& x : st) check(hipStreamDestroy(x), "destroy");
    }
    return 0;
}
