// clock_probe.hip -- what does one s_memtime tick mean on this box, and how fast do dependent VALU ops retire?
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/clock_probe profiles/micro/clock_probe.hip && profiles/micro/clock_probe
// Prints: s_memtime ticks per microsecond (a kernel spins for a fixed tick count, timed with HIP events), s_memrealtime ticks
// per microsecond, and ticks per dependent v_add_u32 / v_fma_f64 in a one-wave chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0));
    unsigned long long t;
    do { t = __builtin_readcyclecounter(); } while (t - t0 < ticks);
    unsigned long long r1;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

__global__ void chain(int n, unsigned long long* out, unsigned* sink, double* dsink) {
    unsigned x = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = x * 3u + 1u;          // dependent v_mad_u32_u24 / v_mul + add chain
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double d = (double)threadIdx.x + 1.5;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) d = __builtin_fma(d, 1.0000001, 0.25);   // dependent v_fma_f64 chain
    }
    const unsigned long long t2 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
    dsink[blockIdx.x * blockDim.x + threadIdx.x] = d;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t1; }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    unsigned long long* d_out; unsigned* d_sink; double* d_dsink;
    hipMalloc(&d_out, blocks * 16); hipMalloc(&d_sink, blocks * 64 * 4); hipMalloc(&d_dsink, blocks * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        const unsigned long long ticks = 20000000ull;
        hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, 0, ticks, d_out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("spin %d blocks: %llu s_memtime ticks in %.3f ms -> %.1f ticks/us; s_memrealtime %llu -> %.1f ticks/us\n", blocks, h[0], ms, h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
    }
    for (int rep = 0; rep < 2; ++rep) {
        const int n = 20000;
        hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, n, d_out, d_sink, d_dsink); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("chain %d blocks: int mul+add %.2f ticks per dependent pair, fp64 fma %.2f ticks per dependent op; kernel %.3f ms -> %.2f ns per int pair+fma\n", blocks,
               (double)h[0] / (n * 16.0), (double)h[1] / (n * 16.0), ms, ms * 1e6 / (n * 16.0));
    }
    return 0;
}
