// occupancy_probe.hip -- how many single-wave workgroups does a CU hold at once, as a function of the LDS each one asks for
// and of its VGPR count?  Every workgroup spins for a fixed number of shader cycles; kernel time / spin time = rounds.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/occupancy_probe profiles/micro/occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int REGS>
__global__ __launch_bounds__(64) void spin(unsigned long long ticks, float* sink) {
    extern __shared__ unsigned char smem[];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 1.0f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc += r[i];
    if (acc == 12345.678f) sink[0] = acc + smem[threadIdx.x];
}

template <int REGS>
void run(int blocks, const char* name) {
    float* sink; (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned long long ticks = 4800000ull;   // 2 ms at 2.4 GHz
    for (int lds : {0, 4096, 8192, 10240, 12288, 16384, 20480, 32768}) {
        (void)hipFuncSetAttribute((const void*)spin<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(spin<REGS>, dim3(256), dim3(64), lds, 0, ticks, sink);   // warm
        (void)hipEventRecord(e0); hipLaunchKernelGGL(spin<REGS>, dim3(blocks), dim3(64), lds, 0, ticks, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s blocks %d lds %5d B: %.2f ms = %.2f rounds -> ~%.1f workgroups per CU at once\n", name, blocks, lds, ms, ms / 2.0, blocks / 256.0 / (ms / 2.0));
    }
}

// smallest multiple of 256 workgroups that needs a second round: capacity per CU = that multiple - 1
template <int REGS>
void capacity(int lds, const char* name) {
    float* sink; (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned long long ticks = 2400000ull;   // 1 ms
    (void)hipFuncSetAttribute((const void*)spin<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)spin<REGS>);
    printf("[numRegs %d] ", fa.numRegs);
    for (int k = 1; k <= 40; ++k) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(spin<REGS>, dim3(256 * k), dim3(64), lds, 0, ticks, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms > 1.7f) { printf("%s, lds %d B: %d workgroups per CU fit at once (%d x 256 needed %.2f ms)\n", name, lds, k - 1, k, ms); return; }
    }
    printf("%s, lds %d B: >= 40 per CU\n", name, lds);
}

int main(int argc, char** argv) {
    if (argc > 1) {
        const int blocks = atoi(argv[1]);
        run<8>(blocks, "12 VGPR");
        run<72>(blocks, "76 VGPR");
        return 0;
    }
    for (int lds : {0, 5120, 8192, 10240, 11264, 12288}) capacity<8>(lds, "12 VGPR");
    for (int lds : {0, 8192}) { capacity<24>(lds, "~28 VGPR"); capacity<40>(lds, "~44 VGPR"); capacity<56>(lds, "~60 VGPR"); capacity<72>(lds, "76 VGPR"); capacity<88>(lds, "~92 VGPR"); capacity<120>(lds, "~124 VGPR"); }
    return 0;
}
