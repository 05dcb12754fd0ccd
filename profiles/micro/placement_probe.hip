// placement_probe.hip -- where does the dispatcher put S single-wave workgroups that are all resident at once?  Config 3's batch is 4 096
// workgroups of 64 threads with 9 408 B of LDS each on 256 CUs: 16 per CU = 4 per SIMD if the placement is even, and the kernel ends with its
// most loaded SIMD if it is not.  Every workgroup records (XCC, SE, CU, SIMD) from the hardware id registers and spins for a fixed time, so
// that all of them are resident together; the host prints the histogram of workgroups per CU and of waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/placement_probe profiles/micro/placement_probe.hip
//   profiles/micro/placement_probe [workgroups=4096] [lds_bytes=9408]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int REGS>
__global__ __launch_bounds__(64) void spin(unsigned long long ticks, unsigned* where, float* sink) {
    extern __shared__ unsigned char smem[];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 1.0f + i;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { where[blockIdx.x * 2] = hw; where[blockIdx.x * 2 + 1] = xcc; }
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

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 4096;
    const int lds = argc > 2 ? atoi(argv[2]) : 9408;
    unsigned* where; (void)hipMalloc(&where, (size_t)blocks * 8);
    float* sink; (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute((const void*)spin<72>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)spin<72>);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned long long ticks = 2400000ull;   // ~1 ms
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(spin<72>, dim3(blocks), dim3(64), lds, 0, ticks, where, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h((size_t)blocks * 2);
        (void)hipMemcpy(h.data(), where, (size_t)blocks * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_cu, per_simd;
        for (int b = 0; b < blocks; ++b) {
            const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[cu_key]++; per_simd[(cu_key << 2) | simd]++;
        }
        std::map<int, int> hist_cu, hist_simd;
        for (auto& kv : per_cu) hist_cu[kv.second]++;
        for (auto& kv : per_simd) hist_simd[kv.second]++;
        printf("rep %d: %d workgroups, lds %d B, %d VGPRs, %.2f ms (1 ms = one round); %zu CUs and %zu SIMDs used\n", rep, blocks, lds, fa.numRegs, ms, per_cu.size(), per_simd.size());
        printf("  workgroups per CU :"); for (auto& kv : hist_cu) printf("  %d x%d", kv.first, kv.second); printf("\n");
        printf("  waves per SIMD    :"); for (auto& kv : hist_simd) printf("  %d x%d", kv.first, kv.second); printf("\n");
    }
    return 0;
}
