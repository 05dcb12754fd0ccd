// wave_max_probe.hip -- cost of a wave64 max-reduction to a uniform value on gfx950, as a dependent chain (what one scenario
// wave of simon_table.hip pays per scheduling cycle) and under 4 waves per SIMD (what the VALU pipe pays).
//   A: 6 DPP steps (quad_perm x2, row_half_mirror, row_mirror, row_bcast15, row_bcast31) + readlane 63   (simon_device.h)
//   B: 4 row DPP steps + 4 readlanes + 3 s_max
//   C: v_permlane32_swap + v_permlane16_swap (gfx950) + 4 row DPP steps + readfirstlane
//   D: C with the row steps first
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/wave_max_probe profiles/micro/wave_max_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define DPP(old, v, ctrl, rowmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rowmask), 0xF, false)

__device__ __forceinline__ unsigned row_max(unsigned v) {
    v = max(v, (unsigned)DPP(0, (int)v, 0xB1, 0xF));
    v = max(v, (unsigned)DPP(0, (int)v, 0x4E, 0xF));
    v = max(v, (unsigned)DPP(0, (int)v, 0x141, 0xF));
    v = max(v, (unsigned)DPP(0, (int)v, 0x140, 0xF));
    return v;
}
template <int V> __device__ __forceinline__ unsigned wmax(unsigned v);
template <> __device__ __forceinline__ unsigned wmax<0>(unsigned v) {
    v = row_max(v);
    v = max(v, (unsigned)DPP(0, (int)v, 0x142, 0xA));
    v = max(v, (unsigned)DPP(0, (int)v, 0x143, 0xC));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <> __device__ __forceinline__ unsigned wmax<1>(unsigned v) {
    v = row_max(v);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
template <> __device__ __forceinline__ unsigned wmax<2>(unsigned v) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = max(r[0], r[1]);
    auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max(q[0], q[1]);
    v = row_max(v);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
template <> __device__ __forceinline__ unsigned wmax<3>(unsigned v) {
    v = row_max(v);
    auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max(q[0], q[1]);
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = max(r[0], r[1]);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

template <int V>
__global__ __launch_bounds__(64) void chain(int iters, unsigned* out, unsigned long long* ticks) {
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        const unsigned m = wmax<V>(x & 0xFFFFFFu);
        acc += m;
        x = x * 1664525u + m + (unsigned)i;        // the next key depends on the reduction: a dependent chain
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc; ticks[blockIdx.x] = t1 - t0; }
}

template <int V>
void run(const char* name, unsigned expect_ref[1]) {
    const int iters = 20000;
    unsigned* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&ticks, 4096 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256 * 4, 256 * 16}) {       // 1 and 4 waves per SIMD
        hipLaunchKernelGGL(chain<V>, dim3(blocks), dim3(64), 0, 0, iters, out, ticks);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(chain<V>, dim3(blocks), dim3(64), 0, 0, iters, out, ticks); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned h; unsigned long long t;
        (void)hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        if (V == 0) expect_ref[0] = h;
        printf("%s  %2d waves/SIMD: %.3f ms, %.1f ns per iteration (loop body incl. 4 other ops), block 0: %.1f ticks/iter, checksum %s\n", name, blocks / 1024, ms,
               ms * 1e6 / iters, (double)t / iters, h == expect_ref[0] ? "same as A" : "DIFFERENT");
    }
}

int main() {
    unsigned ref[1] = {0};
    run<0>("A 6 DPP + readlane            ", ref);
    run<1>("B 4 DPP + 4 readlane + 3 s_max", ref);
    run<2>("C permlane swaps + 4 DPP      ", ref);
    run<3>("D 4 DPP + permlane swaps      ", ref);
    return 0;
}
