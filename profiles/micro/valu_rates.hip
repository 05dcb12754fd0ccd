// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU operations the scenario kernels are
// made of, on gfx950.  Each kernel runs N independent chains per lane so that latency is hidden; 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/micro/valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITER = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int ia, int ib) {
    double x[CH]; float f[CH]; unsigned u[CH]; long long l[CH];
    for (int c = 0; c < CH; ++c) { x[c] = a + c + threadIdx.x; f[c] = (float)x[c]; u[c] = ia + c + threadIdx.x; l[c] = (long long)u[c] * 977 + ib; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) f[c] = __builtin_fmaf(f[c], (float)b, 1.0f);                 // v_fma_f32
            if (OP == 1) x[c] = __builtin_fma(x[c], b, 1.0);                           // v_fma_f64
            if (OP == 2) x[c] = x[c] + b;                                              // v_add_f64
            if (OP == 3) u[c] = u[c] * (unsigned)ib + 1u;                              // v_mul_lo_u32 (+ add)
            if (OP == 4) u[c] = (u[c] ^ (unsigned)ib) + 3u;                            // v_xor + v_add (int32, full rate)
            if (OP == 5) x[c] = (double)(int)(x[c] * 0.5) + 1.5;                       // v_cvt_i32_f64 + v_cvt_f64_i32 + mul + add
            if (OP == 6) x[c] = 1.0 / (x[c] + 3.0);                                    // IEEE f64 division
            if (OP == 7) l[c] = l[c] * (long long)ib + 7;                              // 64-bit integer multiply
            if (OP == 8) u[c] = max(u[c], __builtin_amdgcn_update_dpp(0u, u[c], 0xB1, 0xF, 0xF, false)) + 1u;   // DPP + max
        }
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += x[c] + f[c] + u[c] + (double)l[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, double ops_per_iter) {
    const int blocks = 256 * 4;   // 4 workgroups of 256 threads per CU = 4 waves per SIMD
    double* d; hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 4 waves x ITER x CH x ops ; cycles at the measured rate assume 2.4 GHz nominal
    const double winstr = 4.0 * ITER * CH * ops_per_iter;
    printf("%-46s %8.3f ms  -> %6.2f cycles per wave64 op per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / winstr);
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 1); run<1>("v_fma_f64", 1); run<2>("v_add_f64", 1); run<3>("v_mul_lo_u32 + v_add_u32", 2);
    run<4>("v_xor_b32 + v_add_u32", 2); run<5>("cvt f64->i32->f64 + mul + add (4 ops)", 4); run<6>("IEEE f64 divide + add (as compiled)", 1);
    run<7>("int64 multiply + add (as compiled)", 1); run<8>("DPP quad_perm + max + add (3 ops)", 3);
    return 0;
}
