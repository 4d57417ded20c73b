// developer microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate with NACC independent accumulators per wave
//   mode 0: accumulators wherever the compiler puts them (AGPRs), constant operands
//   mode 1: accumulators forced into VGPRs (inline asm "+v"), constant operands
//   mode 2: mode 0 + A/B operands read from LDS every step
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_rate tools/ubench/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float lds[4096];
    for (int e = threadIdx.x; e < 4096; e += 256) lds[e] = e * 1e-4f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            a = lds[(it * 64 + threadIdx.x) & 4095];
            b = lds[(it * 64 + threadIdx.x + 1024) & 4095];
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 1)
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int MODE> void run(int wgs, float *d)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(wgs), dim3(256), 0, 0, d, 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(wgs), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * 4 * iters * NACC * 4096.0;
    printf("mode %d NACC=%d wgs=%d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", MODE, NACC, wgs, wgs / 256, ms, flop / ms * 1e-9);
}
int main()
{
    float *d; (void)hipMalloc(&d, 4096 * 256 * 4);
    run<8, 0>(256, d); run<8, 0>(512, d);
    run<8, 1>(256, d); run<8, 1>(512, d);
    run<8, 2>(256, d); run<8, 2>(512, d);
    run<4, 1>(512, d); run<2, 1>(512, d);
    return 0;
}
