// micro-benchmark: issue rate of VALU ops on gfx950 as hipcc emits them (developer tool).
// 8 independent dependency chains per lane, 16 waves per SIMD: measures throughput, not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[CHAINS];
    float2v p[CHAINS];
    unsigned u[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = float2v{x[i], x[i] + 1}; u[i] = threadIdx.x + i; }
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 2) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
                if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (OP == 4) asm volatile("v_max3_f32 %0, |%0|, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (OP == 5) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : "v"(a), "v"(b) : "vcc");
                if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(float2v{a, a}), "v"(float2v{b, b}));
                if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(float2v{a, a}));
                if (OP == 8) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
                if (OP == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
                if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ua));
                if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 13) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(ua));
                if (OP == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ua) : );
                if (OP == 17) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
                if (OP == 18) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(float2v{a, a}));
                if (OP == 19) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(u[i]));
                if (OP == 20) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(ub));
                if (OP == 21) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            }
    }
    float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i] + p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int per)
{
    static float *d = nullptr; if (!d) hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 1000;
    k<OP><<<4096, 256>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(a); k<OP><<<4096, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winstr = 4096.0 * 4 * iters * 8 * CHAINS * per;
    printf("%-26s %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / (winstr / 1024.0));
}
int main() {
    run<0>("v_fma_f32", 1); run<17>("v_fma_f32 (sgpr src)", 1); run<12>("v_mul_f32", 1); run<13>("v_add_f32", 1); run<1>("v_sub_f32", 1);
    run<14>("v_max_f32", 1); run<2>("v_rndne_f32", 1); run<8>("v_fract_f32", 1); run<19>("v_cvt_i32_f32", 1); run<3>("v_med3_f32", 1);
    run<4>("v_max3_f32 |x|", 1); run<21>("v_min3_f32", 1); run<5>("v_cmp + v_addc (2 instr)", 2); run<16>("v_cndmask_b32", 1); run<15>("v_alignbit_b32", 1);
    run<9>("v_and_b32", 1); run<10>("v_add_u32", 1); run<11>("v_and_or_b32", 1); run<20>("v_sad_u32", 1);
    run<6>("v_pk_fma_f32", 1); run<7>("v_pk_add_f32", 1); run<18>("v_pk_mul_f32", 1);
    return 0; }
