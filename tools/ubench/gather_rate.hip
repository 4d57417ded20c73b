// micro-benchmark: cost of scattered (gather) global loads on gfx950 (developer tool).
// Every wave issues independent loads from a table that fits L2 (and optionally L1), with different lane->address
// patterns; reports cycles per wave-instruction per CU, i.e. what one pass through the texture addresser / L1 costs.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2_a4 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: coalesced dwordx4 (lane*16)                         1 instr
// MODE 1: random row of 9 floats per lane: x4 + x4 + x1         3 instr  (the register path's neighbour-row gather)
// MODE 2: random row of 9 floats, 3 lanes per row (x4 each)     1 instr covers 21 rows
// MODE 3: random dword per lane                                 1 instr
// MODE 4: random 16-byte record per lane (x4, 16-B aligned)     1 instr
// MODE 5: random 8-byte record per lane (x2)                    1 instr
// MODE 6: random row of 12 floats per lane (48-B stride, x4 x3) 3 instr
// MODE 7: random row of 16 floats per lane (64-B aligned, x4 x4) 4 instr
// MODE 8: 4 consecutive lanes read consecutive 8-B records of a random list (the pair-record reads)  1 instr
template <int MODE> __global__ __launch_bounds__(256) void k(const float *__restrict__ tab, uint32_t rows_mask, int iters, float *out)
{
    const uint32_t lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.0f;
    constexpr int U = 4;   // independent loads (groups) in flight per lane
    for (int it = 0; it < iters; ++it) {
        float v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t h = mix((uint32_t)(it * U + u) * 0x9E3779B9u + gw * 7919u);
            v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.0f;
            if (MODE == 0) {
                const float4_a4 t = *reinterpret_cast<const float4_a4 *>(tab + ((h & rows_mask) & ~63u) * 4 + lane * 4);
                v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
            } else if (MODE == 1 || MODE == 6 || MODE == 7) {
                const uint32_t stride = MODE == 1 ? 9 : MODE == 6 ? 12 : 16;
                const uint32_t r = mix(h + lane * 0x85ebca6bu) & rows_mask;
                const float *p = tab + (size_t)r * stride;
                const float4_a4 a = *reinterpret_cast<const float4_a4 *>(p), b = *reinterpret_cast<const float4_a4 *>(p + 4);
                v[u][0] = a.x + b.x; v[u][1] = a.y + b.y; v[u][2] = a.z + b.z; v[u][3] = a.w + b.w;
                if (MODE == 1) v[u][0] += p[8];
                else {
                    const float4_a4 c = *reinterpret_cast<const float4_a4 *>(p + 8);
                    v[u][1] += c.x + c.y + c.z + c.w;
                    if (MODE == 7) { const float4_a4 d = *reinterpret_cast<const float4_a4 *>(p + 12); v[u][2] += d.x + d.y + d.z + d.w; }
                }
            } else if (MODE == 2) {
                const uint32_t r = mix(h + (lane / 3) * 0x85ebca6bu) & rows_mask;
                const uint32_t ch = lane % 3;
                const float *p = tab + (size_t)r * 9 + (ch == 2 ? 5 : ch * 4);   // chunks [0,4) [4,8) [5,9)
                const float4_a4 a = *reinterpret_cast<const float4_a4 *>(p);
                v[u][0] = a.x; v[u][1] = a.y; v[u][2] = a.z; v[u][3] = a.w;
            } else if (MODE == 3) {
                const uint32_t r = mix(h + lane * 0x85ebca6bu) & rows_mask;
                v[u][0] = tab[r];
            } else if (MODE == 4) {
                const uint32_t r = mix(h + lane * 0x85ebca6bu) & rows_mask;
                const float4 a = *reinterpret_cast<const float4 *>(tab + (size_t)r * 4);
                v[u][0] = a.x; v[u][1] = a.y; v[u][2] = a.z; v[u][3] = a.w;
            } else if (MODE == 5) {
                const uint32_t r = mix(h + lane * 0x85ebca6bu) & rows_mask;
                const float2 a = *reinterpret_cast<const float2 *>(tab + (size_t)r * 2);
                v[u][0] = a.x; v[u][1] = a.y;
            } else if (MODE == 8) {
                const uint32_t r = (mix(h + (lane & 15) * 0x85ebca6bu) & rows_mask & ~3u) + (lane >> 4);
                const float2 a = *reinterpret_cast<const float2 *>(tab + (size_t)r * 2);
                v[u][0] = a.x; v[u][1] = a.y;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char *name, int instr_per_group, double rows_per_group, uint32_t rows, const float *tab, float *out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 400, grid = 256 * 8;    // 8 workgroups per CU
    k<MODE><<<grid, 256>>>(tab, rows - 1, 10, out);
    hipEventRecord(a); k<MODE><<<grid, 256>>>(tab, rows - 1, iters, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double groups = (double)grid * 4 * iters * 4;      // wave-level groups
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-52s rows %7u: %6.1f cycles / wave-instr / CU, %7.1f cycles per 64 rows / CU\n", name, rows,
           cyc / (groups * instr_per_group / 256.0), cyc / (groups * rows_per_group / 64.0 / 256.0));
}

int main()
{
    float *tab, *out;
    hipMalloc(&tab, 64u << 20); hipMemset(tab, 0, 64u << 20); hipMalloc(&out, 256 * 8 * 256 * 4);
    for (uint32_t rows : {512u, 2048u, 65536u}) {     // 9-float rows: 18 KB (L1), 72 KB (one cloud of cfg2), 2.3 MB (L2)
        run<0>("coalesced x4", 1, 64, rows, tab, out);
        run<3>("random dword per lane", 1, 64, rows, tab, out);
        run<5>("random 8-B record per lane (x2)", 1, 64, rows, tab, out);
        run<4>("random 16-B record per lane (x4)", 1, 64, rows, tab, out);
        run<8>("4 lanes x consecutive 8-B records, 16 random lists", 1, 64, rows, tab, out);
        run<1>("random 9-float row per lane (x4 x4 x1)", 3, 64, rows, tab, out);
        run<2>("random 9-float row per 3 lanes (x4 each)", 1, 21, rows, tab, out);
        run<6>("random 12-float row per lane (x4 x4 x4)", 3, 64, rows, tab, out);
        run<7>("random 16-float row per lane, 64-B aligned (x4 x 4)", 4, 64, rows, tab, out);
    }
    return 0;
}
