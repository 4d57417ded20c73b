// op_boundary_bench.cc -- what a framework that runs the model OP BY OP gets from libconv3p_hip.so, measured without
// Python: the cfg2 step (BASELINE.json config 2: B = 32 x N = 2048, pointcnn2_acsd.py:48-67) as the eight op calls
// TensorFlow's executor would issue through integration/tf_conv3p_shim.cc's default variant -- Conv3p x 4, Conv3pGrad
// x 4 against ONE persistent cache, nothing from the (unchanged) Python caller -- with SELU / SELU-gradient as separate
// ops between them (TensorFlow owns those), a different batch every step, one stream, HIP-event timed.  The shim holds a
// reference to the points tensor it validated last and passes CONV3P_CACHE_POINTS_UNCHANGED for calls on that very
// buffer (tf_conv3p_shim.cc, FlagsFor): here, a call whose `points` pointer equals the previous call's -- every batch
// stays allocated, as the held reference guarantees -- so the first op of a step is validated by content hash (and
// rebuilds), the other seven are not.  4th argument `unhinted`: no call carries the hint (-DCONV3P_SHIM_NO_IDENTITY_HINT,
// the rounds-3-to-5 figure).  bench.py runs this binary outside its timed region and reports both.
//
//   make -C integration op_boundary_bench && integration/op_boundary_bench [steps] [warmup] [clouds.bin|-] [unhinted]
//
// Links the C ABI of include/conv3p.h only (no torch).  Data: the clouds bench.py hands over in a file (the batches of
// its own figures), or, run by hand, points on unit-sphere / box surfaces from a small generator (lighter neighbourhoods
// than pointwise_amd/synth.py's: the figures depend on the neighbour statistics, which it prints).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "conv3p.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define OK(x) do { int r_ = (x); if (r_ != CONV3P_OK) { fprintf(stderr, "%s: %s\n", #x, conv3p_status_string(r_)); return 3; } } while (0)

namespace {
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint32_t u32() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }
    float uni() { return (float)((u32() >> 8) * (1.0 / 16777216.0)); }           // [0, 1)
    float sym() { return 2.0f * uni() - 1.0f; }
    float gauss() { float a = uni() + 1e-7f, b = uni(); return std::sqrt(-2.0f * std::log(a)) * std::cos(6.2831853f * b); }
};
// surface-sampled objects normalised to the unit ball: a mixture of a sphere and a box per cloud
void make_clouds(std::vector<float> &P, int B, int N, uint64_t seed)
{
    P.resize((size_t)B * N * 3);
    Rng r(seed);
    for (int b = 0; b < B; ++b) {
        const float mix = r.uni();
        const float ex = 0.4f + 0.5f * r.uni(), ey = 0.4f + 0.5f * r.uni(), ez = 0.4f + 0.5f * r.uni();
        for (int i = 0; i < N; ++i) {
            float x, y, z;
            if (r.uni() < mix) {
                x = r.gauss(); y = r.gauss(); z = r.gauss();
                const float n = std::sqrt(x * x + y * y + z * z) + 1e-9f;
                x /= n; y /= n; z /= n;
            } else {
                x = ex * r.sym(); y = ey * r.sym(); z = ez * r.sym();
                const int face = (int)(r.u32() % 6u);
                if (face < 2) x = face == 0 ? -ex : ex;
                else if (face < 4) y = face == 2 ? -ey : ey;
                else z = face == 4 ? -ez : ez;
                const float n = std::sqrt(ex * ex + ey * ey + ez * ez);
                x /= n; y /= n; z /= n;
            }
            float *p = &P[((size_t)b * N + i) * 3];
            p[0] = x; p[1] = y; p[2] = z;
        }
    }
}
}  // namespace

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 50, warmup = argc > 2 ? atoi(argv[2]) : 10;
    const int B = 32, N = 2048, H = 9, NB = 3;   // NB batches cycled: every step meets new points
    const int cin[4] = {3, H, H, H};
    const int32_t strides[4][3] = {{1, 1, 1}, {2, 2, 2}, {3, 3, 3}, {4, 4, 4}};
    const float voxel = 0.1f;
    hipStream_t s;
    HIPOK(hipStreamCreate(&s));

    const size_t rows = (size_t)B * N;
    float *dP[NB], *dUp[4], *dAct[4], *dPre[4], *dG, *dCarry[2], *dW[4], *dGW[4];
    // optional 3rd argument: the clouds to use, as bench.py writes them (int32 {batches, B, N}, then float32 xyz) -- the
    // same batches its other figures are measured on; without it: the generator above
    const bool have_file = argc > 3 && std::string(argv[3]) != "-";
    const bool identity_hint = !(argc > 4 && std::string(argv[4]) == "unhinted");
    FILE *df = have_file ? fopen(argv[3], "rb") : nullptr;
    if (have_file) {
        int32_t hdr[3] = {0, 0, 0};
        if (!df || fread(hdr, 4, 3, df) != 3 || hdr[0] < NB || hdr[1] != B || hdr[2] != N) {
            fprintf(stderr, "%s: not a {>=%d, %d, %d} cloud file\n", argv[3], NB, B, N);
            return 1;
        }
    }
    for (int i = 0; i < NB; ++i) {
        std::vector<float> P;
        if (df) {
            P.resize(rows * 3);
            if (fread(P.data(), 4, rows * 3, df) != rows * 3) return 1;
        } else {
            make_clouds(P, B, N, 40 + i);
        }
        HIPOK(hipMalloc(&dP[i], rows * 3 * 4));
        HIPOK(hipMemcpy(dP[i], P.data(), rows * 3 * 4, hipMemcpyHostToDevice));
    }
    Rng r(7);
    for (int l = 0; l < 4; ++l) {
        std::vector<float> w((size_t)27 * cin[l] * H), up(rows * H);
        const float a = std::sqrt(3.0f / (27.0f * cin[l]));
        for (auto &v : w) v = a * r.sym();
        for (auto &v : up) v = r.sym();
        HIPOK(hipMalloc(&dW[l], w.size() * 4));
        HIPOK(hipMemcpy(dW[l], w.data(), w.size() * 4, hipMemcpyHostToDevice));
        HIPOK(hipMalloc(&dGW[l], w.size() * 4));
        HIPOK(hipMalloc(&dUp[l], rows * H * 4));
        HIPOK(hipMemcpy(dUp[l], up.data(), rows * H * 4, hipMemcpyHostToDevice));
        HIPOK(hipMalloc(&dAct[l], rows * H * 4));
        HIPOK(hipMalloc(&dPre[l], rows * H * 4));
    }
    HIPOK(hipMalloc(&dG, rows * H * 4));
    HIPOK(hipMalloc(&dCarry[0], rows * H * 4));
    HIPOK(hipMalloc(&dCarry[1], rows * H * 4));

    conv3p_cache_config cfg = {4, 27, 0, H, H, 0};   // four stencils, no hints: what the shim's default variant passes
    const size_t cbytes = conv3p_cache_bytes(4, B, N, &cfg);
    void *cache = nullptr;
    HIPOK(hipMalloc(&cache, cbytes));
    OK(conv3p_cache_init(cache, cbytes, s));

    const float *held = nullptr;   // the shim's held points buffer
    auto flags_for = [&](const float *P) {
        conv3p_cache_config c = cfg;
        if (identity_hint && held == P) c.flags |= CONV3P_CACHE_POINTS_UNCHANGED;
        held = P;
        return c;
    };
    auto step = [&](int it) -> int {
        const float *P = dP[it % NB];
        const float *x = P;   // layer 1: input == points (modelnet_provider.py:212-213)
        for (int l = 0; l < 4; ++l) {
            const conv3p_cache_config cf = flags_for(P);
            OK(conv3p_forward_cached_f32(P, x, dW[l], strides[l], voxel, B, N, cin[l], H, 3, 3, 3, dPre[l], cache, cbytes, &cf, s));
            OK(conv3p_selu_f32(dPre[l], dAct[l], rows * H, s));
            x = dAct[l];
        }
        const float *carry = nullptr;
        for (int l = 3; l >= 0; --l) {
            if (carry) OK(conv3p_selu_grad_add_f32(dAct[l], dUp[l], carry, dG, rows * H, s));
            else OK(conv3p_selu_grad_f32(dAct[l], dUp[l], dG, rows * H, s));
            float *dx = dCarry[l & 1];
            const conv3p_cache_config cf = flags_for(P);
            OK(conv3p_backward_cached_f32(dG, P, l > 0 ? dAct[l - 1] : P, dW[l], strides[l], voxel, B, N, cin[l], H, 3, 3, 3, dx,
                                          dGW[l], cache, cbytes, &cf, s));
            carry = dx;
        }
        return 0;
    };
    for (int i = 0; i < warmup; ++i)
        if (int rc = step(i)) return rc;
    hipEvent_t e0, e1;
    HIPOK(hipEventCreate(&e0));
    HIPOK(hipEventCreate(&e1));
    HIPOK(hipStreamSynchronize(s));
    HIPOK(hipEventRecord(e0, s));
    const auto h0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i)
        if (int rc = step(warmup + i)) return rc;
    const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    HIPOK(hipEventRecord(e1, s));
    HIPOK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPOK(hipEventElapsedTime(&ms, e0, e1));

    // neighbour statistics of the first batch (they drive the cost): mean population of layer 2's stencil
    std::vector<int32_t> cnt(rows * 27);
    int32_t *dcnt = nullptr;
    HIPOK(hipMalloc(&dcnt, rows * 27 * 4));
    const size_t wsb = conv3p_workspace_bytes(CONV3P_PASS_NEIGHBOR_COUNT, 4, B, N, 0, 0, 3, 3, 3);
    void *ws = nullptr;
    HIPOK(hipMalloc(&ws, wsb));
    OK(conv3p_neighbor_count_f32(dP[0], strides[1], voxel, B, N, 3, 3, 3, dcnt, ws, wsb, s));
    HIPOK(hipStreamSynchronize(s));
    HIPOK(hipMemcpy(cnt.data(), dcnt, rows * 27 * 4, hipMemcpyDeviceToHost));
    double tot = 0;
    for (int32_t v : cnt) tot += v;
    printf("{\"op_boundary_native_ms_per_step\": %.4f, \"host_enqueue_ms_per_step\": %.4f, \"steps\": %d, \"warmup\": %d, \"workload\": \"cfg2: B=32 x N=2048, 4 x Conv3p + 4 x "
           "Conv3pGrad through the *_cached_* entry points (one persistent cache; %s), SELU / SELU-grad as separate "
           "ops, a different batch every step, one stream, HIP events\", \"neighbours_per_point_stride2\": %.2f}\n",
           ms / steps, host_ms / steps, steps, warmup,
           identity_hint ? "POINTS_UNCHANGED for calls on the points buffer the shim holds, the first op of a step validated by content hash"
                         : "no call hinted: every call validated by content hash",
           tot / (double)rows);
    return 0;
}
