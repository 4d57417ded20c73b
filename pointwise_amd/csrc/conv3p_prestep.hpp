// conv3p_prestep.hpp -- the host pre-step of the reference's data providers, as gfx950 kernels.
//
// The reference prepares every batch on the host with per-cloud Python loops:
//   rotate_point_cloud / jitter_point_cloud   /root/reference/modelnet_provider.py:23-75   (training augmentation)
//   sort_point_cloud_xyz / sort_point_cloud_xyz2   /root/reference/util.py:55-109          (optional cloud ordering)
// Once the op itself takes well under a millisecond per step these loops are the feed-side bottleneck, so the same
// transformations are provided on the device, operating on the batch where it already lives.  Random numbers stay
// the caller's (rotation angles on the host, Gaussian noise as a device tensor): the kernels are deterministic
// functions of their inputs, which is what makes them checkable against the reference functions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conv3p {

// out[b,i,:] = (float)( clip(sigma * noise[b,i,:], -clip, +clip) + (double)(float)(p[b,i,:] . R_b) )
//   R_b = [[c,0,s],[0,1,0],[-s,0,c]] (rotation about the up axis, modelnet_provider.py:34-39); the product is
//   evaluated in double and stored as float32 (rotated_data is a float32 array, :32), the jitter is added in
//   double (np.random.randn is float64, :73-74) and the sum becomes float32 when the batch is fed.
//   cs[b] = {cos, sin} (host-computed, device array); cs == nullptr: no rotation; noise == nullptr: no jitter.
__global__ __launch_bounds__(256) void augment_kernel(const float *in, const double2 *__restrict__ cs,
                                                      const double *__restrict__ noise, double sigma, double clip,
                                                      float *out,   // may alias `in` (a thread reads its point before it writes it)
                                                      size_t total, int N)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        float r[3] = {x, y, z};
        if (cs != nullptr) {
            const double2 t = cs[i / (size_t)N];
            // row vector times matrix, terms added in index order (np.dot on an (N,3) x (3,3) product)
            r[0] = (float)(((double)x * t.x + (double)y * 0.0) + (double)z * -t.y);
            r[1] = (float)(((double)x * 0.0 + (double)y * 1.0) + (double)z * 0.0);
            r[2] = (float)(((double)x * t.y + (double)y * 0.0) + (double)z * t.x);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double v = (double)r[a];
            if (noise != nullptr) {
                double j = sigma * noise[3 * i + a];
                j = j < -clip ? -clip : (j > clip ? clip : j);          // np.clip
                v = j + v;
            }
            out[3 * i + a] = (float)v;
        }
    }
}

// Total order of float keys as unsigned integers (negative values reversed, -0 < +0, NaN after +inf like numpy's sort).
__device__ __forceinline__ uint32_t float_key(float v)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFEu;   // every NaN (either sign) sorts last, as numpy.argsort puts it
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct SortKey {
    uint32_t x, y, z, idx;
};
__device__ __forceinline__ bool key_less(const SortKey &a, const SortKey &b)
{
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.z != b.z) return a.z < b.z;
    return a.idx < b.idx;
}

// order[b][r] = index of the point of cloud b that comes r-th when the cloud is sorted by x, ties by y, ties by z
// (util.py:66-68: argsort by z, then STABLE argsort by y, then by x), remaining ties by original index.  One
// workgroup per cloud, bitonic network on 16-byte keys in LDS: npad (power of two >= N) * 16 bytes, N <= 8192.
// `data` rows have `ld` floats, the first three being x, y, z.  -0.0 sorts with +0.0 as in numpy (equal keys).
__global__ __launch_bounds__(1024) void sort_xyz_kernel(const float *__restrict__ data, int N, int ld, int npad,
                                                        int32_t *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SortKey *keys = reinterpret_cast<SortKey *>(smem);
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const float *cloud = data + (size_t)b * N * ld;
    for (int i = tid; i < npad; i += nthr) {
        SortKey k{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // padding sorts last
        if (i < N) {
            // numpy compares values: -0.0 == +0.0 (adding +0.0 turns -0.0 into +0.0 and changes nothing else)
            k.x = float_key(cloud[(size_t)i * ld + 0] + 0.0f);
            k.y = float_key(cloud[(size_t)i * ld + 1] + 0.0f);
            k.z = float_key(cloud[(size_t)i * ld + 2] + 0.0f);
            k.idx = (uint32_t)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (npad >> 1); t += nthr) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const SortKey a = keys[i], c = keys[l];
                const bool up = (i & k) == 0;
                if (key_less(c, a) == up) {
                    keys[i] = c;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < N; i += nthr) order[(size_t)b * N + i] = (int32_t)keys[i].idx;
}

// dst[b][r][:] = src[b][order[b][r]][:], rows of `row_bytes` bytes (any element type: points, labels, attributes)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char *__restrict__ src, const int32_t *__restrict__ order,
                                                          int N, int row_bytes, char *__restrict__ dst, size_t rows)
{
    const size_t total = rows * (size_t)row_bytes;
    if ((row_bytes & 3) == 0) {
        const int rw = row_bytes >> 2;
        const size_t tw = rows * (size_t)rw;
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d = reinterpret_cast<uint32_t *>(dst);
        for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tw; e += (size_t)gridDim.x * blockDim.x) {
            const size_t r = e / (size_t)rw;
            const size_t b = r / (size_t)N;
            d[e] = s[(b * N + (size_t)order[r]) * rw + (e - r * rw)];
        }
        return;
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)row_bytes;
        const size_t b = r / (size_t)N;
        dst[e] = src[(b * N + (size_t)order[r]) * row_bytes + (e - r * row_bytes)];
    }
}

}  // namespace conv3p
