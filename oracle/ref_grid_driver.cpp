// Driver around the reference's OWN Grid template -- TEST INFRASTRUCTURE ONLY.
//
// This file is never compiled on its own.  oracle/Makefile builds ONE translation
// unit on the fly, on stdin, as
//     <TF-free line range of the reference .cpp, read in place>  +  <this file>
// so the template that runs is the reference's code, unmodified:
//     tf_conv3p_atrous.cpp:7-388  (std includes, CpuAlloc, Array, Grid)   -> REF_ATROUS
//     tf_conv3p_grid.cpp:7-381    (the non-atrous twin)                   -> REF_PLAIN
// Those ranges contain no TensorFlow symbol.  Lines 1-5 (the TF includes) and
// everything from `using namespace tensorflow;` on (the OpKernel classes) are
// NOT compiled: TensorFlow is absent from this image, so the full op is
// unbuildable here and no stand-in headers are written.  Nothing from the
// reference is copied into the repository; the only outputs are the shared
// objects under oracle/_ref/ (git-ignored).
//
// What this pins: Grid::Grid, Grid::neighbor, Grid::neighbor_count -- i.e. the
// bit-sensitive part of the op (cell ids, box edges, inclusive test, tap index,
// clamp, hole test, visit order, per-tap counts).

template <typename T>
static long ref_lists(const T *pts, int n, T voxel, int fx, int fy, int fz, int sx, int sy, int sz,
                      long *offsets, int *idx, int *tap, long cap, int *counts)
{
    const int ntap = fx * fy * fz;
    Grid<CpuAlloc, T> grid(Array<CpuAlloc, T>((T *)pts, n), voxel);
    Array<CpuAlloc, int> point_index, filter_cell, filter_cell_count;
    point_index.alloc(n);
    filter_cell.alloc(n);
    filter_cell_count.resize(ntap);
    long total = 0;
    offsets[0] = 0;
    for (int i = 0; i < n; ++i) {
#ifdef REF_PLAIN
        (void)sx; (void)sy; (void)sz;
        grid.neighbor(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], fx, fy, fz, voxel,
                      point_index, filter_cell, filter_cell_count);
#else
        grid.neighbor(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], fx, fy, fz, sx, sy, sz, voxel,
                      point_index, filter_cell, filter_cell_count);
#endif
        if (total + point_index.size > cap) { total = -2; break; }
        for (int t = 0; t < point_index.size; ++t) {
            idx[total + t] = point_index[t];
            tap[total + t] = filter_cell[t];
        }
        total += point_index.size;
        offsets[i + 1] = total;
    }
    if (counts && total >= 0) {
        Array<CpuAlloc, int> all(counts, n * ntap);
#ifdef REF_PLAIN
        grid.neighbor_count(fx, fy, fz, voxel, all);
#else
        grid.neighbor_count(fx, fy, fz, sx, sy, sz, voxel, all);
#endif
    }
    point_index.free();
    filter_cell.free();
    filter_cell_count.free();
    return total;
}

extern "C" long ref_grid_lists_f32(const float *pts, int n, float voxel, int fx, int fy, int fz,
                                   int sx, int sy, int sz, long *offsets, int *idx, int *tap,
                                   long cap, int *counts)
{
    return ref_lists<float>(pts, n, voxel, fx, fy, fz, sx, sy, sz, offsets, idx, tap, cap, counts);
}
extern "C" long ref_grid_lists_f64(const double *pts, int n, double voxel, int fx, int fy, int fz,
                                   int sx, int sy, int sz, long *offsets, int *idx, int *tap,
                                   long cap, int *counts)
{
    return ref_lists<double>(pts, n, voxel, fx, fy, fz, sx, sy, sz, offsets, idx, tap, cap, counts);
}
extern "C" int ref_grid_is_plain(void)
{
#ifdef REF_PLAIN
    return 1;
#else
    return 0;
#endif
}
