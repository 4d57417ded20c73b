// Driver around the reference's OWN accumulation loops -- TEST INFRASTRUCTURE ONLY.
//
// This file is never compiled on its own.  oracle/Makefile builds ONE translation unit on the
// fly, on stdin, by splicing line ranges of the reference source -- read in place from
// /root/reference, never written to disk -- between the parts of this file:
//
//     reference lines  A   std includes, CpuAlloc, Array, Grid            (no TensorFlow symbol)
//     this file, part 1    FlatView + head of ref_forward<T>() with its locals
//     reference lines  F   the forward  batch loop of Conv3pOp::Compute     (memset + loop)
//     this file, part 2    tail of ref_forward, head of ref_backward<T>() with its locals
//     reference lines  B   the backward batch loop of Conv3pGradOp::Compute (+ OpenMP reduction)
//     this file, part 3    tail of ref_backward, extern "C" entry points
//
//     variant    A                          F                           B
//     atrous     tf_conv3p_atrous.cpp:7-388  tf_conv3p_atrous.cpp:451-504  tf_conv3p_atrous.cpp:608-716
//     plain      tf_conv3p_grid.cpp:7-381    tf_conv3p_grid.cpp:438-491    tf_conv3p_grid.cpp:585-689
//
// What runs is therefore the reference's loop TEXT, unmodified: the per-term `w * x / (T)fsize`,
// the c-outer / k-inner order, the ii-centred re-binning without an inclusion re-test, the hole
// test, the `count == 0` skip, the per-thread grad_filter partials and their tid-ordered sum.
//
// What is the DRIVER's (stated plainly, so that a reviewer can rule on it): the function
// signatures and the LOCAL VARIABLES those lines read and write.  In the reference these locals
// are initialised from TensorFlow tensors (Compute(), .cpp:409-450 / :529-606: dim_size() calls,
// flat<T>() views, the two output allocations); here they are initialised from plain pointers and
// ints with the same names, types and meanings.  `*_flat` are callables returning `T&` for element
// i, which is all the spliced lines use of Eigen's TensorMap.  No TensorFlow header, type or macro
// is declared, faked or stood in for; the Compute() prologues (argument unpacking, OP_REQUIRES)
// are NOT executed (the host mirror's checks are tested separately, tests/test_host.py).  The
// backward's two output memsets (.cpp:580, :590) sit between OP_REQUIRES lines and are restated
// by the entry points below.
//
// Outputs: oracle/_ref/libref_compute_{atrous,plain,atrous_omp}.so (git-ignored, travel with the
// snapshot).  The *_omp variant is built exactly like the reference CPU object
// (tf_conv3p_compile.sh:32: -O3 -fopenmp -DCONV_OPENMP) and is what bench.py times as
// cpu_baseline.kind = "reference".

template <typename T> struct FlatView {
    T *base;
    T &operator()(long i) const { return base[i]; }
};

template <typename T>
static void ref_forward(const T *points_p, const T *input_p, const T *filter_p, const int *stride_p,
                        T voxel_p, int B, int N, int Cin, int Cout, int fz, int fy, int fx, T *output_p)
{
    // locals of Conv3pOp::Compute (.cpp:409-450), same names
    int batch_size = B;
    int num_points = N;
    FlatView<const T> points_flat{points_p};
    FlatView<const T> input_flat{input_p};
    int filter_z = fz, filter_y = fy, filter_x = fx;
    int filter_c_in = Cin, filter_c_out = Cout;
    int filter_count = filter_x * filter_y * filter_z;
    const T *filter = filter_p;
    int stride_x = stride_p[0], stride_y = stride_p[1], stride_z = stride_p[2];
    T voxel_size = voxel_p;
    FlatView<T> output_flat{output_p};
    (void)stride_x; (void)stride_y; (void)stride_z; (void)filter_z; (void)filter_y;
//@@REF_FORWARD_BODY@@
}

template <typename T>
static void ref_backward(const T *grad_p, const T *points_p, const T *input_p, const T *filter_p,
                         const int *stride_p, T voxel_p, int B, int N, int Cin, int Cout, int fz, int fy,
                         int fx, T *grad_input_p, T *grad_filter_p)
{
    // locals of Conv3pGradOp::Compute (.cpp:529-606), same names
    FlatView<const T> grad_from_next_tensor_flat{grad_p};
    FlatView<const T> points_flat{points_p};
    FlatView<const T> input_flat{input_p};
    int batch_size = B;
    int num_points = N;
    const T *filter_arr = filter_p;
    int stride_x = stride_p[0], stride_y = stride_p[1], stride_z = stride_p[2];
    T voxel_size = voxel_p;
    int filter_z = fz, filter_y = fy, filter_x = fx;
    int filter_full_x = (filter_x - 1) * stride_x + 1;
    int filter_full_y = (filter_y - 1) * stride_y + 1;
    int filter_full_z = (filter_z - 1) * stride_z + 1;
    int filter_c_in = Cin, filter_c_out = Cout;
    int n_weights = filter_z * filter_y * filter_x * filter_c_in * filter_c_out;
    int filter_count = filter_x * filter_y * filter_z;
    FlatView<T> grad_input_flat{grad_input_p};
    T *grad_filter_arr = grad_filter_p;
    (void)filter_full_x; (void)filter_full_y; (void)filter_full_z;
    // .cpp:580 / :590 (the reference zeroes both outputs before accumulating)
    memset(grad_input_p, 0, sizeof(T) * (size_t)B * N * Cin);
    memset(grad_filter_p, 0, sizeof(T) * (size_t)n_weights);
//@@REF_BACKWARD_BODY@@
}

#define REF_ENTRY(SFX, T)                                                                                   \
    extern "C" int ref_compute_forward_##SFX(const T *points, const T *input, const T *filter,              \
                                             const int *stride_xyz, T voxel, int B, int N, int Cin,         \
                                             int Cout, int fz, int fy, int fx, T *output)                   \
    {                                                                                                       \
        ref_forward<T>(points, input, filter, stride_xyz, voxel, B, N, Cin, Cout, fz, fy, fx, output);      \
        return 0;                                                                                           \
    }                                                                                                       \
    extern "C" int ref_compute_backward_##SFX(const T *grad, const T *points, const T *input,               \
                                              const T *filter, const int *stride_xyz, T voxel, int B,       \
                                              int N, int Cin, int Cout, int fz, int fy, int fx,             \
                                              T *grad_input, T *grad_filter)                                \
    {                                                                                                       \
        ref_backward<T>(grad, points, input, filter, stride_xyz, voxel, B, N, Cin, Cout, fz, fy, fx,        \
                        grad_input, grad_filter);                                                           \
        return 0;                                                                                           \
    }
REF_ENTRY(f32, float)
REF_ENTRY(f64, double)

extern "C" int ref_compute_is_plain(void)
{
#ifdef REF_PLAIN
    return 1;
#else
    return 0;
#endif
}
extern "C" int ref_compute_is_openmp(void)
{
#ifdef CONV_OPENMP
    return 1;
#else
    return 0;
#endif
}
extern "C" int ref_compute_threads(void)
{
#ifdef CONV_OPENMP
    return (int)std::thread::hardware_concurrency();   // what Conv3pGradOp forces, .cpp:611-619
#else
    return 1;
#endif
}
