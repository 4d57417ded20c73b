// tf_conv3p_shim.cc -- TensorFlow custom-op shim over the C ABI of include/conv3p.h.
//
// NOT BUILT IN THIS REPOSITORY: TensorFlow (headers + libtensorflow_framework) is not part of the
// image, so this file has never been compiled here; it is the binding a maintainer of
// hkust-vgd/pointwise adds to get a drop-in tf_conv3p.so (see INTEGRATION.md for the build line).
//
// It keeps the reference's operator surface exactly:
//   REGISTER_OP("Conv3p") / REGISTER_OP("Conv3pGrad")  -- same op names, attr T:{float,double},
//   same input names, order and dtypes, same outputs   (tf_ops/conv3p/register_op.cpp:44-75),
// so pointcnn2_acsd.py:10-31 and scene_seg/pointcnn_scene_seg_acsd.py:9-30 load it unchanged
// (tf.load_op_library + conv3p_module.conv3p / conv3p_grad + the Python-side RegisterGradient).
//
// Differences from the reference's registration, all supersets:
//   * `stride` and `voxel_size` are declared HostMemory on the GPU kernel, so their values are
//     read without the blocking cudaMemcpy D2H of tf_conv3p_atrous.cu:577,586;
//   * a shape function is attached (the reference registers none);
//   * DEFAULT: the neighbour geometry lives in a persistent cache tensor per (device, B, N, dtype, taps) -- a
//     process-global map, allocate_persistent, the conv3p_*_cached_* entry points.  The Python caller gives no hint.
//     The shim derives one from TensorFlow's own guarantees: it KEEPS A REFERENCE to the `points` tensor it validated
//     last (a Tensor copy shares and reference-counts the TensorBuffer).  While that reference is held the allocator
//     cannot hand the buffer to anyone else, and a buffer with more than one reference is never written in place
//     (input tensors are immutable values; in-place forwarding needs a reference count of one).  A call whose `points`
//     shares that very buffer (same bytes: Tensor::SharesBufferWith + equal data pointer and size) therefore sees
//     unchanged content and passes CONV3P_CACHE_POINTS_UNCHANGED -- no hash launch and none of the three launches
//     that would find nothing to do (21 us per op call on cfg2).  Any other `points` is validated on the device by
//     content hash as before and becomes the held one: the first op of a step pays one hash + sort + one search launch
//     for all strides, the other seven nothing (bench.py: op_boundary_native_ms_per_step, and
//     op_boundary_native_unhinted_ms_per_step for -DCONV3P_SHIM_NO_IDENTITY_HINT, every call hashed);
//   * -DCONV3P_SHIM_STATELESS: the stateless entry points, scratch from one allocate_temp per Compute
//     (conv3p_workspace_bytes), nothing kept between calls.
#include <map>
#include <mutex>
#include <tuple>
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/register_types.h"
#include "tensorflow/core/framework/shape_inference.h"

#include "conv3p.h"   // -I <repo>/include

using namespace tensorflow;

REGISTER_OP("Conv3p")
    .Attr("T: {float, double}")
    .Input("points: T")
    .Input("input: T")
    .Input("filter: T")
    .Input("stride: int32")
    .Input("voxel_size: T")
    .Output("output: T")
    .SetShapeFn([](shape_inference::InferenceContext *c) {
        shape_inference::ShapeHandle pts, flt;
        TF_RETURN_IF_ERROR(c->WithRank(c->input(0), 3, &pts));
        TF_RETURN_IF_ERROR(c->WithRank(c->input(2), 5, &flt));
        c->set_output(0, c->MakeShape({c->Dim(pts, 0), c->Dim(pts, 1), c->Dim(flt, 4)}));
        return Status::OK();
    });

REGISTER_OP("Conv3pGrad")
    .Attr("T: {float, double}")
    .Input("grad_from_next: T")
    .Input("points: T")
    .Input("input: T")
    .Input("filter: T")
    .Input("stride: int32")
    .Input("voxel_size: T")
    .Output("grad_input: T")
    .Output("grad_filter: T")
    .SetShapeFn([](shape_inference::InferenceContext *c) {
        c->set_output(0, c->input(2));
        c->set_output(1, c->input(3));
        return Status::OK();
    });

namespace {

template <typename T> struct Abi;
template <> struct Abi<float> {
    static constexpr int kElem = 4;
    static constexpr auto forward = conv3p_forward_f32;
    static constexpr auto backward = conv3p_backward_f32;
    static constexpr auto forward_cached = conv3p_forward_cached_f32;
    static constexpr auto backward_cached = conv3p_backward_cached_f32;
};
template <> struct Abi<double> {
    static constexpr int kElem = 8;
    static constexpr auto forward = conv3p_forward_f64;
    static constexpr auto backward = conv3p_backward_f64;
    static constexpr auto forward_cached = conv3p_forward_cached_f64;
    static constexpr auto backward_cached = conv3p_backward_cached_f64;
};

#ifndef CONV3P_SHIM_STATELESS
// One persistent neighbour cache per (device context, B, N, element size, taps): every conv3p op of a model step is
// fed by the same points tensor (pointcnn2_acsd.py:48-66), so the ops of a step find the sort and their stride's
// search done by the first op that needed them; a new batch changes the content hash and everything is rebuilt.
// The cache is sized for the widest layer seen so far and re-allocated when a wider one arrives.  `mu` is held
// across the whole C-ABI call: two Compute()s enqueueing on one stream must not interleave their kernels on the
// same cache (they share its scratch region).
struct CacheSlot {
    PersistentTensor tensor;
    size_t bytes = 0;
    conv3p_cache_config cfg{};
    Tensor held_points;        // the points tensor whose content the cache was last validated against (reference held)
};

// flags for this call: CONV3P_CACHE_POINTS_UNCHANGED iff `points` IS the buffer validated last (see the file header)
int FlagsFor(CacheSlot *cs, const Tensor &points, const void *data)
{
#ifdef CONV3P_SHIM_NO_IDENTITY_HINT
    (void)cs; (void)points; (void)data;
    return 0;
#else
    if (cs->held_points.IsInitialized() && cs->held_points.SharesBufferWith(points) &&
        cs->held_points.flat<int8>().data() == static_cast<const int8 *>(data) && cs->held_points.TotalBytes() == points.TotalBytes())
        return CONV3P_CACHE_POINTS_UNCHANGED;
    cs->held_points = points;  // drops the reference to the previous batch's tensor
    return 0;
#endif
}
typedef std::tuple<const void *, int, int, int, int> CacheKey;
std::mutex g_cache_mu;
std::map<CacheKey, CacheSlot> g_cache;

void *StreamOf(OpKernelContext *ctx);

// returns the 256-byte aligned cache buffer for this call (allocating or growing it), or nullptr with ctx failed
char *CacheFor(OpKernelContext *ctx, int elem, int B, int N, int taps, int Cin, int Cout, CacheSlot **slot)
{
    const CacheKey key(ctx->op_device_context(), B, N, elem, taps);
    CacheSlot &cs = g_cache[key];
    if (cs.bytes == 0 || cs.cfg.max_Cin < Cin || cs.cfg.max_Cout < Cout) {
        conv3p_cache_config cfg{};
        cfg.slots = 8;                       // strides 1..4 of the models and room to spare (LRU beyond)
        cfg.max_taps = taps;
        cfg.pairs_per_point = 0;             // default capacity
        cfg.max_Cin = Cin > cs.cfg.max_Cin ? Cin : cs.cfg.max_Cin;
        cfg.max_Cout = Cout > cs.cfg.max_Cout ? Cout : cs.cfg.max_Cout;
        cfg.flags = 0;                       // per call: CONV3P_CACHE_POINTS_UNCHANGED only by FlagsFor(); no kernel hint: the
                                             // library picks the backward kernel on the device from the lists
        const size_t need = conv3p_cache_bytes(elem, B, N, &cfg);
        if (need == 0) {
            ctx->CtxFailure(errors::InvalidArgument("Conv3p: cannot size the neighbour cache"));
            return nullptr;
        }
        if (cs.bytes != 0) {
            Tensor *old = cs.tensor.AccessTensor(ctx);
            char *op_ = reinterpret_cast<char *>(old->flat<int8>().data());
            (void)conv3p_cache_forget(op_ + (256 - reinterpret_cast<uintptr_t>(op_) % 256) % 256);   // the address the library saw
        }
        Tensor *fresh = nullptr;
        const Status st = ctx->allocate_persistent(DT_INT8, TensorShape({(int64)need + 256}), &cs.tensor, &fresh);
        if (!st.ok()) {
            cs.bytes = 0;
            ctx->CtxFailureWithWarning(st);
            return nullptr;
        }
        // The allocator may have recycled a region that overlaps a cache freed earlier (e.g. the one just outgrown):
        // its hashes and slot marks can survive while its lists were overwritten by temporaries, and the device-side
        // validation would then trust them.  include/conv3p.h asks for a zero-filled buffer; conv3p_cache_init does it on
        // the op's stream, ordered before the search that follows.
        char *raw = reinterpret_cast<char *>(fresh->flat<int8>().data());
        char *aligned = raw + (256 - reinterpret_cast<uintptr_t>(raw) % 256) % 256;
        const int rc = conv3p_cache_init(aligned, need, StreamOf(ctx));
        if (rc != CONV3P_OK) {
            cs.bytes = 0;
            ctx->CtxFailure(errors::Internal("Conv3p: cannot initialise the neighbour cache: ", conv3p_status_string(rc)));
            return nullptr;
        }
        cs.bytes = need;
        cs.cfg = cfg;
        cs.held_points = Tensor();   // a new cache buffer has validated nothing yet
    }
    *slot = &cs;
    char *p = reinterpret_cast<char *>(cs.tensor.AccessTensor(ctx)->flat<int8>().data());
    return p + (256 - reinterpret_cast<uintptr_t>(p) % 256) % 256;
}
#endif

Status FromCode(int rc, const char *what)
{
    if (rc == CONV3P_OK) return Status::OK();
    if (rc == CONV3P_ERR_INVALID_ARGUMENT) return errors::InvalidArgument(what, ": ", conv3p_status_string(rc));
    return errors::Internal(what, ": ", conv3p_status_string(rc));
}

void *StreamOf(OpKernelContext *ctx)
{
    // ROCm build of TensorFlow: the StreamExecutor stream wraps a hipStream_t
    return *reinterpret_cast<void **>(ctx->op_device_context()->stream()->implementation()->GpuStreamMemberHack());
}

template <typename T> class Conv3pHipOp : public OpKernel {
 public:
    explicit Conv3pHipOp(OpKernelConstruction *c) : OpKernel(c) {}
    void Compute(OpKernelContext *ctx) override
    {
        const Tensor &points = ctx->input(0), &input = ctx->input(1), &filter = ctx->input(2);
        const Tensor &stride = ctx->input(3), &voxel = ctx->input(4);
        // the reference's checks and messages (tf_conv3p_atrous.cpp:410-443)
        OP_REQUIRES(ctx, points.dims() == 3, errors::InvalidArgument("Conv3p expects (batch_size, num_points, 3) points shape"));
        OP_REQUIRES(ctx, input.dim_size(0) == points.dim_size(0), errors::InvalidArgument("Conv3p expects points and input tensor to have the same batch size"));
        OP_REQUIRES(ctx, input.dim_size(1) == points.dim_size(1), errors::InvalidArgument("Conv3p expects points and input tensor to have the same number of points"));
        OP_REQUIRES(ctx, filter.dim_size(3) == input.dim_size(2), errors::InvalidArgument("Conv3p expects filter channels to be matched with input channels"));
        OP_REQUIRES(ctx, stride.dim_size(0) == 3, errors::InvalidArgument("Conv3p expects stride tensor to have size 3."));
        OP_REQUIRES(ctx, voxel.dim_size(0) == 1, errors::InvalidArgument("Conv3p expects voxel tensor to have dimension 1."));
        const int B = points.dim_size(0), N = points.dim_size(1), Cin = input.dim_size(2), Cout = filter.dim_size(4);
        const int fz = filter.dim_size(0), fy = filter.dim_size(1), fx = filter.dim_size(2);
        Tensor *out = nullptr;
        OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({B, N, Cout}), &out));
#ifndef CONV3P_SHIM_STATELESS
        std::lock_guard<std::mutex> lk(g_cache_mu);
        CacheSlot *cs = nullptr;
        char *cp = CacheFor(ctx, Abi<T>::kElem, B, N, fz * fy * fx, Cin, Cout, &cs);
        if (cp == nullptr) return;
        conv3p_cache_config cfg = cs->cfg;
        cfg.flags |= FlagsFor(cs, points, points.flat<T>().data());
        const int rc = Abi<T>::forward_cached(points.flat<T>().data(), input.flat<T>().data(), filter.flat<T>().data(),
                                              stride.flat<int32>().data() /* host memory */, voxel.flat<T>()(0), B, N,
                                              Cin, Cout, fz, fy, fx, out->flat<T>().data(), cp, cs->bytes, &cfg,
                                              StreamOf(ctx));
#else
        const size_t need = conv3p_workspace_bytes(CONV3P_PASS_FORWARD, Abi<T>::kElem, B, N, Cin, Cout, fz, fy, fx);
        Tensor ws;
        OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT8, TensorShape({(int64)need + 256}), &ws));
        char *wp = reinterpret_cast<char *>(ws.flat<int8>().data());
        wp += (256 - reinterpret_cast<uintptr_t>(wp) % 256) % 256;
        const int rc = Abi<T>::forward(points.flat<T>().data(), input.flat<T>().data(), filter.flat<T>().data(),
                                       stride.flat<int32>().data() /* host memory */, voxel.flat<T>()(0), B, N, Cin,
                                       Cout, fz, fy, fx, out->flat<T>().data(), wp, need, StreamOf(ctx));
#endif
        OP_REQUIRES_OK(ctx, FromCode(rc, "Conv3p"));
    }
};

template <typename T> class Conv3pGradHipOp : public OpKernel {
 public:
    explicit Conv3pGradHipOp(OpKernelConstruction *c) : OpKernel(c) {}
    void Compute(OpKernelContext *ctx) override
    {
        const Tensor &grad = ctx->input(0), &points = ctx->input(1), &input = ctx->input(2), &filter = ctx->input(3);
        const Tensor &stride = ctx->input(4), &voxel = ctx->input(5);
        OP_REQUIRES(ctx, stride.dim_size(0) == 3, errors::InvalidArgument("Conv3p expects stride tensor to have size 3."));
        OP_REQUIRES(ctx, voxel.dim_size(0) == 1, errors::InvalidArgument("Conv3p expects voxel tensor to have dimension 1."));
        const int B = points.dim_size(0), N = points.dim_size(1), Cin = filter.dim_size(3), Cout = filter.dim_size(4);
        const int fz = filter.dim_size(0), fy = filter.dim_size(1), fx = filter.dim_size(2);
        // tf_conv3p_atrous.cpp:583-585
        OP_REQUIRES(ctx, grad.dim_size(0) == B, errors::InvalidArgument("backprop grad tensor has wrong size for dim 0"));
        OP_REQUIRES(ctx, grad.dim_size(1) == N, errors::InvalidArgument("backprop grad tensor has wrong size for dim 1"));
        OP_REQUIRES(ctx, grad.dim_size(2) == Cout, errors::InvalidArgument("backprop grad tensor has wrong size for dim 2"));
        Tensor *dx = nullptr, *dw = nullptr;
        OP_REQUIRES_OK(ctx, ctx->allocate_output(0, input.shape(), &dx));
        OP_REQUIRES_OK(ctx, ctx->allocate_output(1, filter.shape(), &dw));
#ifndef CONV3P_SHIM_STATELESS
        std::lock_guard<std::mutex> lk(g_cache_mu);
        CacheSlot *cs = nullptr;
        char *cp = CacheFor(ctx, Abi<T>::kElem, B, N, fz * fy * fx, Cin, Cout, &cs);
        if (cp == nullptr) return;
        conv3p_cache_config cfg = cs->cfg;
        cfg.flags |= FlagsFor(cs, points, points.flat<T>().data());
        const int rc = Abi<T>::backward_cached(grad.flat<T>().data(), points.flat<T>().data(), input.flat<T>().data(),
                                               filter.flat<T>().data(), stride.flat<int32>().data(), voxel.flat<T>()(0),
                                               B, N, Cin, Cout, fz, fy, fx, dx->flat<T>().data(), dw->flat<T>().data(), cp,
                                               cs->bytes, &cfg, StreamOf(ctx));
#else
        const size_t need = conv3p_workspace_bytes(CONV3P_PASS_BACKWARD, Abi<T>::kElem, B, N, Cin, Cout, fz, fy, fx);
        Tensor ws;
        OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT8, TensorShape({(int64)need + 256}), &ws));
        char *wp = reinterpret_cast<char *>(ws.flat<int8>().data());
        wp += (256 - reinterpret_cast<uintptr_t>(wp) % 256) % 256;
        const int rc = Abi<T>::backward(grad.flat<T>().data(), points.flat<T>().data(), input.flat<T>().data(),
                                        filter.flat<T>().data(), stride.flat<int32>().data(), voxel.flat<T>()(0), B, N,
                                        Cin, Cout, fz, fy, fx, dx->flat<T>().data(), dw->flat<T>().data(), wp, need,
                                        StreamOf(ctx));
#endif
        OP_REQUIRES_OK(ctx, FromCode(rc, "Conv3pGrad"));
    }
};

}  // namespace

#define REGISTER_HIP(T)                                                                                   \
    REGISTER_KERNEL_BUILDER(Name("Conv3p").Device(DEVICE_GPU).TypeConstraint<T>("T")                     \
                                .HostMemory("stride").HostMemory("voxel_size"), Conv3pHipOp<T>);          \
    REGISTER_KERNEL_BUILDER(Name("Conv3pGrad").Device(DEVICE_GPU).TypeConstraint<T>("T")                 \
                                .HostMemory("stride").HostMemory("voxel_size"), Conv3pGradHipOp<T>);
REGISTER_HIP(float)
REGISTER_HIP(double)
// The DEVICE_CPU kernels of the drop-in library are the reference's own tf_conv3p_atrous.cpp object,
// linked unchanged (tf_conv3p_compile.sh:32): this repository ships no CPU implementation of the op.
