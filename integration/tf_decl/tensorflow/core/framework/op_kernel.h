// DECLARATIONS ONLY -- a subset of TensorFlow 1.x's public C++ op API (tensorflow/core/framework/*.h), written
// from the API documentation for ONE purpose: `make -C integration check` type-checks tf_conv3p_shim.cc
// (-fsyntax-only) in an image that has no TensorFlow.  Nothing here is linked, run or shipped; it is not a
// TensorFlow build and proves nothing about a real one beyond "the shim is well-formed C++ against these
// signatures".  Signatures follow TF 1.9-1.15 (the versions the reference names, README.md:7).
#pragma once
#include <cstddef>
#include <cstdint>
#include <initializer_list>
#include <string>

namespace tensorflow {
typedef long long int64;
typedef int int32;
typedef signed char int8;
enum DataType { DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_INT8 = 6 };
extern const char *const DEVICE_CPU;
extern const char *const DEVICE_GPU;

class Status {
 public:
    static Status OK();
    bool ok() const;
};
namespace errors {
template <typename... Args> Status InvalidArgument(Args... args);
template <typename... Args> Status Internal(Args... args);
}  // namespace errors

class TensorShape {
 public:
    TensorShape();
    TensorShape(std::initializer_list<int64> dims);
};
template <typename T> struct FlatView {
    T *data() const;
    T &operator()(std::size_t i) const;
};
class Tensor {
 public:
    Tensor();
    int dims() const;
    int64 dim_size(int d) const;
    const TensorShape &shape() const;
    template <typename T> FlatView<T> flat();
    template <typename T> FlatView<const T> flat() const;
    // tensorflow/core/framework/tensor.h: copies share (and reference-count) the underlying TensorBuffer
    Tensor(const Tensor &other);
    Tensor &operator=(const Tensor &other);
    bool IsInitialized() const;
    bool SharesBufferWith(const Tensor &b) const;
    std::size_t TotalBytes() const;
};

namespace se_decl {   // stream_executor::Stream -> platform stream handle (ROCm build: a hipStream_t)
struct StreamInterface { void **GpuStreamMemberHack(); };
struct Stream { StreamInterface *implementation(); };
}  // namespace se_decl
class DeviceContext {
 public:
    se_decl::Stream *stream() const;
};

class OpKernelConstruction {};
class OpKernelContext;
class PersistentTensor {   // tensorflow/core/framework/op_kernel.h: a tensor that outlives one Compute()
 public:
    PersistentTensor();
    Tensor *AccessTensor(OpKernelContext *context);
};
class OpKernelContext {
 public:
    const Tensor &input(int index);
    Status allocate_output(int index, const TensorShape &shape, Tensor **tensor);
    Status allocate_temp(DataType type, const TensorShape &shape, Tensor *out_temp);
    Status allocate_persistent(DataType type, const TensorShape &shape, PersistentTensor *out_persistent, Tensor **out_tensor);
    DeviceContext *op_device_context();
    void CtxFailure(const Status &s);
    void CtxFailureWithWarning(const Status &s);
};
class OpKernel {
 public:
    explicit OpKernel(OpKernelConstruction *context);
    virtual ~OpKernel();
    virtual void Compute(OpKernelContext *context) = 0;
};

#define OP_REQUIRES(CTX, EXP, STATUS)          \
    do {                                        \
        if (!(EXP)) {                           \
            (CTX)->CtxFailure((STATUS));        \
            return;                             \
        }                                       \
    } while (0)
#define OP_REQUIRES_OK(CTX, ...)                           \
    do {                                                    \
        ::tensorflow::Status _s(__VA_ARGS__);              \
        if (!_s.ok()) {                                     \
            (CTX)->CtxFailureWithWarning(_s);               \
            return;                                         \
        }                                                   \
    } while (0)
#define TF_RETURN_IF_ERROR(...)                            \
    do {                                                    \
        const ::tensorflow::Status _status = (__VA_ARGS__); \
        if (!_status.ok()) return _status;                  \
    } while (0)

namespace register_kernel {
class Name {
 public:
    explicit Name(const char *op);
    Name &Device(const char *device_type);
    template <typename T> Name &TypeConstraint(const char *attr_name);
    Name &HostMemory(const char *arg_name);
};
}  // namespace register_kernel
#define TF_DECL_CAT_(a, b) a##b
#define TF_DECL_CAT(a, b) TF_DECL_CAT_(a, b)
#define REGISTER_KERNEL_BUILDER(kernel_builder, ...)                                                       \
    static const ::tensorflow::register_kernel::Name &TF_DECL_CAT(tf_decl_kernel_, __COUNTER__) =         \
        ::tensorflow::register_kernel::kernel_builder;                                                     \
    static_assert(sizeof(__VA_ARGS__) > 0, "kernel class must be complete")
}  // namespace tensorflow
