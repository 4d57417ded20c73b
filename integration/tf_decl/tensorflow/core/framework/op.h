// DECLARATIONS ONLY -- see op_kernel.h in this directory.
#pragma once
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"
namespace tensorflow {
namespace register_op {
class OpDefBuilderWrapper {
 public:
    explicit OpDefBuilderWrapper(const char *name);
    OpDefBuilderWrapper &Attr(const char *spec);
    OpDefBuilderWrapper &Input(const char *spec);
    OpDefBuilderWrapper &Output(const char *spec);
    OpDefBuilderWrapper &SetShapeFn(Status (*fn)(shape_inference::InferenceContext *));
};
}  // namespace register_op
#define REGISTER_OP(name)                                                                           \
    static const ::tensorflow::register_op::OpDefBuilderWrapper &TF_DECL_CAT(tf_decl_op_, __COUNTER__) = \
        ::tensorflow::register_op::OpDefBuilderWrapper(name)
}  // namespace tensorflow
