// DECLARATIONS ONLY -- see op_kernel.h in this directory.
#pragma once
#include <initializer_list>
#include "tensorflow/core/framework/op_kernel.h"
namespace tensorflow {
namespace shape_inference {
struct ShapeHandle {};
struct DimensionHandle {};
class InferenceContext {
 public:
    ShapeHandle input(int idx) const;
    Status WithRank(ShapeHandle shape, int64 rank, ShapeHandle *out);
    DimensionHandle Dim(ShapeHandle s, int64 idx);
    ShapeHandle MakeShape(std::initializer_list<DimensionHandle> dims);
    void set_output(int idx, ShapeHandle shape);
};
}  // namespace shape_inference
}  // namespace tensorflow
