// Device code of the planar environment (float32 production path + float64 parity build).
#include "atacom_ops_impl.h"
namespace atacom {
const EnvOps* ops_planar(int dtype) {
    return dtype == ATACOM_F64 ? Ops<double, Planar>::table() : Ops<float, Planar>::table();
}
}  // namespace atacom
