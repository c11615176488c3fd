// Device code of the circle environment (float32 production path + float64 parity build).
#include "atacom_ops_impl.h"
namespace atacom {
const EnvOps* ops_circle(int dtype) {
    return dtype == ATACOM_F64 ? Ops<double, Circle>::table() : Ops<float, Circle>::table();
}
const EnvOps* ops_circle_ec(int dtype) {
    return dtype == ATACOM_F64 ? Ops<double, CircleEC>::table() : Ops<float, CircleEC>::table();
}
const EnvOps* ops_circle_t(int dtype) {
    return dtype == ATACOM_F64 ? Ops<double, CircleT>::table() : Ops<float, CircleT>::table();
}
}  // namespace atacom
