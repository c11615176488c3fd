// Device code of the iiwa environment (float32 production path + float64 parity build).
#include "atacom_ops_impl.h"
namespace atacom {
const EnvOps* ops_iiwa(int dtype) {
    return dtype == ATACOM_F64 ? Ops<double, Iiwa>::table() : Ops<float, Iiwa>::table();
}
}  // namespace atacom
