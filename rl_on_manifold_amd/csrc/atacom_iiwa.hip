// Device code of the iiwa environment, float32 production path.  (The float64 parity build is its own translation
// unit, atacom_iiwa_f64.hip, so that the two -- the largest units of the library -- compile in parallel.)
#include "atacom_iiwa_group.h"
namespace atacom {
// the lane-group kernels are defined in atacom_iiwa_group.hip (a unit with its own scheduler option)
ATACOM_IIWA_GROUP_KERNELS(extern template)
const EnvOps* ops_iiwa_f64();
const EnvOps* ops_iiwa(int dtype) { return dtype == ATACOM_F64 ? ops_iiwa_f64() : Ops<float, Iiwa>::table(); }
}  // namespace atacom
