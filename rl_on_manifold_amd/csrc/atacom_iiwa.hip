// Device code of the iiwa environment, float32 production path.  (The float64 parity build is its own translation
// unit, atacom_iiwa_f64.hip, so that the two -- the largest units of the library -- compile in parallel.)
#include "atacom_ops_impl.h"
namespace atacom {
const EnvOps* ops_iiwa_f64();
const EnvOps* ops_iiwa(int dtype) { return dtype == ATACOM_F64 ? ops_iiwa_f64() : Ops<float, Iiwa>::table(); }
}  // namespace atacom
