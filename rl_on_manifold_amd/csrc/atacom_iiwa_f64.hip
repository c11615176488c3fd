// Device code of the iiwa environment, float64 parity build (see atacom_iiwa.hip).
#include "atacom_ops_impl.h"
namespace atacom {
const EnvOps* ops_iiwa_f64() { return Ops<double, Iiwa>::table(); }
}  // namespace atacom
