// Rigid-body mode of the iiwa environment (row N4), float64 parity build, the reference's chart (see atacom_iiwa_dyn.hip).
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_iiwa_dyn_f64() { return Variant<double, Iiwa, true, 0>::table(); }
}  // namespace atacom
