// Rigid-body mode of the iiwa environment (row N4) with the canonical chart (cfg.chart_mode = 1), both precisions.
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_iiwa_dyn_chart(int dtype) {
    return dtype == ATACOM_F64 ? Variant<double, Iiwa, true, 1>::table() : Variant<float, Iiwa, true, 1>::table();
}
}  // namespace atacom
