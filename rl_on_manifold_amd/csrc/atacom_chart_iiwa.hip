// Canonical-chart kernels (cfg.chart_mode = 1, atacom_chart.h) of the iiwa environment, both precisions.
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_chart_iiwa(int dtype) {
    return dtype == ATACOM_F64 ? Variant<double, Iiwa, false, 1>::table() : Variant<float, Iiwa, false, 1>::table();
}
}  // namespace atacom
