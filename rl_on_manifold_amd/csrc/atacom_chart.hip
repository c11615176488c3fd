// Canonical-chart kernels (cfg.chart_mode = 1, atacom_chart.h) of the circle and planar environments, both precisions,
// and the dispatch over the environments (iiwa: atacom_chart_iiwa.hip, compiled in parallel).
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_chart_iiwa(int dtype);
const VariantOps* ops_chart(int env_id, int dtype) {
    const bool d = dtype == ATACOM_F64;
    switch (env_id) {
        case ATACOM_ENV_CIRCLE: return d ? Variant<double, Circle, false, 1>::table() : Variant<float, Circle, false, 1>::table();
        case ATACOM_ENV_PLANAR: return d ? Variant<double, Planar, false, 1>::table() : Variant<float, Planar, false, 1>::table();
        case ATACOM_ENV_IIWA: return ops_chart_iiwa(dtype);
        default: return nullptr;              // the E / T baselines have no null space
    }
}
}  // namespace atacom
