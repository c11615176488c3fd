// The iiwa stepping kernels with the domain-randomisation options compiled in, float64 parity build (atacom_noise_iiwa.hip).
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_noise_iiwa_f64(int chart_mode) {
    return chart_mode == 1 ? Variant<double, Iiwa, false, 1, true>::table() : Variant<double, Iiwa, false, 0, true>::table();
}
}  // namespace atacom
