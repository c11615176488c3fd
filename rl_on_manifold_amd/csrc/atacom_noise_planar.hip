// The planar stepping kernels with the domain-randomisation options compiled in (cfg.obs_noise / obs_delay / env_noise;
// atacom_air_hockey.py:12-14), both precisions, both charts, and the dispatch over the environments.
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_noise_iiwa_f32(int chart_mode);
const VariantOps* ops_noise_iiwa_f64(int chart_mode);
const VariantOps* ops_noise(int env_id, int dtype, int chart_mode) {
    const bool d = dtype == ATACOM_F64;
    if (env_id == ATACOM_ENV_IIWA) return d ? ops_noise_iiwa_f64(chart_mode) : ops_noise_iiwa_f32(chart_mode);
    if (env_id != ATACOM_ENV_PLANAR) return nullptr;
    if (chart_mode == 1) return d ? Variant<double, Planar, false, 1, true>::table() : Variant<float, Planar, false, 1, true>::table();
    return d ? Variant<double, Planar, false, 0, true>::table() : Variant<float, Planar, false, 0, true>::table();
}
}  // namespace atacom
