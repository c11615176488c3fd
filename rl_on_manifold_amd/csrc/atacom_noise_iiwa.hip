// The iiwa stepping kernels with the domain-randomisation options compiled in (cfg.obs_noise / obs_delay / env_noise;
// iiwa_hit_atacom.py:11-13), float32, both charts.  A translation unit of their own: handles that leave the options off --
// the default -- run kernels that do not contain them (atacom_kernels.h: EnvRef).
#include "atacom_ops_impl.h"
namespace atacom {
const VariantOps* ops_noise_iiwa_f32(int chart_mode) {
    return chart_mode == 1 ? Variant<float, Iiwa, false, 1, true>::table() : Variant<float, Iiwa, false, 0, true>::table();
}
}  // namespace atacom
