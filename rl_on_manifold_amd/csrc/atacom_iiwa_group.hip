// Device code of the iiwa environment's lane-group kernels (4 / 8 lanes per environment, float32, reference chart, kinematic):
// the headline kernels, compiled with the iterative-ilp scheduler (build.py: UNIT_FLAGS) -- see atacom_iiwa_group.h.
#include "atacom_iiwa_group.h"
namespace atacom {
ATACOM_IIWA_GROUP_KERNELS(template)
}  // namespace atacom
