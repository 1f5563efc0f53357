// kernel instantiations for KB_DRIVE_MIXED beyond 128 kilobots: the full workgroup at 256 VGPRs (two waves per SIMD: one env per CU,
// no spills), the code path with per-body masses; with and without the sleep state
#include "kb_step_kernel.h"

namespace kb {
template <int LIGHT_TYPE>
static kb_step_fn pick(int sleep) {
    return sleep ? kb_step_kernel<KB_DRIVE_MIXED, LIGHT_TYPE, true, 0, 3, true, true, true>
                 : kb_step_kernel<KB_DRIVE_MIXED, LIGHT_TYPE, true, 0, 3, true, true, false>;
}
kb_step_fn kb_pick_mixed_large(int light_type, int sleep) {
    if (light_type == KB_LIGHT_CIRCULAR) return pick<KB_LIGHT_CIRCULAR>(sleep);
    if (light_type == KB_LIGHT_NONE) return pick<KB_LIGHT_NONE>(sleep);
    return pick<KB_LIGHT_GENERAL>(sleep);      // GradientLight, MomentumLight, CompositeLight
}
}  // namespace kb
