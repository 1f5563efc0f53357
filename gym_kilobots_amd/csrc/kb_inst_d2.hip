// kernel instantiations for drive law KB_DRIVE_MOTORS
#include "kb_step_kernel.h"

namespace kb {
kb_step_fn kb_pick_motors(int light_type, int objects) {
    if ((objects & ~KB_PICK_SLEEP) >= 5) return kb_pick_motors_discs(light_type, objects);      // all objects are discs: kb_inst_d2_discs.hip
    if (light_type == KB_LIGHT_CIRCULAR) return kb_pick_obj<KB_DRIVE_MOTORS, KB_LIGHT_CIRCULAR>(objects);
    if (light_type == KB_LIGHT_NONE) return kb_pick_obj<KB_DRIVE_MOTORS, KB_LIGHT_NONE>(objects);
    // GradientLight, MomentumLight, CompositeLight: one general kernel
    return kb_pick_obj<KB_DRIVE_MOTORS, KB_LIGHT_GENERAL>(objects);
}
}  // namespace kb
