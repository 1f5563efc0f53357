// kernel instantiations for drive law KB_DRIVE_SIMPLE_PHOTOTAXIS
#include "kb_step_kernel.h"

namespace kb {
kb_step_fn kb_pick_simple_phototaxis(int light_type, int objects) {
#ifdef KB_ONLY_JAM      // measurement builds (tools/ab_jam.py): only the instantiations of the jammed-swarm probe
    return light_type == KB_LIGHT_CIRCULAR ? (objects == 3 ? kb_step_kernel<KB_DRIVE_SIMPLE_PHOTOTAXIS, KB_LIGHT_CIRCULAR, false, 0, 2>
                                                             : (objects == 0 ? kb_step_kernel<KB_DRIVE_SIMPLE_PHOTOTAXIS, KB_LIGHT_CIRCULAR, false> : nullptr)) : nullptr;
#else
    if ((objects & ~KB_PICK_SLEEP) >= 5) return kb_pick_simple_phototaxis_discs(light_type, objects);      // all objects are discs: kb_inst_d3_discs.hip
    if (light_type == KB_LIGHT_CIRCULAR) return kb_pick_obj<KB_DRIVE_SIMPLE_PHOTOTAXIS, KB_LIGHT_CIRCULAR>(objects);
    // GradientLight, MomentumLight, CompositeLight: one general kernel
    return kb_pick_obj<KB_DRIVE_SIMPLE_PHOTOTAXIS, KB_LIGHT_GENERAL>(objects);
#endif
}
#ifdef KB_ONLY_JAM      // (the measurement library is linked from this unit and the C ABI alone)
kb_step_fn kb_pick_velocity(int, int) { return nullptr; }
kb_step_fn kb_pick_velocity_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_accel(int, int) { return nullptr; }
kb_step_fn kb_pick_accel_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_motors(int, int) { return nullptr; }
kb_step_fn kb_pick_motors_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_simple_phototaxis_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_phototaxis(int, int) { return nullptr; }
kb_step_fn kb_pick_phototaxis_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_mixed(int, int) { return nullptr; }
kb_step_fn kb_pick_mixed_large(int, int) { return nullptr; }
#endif
}  // namespace kb
