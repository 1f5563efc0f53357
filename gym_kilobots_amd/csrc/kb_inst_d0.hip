// kernel instantiations for drive law KB_DRIVE_VELOCITY
#include "kb_step_kernel.h"

namespace kb {
kb_step_fn kb_pick_velocity(int light_type, int objects) {
#ifdef KB_ONLY_BENCH      // measurement builds: only the benchmark instantiation (fixed size, no objects, no sleep state, no sensing)
    if (light_type == KB_PICK_FIXED_1024 && objects == KB_PICK_SLEEP) return kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, false, 1024, 0, true, false, true>;
    return (light_type == KB_PICK_FIXED_1024 && objects == 0) ? kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, false, 1024, 0, true, false> : nullptr;
#else
    if ((objects & ~KB_PICK_SLEEP) >= 5) return kb_pick_velocity_discs(light_type, objects);      // all objects are discs: kb_inst_d0_discs.hip
    if ((objects & KB_PICK_SLEEP) && (light_type == KB_PICK_FIXED_1024 || light_type == KB_PICK_FIXED_1024_SENSE))      // (no objects: kb_abi.hip)
        return light_type == KB_PICK_FIXED_1024 ? kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, false, 1024, 0, true, false, true>
                                                : kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, false, 1024, 0, true, true, true>;
    if (light_type == KB_PICK_FIXED_1024) {
        return objects ? kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, true, 1024, 0, true, false>
                       : kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, false, 1024, 0, true, false>;
    }
    if (light_type == KB_PICK_FIXED_1024_SENSE) {
        return objects ? kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, true, 1024> : kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, false, 1024>;
    }
    if (light_type == KB_LIGHT_CIRCULAR) return kb_pick_obj<KB_DRIVE_VELOCITY, KB_LIGHT_CIRCULAR>(objects);
    if (light_type == KB_LIGHT_NONE) return kb_pick_obj<KB_DRIVE_VELOCITY, KB_LIGHT_NONE>(objects);
    // GradientLight, MomentumLight, CompositeLight: one general kernel
    return kb_pick_obj<KB_DRIVE_VELOCITY, KB_LIGHT_GENERAL>(objects);
#endif
}
#ifdef KB_ONLY_BENCH      // (the measurement library is linked from this unit and the C ABI alone)
kb_step_fn kb_pick_velocity_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_accel(int, int) { return nullptr; }
kb_step_fn kb_pick_accel_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_motors(int, int) { return nullptr; }
kb_step_fn kb_pick_motors_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_simple_phototaxis(int, int) { return nullptr; }
kb_step_fn kb_pick_simple_phototaxis_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_phototaxis(int, int) { return nullptr; }
kb_step_fn kb_pick_phototaxis_discs(int, int) { return nullptr; }
kb_step_fn kb_pick_mixed(int, int) { return nullptr; }
kb_step_fn kb_pick_mixed_large(int, int) { return nullptr; }
#endif
}  // namespace kb
