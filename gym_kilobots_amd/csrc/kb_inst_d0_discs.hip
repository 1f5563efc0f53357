// kernel instantiations for drive law KB_DRIVE_VELOCITY in scenes whose objects are all discs (BASELINE config 4):
// the kilobot - polygon contact code and its per-slot registers fold away (POLY = false)
#include "kb_step_kernel.h"

namespace kb {
template <int LIGHT_TYPE>
static kb_step_fn pick(int objects) {    // 5: objects, 6: objects + one-wave workgroup; | KB_PICK_SLEEP: with the sleep state
    if (objects & KB_PICK_SLEEP)
        return (objects & ~KB_PICK_SLEEP) == 6 ? kb_step_kernel<KB_DRIVE_VELOCITY, LIGHT_TYPE, true, 0, 1, false, true, true>
                                               : kb_step_kernel<KB_DRIVE_VELOCITY, LIGHT_TYPE, true, 0, 0, false, true, true>;
    return objects == 6 ? kb_step_kernel<KB_DRIVE_VELOCITY, LIGHT_TYPE, true, 0, 1, false>
                        : kb_step_kernel<KB_DRIVE_VELOCITY, LIGHT_TYPE, true, 0, 0, false>;
}
kb_step_fn kb_pick_velocity_discs(int light_type, int objects) {
    if (light_type == KB_PICK_FIXED_1024) return kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, true, 1024, 0, false, false>;
    if (light_type == KB_PICK_FIXED_1024_SENSE) return kb_step_kernel<KB_DRIVE_VELOCITY, KB_LIGHT_NONE, true, 1024, 0, false>;
    if (light_type == KB_LIGHT_CIRCULAR) return pick<KB_LIGHT_CIRCULAR>(objects);
    if (light_type == KB_LIGHT_NONE) return pick<KB_LIGHT_NONE>(objects);
    return pick<KB_LIGHT_GENERAL>(objects);
}
}  // namespace kb
