/*
 * kilobots_hip.h -- C ABI of libkilobots_hip.so: the MI355X (gfx950) batched Kilobot world step.
 *
 * This is the drop-in boundary for the one hot path this library replaces: the substep loop of
 * gym-kilobots' `KilobotsEnv.step` (reference gym_kilobots/envs/kilobots_env.py:161-215), i.e.
 * light step -> light sensing -> per-kilobot drive law -> Box2D `world.Step(0.1, 10, 10)` ->
 * pose read-back, executed for num_envs x num_bots agents per launch.
 *
 * The reference has no FFI of its own (it is Python calling the SWIG module `Box2D`); every
 * entry point below cites the reference Python interface it replaces.  The Python binding
 * (ctypes) is gym_kilobots_amd/_native.py; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative KB_E* code otherwise;
 *     kb_last_error() returns a thread-local message for the last failure;
 *   - nothing throws across the ABI;
 *   - all `d_*` pointers are DEVICE pointers owned by the caller (e.g. torch tensors); the
 *     library allocates no device memory.  Buffers bound with kb_bind() must stay alive until
 *     kb_destroy() or the next kb_bind();
 *   - all work is enqueued asynchronously on the `hipStream_t` passed as `void *stream`
 *     (NULL = the null stream); no call synchronises the device;
 *   - a handle is not thread-safe; distinct handles are independent;
 *   - arrays are SoA, shape [num_envs][num_bots], env-major, float32 unless noted;
 *   - body poses are Box2D world units = metres x 25 (reference gym_kilobots/lib/body.py:7),
 *     exactly what the reference's b2Body holds; kb_get_poses() converts to metres like
 *     Body.get_pose (body.py:63-65).
 */
#ifndef KILOBOTS_HIP_H
#define KILOBOTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KB_OK 0
#define KB_EINVAL (-1)      /* bad argument / unsupported configuration */
#define KB_ENOTBOUND (-2)   /* kb_bind() has not been called or a required buffer is NULL */
#define KB_EHIP (-3)        /* a HIP runtime call failed */
#define KB_ELDS (-4)        /* configuration does not fit the 160 KiB LDS of a CU */

/* drive laws: which reference Kilobot subclass every bot of the handle is */
enum kb_drive_mode {
    KB_DRIVE_VELOCITY = 0,          /* SimpleVelocityControlKilobot      kilobot.py:213-263 */
    KB_DRIVE_ACCEL = 1,             /* SimpleAccelerationControlKilobot  kilobot.py:266-300 */
    KB_DRIVE_MOTORS = 2,            /* Kilobot.step motor law            kilobot.py:86-127  */
    KB_DRIVE_SIMPLE_PHOTOTAXIS = 3, /* SimplePhototaxisKilobot           kilobot.py:171-210 */
    KB_DRIVE_PHOTOTAXIS = 4,        /* PhototaxisKilobot                 kilobot.py:303-333 */
    KB_DRIVE_MIXED = 5              /* any mix of the five in one env (KilobotsEnv.step steps whatever is in _kilobots,
                                       kilobots_env.py:183-184): the law of every kilobot comes from kb_buffers.bot_mode, the
                                       fixture density of its class from kb_config.mode_density; every per-law state buffer is
                                       required; spill-free 256-VGPR instantiations: one-wave workgroups up to 128 kilobots,
                                       the full workgroup (one env per CU) beyond */
};

enum kb_light_type {
    KB_LIGHT_NONE = 0,
    KB_LIGHT_CIRCULAR = 1,          /* CircularGradientLight             light.py:151-195   */
    KB_LIGHT_GRADIENT = 2,          /* GradientLight                     light.py:218-271   */
    KB_LIGHT_MOMENTUM = 3,          /* MomentumLight                     light.py:274-319   */
    KB_LIGHT_COMPOSITE = 4          /* CompositeLight of <= 4 circular / momentum lights, light.py:99-148 */
};
#define KB_MAX_LIGHTS 4

/* kb_step flags */
#define KB_STEP_NO_DRIVE 1          /* world.Step only, kilobot body velocities = 0:
                                       KilobotsEnv._step_world, kilobots_env.py:217-219 */

#define KB_MAX_OBJECTS 8
#define KB_MAX_POLY_VERTS 4
/* fixture of a pushable object (body.py): Circle :181-192, Quad / CornerQuad :129-178 (b2PolygonShape::SetAsBox),
 * single-fixture convex Polygon :217-262 (b2PolygonShape::Set) */
enum kb_shape { KB_SHAPE_CIRCLE = 0, KB_SHAPE_BOX = 1, KB_SHAPE_POLYGON = 2 };
#define KB_OWS_COLS 12      /* object warm-start table: column = partner object 0..7, or 8 + wall */
#define KB_OWS_WORDS 6      /* per entry: (feature id, normal impulse, tangent impulse) of <= 2 manifold points; id < 0 = none */
#define KB_MAX_BOTS 1024

/* Scene constants.  Defaults of the reference are given in brackets. */
typedef struct kb_config {
    int32_t num_envs, num_bots, num_objects;    /* num_objects: 0..8 pushable bodies per env (the same shapes in every env) */
    float world_width, world_height;            /* metres [2.0, 1.5]   kilobots_env.py:19 */
    float dt;                                   /* [0.1]               kilobots_env.py:25,32 */
    int32_t vel_iters, pos_iters;               /* [10, 10]            kilobots_env.py:26-27 */
    int32_t drive_mode, light_type;
    float bot_radius;                           /* metres [0.0165]     kilobot.py:9 */
    float bot_density;                          /* [1.0; 2.0 for the velocity/accel bots] kilobot.py:25,214 */
    float bot_linear_damping, bot_angular_damping; /* [0.8, 0.8]       kilobot.py:29-30 */
    float light_radius;                         /* metres [0.2]        light.py:152 */
    float light_lo[2], light_hi[2];             /* light position bounds, light.py:44-46 */
    float light_act_lo[2], light_act_hi[2];     /* [-0.01, 0.01]       light.py:49-54 */
    float light_max_velocity;                   /* MomentumLight.max_velocity, light.py:289-292 */
    int32_t ws_slots;                           /* warm-start slots per bot [8] */
    float obj_radius[KB_MAX_OBJECTS];           /* metres; Circle(radius=...), body.py:181-192 */
    float obj_density, obj_friction;            /* [2, 0.01] body.py:11-12 */
    float obj_linear_damping, obj_angular_damping; /* [0.8, 0.8] body.py:15-16 */
    int32_t toi_walls;                          /* [1] b2World::SolveTOI against the static walls (continuousPhysics) */
    int32_t solver_mode;                        /* 0 = automatic.  Test knobs (results are identical in every mode):
                                                   1 list solver, one wave per island set; 2 list solver, whole
                                                   workgroup per sweep; 3 / 4 = 1 / 2 with contacts staged in `scratch` */
    /* KB_LIGHT_COMPOSITE only: the component lights (the fields above then are unused) */
    int32_t light_count;                        /* 1..KB_MAX_LIGHTS */
    int32_t light_kind[KB_MAX_LIGHTS];          /* KB_LIGHT_CIRCULAR or KB_LIGHT_MOMENTUM */
    float lightc_radius[KB_MAX_LIGHTS], lightc_max_velocity[KB_MAX_LIGHTS];
    float lightc_lo[KB_MAX_LIGHTS][2], lightc_hi[KB_MAX_LIGHTS][2];
    float lightc_act_lo[KB_MAX_LIGHTS][2], lightc_act_hi[KB_MAX_LIGHTS][2];
    /* objects other than circles */
    int32_t obj_shape[KB_MAX_OBJECTS];          /* enum kb_shape [circle] */
    int32_t obj_nverts[KB_MAX_OBJECTS];         /* KB_SHAPE_POLYGON: 3..KB_MAX_POLY_VERTS */
    float obj_verts[KB_MAX_OBJECTS][KB_MAX_POLY_VERTS][2]; /* body frame, Box2D WORLD UNITS: metres x 25 evaluated in double and
                                                   then rounded, exactly what body.py:137,246 hands to Box2D.
                                                   BOX: [0] = (width / 2, height / 2) x 25 (Quad, body.py:136-137).
                                                   POLYGON: counter-clockwise hull in b2PolygonShape::Set order, centred on
                                                   its centroid (body.py:226-241) */
    float wall_friction;                        /* [0.2] b2FixtureDef default of the arena chain, kilobots_env.py:46-51 */
    /* bodies with several convex fixtures (LForm, TForm, CForm: body.py:277-334).  With num_fixtures > 0 the arrays
     * obj_shape / obj_nverts / obj_verts / obj_radius are indexed by FIXTURE and fixture f belongs to object
     * obj_fixture_body[f] (fixtures of one object need not be adjacent); num_fixtures == 0 means one fixture per
     * object.  Mass, centre of mass and inertia follow b2Body::ResetMassData; ox / oy hold the body ORIGIN like
     * Body.get_pose (body.py:63-65), ovx / ovy the velocity of the centre of mass like b2Body::GetLinearVelocity. */
    int32_t num_fixtures;                       /* 0, or num_objects..KB_MAX_OBJECTS (at most 8 fixtures per env in total) */
    int32_t obj_fixture_body[KB_MAX_OBJECTS];
    int32_t damping_model;                      /* [KB_DAMPING_PADE] how b2Island::Solve applies linearDamping / angularDamping
                                                   (body.py:15-16,35-36): Box2D >= 2.3.1 `v *= 1 / (1 + h c)`, Box2D <= 2.3.0
                                                   `v *= clamp(1 - h c, 0, 1)`.  The reference does not pin box2d-py (setup.py:5);
                                                   INTEGRATION.md says how to tell which one an installed wheel uses. */
    float sense_radius;                         /* metres, centre to centre; 0 = off.  IR-range neighbour sensing (no counterpart in
                                                   the reference; nearest: Body.collides_with, body.py:87-90): at the sensing point
                                                   of every substep (kilobots_env.py:174-180, before the drive law) kilobot i counts
                                                   the kilobots j != i of its env with |p_j - p_i|^2 <= (25 R)^2 in fp32 world units;
                                                   kb_buffers.nbr_count holds the counts of the last substep */
    int32_t contact_capacity;                   /* [0] contacts (and warm-start entries) per env; 0 = the default rule
                                                   max(4 N + 64, min(N (N - 1) / 2 + 4 N, 2304)) + 40 objects.  A spawn that
                                                   overlaps more kilobots than that sets status bit 0; raise it (<= 65528) then:
                                                   the entries live in HBM (36 B each), not in LDS */
    float mode_density[5];                      /* KB_DRIVE_MIXED: fixture density of the kilobots of drive law k (Kilobot._density 1.0,
                                                   SimpleVelocityControlKilobot._density 2.0: kilobot.py:25, :214); 0 = bot_density */
    int32_t allow_sleep;                        /* [1 in the helpers that fill a default config; 0 leaves the state out] b2World(gravity, doSleep=True) of kilobots_env.py:45: bodies carry
                                                   b2Body::m_sleepTime (kb_buffers.sleep_time / osleep, seconds; < 0: asleep).  An
                                                   island whose bodies all stayed below b2_linearSleepTolerance /
                                                   b2_angularSleepTolerance for b2_timeToSleep = 0.5 s and whose position
                                                   constraints converged falls asleep (b2Island::Solve: velocities zeroed); islands
                                                   without an awake body are not simulated (b2World::Solve); a body wakes when a
                                                   non-zero velocity is assigned to it (Kilobot.step -> b2Body::SetLinearVelocity /
                                                   SetAngularVelocity) or an awake island reaches it.  The envs of
                                                   gym_kilobots_amd.envs switch it on like the reference; bench.py's headline
                                                   runs without the state (sleeping cannot change a trajectory in which every
                                                   kilobot is commanded to move in every substep: DESIGN.md 4c) and reports the
                                                   instantiation with it next to it */
} kb_config;

enum kb_damping_model { KB_DAMPING_PADE = 0, KB_DAMPING_LINEAR = 1 };

/* Device buffers of one handle.  NULL is allowed for buffers the configuration never touches
 * (noted per field).  Replaces the per-object state of the reference: b2Body position/angle
 * (body.py:51-72), Kilobot._velocity / _acceleration / _motor_* (kilobot.py:37-38,225-229,277),
 * PhototaxisKilobot counters (kilobot.py:307-313), light position (light.py:40-42). */
typedef struct kb_buffers {
    float *x, *y, *theta;               /* required */
    float *v, *w;                       /* VELOCITY / ACCEL modes: commanded (v [m/s], omega [rad/s]) */
    float *acc_v, *acc_w;               /* ACCEL mode */
    uint8_t *motor_l, *motor_r;         /* MOTORS / PHOTOTAXIS modes */
    float *pt_threshold;                /* PHOTOTAXIS mode */
    int32_t *pt_update, *pt_nochange;   /* PHOTOTAXIS mode */
    uint8_t *pt_dir;                    /* PHOTOTAXIS mode: 0 = 'left', 1 = 'right' */
    float *light_x, *light_y;           /* [num_envs][kb_light_count()], metres; GradientLight: its angle in light_x */
    float *light_vx, *light_vy;         /* [num_envs][kb_light_count()] MomentumLight velocity (light.py:284-287) */
    float *ox, *oy, *otheta, *ovx, *ovy, *ow; /* objects: [num_envs][num_objects] pose (world units, radians) and body velocity */
    /* warm-start store (Box2D keeps the accumulated normal impulse in each b2Contact): per env a packed
     * list of kb_contact_capacity() entries, owner bots ascending, ws_cnt[bot] entries per owner */
    uint32_t *ws_key;                   /* required: [num_envs][kb_contact_capacity()] */
    float *ws_acc;                      /* required: [num_envs][kb_contact_capacity()] */
    uint8_t *ws_cnt;                    /* required: [num_envs][num_bots]; zero it to forget all contacts */
    float *light_value, *light_gx, *light_gy; /* optional outputs: last sensed light (kilobots_env.py:176-180) */
    float *cmd_vx, *cmd_vy, *cmd_w;     /* optional outputs: body velocity written by the drive law */
    int32_t *status;                    /* required: [num_envs]; bit0 contact capacity overflow,
                                           bit1 warm-start slot overflow, bit2 a device staging limit was hit (more than 64 kilobots
                                           on one fixture, 255 kilobot-object contacts in one env, or 63 partners in one cell pair),
                                           bit3 more kilobots near the walls in one substep than half the LDS contact staging holds
                                           (512 at 1024 kilobots; the continuous step is skipped for the rest) */
    void *scratch;                      /* required: kb_scratch_bytes() bytes; contact staging of envs whose
                                           contacts do not fit the LDS staging area (contents are transient) */
    float *ows_acc;                     /* objects: [num_envs][8][KB_OWS_COLS][KB_OWS_WORDS] manifold impulses of the
                                           object-object (column = higher partner) and object-wall (column 8 + wall)
                                           contacts: b2ManifoldPoint id / normalImpulse / tangentImpulse; fill with -1
                                           to forget them */
    uint32_t *nbr_count;                /* sense_radius > 0: [num_envs][num_bots] neighbours within IR range (output; required then) */
    float *sleep_time;                  /* allow_sleep: [num_envs][num_bots] b2Body::m_sleepTime in seconds, < 0 = asleep (required then;
                                           zero-fill = awake) */
    float *osleep;                      /* allow_sleep with objects: the same for the objects, [num_envs][num_objects] */
    uint8_t *bot_mode;                  /* KB_DRIVE_MIXED: [num_envs][num_bots] drive law of every kilobot (KB_DRIVE_VELOCITY ..
                                           KB_DRIVE_PHOTOTAXIS; required then) */
} kb_buffers;

typedef struct kb_sim kb_sim;

/* Replaces KilobotsEnv.__init__ world construction (kilobots_env.py:45-51: b2World + chain-loop
 * walls) and Body/Circle/Kilobot fixture constants (body.py:32-38,187-192; kilobot.py:9-30). */
int kb_create(const kb_config *cfg, kb_sim **out);
void kb_destroy(kb_sim *sim);

/* Attach caller-owned device buffers (copied by value). */
int kb_bind(kb_sim *sim, const kb_buffers *buf);

/* SimpleVelocityControlKilobot.set_action / SimpleAccelerationControlKilobot.set_action
 * (kilobot.py:235-241, 283-289), as called per bot by DirectControlKilobotsEnv.step
 * (direct_control_kilobots_env.py:18-27).  d_actions: [num_envs][num_bots][2] or NULL (= None). */
int kb_set_actions(kb_sim *sim, const float *d_actions, void *stream);

/* n_substeps iterations of the KilobotsEnv.step loop body (kilobots_env.py:168-190) in ONE launch.
 * d_actions (optional): kilobot actions applied first, as kb_set_actions.
 * d_light_action (optional): [num_envs][kb_light_action_dim()] light action (2 per positional light in
 * component order, 1 for a GradientLight), applied every substep (kilobots_env.py:171-172);
 * NULL = action None. */
int kb_step(kb_sim *sim, const float *d_actions, const float *d_light_action, int n_substeps, int flags,
            void *stream);

/* IR-range neighbour sensing on the CURRENT poses, without stepping (e.g. right after a reset): d_count
 * [num_envs][num_bots] uint32 = number of kilobots j != i of the same env with |p_j - p_i|^2 <= (25 radius_m)^2.
 * The same predicate kb_step evaluates at the sensing point of every substep when kb_config.sense_radius > 0.
 * No reference counterpart (the reference has no neighbour sensing; nearest: Body.collides_with, body.py:87-90). */
int kb_sense(kb_sim *sim, float radius_m, uint32_t *d_count, void *stream);

/* The sensing point of ONE substep on its own, for kilobots that are programmed on the host (a Kilobot subclass with its
 * own _loop, kilobot.py:86-88,164-168): Light.step with d_light_action ([num_envs][kb_light_action_dim()], NULL = action None:
 * the light stays) and value_and_gradients at every kilobot's light sensor (kilobots_env.py:171-180) into
 * kb_buffers.light_value / light_gx / light_gy, exactly the arithmetic the fused step uses.  The host then runs the
 * kilobots' _loop (get_ambientlight -> light_value, set_motors -> motor_l / motor_r) and calls
 * kb_step(sim, NULL, NULL, 1, 0, stream): motor law + world.Step of that substep (the light is not stepped again). */
int kb_light_sense(kb_sim *sim, const float *d_light_action, void *stream);

/* KilobotsEnv.reset() for every env of the handle, on the device (kilobots_env.py:150-159 with the spawn rule of
 * YamlKilobotsEnv._init_kilobots, yaml_kilobots_env.py:346-352): positions ~ N(mean, std) per coordinate, clipped to
 * the world bounds -/+ 0.02 m, theta = 0 (body.py:28-29) or U(-pi, pi); commands and accelerations zeroed
 * (random_velocity: the U([0, 0.01] x [-pi/2, pi/2]) initial command of kilobot.py:225-229); motors as after
 * Kilobot._setup -> turn_left; phototaxis counters cleared; warm-start impulses forgotten; status cleared.  Object
 * and light state is not touched.  resolve != 0 appends the "step to resolve" of kilobots_env.py:156-157 (one world.Step
 * with the kilobots at rest).
 * Random numbers: Philox4x32-10, key = seed (lo, hi), counter = (env_offset + env, bot, 0, 0); the four outputs give the
 * two Box-Muller normals, the heading and the command.  A shard created with env_offset = its first global env index
 * reproduces the corresponding rows of the unsharded reset bit for bit (the reference itself is not reproducible:
 * seed() only stores the number, kilobots_env.py:145-148). */
typedef struct kb_reset_params {
    uint64_t seed;
    int32_t env_offset;
    float mean[2];          /* metres */
    float std;              /* metres */
    int32_t random_theta;
    int32_t random_velocity;
    int32_t resolve;
} kb_reset_params;
int kb_reset(kb_sim *sim, const kb_reset_params *rp, void *stream);

/* KilobotsEnv.get_state()['kilobots'] (kilobots_env.py:115-118 -> body.py:63-72):
 * d_out [num_envs][num_bots][3] = (x [m], y [m], theta). */
int kb_get_poses(kb_sim *sim, float *d_out, void *stream);

/* The whole of KilobotsEnv.get_state() of the kilobots and objects (kilobots_env.py:115-118 -> body.py:63-72) and the
 * status word in ONE buffer, for hosts that read the state back after every step (one copy instead of three):
 * d_out [num_envs][3 num_bots + 3 num_objects + 1] float32 = per env the kilobots' (x [m], y [m], theta), the objects'
 * (x [m], y [m], theta), then the bit pattern of the env's int32 kb_buffers.status word. */
int kb_get_state(kb_sim *sim, float *d_out, void *stream);

/* Introspection */
int kb_lds_bytes(const kb_sim *sim);            /* dynamic LDS per workgroup (one env per workgroup) */
int kb_light_action_dim(const kb_sim *sim);     /* floats per env in d_light_action */
int kb_light_count(const kb_sim *sim);          /* light components per env (0 without a light) */
int kb_contact_capacity(const kb_sim *sim);     /* contacts (and warm-start entries) per env */
int kb_lds_staging_entries(const kb_sim *sim);  /* contacts of one env that are staged in LDS; an env with more takes its slice of
                                                   kb_buffers.scratch for that substep (same results, slower).  kb_create trades
                                                   entries for resident envs per CU, never below num_bots + 64 (688 at 1024) */
size_t kb_scratch_bytes(const kb_sim *sim);     /* size of kb_buffers.scratch: 32 B per contact of the capacity (staging record; level-sorted record of the cooperative sweeps) */
int kb_block_threads(const kb_sim *sim);
int kb_resident_envs_per_cu(kb_sim *sim);        /* workgroups (= envs) of this handle's kernel that one CU holds at a time (HIP occupancy query; needs a GPU) */
int kb_set_block_threads(kb_sim *sim, int threads);  /* multiple of 64 in [64, 512], num_bots <= 2 * threads.  kb_create picks
                                                         the measured best (one kilobot per thread, power-of-two wave count,
                                                         as many resident envs as the LDS allows); results never depend on
                                                         it.  kb_lds_bytes() follows (tables sized by the wave count). */
const char *kb_last_error(void);
const char *kb_version(void);

#ifdef __cplusplus
}
#endif
#endif
