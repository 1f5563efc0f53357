/*
 * kb_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see kb_oracle.h).
 *
 * Restates, in fp32 like Box2D, one substep of gym-kilobots'
 *     KilobotsEnv.step            gym_kilobots/envs/kilobots_env.py:168-190
 * for `num_envs` independent worlds of `num_bots` kilobots (+ circular objects):
 *     light.step                  gym_kilobots/lib/light.py:59-75, 237-253, 300-316
 *     light.value_and_gradients   gym_kilobots/lib/light.py:176-189
 *     kb.step (5 drive laws)      gym_kilobots/lib/kilobot.py:86-127,191-203,253-258,294-300,318-333
 *     world.Step(dt, 10, 10)      kilobots_env.py:187 -> Box2D 2.3.1 (external; restated from
 *                                 its published algorithm: b2World::Step / b2Island::Solve /
 *                                 b2ContactSolver / b2CollideCircles / b2CollideEdgeAndCircle)
 *
 * Deliberate, documented differences from Box2D (DESIGN.md "Contact order"):
 *   - Box2D visits contacts in the order its dynamic-tree broadphase happened to create them
 *     (an implementation artefact, not reproducible without Box2D).  The oracle defines a
 *     canonical Gauss-Seidel order instead: contacts are sorted by (class, group, A, B) where
 *     class/group come from a uniform grid (see contact_class()).  Groups of one class never
 *     share a body, which is what lets the HIP kernel run a class in parallel and still be
 *     order-identical to this sequential code.
 *   - circle-circle and circle-wall contacts are treated as exactly central (Box2D's
 *     cross(r, n) is ~1e-8 there), so kilobots never receive torque from contacts.
 *   - arena walls are the four infinite lines of the closed chain loop of
 *     kilobots_env.py:46-51 with exact axis normals.
 *   - sleeping (b2World doSleep=True, kilobots_env.py:45) is restated: b2Island::Solve's allowSleep block, the awake seeds of
 *     b2World::Solve, the wake rules of the velocity setters and of b2Contact::Update (a contact that stops touching wakes
 *     both bodies); kbo_config.allow_sleep = 0 drops the state.
 *   - continuous step (b2World::SolveTOI) only against the static walls, which is all Box2D does for
 *     non-bullet bodies; b2TimeOfImpact's control flow is followed with closed-form wall distances (circle
 *     centre; support vertex of a polygon, i.e. the e_faceA separation function on the wall) instead of GJK.
 *     Kilobots: frictionless point contact.  Objects: the TOI sub-solve runs on the manifold constraints of the
 *     body's wall contacts (friction, block solver, rotation), see toi_walls_object.
 *   - pushable objects are circles, boxes or single convex polygons (body.py:129-192, 217-262) with Box2D's
 *     full contact model among themselves and against the walls (manifolds with feature ids, Coulomb friction
 *     sqrt(f1 f2), two-point block solver, rotation); kilobot contacts are frictionless in the reference
 *     (kilobot.py:26) and central on the kilobot side.  Bodies may carry several convex fixtures (LForm, TForm, CForm,
 *     body.py:277-334) with b2Body::ResetMassData's centre of mass.
 */
#include "kb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- Box2D constants (b2Settings.h, 2.3.1) ------------------------------------------- */
#define B2_PI 3.14159265359f
#define B2_LINEAR_SLOP 0.005f
#define B2_POLYGON_RADIUS (2.0f * B2_LINEAR_SLOP)
#define B2_TIME_TO_SLEEP 0.5f                                   /* b2Settings.h: b2_timeToSleep */
#define B2_LINEAR_SLEEP_TOL 0.01f                               /* b2_linearSleepTolerance */
#define B2_ANGULAR_SLEEP_TOL (2.0f / 180.0f * 3.14159265359f)   /* b2_angularSleepTolerance (b2_pi) */
#define B2_BAUMGARTE 0.2f
#define B2_MAX_LINEAR_CORRECTION 0.2f
#define B2_MAX_TRANSLATION 2.0f
#define B2_MAX_TRANSLATION_SQ (B2_MAX_TRANSLATION * B2_MAX_TRANSLATION)
#define B2_MAX_ROTATION (0.5f * B2_PI)
#define B2_MAX_ROTATION_SQ (B2_MAX_ROTATION * B2_MAX_ROTATION)
#define B2_EPSILON 1.19209290e-07f
#define B2_TOI_BAUMGARTE 0.75f
#define B2_MAX_SUBSTEPS 8         /* b2World::SolveTOI k_maxSubSteps */
#define WORLD_SCALE 25.0f              /* body.py:7 */

#define CELL_SIZE 0.875f               /* world units; >= 2 * bot radius (0.825) */
#define MAX_CELLS 8192

#define KEY_WALL 0x10000u
#define KEY_OBJ 0x20000u
#define KEY_EMPTY 0xFFFFFFFFu

/* contact classes (canonical order) */
#define CLS_SAME 0
#define CLS_E 1   /* +1 for odd base-cell parity */
#define CLS_N 3
#define CLS_NE 5
#define CLS_NW 7
#define CLS_WALL 9
#define CLS_BOT_OBJ 10          /* kilobot - object, group = object, sequential inside a group */
/* after them, in every sweep: the manifold constraints object - object (pairs in lexicographic order), then
 * object - wall (by object, wall) */
#define OWS KBO_OWS_COLS
#define OWW KBO_OWS_WORDS
#define MAX_MC (KBO_MAX_OBJECTS * (KBO_MAX_OBJECTS - 1) / 2 + 4 * KBO_MAX_OBJECTS)
#define B2_VELOCITY_THRESHOLD 1.0f

/* ---- sin/cos: Cephes single-precision algorithm (public domain, S. Moshier), restated ---- */
void kbo_sincosf(float xx, float *sp, float *cp) {
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    const float FOPI = 1.27323954473516f;
    float x = fabsf(xx);
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    float ssign = xx < 0.0f ? -1.0f : 1.0f, csign = 1.0f;
    if (j > 3) { ssign = -ssign; csign = -csign; j -= 4; }
    if (j > 1) csign = -csign;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
               - 0.5f * z + 1.0f;
    float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
    if (j == 1 || j == 2) { *sp = ssign * pc; *cp = csign * ps; }
    else { *sp = ssign * ps; *cp = csign * pc; }
}

static inline float clampf(float a, float lo, float hi) { return fmaxf(lo, fminf(a, hi)); }


/* ---- small 2-D algebra in Box2D's operation order (b2Math.h) ---------------------------------- */
typedef struct { float x, y; } v2;
typedef struct { v2 p; float s, c; } xf_t;            /* b2Transform: position + b2Rot (sin, cos) */
static inline v2 V2(float x, float y) { v2 r; r.x = x; r.y = y; return r; }
static inline v2 v_add(v2 a, v2 b) { return V2(a.x + b.x, a.y + b.y); }
static inline v2 v_sub(v2 a, v2 b) { return V2(a.x - b.x, a.y - b.y); }
static inline v2 v_scale(float s, v2 a) { return V2(s * a.x, s * a.y); }
static inline v2 v_neg(v2 a) { return V2(-a.x, -a.y); }
static inline float v_dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline float v_cross(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static inline v2 v_cross_vs(v2 a, float s) { return V2(s * a.y, -s * a.x); }   /* b2Cross(vector, scalar) */
static inline v2 v_cross_sv(float s, v2 a) { return V2(-s * a.y, s * a.x); }   /* b2Cross(scalar, vector) */
static inline v2 v_normalize(v2 a) {                                            /* b2Vec2::Normalize */
    float len = sqrtf(a.x * a.x + a.y * a.y);
    if (len < B2_EPSILON) return a;
    float inv = 1.0f / len;
    return V2(a.x * inv, a.y * inv);
}
static inline xf_t xf_make(float px, float py, float ang) {
    xf_t t; t.p = V2(px, py); kbo_sincosf(ang, &t.s, &t.c); return t;
}
static inline v2 rot_mul(const xf_t *t, v2 v) { return V2(t->c * v.x - t->s * v.y, t->s * v.x + t->c * v.y); }
static inline v2 rot_mulT(const xf_t *t, v2 v) { return V2(t->c * v.x + t->s * v.y, -t->s * v.x + t->c * v.y); }
static inline v2 xf_mul(const xf_t *t, v2 v) {
    return V2((t->c * v.x - t->s * v.y) + t->p.x, (t->s * v.x + t->c * v.y) + t->p.y);
}
static inline v2 xf_mulT(const xf_t *t, v2 v) {
    float px = v.x - t->p.x, py = v.y - t->p.y;
    return V2(t->c * px + t->s * py, -t->s * px + t->c * py);
}
static inline xf_t xf_mulT_xf(const xf_t *A, const xf_t *B) {                   /* b2MulT(A, B) = inv(A) B */
    xf_t C;
    C.s = A->c * B->s - A->s * B->c; C.c = A->c * B->c + A->s * B->s;
    C.p = rot_mulT(A, v_sub(B->p, A->p));
    return C;
}

/* fixture shape of a pushable object: circle (b2CircleShape) or convex polygon (b2PolygonShape, skin radius) */
typedef struct { int kind, n; v2 v[KBO_MAX_POLY_VERTS], nrm[KBO_MAX_POLY_VERTS]; float radius, bound; } shape_t;

/* ---- derived per-config parameters ------------------------------------------------------- */
typedef struct {
    float xmin, ymin, xmax, ymax;   /* arena, world units */
    float cell, inv_cell; int gw, gh;
    float r_bot, im_bot;            /* world radius, inverse mass */
    float kl_bot, ka_bot;           /* Pade damping factors 1/(1+h c), b2Island.cpp */
    float im_mode[5];               /* inverse mass of a kilobot by drive law (KBO_DRIVE_MIXED; otherwise all im_bot) */
    float h;
    float r_obj[KBO_MAX_OBJECTS];   /* per fixture: contact radius (circle) / bounding radius about the body's centre of mass */
    float im_obj[KBO_MAX_OBJECTS], ii_obj[KBO_MAX_OBJECTS];   /* per object: inverse mass, inverse inertia about the centre of mass */
    v2 lc_obj[KBO_MAX_OBJECTS];     /* per object: b2Sweep::localCenter (centre of mass in the body frame) */
    float kl_obj, ka_obj;
    int nfix, fix_body[KBO_MAX_OBJECTS];
    shape_t shape[KBO_MAX_OBJECTS]; /* per fixture, body frame */
    float mu_oo, mu_ow;             /* b2MixFriction: sqrt(f1 * f2) */
} derived_t;

/* b2PolygonShape::ComputeMass (triangle fan about the vertex average): mass, centroid, inertia about the body origin */
static void polygon_mass(const shape_t *sh, float density, float *mass, v2 *centroid, float *inertia) {
    v2 center = V2(0.0f, 0.0f), s = V2(0.0f, 0.0f);
    float area = 0.0f, I = 0.0f;
    for (int i = 0; i < sh->n; ++i) s = v_add(s, sh->v[i]);
    s = v_scale(1.0f / (float)sh->n, s);
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < sh->n; ++i) {
        v2 e1 = v_sub(sh->v[i], s), e2 = v_sub(sh->v[i + 1 < sh->n ? i + 1 : 0], s);
        float D = v_cross(e1, e2);
        float triangleArea = 0.5f * D;
        area += triangleArea;
        center = v_add(center, v_scale(triangleArea * k_inv3, v_add(e1, e2)));
        float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x;
        float inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
        I += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    float m = density * area;
    center = v_scale(1.0f / area, center);
    v2 c = v_add(center, s);
    float Io = density * I;
    Io += m * (v_dot(c, c) - v_dot(center, center));
    *mass = m; *centroid = c; *inertia = Io;
}

static float damping_factor(int model, float h, float c) {
    if (model == 1) return clampf(1.0f - h * c, 0.0f, 1.0f);
    return 1.0f / (1.0f + h * c);
}

static void derive(const kbo_config *c, derived_t *d) {
    float W = c->world_width * WORLD_SCALE, H = c->world_height * WORLD_SCALE;
    d->xmin = -0.5f * W; d->xmax = 0.5f * W; d->ymin = -0.5f * H; d->ymax = 0.5f * H;
    float cell = CELL_SIZE;
    float dmin = 2.0f * c->bot_radius * WORLD_SCALE;
    while (cell < dmin) cell *= 2.0f;
    for (;;) {
        d->inv_cell = 1.0f / cell;
        d->gw = (int)ceilf(W * d->inv_cell); if (d->gw < 1) d->gw = 1;
        d->gh = (int)ceilf(H * d->inv_cell); if (d->gh < 1) d->gh = 1;
        if ((long)d->gw * d->gh <= MAX_CELLS) break;
        cell *= 2.0f;
    }
    d->cell = cell;
    d->h = c->dt;
    d->r_bot = c->bot_radius * WORLD_SCALE;
    /* b2CircleShape::ComputeMass: mass = density * pi * r^2 */
    float m = c->bot_density * B2_PI * d->r_bot * d->r_bot;
    d->im_bot = m > 0.0f ? 1.0f / m : 0.0f;
    for (int k = 0; k < 5; ++k) {
        d->im_mode[k] = d->im_bot;
        if (c->drive_mode == KBO_DRIVE_MIXED && c->mode_density[k] > 0.0f) {
            const float mk = c->mode_density[k] * B2_PI * d->r_bot * d->r_bot;
            d->im_mode[k] = mk > 0.0f ? 1.0f / mk : 0.0f;
        }
    }
    /* b2Island::Solve: `v *= 1.0f / (1.0f + h * damping)` since Box2D 2.3.1; `v *= b2Clamp(1.0f - h * damping, 0, 1)` before */
    d->kl_bot = damping_factor(c->damping_model, d->h, c->bot_linear_damping);
    d->ka_bot = damping_factor(c->damping_model, d->h, c->bot_angular_damping);
    d->nfix = c->num_fixtures > 0 ? c->num_fixtures : c->num_objects;
    float bm[KBO_MAX_OBJECTS], bi[KBO_MAX_OBJECTS];
    v2 bc[KBO_MAX_OBJECTS];
    for (int m = 0; m < KBO_MAX_OBJECTS; ++m) { bm[m] = 0.0f; bi[m] = 0.0f; bc[m] = V2(0.0f, 0.0f); d->fix_body[m] = m; }
    for (int k = 0; k < KBO_MAX_OBJECTS; ++k) {
        shape_t *sh = &d->shape[k];
        memset(sh, 0, sizeof(*sh));
        if (k >= d->nfix) { d->r_obj[k] = 0.0f; continue; }
        if (c->num_fixtures > 0) d->fix_body[k] = c->obj_fixture_body[k];
        sh->kind = c->obj_shape[k];
        float mo, io;
        v2 ce = V2(0.0f, 0.0f);
        if (sh->kind == KBO_SHAPE_CIRCLE) {
            float r = c->obj_radius[k] * WORLD_SCALE;
            sh->radius = r;
            mo = c->obj_density * B2_PI * r * r;                           /* b2CircleShape::ComputeMass */
            io = mo * (0.5f * r * r);                                      /* I = mass * (0.5 r^2 + |p|^2), p = 0 */
        } else {
            if (sh->kind == KBO_SHAPE_BOX) {                               /* b2PolygonShape::SetAsBox */
                float hx = c->obj_verts[k][0][0], hy = c->obj_verts[k][0][1];
                sh->n = 4;
                sh->v[0] = V2(-hx, -hy); sh->v[1] = V2(hx, -hy); sh->v[2] = V2(hx, hy); sh->v[3] = V2(-hx, hy);
                sh->nrm[0] = V2(0.0f, -1.0f); sh->nrm[1] = V2(1.0f, 0.0f); sh->nrm[2] = V2(0.0f, 1.0f); sh->nrm[3] = V2(-1.0f, 0.0f);
            } else {                                                       /* b2PolygonShape::Set on an ordered hull */
                sh->n = c->obj_nverts[k] < 3 ? 3 : (c->obj_nverts[k] > KBO_MAX_POLY_VERTS ? KBO_MAX_POLY_VERTS : c->obj_nverts[k]);
                for (int i = 0; i < sh->n; ++i) sh->v[i] = V2(c->obj_verts[k][i][0], c->obj_verts[k][i][1]);
                for (int i = 0; i < sh->n; ++i) {
                    v2 edge = v_sub(sh->v[i + 1 < sh->n ? i + 1 : 0], sh->v[i]);
                    sh->nrm[i] = v_normalize(v_cross_vs(edge, 1.0f));
                }
            }
            sh->radius = B2_POLYGON_RADIUS;
            polygon_mass(sh, c->obj_density, &mo, &ce, &io);
        }
        /* b2Body::ResetMassData: accumulate over the fixtures of the body */
        const int m = d->fix_body[k];
        bm[m] += mo; bc[m] = v_add(bc[m], v_scale(mo, ce)); bi[m] += io;
    }
    for (int m = 0; m < KBO_MAX_OBJECTS; ++m) {
        d->lc_obj[m] = V2(0.0f, 0.0f);
        if (bm[m] > 0.0f) {
            d->im_obj[m] = 1.0f / bm[m];
            d->lc_obj[m] = v_scale(d->im_obj[m], bc[m]);
        } else d->im_obj[m] = 0.0f;
        float io = bi[m] - bm[m] * v_dot(d->lc_obj[m], d->lc_obj[m]);     /* inertia about the centre of mass */
        d->ii_obj[m] = io > 0.0f ? 1.0f / io : 0.0f;
    }
    for (int k = 0; k < d->nfix; ++k) {           /* bounding radius about the centre of mass of the body */
        shape_t *sh = &d->shape[k];
        if (sh->kind == KBO_SHAPE_CIRCLE) { sh->bound = sh->radius; d->r_obj[k] = sh->radius; continue; }
        float far2 = 0.0f;
        for (int i = 0; i < sh->n; ++i) { v2 q = v_sub(sh->v[i], d->lc_obj[d->fix_body[k]]); far2 = fmaxf(far2, v_dot(q, q)); }
        sh->bound = sqrtf(far2) + B2_POLYGON_RADIUS;
        d->r_obj[k] = sh->bound;
    }
    d->mu_oo = sqrtf(c->obj_friction * c->obj_friction);
    d->mu_ow = sqrtf(c->obj_friction * c->wall_friction);
    d->kl_obj = damping_factor(c->damping_model, d->h, c->obj_linear_damping);
    d->ka_obj = damping_factor(c->damping_model, d->h, c->obj_angular_damping);
}

/* ---- light sensing: CircularGradientLight.value_and_gradients, light.py:176-189 ------------- */
static void light_circular(float sx, float sy, float lx, float ly, float R, float *val, float *gx, float *gy) {
    float dx = -1.0f * (sx - lx), dy = -1.0f * (sy - ly);
    float n = sqrtf(dx * dx + dy * dy);
    float v = 1.0f - n / R;
    v = fmaxf(fminf(v, 1.0f), 0.0f);
    *val = v * 255.0f;
    if (n > 0.0f) { dx = dx / n; dy = dy / n; }     /* reference: NaN at n == 0; oracle: zero gradient */
    else { dx = 0.0f; dy = 0.0f; }
    if (n > R) { dx *= 0.0f; dy *= 0.0f; }
    *gx = dx; *gy = dy;
}

/* ---- light models on top of light_circular ------------------------------------------------------
 * per-env light state: light_x/y/vx/vy [num_envs][count]; GradientLight keeps its angle in light_x. */
typedef struct { int kind; float radius, maxv, lo[2], hi[2], alo[2], ahi[2]; } lightc_t;

static int light_components(const kbo_config *c, lightc_t *L) {
    int n = c->light_type == KBO_LIGHT_COMPOSITE ? c->light_count : 1;
    for (int i = 0; i < n; ++i) {
        if (c->light_type == KBO_LIGHT_COMPOSITE) {
            L[i].kind = c->light_kind[i]; L[i].radius = c->lightc_radius[i]; L[i].maxv = c->lightc_max_velocity[i];
            for (int k = 0; k < 2; ++k) {
                L[i].lo[k] = c->lightc_lo[i][k]; L[i].hi[k] = c->lightc_hi[i][k];
                L[i].alo[k] = c->lightc_act_lo[i][k]; L[i].ahi[k] = c->lightc_act_hi[i][k];
            }
        } else {
            L[i].kind = c->light_type; L[i].radius = c->light_radius; L[i].maxv = c->light_max_velocity;
            for (int k = 0; k < 2; ++k) {
                L[i].lo[k] = c->light_lo[k]; L[i].hi[k] = c->light_hi[k];
                L[i].alo[k] = c->light_act_lo[k]; L[i].ahi[k] = c->light_act_hi[k];
            }
        }
    }
    return n;
}

int kbo_light_action_dim(const kbo_config *c) {
    if (c->light_type == KBO_LIGHT_NONE) return 0;
    if (c->light_type == KBO_LIGHT_GRADIENT) return 1;
    return 2 * (c->light_type == KBO_LIGHT_COMPOSITE ? c->light_count : 1);
}

/* Light.step of every component (light.py:59-75, 237-253, 300-316, 122-127) */
static void light_step_env(const kbo_config *cfg, kbo_state *st, int e, const float *action, float h) {
    lightc_t L[KBO_MAX_LIGHTS];
    const int n = light_components(cfg, L);
    const int adim = kbo_light_action_dim(cfg);
    const float *a = action + (size_t)e * adim;
    if (cfg->light_type == KBO_LIGHT_GRADIENT) {
        /* GradientLight.step: absolute angle action clamped to +-2 pi, wrapped once into [-pi, pi] */
        const float pi = 3.14159265358979323846f;
        float ang = fminf(fmaxf(a[0], -2.0f * pi), 2.0f * pi);
        if (ang < -pi) ang += 2.0f * pi;
        if (ang > pi) ang -= 2.0f * pi;
        st->light_x[e] = ang;
        return;
    }
    for (int i = 0; i < n; ++i) {
        const size_t li = (size_t)e * n + i;
        float ax = fminf(fmaxf(a[2 * i + 0], L[i].alo[0]), L[i].ahi[0]);
        float ay = fminf(fmaxf(a[2 * i + 1], L[i].alo[1]), L[i].ahi[1]);
        float lx, ly;
        if (L[i].kind == KBO_LIGHT_MOMENTUM) {
            float vx = st->light_vx[li] + ax * h, vy = st->light_vy[li] + ay * h;
            float nv = sqrtf(vx * vx + vy * vy);
            if (nv > L[i].maxv) { float sc = L[i].maxv / nv; vx *= sc; vy *= sc; }
            st->light_vx[li] = vx; st->light_vy[li] = vy;
            lx = st->light_x[li] + vx * h; ly = st->light_y[li] + vy * h;
        } else {
            lx = st->light_x[li] + ax * h; ly = st->light_y[li] + ay * h;
        }
        st->light_x[li] = fminf(fmaxf(lx, L[i].lo[0]), L[i].hi[0]);
        st->light_y[li] = fminf(fmaxf(ly, L[i].lo[1]), L[i].hi[1]);
    }
}

/* value_and_gradients at one sensor position (metres): light.py:176-189, 137-141; GradientLight: the evident
 * intent of light.py:255-260 (projection on the gradient direction; the reference's dot() only runs for N == 2) */
static void light_sense(const kbo_config *cfg, const kbo_state *st, int e, float sx, float sy, float *val, float *gx, float *gy) {
    lightc_t L[KBO_MAX_LIGHTS];
    const int n = light_components(cfg, L);
    if (cfg->light_type == KBO_LIGHT_GRADIENT) {
        float s, c; kbo_sincosf(st->light_x[e], &s, &c);
        *val = c * sx + s * sy; *gx = c; *gy = s;
        return;
    }
    float vsum = 0.0f, vbest = 0.0f, bgx = 0.0f, bgy = 0.0f;
    for (int i = 0; i < n; ++i) {
        float v, x, y;
        light_circular(sx, sy, st->light_x[(size_t)e * n + i], st->light_y[(size_t)e * n + i], L[i].radius, &v, &x, &y);
        vsum = i == 0 ? v : vsum + v;
        if (i == 0 || v > vbest) { vbest = v; bgx = x; bgy = y; }     /* np.argmax: first maximum */
    }
    *val = vsum; *gx = bgx; *gy = bgy;
}

/* ---- Kilobot.step motor law, kilobot.py:86-127 (world-unit body velocity out) ---------------- */
static void motor_law(int ml, int mr, float th, float h, float *vx, float *vy, float *w) {
    const float max_lin = 0.01f, max_ang = 0.5f * 3.14159265358979323846f;
    float s, c; kbo_sincosf(th, &s, &c);
    if (ml && mr) {
        /* kilobot.py:97-101: evident intent (the reference raises TypeError at :127) */
        float lin = (float)(mr + ml) / 510.0f * max_lin;
        *vx = (s * lin) * WORLD_SCALE; *vy = (c * lin) * WORLD_SCALE;
        *w = (float)(mr - ml) / 510.0f * max_ang;
    } else if (mr || ml) {
        /* kilobot.py:103-121: pivot about the opposite leg */
        float av, lx, ly = -0.009f;
        if (mr) { av = (float)mr / 255.0f * max_ang; lx = -0.013f; }
        else { av = -(float)ml / 255.0f * max_ang; lx = 0.013f; }
        float ds, dc; kbo_sincosf(av * h, &ds, &dc);
        float tx = lx - (dc * lx - ds * ly), ty = ly - (ds * lx + dc * ly);
        tx *= WORLD_SCALE; ty *= WORLD_SCALE;
        float wx = c * tx - s * ty, wy = s * tx + c * ty;      /* b2Body::GetWorldVector */
        wx = wx / WORLD_SCALE / h; wy = wy / WORLD_SCALE / h;
        *vx = wx * WORLD_SCALE; *vy = wy * WORLD_SCALE; *w = av;
    } else {
        *vx = 0.0f; *vy = 0.0f; *w = 0.0f;                     /* reference: TypeError */
    }
}

/* ---- contacts ----------------------------------------------------------------------------- */
typedef struct {
    int a, b;           /* body indices: bot 0..N-1, object N+m; wall contacts: a = -1-w (static), b = body */
    int cls; int group;
    float nx, ny;       /* velocity-phase normal (A -> B) */
    float acc;          /* accumulated normal impulse */
    float ima, imb;
    float ra, rb;
    int owner, slot;    /* warm-start slot (owner bot, slot index) or -1 */
    /* kilobot - polygon object (b2CollidePolygonAndCircle; Box2D's A = the polygon = body b, B = the kilobot = body a) */
    int poly, fix;      /* fix: the fixture of the object that is touched */
    int skip;           /* sleeping: both bodies belong to an island without an awake body, the contact is not solved */
    v2 ln, lp;          /* manifold: localNormal, localPoint in the polygon's frame */
    v2 rA;              /* velocity phase: contact point relative to the polygon's centre */
    float nmass;        /* velocity phase: normalMass */
} contact_t;

/* manifold constraint: an object-object or object-wall contact with Box2D's full contact model */
typedef struct {
    int type, count;            /* b2Manifold::e_circles 0 / e_faceA 1 / e_faceB 2; pointCount */
    v2 localNormal, localPoint;
    v2 lp[2]; int id[2];        /* b2ManifoldPoint: localPoint, id (contact feature key, here a small code) */
} manifold_t;
typedef struct {
    int a, b;                   /* bodies; a = -1 - wall for a wall */
    int owner, col;             /* warm-start table entry ows[owner fixture][col]: col = partner fixture or 8 + wall */
    manifold_t m;
    float friction, radA, radB;
    float nimp[2], timp[2];
    v2 normal, rA[2], rB[2];    /* b2ContactVelocityConstraint */
    float nmass[2], tmass[2];
    float k11, k12, k22, n11, n12, n21, n22;   /* K and normalMass = inverse(K) */
    int vcount;
    int skip;                   /* sleeping island: not solved */
} mc_t;

static int contact_cmp(const void *pa, const void *pb) {
    const contact_t *x = (const contact_t *)pa, *y = (const contact_t *)pb;
    if (x->cls != y->cls) return x->cls < y->cls ? -1 : 1;
    if (x->group != y->group) return x->group < y->group ? -1 : 1;
    int xa = x->a < 0 ? -1 - x->a : x->a, ya = y->a < 0 ? -1 - y->a : y->a;   /* walls: ascending edge index */
    if (xa != ya) return xa < ya ? -1 : 1;
    if (x->b != y->b) return x->b < y->b ? -1 : 1;
    return 0;
}

typedef struct {
    int N, M, S;
    float *px, *py, *vx, *vy, *bw;    /* working copies, bots then objects */
    int *cell, *cx, *cy;
    int *cell_start, *cell_items;
    contact_t *con; int ncon, cap;
    mc_t mc[MAX_MC]; int nmc;
    const derived_t *d;
    float *ang;            /* working copy of the object angles (index N + m) */
    int *parent; unsigned char *active, *next_active;
    /* sleeping (b2Island::Solve, b2World::Solve): per body sleep time (< 0: asleep), per island root: awake / not solved / min sleep time */
    float *slp, *isl_min; unsigned char *isl_awake, *isl_unsolved;
    int *woff, *wnoff;     /* packed warm-start offsets: previous / next substep */
    unsigned char *ws_hit; /* previous substep's packed list: entry matched by a contact of this substep (b2Contact::Update wasTouching) */
    float *x0, *y0, *a0;   /* poses at the start of the step (continuous step) */
    int status;
} work_t;

static int uf_find(int *p, int i) { while (p[i] != i) { p[i] = p[p[i]]; i = p[i]; } return i; }
static void uf_union(int *p, int a, int b) {
    a = uf_find(p, a); b = uf_find(p, b);
    if (a == b) return;
    if (a < b) p[b] = a; else p[a] = b;
}

/* warm-start store: per env a packed list of (key, impulse) entries, owner bots in ascending order,
 * ws_cnt[owner] entries each; entry (owner, slot) sits at woff[owner] + slot */
static float ws_lookup(const kbo_state *st, const work_t *w, int e, int owner, unsigned key) {
    int cnt = st->ws_cnt[(size_t)e * w->N + owner];
    for (int s = 0; s < cnt && w->woff[owner] + s < w->cap; ++s) {   /* (entries behind the capacity were never stored) */
        size_t idx = (size_t)e * w->cap + (size_t)(w->woff[owner] + s);
        if (st->ws_key[idx] == key) { w->ws_hit[w->woff[owner] + s] = 1; return st->ws_acc[idx]; }
    }
    return -1.0f;  /* accumulated impulses are >= 0 */
}

static void add_contact(work_t *w, int a, int b, int cls, int group, float ima, float imb,
                        float ra, float rb, float acc, int owner, int slot) {
    if (w->ncon >= w->cap) { w->status |= 1; return; }
    contact_t *c = &w->con[w->ncon++];
    c->a = a; c->b = b; c->cls = cls; c->group = group; c->ima = ima; c->imb = imb;
    c->ra = ra; c->rb = rb; c->acc = acc; c->owner = owner; c->slot = slot; c->nx = 0; c->ny = 0;
    c->poly = 0; c->fix = 0; c->ln = V2(0.0f, 0.0f); c->lp = V2(0.0f, 0.0f); c->rA = V2(0.0f, 0.0f); c->nmass = 0.0f;
}

static inline float wall_dist(const derived_t *d, int wl, float x, float y, float *nx, float *ny);

/* ---- narrowphase of the object shapes (b2CollideCircle.cpp, b2CollidePolygon.cpp, b2CollideEdge.cpp) ---------- */

/* b2CollidePolygonAndCircle: polygon A at xfA, circle centre c (world), circle radius rc */
static int collide_poly_circle(const shape_t *P, const xf_t *xfA, v2 c, float rc, v2 *ln, v2 *lp) {
    const v2 cLocal = xf_mulT(xfA, c);
    int normalIndex = 0;
    float separation = -3.402823466e+38f;
    const float radius = P->radius + rc;
    for (int i = 0; i < P->n; ++i) {
        float sp = v_dot(P->nrm[i], v_sub(cLocal, P->v[i]));
        if (sp > radius) return 0;
        if (sp > separation) { separation = sp; normalIndex = i; }
    }
    const int i1 = normalIndex, i2 = i1 + 1 < P->n ? i1 + 1 : 0;
    const v2 v1 = P->v[i1], v2_ = P->v[i2];
    if (separation < B2_EPSILON) {                      /* centre inside the polygon */
        *ln = P->nrm[normalIndex]; *lp = v_scale(0.5f, v_add(v1, v2_));
        return 1;
    }
    const float u1 = v_dot(v_sub(cLocal, v1), v_sub(v2_, v1));
    const float u2 = v_dot(v_sub(cLocal, v2_), v_sub(v1, v2_));
    if (u1 <= 0.0f) {
        v2 dd = v_sub(cLocal, v1);
        if (v_dot(dd, dd) > radius * radius) return 0;
        *ln = v_normalize(dd); *lp = v1;
    } else if (u2 <= 0.0f) {
        v2 dd = v_sub(cLocal, v2_);
        if (v_dot(dd, dd) > radius * radius) return 0;
        *ln = v_normalize(dd); *lp = v2_;
    } else {
        v2 faceCenter = v_scale(0.5f, v_add(v1, v2_));
        float sp = v_dot(v_sub(cLocal, faceCenter), P->nrm[i1]);
        if (sp > radius) return 0;
        *ln = P->nrm[i1]; *lp = faceCenter;
    }
    return 1;
}

/* b2FindMaxSeparation (Box2D >= 2.3.1 brute-force form) */
static float find_max_separation(int *edgeIndex, const shape_t *p1, const xf_t *xf1, const shape_t *p2, const xf_t *xf2) {
    const xf_t xf = xf_mulT_xf(xf2, xf1);
    int bestIndex = 0;
    float maxSeparation = -3.402823466e+38f;
    for (int i = 0; i < p1->n; ++i) {
        v2 n = rot_mul(&xf, p1->nrm[i]);
        v2 v1 = xf_mul(&xf, p1->v[i]);
        float si = 3.402823466e+38f;
        for (int j = 0; j < p2->n; ++j) {
            float sij = v_dot(n, v_sub(p2->v[j], v1));
            if (sij < si) si = sij;
        }
        if (si > maxSeparation) { maxSeparation = si; bestIndex = i; }
    }
    *edgeIndex = bestIndex;
    return maxSeparation;
}

typedef struct { v2 v; int ia, ib, ta, tb; } clipv_t;     /* b2ClipVertex: point + contact feature */
#define FEAT_VERTEX 0
#define FEAT_FACE 1
static inline int feature_key(int ia, int ib, int ta, int tb) { return ia | (ib << 2) | (ta << 4) | (tb << 5); }

static int clip_segment_to_line(clipv_t vOut[2], const clipv_t vIn[2], v2 normal, float offset, int vertexIndexA) {
    int numOut = 0;
    float distance0 = v_dot(normal, vIn[0].v) - offset;
    float distance1 = v_dot(normal, vIn[1].v) - offset;
    if (distance0 <= 0.0f) vOut[numOut++] = vIn[0];
    if (distance1 <= 0.0f) vOut[numOut++] = vIn[1];
    if (distance0 * distance1 < 0.0f) {
        float interp = distance0 / (distance0 - distance1);
        vOut[numOut].v = v_add(vIn[0].v, v_scale(interp, v_sub(vIn[1].v, vIn[0].v)));
        vOut[numOut].ia = vertexIndexA; vOut[numOut].ib = vIn[0].ib;
        vOut[numOut].ta = FEAT_VERTEX; vOut[numOut].tb = FEAT_FACE;
        ++numOut;
    }
    return numOut;
}

/* b2CollidePolygons */
static void collide_polygons(manifold_t *m, const shape_t *pA, const xf_t *xfA, const shape_t *pB, const xf_t *xfB) {
    m->count = 0;
    const float totalRadius = pA->radius + pB->radius;
    int edgeA = 0, edgeB = 0;
    float separationA = find_max_separation(&edgeA, pA, xfA, pB, xfB);
    if (separationA > totalRadius) return;
    float separationB = find_max_separation(&edgeB, pB, xfB, pA, xfA);
    if (separationB > totalRadius) return;
    const shape_t *p1, *p2; const xf_t *xf1, *xf2; int edge1, flip;
    const float k_tol = 0.1f * B2_LINEAR_SLOP;
    if (separationB > separationA + k_tol) { p1 = pB; p2 = pA; xf1 = xfB; xf2 = xfA; edge1 = edgeB; m->type = 2; flip = 1; }
    else { p1 = pA; p2 = pB; xf1 = xfA; xf2 = xfB; edge1 = edgeA; m->type = 1; flip = 0; }
    clipv_t incident[2];
    {   /* b2FindIncidentEdge */
        v2 normal1 = rot_mulT(xf2, rot_mul(xf1, p1->nrm[edge1]));
        int index = 0; float minDot = 3.402823466e+38f;
        for (int i = 0; i < p2->n; ++i) { float dt = v_dot(normal1, p2->nrm[i]); if (dt < minDot) { minDot = dt; index = i; } }
        int i1 = index, i2 = i1 + 1 < p2->n ? i1 + 1 : 0;
        incident[0].v = xf_mul(xf2, p2->v[i1]); incident[0].ia = edge1; incident[0].ib = i1; incident[0].ta = FEAT_FACE; incident[0].tb = FEAT_VERTEX;
        incident[1].v = xf_mul(xf2, p2->v[i2]); incident[1].ia = edge1; incident[1].ib = i2; incident[1].ta = FEAT_FACE; incident[1].tb = FEAT_VERTEX;
    }
    const int iv1 = edge1, iv2 = edge1 + 1 < p1->n ? edge1 + 1 : 0;
    v2 v11 = p1->v[iv1], v12 = p1->v[iv2];
    v2 localTangent = v_normalize(v_sub(v12, v11));
    v2 localNormal = v_cross_vs(localTangent, 1.0f);
    v2 planePoint = v_scale(0.5f, v_add(v11, v12));
    v2 tangent = rot_mul(xf1, localTangent);
    v2 normal = v_cross_vs(tangent, 1.0f);
    v11 = xf_mul(xf1, v11); v12 = xf_mul(xf1, v12);
    float frontOffset = v_dot(normal, v11);
    float sideOffset1 = -v_dot(tangent, v11) + totalRadius;
    float sideOffset2 = v_dot(tangent, v12) + totalRadius;
    clipv_t clip1[2], clip2[2];
    if (clip_segment_to_line(clip1, incident, v_neg(tangent), sideOffset1, iv1) < 2) return;
    if (clip_segment_to_line(clip2, clip1, tangent, sideOffset2, iv2) < 2) return;
    m->localNormal = localNormal; m->localPoint = planePoint;
    int pc = 0;
    for (int i = 0; i < 2; ++i) {
        float separation = v_dot(normal, clip2[i].v) - frontOffset;
        if (separation <= totalRadius) {
            m->lp[pc] = xf_mulT(xf2, clip2[i].v);
            m->id[pc] = flip ? feature_key(clip2[i].ib, clip2[i].ia, clip2[i].tb, clip2[i].ta)
                             : feature_key(clip2[i].ia, clip2[i].ib, clip2[i].ta, clip2[i].tb);
            ++pc;
        }
    }
    m->count = pc;
}

/* wall w: a point on it (vertex v1 of its chain edge, kilobots_env.py:48-51) */
static inline v2 wall_point(const derived_t *d, int wl) {
    switch (wl) {
    case 0: return V2(d->xmin, d->ymax);
    case 1: return V2(d->xmin, d->ymin);
    case 2: return V2(d->xmax, d->ymin);
    default: return V2(d->xmax, d->ymax);
    }
}
static inline v2 wall_normal(int wl) {       /* pointing into the arena */
    switch (wl) {
    case 0: return V2(1.0f, 0.0f);
    case 1: return V2(0.0f, 1.0f);
    case 2: return V2(-1.0f, 0.0f);
    default: return V2(0.0f, -1.0f);
    }
}

/* b2CollideEdgeAndPolygon for a long two-sided edge with the polygon on its inner side: the edge is the reference
 * face, the incident edge is the polygon edge most anti-parallel to the wall normal, both of its vertices are
 * candidates (the side planes of a long edge never clip), kept when within 2 * polygonRadius */
static void collide_wall_poly(manifold_t *m, const derived_t *d, int wl, const shape_t *P, const xf_t *xfB) {
    m->count = 0;
    const v2 n = wall_normal(wl), v1 = wall_point(d, wl);
    const float radius = 2.0f * B2_POLYGON_RADIUS;
    v2 wv[KBO_MAX_POLY_VERTS], wn[KBO_MAX_POLY_VERTS];
    float edgeSep = 3.402823466e+38f;
    for (int i = 0; i < P->n; ++i) {
        wv[i] = xf_mul(xfB, P->v[i]); wn[i] = rot_mul(xfB, P->nrm[i]);
        float sp = v_dot(n, v_sub(wv[i], v1));
        if (sp < edgeSep) edgeSep = sp;
    }
    if (edgeSep > radius) return;
    int bestIndex = 0; float bestValue = v_dot(n, wn[0]);
    for (int i = 1; i < P->n; ++i) { float value = v_dot(n, wn[i]); if (value < bestValue) { bestValue = value; bestIndex = i; } }
    const int i1 = bestIndex, i2 = i1 + 1 < P->n ? i1 + 1 : 0;
    m->type = 1; m->localNormal = n; m->localPoint = v1;
    int pc = 0;
    const int idx[2] = {i1, i2};
    for (int k = 0; k < 2; ++k) {
        float separation = v_dot(n, v_sub(wv[idx[k]], v1));
        if (separation <= radius) { m->lp[pc] = P->v[idx[k]]; m->id[pc] = feature_key(0, idx[k], FEAT_FACE, FEAT_VERTEX); ++pc; }
    }
    m->count = pc;
}

/* b2Body::SynchronizeTransform: q from the angle, p = c - q * localCenter (c = centre of mass) */
static inline xf_t xf_of_body(const derived_t *d, int m, v2 c, float a) {
    xf_t t = xf_make(0.0f, 0.0f, a);
    const v2 lc = d->lc_obj[m];
    t.p = V2(c.x - (t.c * lc.x - t.s * lc.y), c.y - (t.s * lc.x + t.c * lc.y));
    return t;
}
static inline xf_t body_xf(const work_t *w, int b) {
    if (b < 0) { xf_t t; t.p = V2(0.0f, 0.0f); t.s = 0.0f; t.c = 1.0f; return t; }     /* static arena body */
    return xf_of_body(w->d, b - w->N, V2(w->px[b], w->py[b]), w->ang[b]);
}

/* previous manifold of the same pair -> impulses of the points whose feature id is unchanged (b2Contact::Update) */
static void mc_warm(const kbo_state *st, int e, mc_t *c) {
    const float *old = st->ows_acc + (((size_t)e * KBO_MAX_OBJECTS + c->owner) * OWS + c->col) * OWW;
    for (int j = 0; j < c->m.count; ++j) {
        c->nimp[j] = 0.0f; c->timp[j] = 0.0f;
        for (int k = 0; k < 2; ++k)
            if (old[3 * k] >= 0.0f && (int)old[3 * k] == c->m.id[j]) { c->nimp[j] = old[3 * k + 1]; c->timp[j] = old[3 * k + 2]; break; }
    }
}

/* manifold of fixture f against wall wl at the current pose of its body (impulses zero); 0: not touching */
static int wall_manifold(const derived_t *d, const work_t *w, int f, int wl, mc_t *out) {
    const shape_t *sh = &d->shape[f];
    const int N = w->N, m = d->fix_body[f];
    mc_t c; memset(&c, 0, sizeof(c));
    c.a = -1 - wl; c.b = N + m; c.owner = f; c.col = 8 + wl; c.friction = d->mu_ow;
    c.radA = B2_POLYGON_RADIUS; c.radB = sh->radius;
    if (sh->kind == KBO_SHAPE_CIRCLE) {                                        /* b2CollideEdgeAndCircle, region AB */
        float nx, ny; float dist = wall_dist(d, wl, w->px[N + m], w->py[N + m], &nx, &ny);
        float rwo = B2_POLYGON_RADIUS + sh->radius;
        if (dist * dist > rwo * rwo) return 0;
        if (dist < 0.0f) { nx = -nx; ny = -ny; }
        c.m.type = 1; c.m.count = 1; c.m.localNormal = V2(nx, ny); c.m.localPoint = wall_point(d, wl);
        c.m.lp[0] = V2(0.0f, 0.0f); c.m.id[0] = 0;
    } else {
        xf_t xb = body_xf(w, c.b);
        collide_wall_poly(&c.m, d, wl, sh, &xb);
        if (c.m.count == 0) return 0;
    }
    *out = c;
    return 1;
}

/* all object-object and object-wall manifolds of one env, in canonical order: fixture pairs (f1 < f2, different
 * bodies) lexicographically, then (fixture, wall) */
static void detect_mc(const derived_t *d, const kbo_state *st, int e, work_t *w) {
    const int N = w->N, F = d->nfix;
    w->nmc = 0;
    for (int f1 = 0; f1 < F; ++f1)
        for (int f2 = f1 + 1; f2 < F; ++f2) {
            const int m1 = d->fix_body[f1], m2 = d->fix_body[f2];
            if (m1 == m2) continue;                         /* fixtures of one body never collide */
            const shape_t *s1 = &d->shape[f1], *s2 = &d->shape[f2];
            {   /* bounding circles about the centres of mass (stand-in for the broadphase; never rejects a touching pair) */
                float dx = w->px[N + m2] - w->px[N + m1], dy = w->py[N + m2] - w->py[N + m1];
                float rb = s1->bound + s2->bound;
                if (dx * dx + dy * dy > rb * rb) continue;
            }
            mc_t c; memset(&c, 0, sizeof(c));
            c.owner = f1; c.col = f2; c.friction = d->mu_oo;
            if (s1->kind == KBO_SHAPE_CIRCLE && s2->kind == KBO_SHAPE_CIRCLE) {      /* b2CollideCircles */
                float dx = w->px[N + m2] - w->px[N + m1], dy = w->py[N + m2] - w->py[N + m1];
                float rr = s1->radius + s2->radius;
                if (dx * dx + dy * dy > rr * rr) continue;
                c.a = N + m1; c.b = N + m2; c.m.type = 0; c.m.count = 1; c.m.id[0] = 0;
                c.m.localPoint = V2(0.0f, 0.0f); c.m.lp[0] = V2(0.0f, 0.0f);
                c.radA = s1->radius; c.radB = s2->radius;
            } else if (s1->kind != KBO_SHAPE_CIRCLE && s2->kind != KBO_SHAPE_CIRCLE) {
                c.a = N + m1; c.b = N + m2;
                xf_t xa = body_xf(w, c.a), xb = body_xf(w, c.b);
                collide_polygons(&c.m, s1, &xa, s2, &xb);
                if (c.m.count == 0) continue;
                c.radA = s1->radius; c.radB = s2->radius;
            } else {                                                                   /* polygon is fixture A */
                const int fp = s1->kind != KBO_SHAPE_CIRCLE ? f1 : f2, fc = fp == f1 ? f2 : f1;
                c.a = N + d->fix_body[fp]; c.b = N + d->fix_body[fc];
                xf_t xa = body_xf(w, c.a);
                v2 ln, lp;
                if (!collide_poly_circle(&d->shape[fp], &xa, V2(w->px[c.b], w->py[c.b]), d->shape[fc].radius, &ln, &lp)) continue;
                c.m.type = 1; c.m.count = 1; c.m.localNormal = ln; c.m.localPoint = lp; c.m.lp[0] = V2(0.0f, 0.0f); c.m.id[0] = 0;
                c.radA = d->shape[fp].radius; c.radB = d->shape[fc].radius;
            }
            mc_warm(st, e, &c);
            w->mc[w->nmc++] = c;
        }
    for (int f = 0; f < F; ++f)
        for (int wl = 0; wl < 4; ++wl) {
            mc_t c;
            if (!wall_manifold(d, w, f, wl, &c)) continue;
            mc_warm(st, e, &c);
            w->mc[w->nmc++] = c;
        }
}

/* b2CollideCircles + canonical class/group; emits contacts owned by bot `a` in emission order */
static void detect_env(const kbo_config *cfg, const derived_t *d, const kbo_state *st, int e, work_t *w) {
    const int N = w->N, S = w->S;
    const int ncell = d->gw * d->gh;
    {   /* offsets of the previous substep's packed warm-start list */
        int run = 0;
        for (int b = 0; b < N; ++b) { w->woff[b] = run; run += st->ws_cnt[(size_t)e * N + b]; }
        if (run > 0) memset(w->ws_hit, 0, (size_t)(run < w->cap ? run : w->cap));
    }
    /* cells */
    memset(w->cell_start, 0, sizeof(int) * (ncell + 1));
    for (int b = 0; b < N; ++b) {
        int cx = (int)floorf((w->px[b] - d->xmin) * d->inv_cell);
        int cy = (int)floorf((w->py[b] - d->ymin) * d->inv_cell);
        cx = cx < 0 ? 0 : (cx >= d->gw ? d->gw - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= d->gh ? d->gh - 1 : cy);
        w->cx[b] = cx; w->cy[b] = cy; w->cell[b] = cy * d->gw + cx;
        w->cell_start[w->cell[b] + 1]++;
    }
    for (int c = 0; c < ncell; ++c) w->cell_start[c + 1] += w->cell_start[c];
    {   /* stable fill: ascending bot id inside each cell */
        int *fill = (int *)malloc(sizeof(int) * ncell);
        memcpy(fill, w->cell_start, sizeof(int) * ncell);
        for (int b = 0; b < N; ++b) w->cell_items[fill[w->cell[b]]++] = b;
        free(fill);
    }
    const float rr = d->r_bot + d->r_bot, rr2 = rr * rr;
    /* inverse mass of kilobot b: by its drive law in a mixed env (the classes have different densities, kilobot.py:25 / :214) */
#define IM_BOT(b_) (cfg->drive_mode == KBO_DRIVE_MIXED ? d->im_mode[st->bot_mode[(size_t)e * N + (b_)] < 5 ? st->bot_mode[(size_t)e * N + (b_)] : 0] : d->im_bot)
    static const int ddx[5] = {0, 1, 0, 1, -1}, ddy[5] = {0, 0, 1, 1, 1};
    static const int dcls[5] = {CLS_SAME, CLS_E, CLS_N, CLS_NE, CLS_NW};
    w->ncon = 0;
    for (int a = 0; a < N; ++a) {
        int nslot = 0;
        for (int k = 0; k < 5; ++k) {
            int ox = w->cx[a] + ddx[k], oy = w->cy[a] + ddy[k];
            if (ox < 0 || ox >= d->gw || oy < 0 || oy >= d->gh) continue;
            int oc = oy * d->gw + ox;
            for (int it = w->cell_start[oc]; it < w->cell_start[oc + 1]; ++it) {
                int b = w->cell_items[it];
                if (k == 0 && b <= a) continue;
                float dx = w->px[b] - w->px[a], dy = w->py[b] - w->py[a];
                float dd = dx * dx + dy * dy;
                if (dd > rr2) continue;                          /* b2CollideCircles */
                int cls = dcls[k];
                if (k == 1 || k == 3 || k == 4) cls += (w->cx[a] & 1);
                else if (k == 2) cls += (w->cy[a] & 1);
                float acc = ws_lookup(st, w, e, a, (unsigned)b);
                if (acc < 0.0f) acc = ws_lookup(st, w, e, b, (unsigned)a);
                if (acc < 0.0f) acc = 0.0f;
                int slot = nslot < S ? nslot : -1;
                if (slot < 0) w->status |= 2;
                nslot++;
                add_contact(w, a, b, cls, w->cell[a], IM_BOT(a), IM_BOT(b), d->r_bot, d->r_bot, acc, a, slot);
            }
        }
        /* walls: b2CollideEdgeAndCircle (region AB), edges of the chain loop kilobots_env.py:48-51:
         * 0 left (x = xmin), 1 bottom (y = ymin), 2 right (x = xmax), 3 top (y = ymax) */
        const float rw = B2_POLYGON_RADIUS + d->r_bot, rw2 = rw * rw;
        for (int wl = 0; wl < 4; ++wl) {
            float dist = wl == 0 ? w->px[a] - d->xmin : wl == 1 ? w->py[a] - d->ymin
                       : wl == 2 ? d->xmax - w->px[a] : d->ymax - w->py[a];
            if (dist * dist > rw2) continue;
            float acc = ws_lookup(st, w, e, a, KEY_WALL + (unsigned)wl);
            if (acc < 0.0f) acc = 0.0f;
            int slot = nslot < S ? nslot : -1;
            if (slot < 0) w->status |= 2;
            nslot++;
            add_contact(w, -1 - wl, a, CLS_WALL, a, 0.0f, IM_BOT(a), B2_POLYGON_RADIUS, d->r_bot, acc, a, slot);
        }
        /* pushable objects: b2CollideCircles / b2CollidePolygonAndCircle kilobot - fixture f of object m */
        for (int f = 0; f < d->nfix; ++f) {
            const shape_t *sh = &d->shape[f];
            const int m = d->fix_body[f];
            float dx = w->px[N + m] - w->px[a], dy = w->py[N + m] - w->py[a];
            float ro = d->r_bot + d->r_obj[f];                  /* circle: contact radius; polygon: bounding radius */
            if (dx * dx + dy * dy > ro * ro) continue;
            v2 ln = V2(0.0f, 0.0f), lp = V2(0.0f, 0.0f);
            if (sh->kind != KBO_SHAPE_CIRCLE) {
                xf_t xo = body_xf(w, N + m);
                if (!collide_poly_circle(sh, &xo, V2(w->px[a], w->py[a]), d->r_bot, &ln, &lp)) continue;
            }
            float acc = ws_lookup(st, w, e, a, KEY_OBJ + (unsigned)f);
            if (acc < 0.0f) acc = 0.0f;
            int slot = nslot < S ? nslot : -1;
            if (slot < 0) w->status |= 2;
            nslot++;
            const int before = w->ncon;
            add_contact(w, a, N + m, CLS_BOT_OBJ, f, IM_BOT(a), d->im_obj[m], d->r_bot, sh->radius, acc, a, slot);
            if (w->ncon > before) {
                contact_t *c = &w->con[w->ncon - 1];
                c->fix = f;
                if (sh->kind != KBO_SHAPE_CIRCLE) { c->poly = 1; c->ln = ln; c->lp = lp; }
            }
        }
    }
    detect_mc(d, st, e, w);
    qsort(w->con, w->ncon, sizeof(contact_t), contact_cmp);
    (void)cfg;
}

/* wall geometry helper: signed inward distance and inward normal of wall wl at point (x,y) */
static inline float wall_dist(const derived_t *d, int wl, float x, float y, float *nx, float *ny) {
    switch (wl) {
    case 0: *nx = 1.0f; *ny = 0.0f; return x - d->xmin;
    case 1: *nx = 0.0f; *ny = 1.0f; return y - d->ymin;
    case 2: *nx = -1.0f; *ny = 0.0f; return d->xmax - x;
    default: *nx = 0.0f; *ny = -1.0f; return d->ymax - y;
    }
}

/* b2TimeOfImpact for a circle (radius R) whose centre moves linearly from (x0,y0) to (x1,y1) against wall wl.
 * Returns 1 and *t in [0,1] when the state is e_touching, 0 otherwise (separated / overlapped / failed). */
static int toi_wall(const derived_t *d, int wl, float R, float x0, float y0, float x1, float y1, float *tout) {
    const float total = R + B2_POLYGON_RADIUS;
    const float target = fmaxf(B2_LINEAR_SLOP, total - 3.0f * B2_LINEAR_SLOP);
    const float tol = 0.25f * B2_LINEAR_SLOP;
    float nx, ny;
#define DIST_AT(t) wall_dist(d, wl, (1.0f - (t)) * x0 + (t) * x1, (1.0f - (t)) * y0 + (t) * y1, &nx, &ny)
    float t1 = 0.0f;
    for (int iter = 0; iter < 20; ++iter) {
        const float dist = DIST_AT(t1);
        if (fabsf(dist) <= 0.0f) return 0;                       /* overlapped */
        if (fabsf(dist) < target + tol) { *tout = t1; return 1; }  /* touching */
        int done = 0;
        float t2 = 1.0f;
        for (int push = 0; push < 8; ++push) {
            float s2 = DIST_AT(t2);
            if (s2 > target + tol) return 0;                     /* separated at the end of the step */
            if (s2 > target - tol) { t1 = t2; break; }
            float s1 = DIST_AT(t1);
            if (s1 < target - tol) return 0;                     /* failed */
            if (s1 <= target + tol) { *tout = t1; return 1; }
            float a1 = t1, a2 = t2;
            for (int root = 0; root < 50; ++root) {              /* bisection / secant alternating, b2_toiRootIters */
                float t = (root & 1) ? a1 + (target - s1) * (a2 - a1) / (s2 - s1) : 0.5f * (a1 + a2);
                float sv = DIST_AT(t);
                if (fabsf(sv - target) < tol) { t2 = t; break; }
                if (sv > target) { a1 = t; s1 = sv; } else { a2 = t; s2 = sv; }
            }
            (void)done;
        }
    }
#undef DIST_AT
    return 0;                                                    /* failed (iteration cap) */
}

/* b2World::SolveTOI + b2Island::SolveTOI for ONE circular body against the arena walls (the only TOI events Box2D
 * computes for non-bullet bodies).  (x0,y0,a0): pose at the start of the step; (x,y,a): pose after b2Island::Solve;
 * (vx,vy,w): velocity after the solve.  Pose and velocity are updated in place. */
static void toi_walls_body(const kbo_config *cfg, const derived_t *d, float R, float im,
                           float x0, float y0, float a0, float *x, float *y, float *a, float *vx, float *vy, float *w) {
    const float h = d->h;
    const float total = R + B2_POLYGON_RADIUS;
    float alpha0 = 0.0f;
    float c0x = x0, c0y = y0, ca0 = a0, cx = *x, cy = *y, ca = *a;
    for (int ev = 0; ev < B2_MAX_SUBSTEPS; ++ev) {
        float minAlpha = 1.0f;
        for (int wl = 0; wl < 4; ++wl) {
            float t;
            float alpha = 1.0f;
            if (toi_wall(d, wl, R, c0x, c0y, cx, cy, &t)) alpha = fminf(alpha0 + (1.0f - alpha0) * t, 1.0f);
            if (alpha < minAlpha) minAlpha = alpha;
        }
        if (1.0f - 10.0f * B2_EPSILON < minAlpha) break;
        /* b2Body::Advance */
        const float beta = (minAlpha - alpha0) / (1.0f - alpha0);
        c0x += beta * (cx - c0x); c0y += beta * (cy - c0y); ca0 += beta * (ca - ca0);
        alpha0 = minAlpha;
        cx = c0x; cy = c0y; ca = ca0;
        /* manifolds of the static contacts at the TOI pose */
        int touch[4]; float wnx[4], wny[4];
        for (int wl = 0; wl < 4; ++wl) {
            float dist = wall_dist(d, wl, cx, cy, &wnx[wl], &wny[wl]);
            touch[wl] = !(dist * dist > total * total);
            if (dist < 0.0f) { wnx[wl] = -wnx[wl]; wny[wl] = -wny[wl]; }
        }
        /* b2ContactSolver::SolveTOIPositionConstraints, 20 iterations */
        for (int it = 0; it < 20; ++it) {
            float minSep = 0.0f;
            for (int wl = 0; wl < 4; ++wl) {
                if (!touch[wl]) continue;
                float bx, by; float dist = wall_dist(d, wl, cx, cy, &bx, &by);
                float along = (wnx[wl] == bx && wny[wl] == by) ? dist : -dist;
                float sep = along - B2_POLYGON_RADIUS - R;
                minSep = fminf(minSep, sep);
                float C = clampf(B2_TOI_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                float K = 0.0f + im;
                float imp = K > 0.0f ? -C / K : 0.0f;
                cx += im * (imp * wnx[wl]); cy += im * (imp * wny[wl]);
            }
            if (minSep >= -1.5f * B2_LINEAR_SLOP) break;
        }
        c0x = cx; c0y = cy; ca0 = ca;                              /* leap of faith to the new safe state */
        /* velocity constraints without warm starting */
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < cfg->vel_iters; ++it) {
            for (int wl = 0; wl < 4; ++wl) {
                if (!touch[wl]) continue;
                float vn = *vx * wnx[wl] + *vy * wny[wl];
                float k = 0.0f + im;
                float nm = k > 0.0f ? 1.0f / k : 0.0f;
                float lambda = -(nm * vn);
                float newimp = fmaxf(acc[wl] + lambda, 0.0f);
                lambda = newimp - acc[wl];
                acc[wl] = newimp;
                *vx += im * (lambda * wnx[wl]); *vy += im * (lambda * wny[wl]);
            }
        }
        /* integrate the rest of the step */
        const float hh = (1.0f - minAlpha) * h;
        float tx = hh * *vx, ty = hh * *vy;
        if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
            float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
            *vx *= ratio; *vy *= ratio;
        }
        float rot = hh * *w;
        if (rot * rot > B2_MAX_ROTATION_SQ) *w *= B2_MAX_ROTATION / fabsf(rot);
        cx += hh * *vx; cy += hh * *vy; ca += hh * *w;
    }
    *x = cx; *y = cy; *a = ca;
}

/* ---- b2ContactSolver for the manifold constraints (object-object, object-wall) ------------------------------- */
typedef struct { v2 c; float a; v2 v; float w; float m, i; } bstate_t;
static inline bstate_t body_get(const derived_t *d, const work_t *w, int b) {
    bstate_t s; memset(&s, 0, sizeof(s));
    if (b < 0) return s;                                   /* static: zero velocity, zero inverse mass */
    s.c = V2(w->px[b], w->py[b]); s.a = w->ang[b]; s.v = V2(w->vx[b], w->vy[b]); s.w = w->bw[b];
    s.m = d->im_obj[b - w->N]; s.i = d->ii_obj[b - w->N];
    return s;
}
static inline void body_put_vel(work_t *w, int b, const bstate_t *s) {
    if (b < 0) return;
    w->vx[b] = s->v.x; w->vy[b] = s->v.y; w->bw[b] = s->w;
}
static inline void body_put_pos(work_t *w, int b, const bstate_t *s) {
    if (b < 0) return;
    w->px[b] = s->c.x; w->py[b] = s->c.y; w->ang[b] = s->a;
}

/* b2ContactSolver::InitializeVelocityConstraints (b2WorldManifold::Initialize inside) */
static void mc_init_velocity(const derived_t *d, const work_t *w, mc_t *c) {
    const bstate_t A = body_get(d, w, c->a), B = body_get(d, w, c->b);
    const xf_t xfA = body_xf(w, c->a), xfB = body_xf(w, c->b);
    v2 pts[2];
    if (c->m.type == 0) {
        v2 normal = V2(1.0f, 0.0f);
        v2 pointA = xf_mul(&xfA, c->m.localPoint), pointB = xf_mul(&xfB, c->m.lp[0]);
        v2 dd = v_sub(pointB, pointA);
        if (v_dot(dd, dd) > B2_EPSILON * B2_EPSILON) normal = v_normalize(dd);
        v2 cA = v_add(pointA, v_scale(c->radA, normal)), cB = v_sub(pointB, v_scale(c->radB, normal));
        pts[0] = v_scale(0.5f, v_add(cA, cB));
        c->normal = normal;
    } else if (c->m.type == 1) {
        v2 normal = rot_mul(&xfA, c->m.localNormal);
        v2 planePoint = xf_mul(&xfA, c->m.localPoint);
        for (int j = 0; j < c->m.count; ++j) {
            v2 clipPoint = xf_mul(&xfB, c->m.lp[j]);
            v2 cA = v_add(clipPoint, v_scale(c->radA - v_dot(v_sub(clipPoint, planePoint), normal), normal));
            v2 cB = v_sub(clipPoint, v_scale(c->radB, normal));
            pts[j] = v_scale(0.5f, v_add(cA, cB));
        }
        c->normal = normal;
    } else {
        v2 normal = rot_mul(&xfB, c->m.localNormal);
        v2 planePoint = xf_mul(&xfB, c->m.localPoint);
        for (int j = 0; j < c->m.count; ++j) {
            v2 clipPoint = xf_mul(&xfA, c->m.lp[j]);
            v2 cB = v_add(clipPoint, v_scale(c->radB - v_dot(v_sub(clipPoint, planePoint), normal), normal));
            v2 cA = v_sub(clipPoint, v_scale(c->radA, normal));
            pts[j] = v_scale(0.5f, v_add(cA, cB));
        }
        c->normal = v_neg(normal);
    }
    c->vcount = c->m.count;
    const v2 tangent = v_cross_vs(c->normal, 1.0f);
    for (int j = 0; j < c->m.count; ++j) {
        c->rA[j] = v_sub(pts[j], A.c); c->rB[j] = v_sub(pts[j], B.c);
        float rnA = v_cross(c->rA[j], c->normal), rnB = v_cross(c->rB[j], c->normal);
        float kNormal = A.m + B.m + A.i * rnA * rnA + B.i * rnB * rnB;
        c->nmass[j] = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        float rtA = v_cross(c->rA[j], tangent), rtB = v_cross(c->rB[j], tangent);
        float kTangent = A.m + B.m + A.i * rtA * rtA + B.i * rtB * rtB;
        c->tmass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
        /* restitution 0: velocityBias = 0 */
    }
    if (c->m.count == 2) {                                 /* block solver set-up */
        float rn1A = v_cross(c->rA[0], c->normal), rn1B = v_cross(c->rB[0], c->normal);
        float rn2A = v_cross(c->rA[1], c->normal), rn2B = v_cross(c->rB[1], c->normal);
        float k11 = A.m + B.m + A.i * rn1A * rn1A + B.i * rn1B * rn1B;
        float k22 = A.m + B.m + A.i * rn2A * rn2A + B.i * rn2B * rn2B;
        float k12 = A.m + B.m + A.i * rn1A * rn2A + B.i * rn1B * rn2B;
        const float k_maxConditionNumber = 1000.0f;
        if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
            c->k11 = k11; c->k12 = k12; c->k22 = k22;
            float det = k11 * k22 - k12 * k12;              /* b2Mat22::GetInverse */
            if (det != 0.0f) det = 1.0f / det;
            c->n11 = det * k22; c->n12 = -det * k12; c->n21 = -det * k12; c->n22 = det * k11;
        } else {
            c->vcount = 1;                                  /* the constraints are redundant: use one */
        }
    }
}

static inline void mc_apply(bstate_t *A, bstate_t *B, v2 rA, v2 rB, v2 P) {
    A->v = v_sub(A->v, v_scale(A->m, P)); A->w -= A->i * v_cross(rA, P);
    B->v = v_add(B->v, v_scale(B->m, P)); B->w += B->i * v_cross(rB, P);
}

/* b2ContactSolver::WarmStart */
static void mc_warm_start(const derived_t *d, work_t *w, mc_t *c) {
    bstate_t A = body_get(d, w, c->a), B = body_get(d, w, c->b);
    const v2 tangent = v_cross_vs(c->normal, 1.0f);
    for (int j = 0; j < c->vcount; ++j) {
        v2 P = v_add(v_scale(c->nimp[j], c->normal), v_scale(c->timp[j], tangent));
        A.w -= A.i * v_cross(c->rA[j], P); A.v = v_sub(A.v, v_scale(A.m, P));
        B.w += B.i * v_cross(c->rB[j], P); B.v = v_add(B.v, v_scale(B.m, P));
    }
    body_put_vel(w, c->a, &A); body_put_vel(w, c->b, &B);
}

static inline v2 mc_dv(const bstate_t *A, const bstate_t *B, v2 rA, v2 rB) {
    return v_sub(v_sub(v_add(B->v, v_cross_sv(B->w, rB)), A->v), v_cross_sv(A->w, rA));
}

/* b2ContactSolver::SolveVelocityConstraints for one contact */
static void mc_solve_velocity(const derived_t *d, work_t *w, mc_t *c) {
    bstate_t A = body_get(d, w, c->a), B = body_get(d, w, c->b);
    const v2 normal = c->normal, tangent = v_cross_vs(c->normal, 1.0f);
    for (int j = 0; j < c->vcount; ++j) {                  /* friction first */
        v2 dv = mc_dv(&A, &B, c->rA[j], c->rB[j]);
        float vt = v_dot(dv, tangent) - 0.0f;
        float lambda = c->tmass[j] * (-vt);
        float maxFriction = c->friction * c->nimp[j];
        float newImpulse = clampf(c->timp[j] + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c->timp[j];
        c->timp[j] = newImpulse;
        mc_apply(&A, &B, c->rA[j], c->rB[j], v_scale(lambda, tangent));
    }
    if (c->vcount == 1) {
        v2 dv = mc_dv(&A, &B, c->rA[0], c->rB[0]);
        float vn = v_dot(dv, normal);
        float lambda = -c->nmass[0] * (vn - 0.0f);
        float newImpulse = fmaxf(c->nimp[0] + lambda, 0.0f);
        lambda = newImpulse - c->nimp[0];
        c->nimp[0] = newImpulse;
        mc_apply(&A, &B, c->rA[0], c->rB[0], v_scale(lambda, normal));
    } else {                                               /* block solver (total enumeration of the 2x2 LCP) */
        const float ax = c->nimp[0], ay = c->nimp[1];
        v2 dv1 = mc_dv(&A, &B, c->rA[0], c->rB[0]), dv2 = mc_dv(&A, &B, c->rA[1], c->rB[1]);
        float vn1 = v_dot(dv1, normal), vn2 = v_dot(dv2, normal);
        float bx = vn1 - 0.0f, by = vn2 - 0.0f;
        bx -= c->k11 * ax + c->k12 * ay; by -= c->k12 * ax + c->k22 * ay;      /* b -= K a */
        float xx, xy; int solved = 0;
        for (;;) {
            xx = -(c->n11 * bx + c->n21 * by); xy = -(c->n12 * bx + c->n22 * by);   /* x = -normalMass b */
            if (xx >= 0.0f && xy >= 0.0f) { solved = 1; break; }
            xx = -c->nmass[0] * bx; xy = 0.0f; vn1 = 0.0f; vn2 = c->k12 * xx + by;
            if (xx >= 0.0f && vn2 >= 0.0f) { solved = 1; break; }
            xx = 0.0f; xy = -c->nmass[1] * by; vn1 = c->k12 * xy + bx; vn2 = 0.0f;
            if (xy >= 0.0f && vn1 >= 0.0f) { solved = 1; break; }
            xx = 0.0f; xy = 0.0f; vn1 = bx; vn2 = by;
            if (vn1 >= 0.0f && vn2 >= 0.0f) { solved = 1; break; }
            break;                                          /* no solution: give up */
        }
        if (solved) {
            float dx = xx - ax, dy = xy - ay;
            v2 P1 = v_scale(dx, normal), P2 = v_scale(dy, normal);
            A.v = v_sub(A.v, v_scale(A.m, v_add(P1, P2)));
            A.w -= A.i * (v_cross(c->rA[0], P1) + v_cross(c->rA[1], P2));
            B.v = v_add(B.v, v_scale(B.m, v_add(P1, P2)));
            B.w += B.i * (v_cross(c->rB[0], P1) + v_cross(c->rB[1], P2));
            c->nimp[0] = xx; c->nimp[1] = xy;
        }
    }
    body_put_vel(w, c->a, &A); body_put_vel(w, c->b, &B);
}

/* b2ContactSolver::SolvePositionConstraints for one contact; returns its minimum separation */
static float mc_solve_position(const derived_t *d, work_t *w, mc_t *c, float baumgarte) {
    bstate_t A = body_get(d, w, c->a), B = body_get(d, w, c->b);
    float minSeparation = 0.0f;
    for (int j = 0; j < c->m.count; ++j) {
        xf_t xfA, xfB;
        if (c->a < 0) { xfA.p = V2(0.0f, 0.0f); xfA.s = 0.0f; xfA.c = 1.0f; } else xfA = xf_of_body(d, c->a - w->N, A.c, A.a);
        xfB = xf_of_body(d, c->b - w->N, B.c, B.a);
        v2 normal, point; float separation;
        if (c->m.type == 0) {                               /* b2PositionSolverManifold */
            v2 pointA = xf_mul(&xfA, c->m.localPoint), pointB = xf_mul(&xfB, c->m.lp[0]);
            normal = v_normalize(v_sub(pointB, pointA));
            point = v_scale(0.5f, v_add(pointA, pointB));
            separation = v_dot(v_sub(pointB, pointA), normal) - c->radA - c->radB;
        } else if (c->m.type == 1) {
            normal = rot_mul(&xfA, c->m.localNormal);
            v2 planePoint = xf_mul(&xfA, c->m.localPoint);
            v2 clipPoint = xf_mul(&xfB, c->m.lp[j]);
            separation = v_dot(v_sub(clipPoint, planePoint), normal) - c->radA - c->radB;
            point = clipPoint;
        } else {
            normal = rot_mul(&xfB, c->m.localNormal);
            v2 planePoint = xf_mul(&xfB, c->m.localPoint);
            v2 clipPoint = xf_mul(&xfA, c->m.lp[j]);
            separation = v_dot(v_sub(clipPoint, planePoint), normal) - c->radA - c->radB;
            point = clipPoint;
            normal = v_neg(normal);
        }
        v2 rA = v_sub(point, A.c), rB = v_sub(point, B.c);
        minSeparation = fminf(minSeparation, separation);
        float C = clampf(baumgarte * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
        float rnA = v_cross(rA, normal), rnB = v_cross(rB, normal);
        float K = A.m + B.m + A.i * rnA * rnA + B.i * rnB * rnB;
        float impulse = K > 0.0f ? -C / K : 0.0f;
        v2 P = v_scale(impulse, normal);
        A.c = v_sub(A.c, v_scale(A.m, P)); A.a -= A.i * v_cross(rA, P);
        B.c = v_add(B.c, v_scale(B.m, P)); B.a += B.i * v_cross(rB, P);
    }
    body_put_pos(w, c->a, &A); body_put_pos(w, c->b, &B);
    return minSeparation;
}

/* b2TimeOfImpact for polygon fixture f of a body sweeping from (c0, a0) to (c1, a1) against wall wl: the separation
 * function is e_faceA with the wall as the face, i.e. the signed distance of the deepest vertex from the wall line;
 * Box2D's control flow (conservative advancement, push-back over the vertices, bisection / secant root finder).
 * Returns 1 and *t when the state is e_touching. */
static int toi_wall_poly(const derived_t *d, int wl, int f, int m, v2 c0, float a0, v2 c1, float a1, float *tout) {
    const shape_t *sh = &d->shape[f];
    const float total = B2_POLYGON_RADIUS + B2_POLYGON_RADIUS;
    const float target = fmaxf(B2_LINEAR_SLOP, total - 3.0f * B2_LINEAR_SLOP);
    const float tol = 0.25f * B2_LINEAR_SLOP;
    float nx, ny;
    {                                                                 /* b2Sweep::Normalize */
        const float twoPi = 2.0f * B2_PI;
        const float dd = twoPi * floorf(a0 / twoPi);
        a0 -= dd; a1 -= dd;
    }
    wall_dist(d, wl, 0.0f, 0.0f, &nx, &ny);
    const v2 axis = V2(-nx, -ny);                                     /* -normal of the face (wall = proxy A) */
#define XF_AT(t) xf_of_body(d, m, V2((1.0f - (t)) * c0.x + (t) * c1.x, (1.0f - (t)) * c0.y + (t) * c1.y), (1.0f - (t)) * a0 + (t) * a1)
#define VERT_AT(xf, i) wall_dist(d, wl, xf_mul(&(xf), sh->v[i]).x, xf_mul(&(xf), sh->v[i]).y, &nx, &ny)
    float t1 = 0.0f;
    for (int iter = 0; iter < 20; ++iter) {
        xf_t x1 = XF_AT(t1);
        float dist = 3.402823466e+38f;
        for (int i = 0; i < sh->n; ++i) dist = fminf(dist, VERT_AT(x1, i));
        if (dist <= 0.0f) return 0;                                   /* overlapped */
        if (dist < target + tol) { *tout = t1; return 1; }            /* touching */
        float t2 = 1.0f;
        int done = 0;
        for (int push = 0; push < 8; ++push) {
            xf_t x2 = XF_AT(t2);
            /* FindMinSeparation, e_faceA: support vertex of the polygon along -normal in its own frame */
            const v2 axisB = V2(x2.c * axis.x + x2.s * axis.y, -x2.s * axis.x + x2.c * axis.y);
            int idx = 0; float best = v_dot(sh->v[0], axisB);
            for (int i = 1; i < sh->n; ++i) { float val = v_dot(sh->v[i], axisB); if (val > best) { best = val; idx = i; } }
            float s2 = VERT_AT(x2, idx);
            if (s2 > target + tol) return 0;                          /* separated at the end of the step */
            if (s2 > target - tol) { t1 = t2; break; }                /* advance the sweeps */
            float s1 = VERT_AT(x1, idx);                               /* Evaluate(idx, t1) */
            if (s1 < target - tol) return 0;                          /* failed */
            if (s1 <= target + tol) { *tout = t1; return 1; }         /* touching */
            float r1 = t1, r2 = t2;
            for (int root = 0; root < 50; ++root) {                   /* 1D root finder on this vertex */
                float t = (root & 1) ? r1 + (target - s1) * (r2 - r1) / (s2 - s1) : 0.5f * (r1 + r2);
                xf_t xt = XF_AT(t);
                float sv = VERT_AT(xt, idx);
                if (fabsf(sv - target) < tol) { t2 = t; break; }
                if (sv > target) { r1 = t; s1 = sv; } else { r2 = t; s2 = sv; }
            }
            (void)done;
        }
    }
#undef XF_AT
#undef VERT_AT
    return 0;                                                         /* failed (iteration cap) */
}

/* b2World::SolveTOI + b2Island::SolveTOI for ONE pushable object (body N + m) against the arena walls, with the
 * contact model of the manifold constraints (friction, block solver, rotation); impulses start at zero (no warm
 * starting in the TOI sub-solve).  The body's state in the work arrays is (pose after b2Island::Solve, velocity). */
static void toi_walls_object(const kbo_config *cfg, const derived_t *d, kbo_state *st, int e, work_t *w, int m) {
    const int N = w->N, b = N + m;
    const float h = d->h;
    float alpha0 = 0.0f;
    v2 c0 = V2(w->x0[b], w->y0[b]);
    float a0 = w->a0[b];
    for (int ev = 0; ev < B2_MAX_SUBSTEPS; ++ev) {
        const v2 c1 = V2(w->px[b], w->py[b]);
        const float a1 = w->ang[b];
        float minAlpha = 1.0f;
        for (int f = 0; f < d->nfix; ++f) {
            if (d->fix_body[f] != m) continue;
            for (int wl = 0; wl < 4; ++wl) {
                float t, alpha = 1.0f;
                int hit = d->shape[f].kind == KBO_SHAPE_CIRCLE
                        ? toi_wall(d, wl, d->shape[f].radius, c0.x, c0.y, c1.x, c1.y, &t)
                        : toi_wall_poly(d, wl, f, m, c0, a0, c1, a1, &t);
                if (hit) alpha = fminf(alpha0 + (1.0f - alpha0) * t, 1.0f);
                if (alpha < minAlpha) minAlpha = alpha;
            }
        }
        if (1.0f - 10.0f * B2_EPSILON < minAlpha) break;
        /* b2Body::Advance */
        const float beta = (minAlpha - alpha0) / (1.0f - alpha0);
        c0.x += beta * (c1.x - c0.x); c0.y += beta * (c1.y - c0.y); a0 += beta * (a1 - a0);
        alpha0 = minAlpha;
        w->px[b] = c0.x; w->py[b] = c0.y; w->ang[b] = a0;
        /* the contacts of the body with the static walls at the TOI pose (b2Contact::Update) */
        /* (the TOI contact itself always touches: target separation 0.005 < the 0.02 skin of the manifold test.)
         * b2Contact::Update also re-matches the stored impulses to the new manifold's ids; the TOI solve itself starts
         * from zero impulses and stores nothing. */
        mc_t tc[4 * KBO_MAX_OBJECTS];
        int ntc = 0;
        for (int f = 0; f < d->nfix; ++f) {
            if (d->fix_body[f] != m) continue;
            for (int wl = 0; wl < 4; ++wl) {
                float *row = st->ows_acc + (((size_t)e * KBO_MAX_OBJECTS + f) * OWS + 8 + wl) * OWW;
                if (wall_manifold(d, w, f, wl, &tc[ntc])) {
                    mc_t *c = &tc[ntc++];
                    mc_warm(st, e, c);
                    for (int k = 0; k < OWW; ++k) row[k] = -1.0f;
                    for (int j = 0; j < c->m.count; ++j) { row[3 * j] = (float)c->m.id[j]; row[3 * j + 1] = c->nimp[j]; row[3 * j + 2] = c->timp[j]; }
                    c->nimp[0] = c->nimp[1] = c->timp[0] = c->timp[1] = 0.0f;
                } else {
                    for (int k = 0; k < OWW; ++k) row[k] = -1.0f;
                }
            }
        }
        /* b2ContactSolver::SolveTOIPositionConstraints, 20 iterations */
        for (int it = 0; it < 20; ++it) {
            float minSep = 0.0f;
            for (int i = 0; i < ntc; ++i) minSep = fminf(minSep, mc_solve_position(d, w, &tc[i], B2_TOI_BAUMGARTE));
            if (minSep >= -1.5f * B2_LINEAR_SLOP) break;
        }
        c0 = V2(w->px[b], w->py[b]); a0 = w->ang[b];                /* leap of faith to the new safe state */
        /* velocity constraints without warm starting */
        for (int i = 0; i < ntc; ++i) mc_init_velocity(d, w, &tc[i]);
        for (int it = 0; it < cfg->vel_iters; ++it)
            for (int i = 0; i < ntc; ++i) mc_solve_velocity(d, w, &tc[i]);
        /* integrate the rest of the step */
        const float hh = (1.0f - minAlpha) * h;
        float tx = hh * w->vx[b], ty = hh * w->vy[b];
        if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
            float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
            w->vx[b] *= ratio; w->vy[b] *= ratio;
        }
        float rot = hh * w->bw[b];
        if (rot * rot > B2_MAX_ROTATION_SQ) w->bw[b] *= B2_MAX_ROTATION / fabsf(rot);
        w->px[b] += hh * w->vx[b]; w->py[b] += hh * w->vy[b]; w->ang[b] += hh * w->bw[b];
    }
}

/* b2Island::Solve for one env (all islands; islands only matter for the position-iteration early-out) */
static void world_step_env(const kbo_config *cfg, const derived_t *d, kbo_state *st, int e, work_t *w) {
    const int N = w->N;
    const float h = d->h;
    const int sleeping = cfg->allow_sleep != 0;
    detect_env(cfg, d, st, e, w);
    /* islands (connected components over the touching dynamic-dynamic contacts; b2World::Solve does not propagate an
     * island across static bodies): for the per-island early-out of the position solver, and for sleeping */
    const int T = N + w->M;
    for (int b = 0; b < T; ++b) { w->parent[b] = b; }
    for (int i = 0; i < w->ncon; ++i) { w->con[i].skip = 0; if (w->con[i].a >= 0) uf_union(w->parent, w->con[i].a, w->con[i].b); }
    for (int i = 0; i < w->nmc; ++i) { w->mc[i].skip = 0; if (w->mc[i].a >= 0) uf_union(w->parent, w->mc[i].a, w->mc[i].b); }
    for (int b = 0; b < T; ++b) { w->parent[b] = uf_find(w->parent, b); w->active[b] = 1; }
    if (sleeping) {
        /* b2Contact::Update (b2ContactManager::Collide, ahead of b2World::Solve): a contact that was touching in the previous
         * step and is not any more wakes both bodies (SetAwake(true): awake flag, m_sleepTime = 0 if it slept) -- an awake
         * kilobot that leaves a sleeping neighbour wakes it, and with it the neighbour's resting island.  "Was touching" = an
         * entry of the previous substep's packed list that no contact of this substep matched; for object pairs an entry of
         * the manifold table without a manifold now.  (Contacts with the walls: the wall is static, and the kilobot that
         * leaves it moves, i.e. is awake.) */
        for (int b = 0; b < N; ++b) {
            const int cnt = st->ws_cnt[(size_t)e * N + b];
            for (int s_ = 0; s_ < cnt && w->woff[b] + s_ < w->cap; ++s_) {
                if (w->ws_hit[w->woff[b] + s_]) continue;
                const unsigned key = st->ws_key[(size_t)e * w->cap + (size_t)(w->woff[b] + s_)];
                int other = -1;
                if (key < KEY_WALL) other = (int)key;
                else if (key >= KEY_OBJ) other = N + d->fix_body[key - KEY_OBJ];
                if (other < 0 || other >= T) continue;
                if (w->slp[b] < 0.0f) w->slp[b] = 0.0f;
                if (w->slp[other] < 0.0f) w->slp[other] = 0.0f;
            }
        }
        for (int f1 = 0; f1 < d->nfix; ++f1)
            for (int f2 = f1 + 1; f2 < d->nfix; ++f2) {
                const float *old = st->ows_acc + (((size_t)e * KBO_MAX_OBJECTS + f1) * OWS + f2) * OWW;
                if (!(old[0] >= 0.0f) || d->fix_body[f1] == d->fix_body[f2]) continue;
                int now = 0;
                for (int i = 0; i < w->nmc; ++i) now |= (w->mc[i].owner == f1 && w->mc[i].col == f2);
                if (now) continue;
                const int b1 = N + d->fix_body[f1], b2 = N + d->fix_body[f2];
                if (w->slp[b1] < 0.0f) w->slp[b1] = 0.0f;
                if (w->slp[b2] < 0.0f) w->slp[b2] = 0.0f;
            }
        /* b2World::Solve: islands grow from AWAKE seeds and wake every body they reach (b2Body::SetAwake(true): flag set,
         * m_sleepTime = 0 if it was asleep); components without an awake body are not simulated at all: their bodies do
         * not move, their contacts keep their impulses */
        for (int b = 0; b < T; ++b) w->isl_awake[b] = 0;
        for (int b = 0; b < T; ++b) if (!(w->slp[b] < 0.0f)) w->isl_awake[w->parent[b]] = 1;
        for (int b = 0; b < T; ++b) {
            if (w->isl_awake[w->parent[b]]) { if (w->slp[b] < 0.0f) w->slp[b] = 0.0f; }
            else { w->vx[b] = 0.0f; w->vy[b] = 0.0f; w->bw[b] = 0.0f; }
            w->active[b] = w->isl_awake[b];        /* (only read at island roots) */
        }
        for (int i = 0; i < w->ncon; ++i) w->con[i].skip = !w->isl_awake[w->parent[w->con[i].b]];
        for (int i = 0; i < w->nmc; ++i) w->mc[i].skip = !w->isl_awake[w->parent[w->mc[i].b]];
    }

    /* integrate velocities: no forces; Pade damping (b2Island.cpp: v *= 1/(1 + h*c)) */
    for (int b = 0; b < N; ++b) {
        const int law = cfg->drive_mode == KBO_DRIVE_MIXED ? st->bot_mode[(size_t)e * N + b] : cfg->drive_mode;
        float kl = (law == KBO_DRIVE_SIMPLE_PHOTOTAXIS) ? 1.0f / (1.0f + h * 0.0f) : d->kl_bot;
        w->vx[b] *= kl; w->vy[b] *= kl; w->bw[b] *= d->ka_bot;
    }
    for (int m = 0; m < w->M; ++m) { w->vx[N + m] *= d->kl_obj; w->vy[N + m] *= d->kl_obj; w->bw[N + m] *= d->ka_obj; }

    /* b2ContactSolver::InitializeVelocityConstraints: normals from current poses */
    for (int i = 0; i < w->ncon; ++i) {
        contact_t *c = &w->con[i];
        if (c->a < 0) {
            float nx, ny; float dist = wall_dist(d, -1 - c->a, w->px[c->b], w->py[c->b], &nx, &ny);
            if (dist < 0.0f) { nx = -nx; ny = -ny; }      /* b2CollideEdgeAndCircle: flip towards the centre */
            c->nx = nx; c->ny = ny;
        } else if (c->poly) {
            /* b2WorldManifold::Initialize, e_faceA with A = the polygon (body b), B = the kilobot (body a) */
            const xf_t xo = body_xf(w, c->b);
            v2 normal = rot_mul(&xo, c->ln);
            v2 planePoint = xf_mul(&xo, c->lp);
            v2 clipPoint = V2(w->px[c->a], w->py[c->a]);
            v2 cA = v_add(clipPoint, v_scale(c->rb - v_dot(v_sub(clipPoint, planePoint), normal), normal));
            v2 cB = v_sub(clipPoint, v_scale(c->ra, normal));
            v2 point = v_scale(0.5f, v_add(cA, cB));
            c->rA = v_sub(point, V2(w->px[c->b], w->py[c->b]));
            c->nx = normal.x; c->ny = normal.y;            /* from the polygon to the kilobot */
            float rnA = v_cross(c->rA, normal);
            float kNormal = c->imb + c->ima + d->ii_obj[c->b - N] * rnA * rnA;      /* mA + mB + iA rnA^2 (kilobot side central) */
            c->nmass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        } else {
            float dx = w->px[c->b] - w->px[c->a], dy = w->py[c->b] - w->py[c->a];
            if (dx * dx + dy * dy > B2_EPSILON * B2_EPSILON) {   /* b2WorldManifold::Initialize */
                float len = sqrtf(dx * dx + dy * dy);
                float inv = 1.0f / len;
                c->nx = dx * inv; c->ny = dy * inv;
            } else { c->nx = 1.0f; c->ny = 0.0f; }
        }
    }
    for (int i = 0; i < w->nmc; ++i) mc_init_velocity(d, w, &w->mc[i]);
    /* WarmStart */
    for (int i = 0; i < w->ncon; ++i) {
        contact_t *c = &w->con[i];
        if (c->skip) continue;
        float Px = c->acc * c->nx, Py = c->acc * c->ny;
        if (c->poly) {                                      /* A = polygon b, B = kilobot a */
            w->bw[c->b] -= d->ii_obj[c->b - N] * (c->rA.x * Py - c->rA.y * Px);
            w->vx[c->b] -= c->imb * Px; w->vy[c->b] -= c->imb * Py;
            w->vx[c->a] += c->ima * Px; w->vy[c->a] += c->ima * Py;
            continue;
        }
        if (c->a >= 0) { w->vx[c->a] -= c->ima * Px; w->vy[c->a] -= c->ima * Py; }
        w->vx[c->b] += c->imb * Px; w->vy[c->b] += c->imb * Py;
    }
    for (int i = 0; i < w->nmc; ++i) if (!w->mc[i].skip) mc_warm_start(d, w, &w->mc[i]);
    /* SolveVelocityConstraints */
    for (int it = 0; it < cfg->vel_iters; ++it) {
        for (int i = 0; i < w->ncon; ++i) {
            contact_t *c = &w->con[i];
            if (c->skip) continue;
            if (c->poly) {                                  /* one point, friction sqrt(0 * f) = 0; A = polygon b, B = kilobot a */
                const float wA = w->bw[c->b];
                float dvx = (w->vx[c->a] - w->vx[c->b]) - (-wA * c->rA.y), dvy = (w->vy[c->a] - w->vy[c->b]) - (wA * c->rA.x);
                float vn = dvx * c->nx + dvy * c->ny;
                float lambda = -(c->nmass * vn);
                float newimp = fmaxf(c->acc + lambda, 0.0f);
                lambda = newimp - c->acc;
                c->acc = newimp;
                float Px = lambda * c->nx, Py = lambda * c->ny;
                w->vx[c->b] -= c->imb * Px; w->vy[c->b] -= c->imb * Py;
                w->bw[c->b] = wA - d->ii_obj[c->b - N] * (c->rA.x * Py - c->rA.y * Px);
                w->vx[c->a] += c->ima * Px; w->vy[c->a] += c->ima * Py;
                continue;
            }
            float vax = 0.0f, vay = 0.0f;
            if (c->a >= 0) { vax = w->vx[c->a]; vay = w->vy[c->a]; }
            float dvx = w->vx[c->b] - vax, dvy = w->vy[c->b] - vay;
            float vn = dvx * c->nx + dvy * c->ny;
            float k = c->ima + c->imb;
            float nm = k > 0.0f ? 1.0f / k : 0.0f;
            float lambda = -(nm * vn);
            float newimp = fmaxf(c->acc + lambda, 0.0f);
            lambda = newimp - c->acc;
            c->acc = newimp;
            float Px = lambda * c->nx, Py = lambda * c->ny;
            if (c->a >= 0) { w->vx[c->a] = vax - c->ima * Px; w->vy[c->a] = vay - c->ima * Py; }
            w->vx[c->b] += c->imb * Px; w->vy[c->b] += c->imb * Py;
        }
        for (int i = 0; i < w->nmc; ++i) if (!w->mc[i].skip) mc_solve_velocity(d, w, &w->mc[i]);
    }
    /* StoreImpulses -> warm-start cache of the next substep */
    memset(st->ws_cnt + (size_t)e * N, 0, (size_t)N);
    for (int m = 0; m < d->nfix; ++m)
        for (int k = 0; k < OWS * OWW; ++k) st->ows_acc[((size_t)e * KBO_MAX_OBJECTS + m) * OWS * OWW + k] = -1.0f;
    for (int i = 0; i < w->nmc; ++i) {
        const mc_t *c = &w->mc[i];
        float *dst = st->ows_acc + (((size_t)e * KBO_MAX_OBJECTS + c->owner) * OWS + c->col) * OWW;
        for (int j = 0; j < c->m.count; ++j) { dst[3 * j] = (float)c->m.id[j]; dst[3 * j + 1] = c->nimp[j]; dst[3 * j + 2] = c->timp[j]; }
    }
    for (int i = 0; i < w->ncon; ++i) {
        contact_t *c = &w->con[i];
        if (c->slot < 0) continue;
        size_t ci = (size_t)e * N + c->owner;
        if (st->ws_cnt[ci] < c->slot + 1) st->ws_cnt[ci] = (uint8_t)(c->slot + 1);
    }
    {
        int run = 0;
        for (int b = 0; b < N; ++b) { w->wnoff[b] = run; run += st->ws_cnt[(size_t)e * N + b]; }
    }
    for (int i = 0; i < w->ncon; ++i) {
        contact_t *c = &w->con[i];
        if (c->slot < 0) continue;
        int pos = w->wnoff[c->owner] + c->slot;
        if (pos >= w->cap) continue;
        size_t idx = (size_t)e * w->cap + (size_t)pos;
        unsigned key = c->a < 0 ? KEY_WALL + (unsigned)(-1 - c->a) : (c->b >= N ? KEY_OBJ + (unsigned)c->fix : (unsigned)c->b);
        st->ws_key[idx] = key; st->ws_acc[idx] = c->acc;
    }
    /* integrate positions */
    for (int b = 0; b < N + w->M; ++b) { w->x0[b] = w->px[b]; w->y0[b] = w->py[b]; }
    for (int b = 0; b < N; ++b) w->a0[b] = st->theta[(size_t)e * N + b];
    for (int m = 0; m < w->M; ++m) w->a0[N + m] = w->ang[N + m];
    for (int b = 0; b < N; ++b) {
        float tx = h * w->vx[b], ty = h * w->vy[b];
        if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
            float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
            w->vx[b] *= ratio; w->vy[b] *= ratio;
        }
        float rot = h * w->bw[b];
        if (rot * rot > B2_MAX_ROTATION_SQ) {
            float ratio = B2_MAX_ROTATION / fabsf(rot);
            w->bw[b] *= ratio;
        }
        w->px[b] += h * w->vx[b]; w->py[b] += h * w->vy[b];
        st->theta[(size_t)e * N + b] += h * w->bw[b];
    }
    for (int m = 0; m < w->M; ++m) {
        const int b = N + m;
        float tx = h * w->vx[b], ty = h * w->vy[b];
        if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
            float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
            w->vx[b] *= ratio; w->vy[b] *= ratio;
        }
        float rot = h * w->bw[b];
        if (rot * rot > B2_MAX_ROTATION_SQ) {
            float ratio = B2_MAX_ROTATION / fabsf(rot);
            w->bw[b] *= ratio;
        }
        w->px[b] += h * w->vx[b]; w->py[b] += h * w->vy[b];
        w->ang[b] += h * w->bw[b];
    }
    /* SolvePositionConstraints, b2ContactSolver.cpp; per island: break when minSeparation >= -3 slop
     * (islands without an awake body start inactive: w->active was set with the islands above) */
    for (int b = 0; b < T; ++b) w->isl_unsolved[b] = (unsigned char)(cfg->pos_iters <= 0);
    for (int it = 0; it < cfg->pos_iters; ++it) {
        int any = 0;
        memset(w->next_active, 0, (size_t)T);
        for (int i = 0; i < w->ncon; ++i) {
            contact_t *c = &w->con[i];
            int isl = w->parent[c->b];
            if (!w->active[isl]) continue;
            float nx, ny, sep;
            if (c->poly) {
                /* b2PositionSolverManifold e_faceA, A = polygon (body b), B = kilobot (body a); point = clipPoint */
                const xf_t xo = body_xf(w, c->b);
                v2 normal = rot_mul(&xo, c->ln);
                v2 planePoint = xf_mul(&xo, c->lp);
                v2 clipPoint = V2(w->px[c->a], w->py[c->a]);
                sep = v_dot(v_sub(clipPoint, planePoint), normal) - c->rb - c->ra;
                v2 rA = v_sub(clipPoint, V2(w->px[c->b], w->py[c->b]));
                if (sep < -3.0f * B2_LINEAR_SLOP) { w->next_active[isl] = 1; any = 1; }
                float C = clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                float rnA = v_cross(rA, normal);
                float K = c->imb + c->ima + d->ii_obj[c->b - N] * rnA * rnA;
                float imp = K > 0.0f ? -C / K : 0.0f;
                v2 P = v_scale(imp, normal);
                w->px[c->b] -= c->imb * P.x; w->py[c->b] -= c->imb * P.y;
                w->ang[c->b] -= d->ii_obj[c->b - N] * v_cross(rA, P);
                w->px[c->a] += c->ima * P.x; w->py[c->a] += c->ima * P.y;
                continue;
            }
            if (c->a < 0) {
                float bx, by; float dist = wall_dist(d, -1 - c->a, w->px[c->b], w->py[c->b], &bx, &by);
                /* manifold normal is fixed at detection time (stored in c->nx, c->ny) */
                nx = c->nx; ny = c->ny;
                float along = (nx == bx && ny == by) ? dist : -dist;
                sep = along - c->ra - c->rb;
            } else {
                float dx = w->px[c->b] - w->px[c->a], dy = w->py[c->b] - w->py[c->a];
                float len = sqrtf(dx * dx + dy * dy);
                nx = dx; ny = dy;
                if (!(len < B2_EPSILON)) { float inv = 1.0f / len; nx = dx * inv; ny = dy * inv; }   /* b2Vec2::Normalize */
                sep = (dx * nx + dy * ny) - c->ra - c->rb;
            }
            if (sep < -3.0f * B2_LINEAR_SLOP) { w->next_active[isl] = 1; any = 1; }
            float C = clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
            float K = c->ima + c->imb;
            float imp = K > 0.0f ? -C / K : 0.0f;
            float Px = imp * nx, Py = imp * ny;
            if (c->a >= 0) { w->px[c->a] -= c->ima * Px; w->py[c->a] -= c->ima * Py; }
            w->px[c->b] += c->imb * Px; w->py[c->b] += c->imb * Py;
        }
        for (int i = 0; i < w->nmc; ++i) {
            mc_t *c = &w->mc[i];
            int isl = w->parent[c->b];
            if (!w->active[isl]) continue;
            float minSep = mc_solve_position(d, w, c, B2_BAUMGARTE);
            if (minSep < -3.0f * B2_LINEAR_SLOP) { w->next_active[isl] = 1; any = 1; }
        }
        memcpy(w->active, w->next_active, (size_t)T);
        if (it == cfg->pos_iters - 1) memcpy(w->isl_unsolved, w->next_active, (size_t)T);   /* b2Island::Solve: positionSolved stays false */
        if (!any) break;
    }
    if (sleeping) {
        /* b2Island::Solve, the allowSleep block: a body slower than the tolerances accumulates sleep time; an island whose
         * bodies have all rested for b2_timeToSleep and whose position constraints converged is put to sleep
         * (b2Body::SetAwake(false): sleep time, velocities and forces zeroed) */
        const float linTolSqr = B2_LINEAR_SLEEP_TOL * B2_LINEAR_SLEEP_TOL, angTolSqr = B2_ANGULAR_SLEEP_TOL * B2_ANGULAR_SLEEP_TOL;
        for (int b = 0; b < T; ++b) w->isl_min[b] = 3.402823466e+38f;
        for (int b = 0; b < T; ++b) {
            const int r = w->parent[b];
            if (!w->isl_awake[r]) continue;
            if (w->bw[b] * w->bw[b] > angTolSqr || w->vx[b] * w->vx[b] + w->vy[b] * w->vy[b] > linTolSqr) {
                w->slp[b] = 0.0f; w->isl_min[r] = 0.0f;
            } else {
                w->slp[b] += h; w->isl_min[r] = fminf(w->isl_min[r], w->slp[b]);
            }
        }
        for (int b = 0; b < T; ++b) {
            const int r = w->parent[b];
            if (w->isl_awake[r] && w->isl_min[r] >= B2_TIME_TO_SLEEP && !w->isl_unsolved[r]) {
                w->slp[b] = -1.0f; w->vx[b] = 0.0f; w->vy[b] = 0.0f; w->bw[b] = 0.0f;
            }
        }
    }
    /* b2World::SolveTOI: continuous step of every dynamic body against the static walls */
    if (cfg->toi_walls) {
        /* (b2World::SolveTOI skips contacts without an awake dynamic body) */
        for (int b = 0; b < N; ++b)
            if (!(sleeping && w->slp[b] < 0.0f))
                toi_walls_body(cfg, d, d->r_bot, cfg->drive_mode == KBO_DRIVE_MIXED ? d->im_mode[st->bot_mode[(size_t)e * N + b] < 5 ? st->bot_mode[(size_t)e * N + b] : 0] : d->im_bot,
                               w->x0[b], w->y0[b], w->a0[b], &w->px[b], &w->py[b],
                               &st->theta[(size_t)e * N + b], &w->vx[b], &w->vy[b], &w->bw[b]);
        for (int m = 0; m < w->M; ++m) if (!(sleeping && w->slp[N + m] < 0.0f)) toi_walls_object(cfg, d, st, e, w, m);
    }
}

/* one substep of the kilobots_env.py:168-190 loop for env e */
/* neighbour counts of one env: all pairs, fp32 world units */
static void sense_env(const float *x, const float *y, int N, float radius_m, uint32_t *out) {
    const float Rw = radius_m * WORLD_SCALE, R2 = Rw * Rw;
    for (int a = 0; a < N; ++a) {
        uint32_t cnt = 0;
        for (int b = 0; b < N; ++b) {
            if (b == a) continue;
            const float dx = x[b] - x[a], dy = y[b] - y[a];
            const float dd = dx * dx + dy * dy;
            if (!(dd > R2)) cnt++;
        }
        out[a] = cnt;
    }
}

int kbo_sense(const kbo_config *cfg, const kbo_state *st, float radius_m, uint32_t *out) {
    if (!cfg || !st || !out || !(radius_m > 0.0f)) return -1;
    for (int e = 0; e < cfg->num_envs; ++e) {
        const size_t o = (size_t)e * cfg->num_bots;
        sense_env(st->x + o, st->y + o, cfg->num_bots, radius_m, out + o);
    }
    return 0;
}

/* ---- kb_reset: Philox4x32-10, Cephes logf, Box-Muller ------------------------------------------------------------ */
void kbo_philox4x32_10(const uint32_t cin[4], const uint32_t kin[2], uint32_t out[4]) {
    uint32_t c0 = cin[0], c1 = cin[1], c2 = cin[2], c3 = cin[3], k0 = kin[0], k1 = kin[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* natural logarithm of a positive normal float, Cephes single-precision algorithm */
float kbo_logf(float xx) {
    uint32_t bits; memcpy(&bits, &xx, 4);
    int e = (int)((bits >> 23) & 255u) - 126;
    uint32_t mb = (bits & 0x807FFFFFu) | 0x3F000000u;
    float x; memcpy(&x, &mb, 4);                                        /* mantissa in [0.5, 1) */
    if (x < 0.707106781186547524f) { e -= 1; x = x + x - 1.0f; }
    else x = x - 1.0f;
    float z = x * x;
    float y = ((((((((7.0376836292E-2f * x - 1.1514610310E-1f) * x + 1.1676998740E-1f) * x - 1.2420140846E-1f) * x
                  + 1.4249322787E-1f) * x - 1.6668057665E-1f) * x + 2.0000714765E-1f) * x - 2.4999993993E-1f) * x
               + 3.3333331174E-1f) * x * z;
    const float fe = (float)e;
    if (e) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = x + y;
    if (e) z += 0.693359375f * fe;
    return z;
}

int kbo_reset(const kbo_config *cfg, kbo_state *st, const kbo_reset_params *rp) {
    if (!cfg || !st || !rp) return -1;
    const int E = cfg->num_envs, N = cfg->num_bots;
    const uint32_t key[2] = {(uint32_t)(rp->seed & 0xFFFFFFFFull), (uint32_t)(rp->seed >> 32)};
    /* world_bounds -/+ 0.02 (yaml_kilobots_env.py:350-351), metres */
    const float lo_x = -0.5f * cfg->world_width + 0.02f, hi_x = 0.5f * cfg->world_width - 0.02f;
    const float lo_y = -0.5f * cfg->world_height + 0.02f, hi_y = 0.5f * cfg->world_height - 0.02f;
    for (int e = 0; e < E; ++e) {
        for (int b = 0; b < N; ++b) {
            const size_t i = (size_t)e * N + b;
            const uint32_t ctr[4] = {(uint32_t)(rp->env_offset + e), (uint32_t)b, 0u, 0u};
            uint32_t r[4];
            kbo_philox4x32_10(ctr, key, r);
            const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f);      /* (0, 1] */
            const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);             /* [0, 1) */
            const float rad = sqrtf(-2.0f * kbo_logf(u1));
            float sn, cs;
            kbo_sincosf(6.28318530717958647692f * u2, &sn, &cs);
            float xm = rp->mean[0] + rp->std * (rad * cs), ym = rp->mean[1] + rp->std * (rad * sn);
            xm = fminf(fmaxf(xm, lo_x), hi_x); ym = fminf(fmaxf(ym, lo_y), hi_y);
            st->x[i] = xm * WORLD_SCALE; st->y[i] = ym * WORLD_SCALE;
            float th = 0.0f;                                                          /* body.py:28-29 */
            if (rp->random_theta) th = ((float)(r[2] >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * 3.14159265358979323846f;
            st->theta[i] = th;
            st->ws_cnt[i] = 0;
            if (cfg->allow_sleep && st->sleep_time) st->sleep_time[i] = 0.0f;          /* new bodies are awake (b2BodyDef::awake) */
            if (st->nbr_count) st->nbr_count[i] = 0u;                                  /* (the resolve step does not sense) */
            const int law = cfg->drive_mode == KBO_DRIVE_MIXED ? st->bot_mode[i] : cfg->drive_mode;
            if (law == KBO_DRIVE_VELOCITY || law == KBO_DRIVE_ACCEL) {
                float v = 0.0f, w = 0.0f;
                if (rp->random_velocity) {                                            /* kilobot.py:225-229 */
                    v = (float)(r[3] & 0xFFFFu) * (1.0f / 65536.0f) * 0.01f;
                    w = ((float)(r[3] >> 16) * (1.0f / 65536.0f) * 2.0f - 1.0f) * (0.5f * 3.14159265358979323846f);
                }
                st->v[i] = v; st->w[i] = w;
            }
            if (law == KBO_DRIVE_ACCEL) { st->acc_v[i] = 0.0f; st->acc_w[i] = 0.0f; }
            if (law == KBO_DRIVE_MOTORS || law == KBO_DRIVE_PHOTOTAXIS) { st->motor_l[i] = 255; st->motor_r[i] = 0; }
            if (law == KBO_DRIVE_PHOTOTAXIS) {
                st->pt_threshold[i] = -INFINITY; st->pt_update[i] = 0; st->pt_nochange[i] = 0; st->pt_dir[i] = 0;
            }
        }
        if (st->status) st->status[e] = 0;
        if (cfg->num_objects > 0 && st->ows_acc)
            for (int k = 0; k < KBO_MAX_OBJECTS * KBO_OWS_COLS * KBO_OWS_WORDS; ++k)
                st->ows_acc[(size_t)e * (KBO_MAX_OBJECTS * KBO_OWS_COLS * KBO_OWS_WORDS) + k] = -1.0f;
    }
    return 0;
}

static void substep_env(const kbo_config *cfg, const derived_t *d, kbo_state *st, const float *light_action,
                        int flags, int e, work_t *w) {
    const int N = w->N;
    const size_t o = (size_t)e * N;
    const float h = d->h;
    /* light.step, kilobots_env.py:171-172 */
    if (light_action && cfg->light_type != KBO_LIGHT_NONE && !(flags & KBO_STEP_NO_DRIVE))
        light_step_env(cfg, st, e, light_action, h);
    /* IR-range neighbour sensing at the sensing point of the substep (with the light: kilobots_env.py:174-180) */
    if (st->nbr_count && cfg->sense_radius > 0.0f && !(flags & KBO_STEP_NO_DRIVE))
        sense_env(st->x + o, st->y + o, N, cfg->sense_radius, st->nbr_count + o);
    for (int b = 0; b < N; ++b) {
        float th = st->theta[o + b];
        float bvx = 0.0f, bvy = 0.0f, bw = 0.0f;
        w->px[b] = st->x[o + b]; w->py[b] = st->y[o + b];
        const int law = cfg->drive_mode == KBO_DRIVE_MIXED ? st->bot_mode[o + b] : cfg->drive_mode;
        if (!(flags & KBO_STEP_NO_DRIVE)) {
            float lval = 0.0f, lgx = 0.0f, lgy = 0.0f;
            if (cfg->light_type != KBO_LIGHT_NONE) {
                /* kilobots_env.py:174-180; sensor position kilobot.py:54-55 / :188-189 */
                float sx = w->px[b], sy = w->py[b];
                if (law != KBO_DRIVE_SIMPLE_PHOTOTAXIS) {
                    float s, c; kbo_sincosf(th, &s, &c);
                    float lx0 = 0.0f, ly0 = -d->r_bot;
                    sx = (c * lx0 - s * ly0) + w->px[b];
                    sy = (s * lx0 + c * ly0) + w->py[b];
                }
                light_sense(cfg, st, e, sx / WORLD_SCALE, sy / WORLD_SCALE, &lval, &lgx, &lgy);
                if (st->light_value) { st->light_value[o + b] = lval; st->light_gx[o + b] = lgx; st->light_gy[o + b] = lgy; }
            }
            switch (law) {
            case KBO_DRIVE_ACCEL: {
                /* kilobot.py:294-300 */
                float v = st->v[o + b] + st->acc_v[o + b] * h, ww = st->w[o + b] + st->acc_w[o + b] * h;
                const float mw = 0.5f * 3.14159265358979323846f;
                v = fminf(fmaxf(v, 0.0f), 0.01f); ww = fminf(fmaxf(ww, -mw), mw);
                st->v[o + b] = v; st->w[o + b] = ww;
            } /* fallthrough */
            case KBO_DRIVE_VELOCITY: {
                /* kilobot.py:253-258 */
                float s, c; kbo_sincosf(th, &s, &c);
                float sp = st->v[o + b] * WORLD_SCALE;
                bvx = c * sp; bvy = s * sp; bw = st->w[o + b];
            } break;
            case KBO_DRIVE_PHOTOTAXIS: {
                /* kilobot.py:318-333 */
                int upd = st->pt_update[o + b];
                if (upd % 6 == 0) {
                    float meas = lval;
                    if (meas > st->pt_threshold[o + b] || st->pt_nochange[o + b] >= 15) {
                        st->pt_threshold[o + b] = meas + 0.01f;
                        if (st->pt_dir[o + b] == 0) { st->pt_dir[o + b] = 1; st->motor_l[o + b] = 0; st->motor_r[o + b] = 255; }
                        else { st->pt_dir[o + b] = 0; st->motor_l[o + b] = 255; st->motor_r[o + b] = 0; }
                        st->pt_nochange[o + b] = 0;
                    } else st->pt_nochange[o + b] += 1;
                }
                st->pt_update[o + b] = upd + 1;
            } /* fallthrough */
            case KBO_DRIVE_MOTORS:
                motor_law(st->motor_l[o + b], st->motor_r[o + b], th, h, &bvx, &bvy, &bw);
                break;
            case KBO_DRIVE_SIMPLE_PHOTOTAXIS: {
                /* kilobot.py:191-203 */
                float n = sqrtf(lgx * lgx + lgy * lgy);
                float mx = lgx, my = lgy;
                if (n > 0.01f) { mx = lgx / n * 0.01f; my = lgy / n * 0.01f; }
                bvx = mx * WORLD_SCALE; bvy = my * WORLD_SCALE; bw = 0.0f;
            } break;
            default: break;
            }
        }
        w->vx[b] = bvx; w->vy[b] = bvy; w->bw[b] = bw;
        if (st->cmd_vx) { st->cmd_vx[o + b] = bvx; st->cmd_vy[o + b] = bvy; st->cmd_w[o + b] = bw; }
        if (cfg->allow_sleep) {
            /* kilobot.py:123-127 assigns body.angularVelocity / body.linearVelocity: b2Body::SetAngularVelocity /
             * SetLinearVelocity wake a sleeping body iff the assigned value is non-zero (w * w > 0, b2Dot(v, v) > 0) */
            float sl = st->sleep_time[o + b];
            if (!(flags & KBO_STEP_NO_DRIVE) && sl < 0.0f && (bw * bw > 0.0f || bvx * bvx + bvy * bvy > 0.0f)) sl = 0.0f;
            w->slp[b] = sl;
            /* a body that stays asleep has zero velocity (SetAwake(false) zeroed it and only a zero value was assigned since;
             * a non-zero command whose square underflows counts as zero here) */
            if (sl < 0.0f) { w->vx[b] = 0.0f; w->vy[b] = 0.0f; w->bw[b] = 0.0f; }
        }
    }
    for (int m = 0; m < w->M; ++m) {   /* objects keep their velocities between substeps (plain b2Body) */
        const size_t oi = (size_t)e * w->M + m;
        w->px[N + m] = st->ox[oi]; w->py[N + m] = st->oy[oi];
        w->vx[N + m] = st->ovx[oi]; w->vy[N + m] = st->ovy[oi]; w->bw[N + m] = st->ow[oi];
        w->ang[N + m] = st->otheta[oi];
        if (cfg->allow_sleep) w->slp[N + m] = st->osleep[oi];
        if (d->lc_obj[m].x != 0.0f || d->lc_obj[m].y != 0.0f) {    /* state holds the body origin, the solver the centre of mass */
            xf_t t = xf_make(st->ox[oi], st->oy[oi], st->otheta[oi]);
            v2 cm = xf_mul(&t, d->lc_obj[m]);
            w->px[N + m] = cm.x; w->py[N + m] = cm.y;
        }
    }
    world_step_env(cfg, d, st, e, w);
    for (int b = 0; b < N; ++b) { st->x[o + b] = w->px[b]; st->y[o + b] = w->py[b]; }
    if (cfg->allow_sleep) for (int b = 0; b < N; ++b) st->sleep_time[o + b] = w->slp[b];
    for (int m = 0; m < w->M; ++m) {
        const size_t oi = (size_t)e * w->M + m;
        if (cfg->allow_sleep) st->osleep[oi] = w->slp[N + m];
        st->ox[oi] = w->px[N + m]; st->oy[oi] = w->py[N + m];
        st->ovx[oi] = w->vx[N + m]; st->ovy[oi] = w->vy[N + m]; st->ow[oi] = w->bw[N + m];
        st->otheta[oi] = w->ang[N + m];
        if (d->lc_obj[m].x != 0.0f || d->lc_obj[m].y != 0.0f) {    /* b2Body::SynchronizeTransform */
            xf_t t = xf_of_body(d, m, V2(w->px[N + m], w->py[N + m]), w->ang[N + m]);
            st->ox[oi] = t.p.x; st->oy[oi] = t.p.y;
        }
    }
    if (st->status) st->status[e] |= w->status;
}

/* The sensing point of one substep on its own (specification of kb_light_sense, include/kilobots_hip.h): light.step with
 * the action (NULL = None: the light stays) and value_and_gradients at every kilobot's sensor, kilobots_env.py:171-180,
 * exactly as substep_env does them -- for kilobots whose _loop runs on the host before the motor law. */
int kbo_light_sense(const kbo_config *cfg, kbo_state *st, const float *light_action) {
    if (!cfg || !st || cfg->light_type == KBO_LIGHT_NONE || !st->light_value || !st->light_gx || !st->light_gy) return -1;
    derived_t d;
    derive(cfg, &d);
    const int N = cfg->num_bots;
    for (int e = 0; e < cfg->num_envs; ++e) {
        const size_t o = (size_t)e * N;
        if (light_action) light_step_env(cfg, st, e, light_action, d.h);
        for (int b = 0; b < N; ++b) {
            const float th = st->theta[o + b], px = st->x[o + b], py = st->y[o + b];
            float sx = px, sy = py;
            if ((cfg->drive_mode == KBO_DRIVE_MIXED ? st->bot_mode[o + b] : cfg->drive_mode) != KBO_DRIVE_SIMPLE_PHOTOTAXIS) {
                float s, c; kbo_sincosf(th, &s, &c);
                float lx0 = 0.0f, ly0 = -d.r_bot;
                sx = (c * lx0 - s * ly0) + px;
                sy = (s * lx0 + c * ly0) + py;
            }
            light_sense(cfg, st, e, sx / WORLD_SCALE, sy / WORLD_SCALE, &st->light_value[o + b], &st->light_gx[o + b], &st->light_gy[o + b]);
        }
    }
    return 0;
}

int kbo_contact_capacity(const kbo_config *cfg) {
    /* contact capacity per env (must equal the HIP kernel's, kb_contact_capacity) */
    long N = cfg->num_bots;
    if (cfg->contact_capacity > 0) return (int)((cfg->contact_capacity + 7) & ~7);
    long cap = N * (N - 1) / 2 + 4L * N;
    if (cap > 2304) cap = 2304;
    if (cap < 4L * N + 64) cap = 4L * N + 64;
    cap += 40L * cfg->num_objects;
    cap = (cap + 7) & ~7L;
    return (int)cap;
}

static int work_alloc(work_t *w, const kbo_config *cfg, const derived_t *d) {
    memset(w, 0, sizeof(*w));
    int N = cfg->num_bots, M = cfg->num_objects, T = N + M;
    w->N = N; w->M = M; w->S = cfg->ws_slots; w->d = d;
    w->px = (float *)malloc(sizeof(float) * T * 9);
    w->py = w->px + T; w->vx = w->py + T; w->vy = w->vx + T; w->bw = w->vy + T;
    w->x0 = w->bw + T; w->y0 = w->x0 + T; w->a0 = w->y0 + T; w->ang = w->a0 + T;
    for (int i = 0; i < T; ++i) w->ang[i] = 0.0f;
    w->cell = (int *)malloc(sizeof(int) * T * 3); w->cx = w->cell + T; w->cy = w->cx + T;
    w->cell_start = (int *)malloc(sizeof(int) * (d->gw * d->gh + 1));
    w->cell_items = (int *)malloc(sizeof(int) * T);
    w->cap = kbo_contact_capacity(cfg);
    w->con = (contact_t *)malloc(sizeof(contact_t) * w->cap);
    w->parent = (int *)malloc(sizeof(int) * T * 3); w->woff = w->parent + T; w->wnoff = w->woff + T;
    w->active = (unsigned char *)malloc(4 * (size_t)T); w->next_active = w->active + T;
    w->isl_awake = w->next_active + T; w->isl_unsolved = w->isl_awake + T;
    w->slp = (float *)malloc(sizeof(float) * T * 2); w->isl_min = w->slp + T;
    w->ws_hit = (unsigned char *)malloc((size_t)w->cap + 1);
    return (w->px && w->cell && w->cell_start && w->cell_items && w->con && w->parent && w->active && w->slp && w->ws_hit) ? 0 : -1;
}
static void work_free(work_t *w) {
    free(w->px); free(w->cell); free(w->cell_start); free(w->cell_items); free(w->con); free(w->parent); free(w->active); free(w->slp); free(w->ws_hit);
}

int kbo_step(const kbo_config *cfg, kbo_state *st, const float *light_action, int n_substeps, int flags,
             int num_threads) {
    if (!cfg || !st || cfg->num_bots < 1 || cfg->num_envs < 1 || cfg->num_objects < 0 || cfg->num_objects > KBO_MAX_OBJECTS) return -1;
    if (cfg->num_objects > 0 && (!st->ox || !st->oy || !st->otheta || !st->ovx || !st->ovy || !st->ow || !st->ows_acc)) return -1;
    if (cfg->allow_sleep && (!st->sleep_time || (cfg->num_objects > 0 && !st->osleep))) return -1;
    if (cfg->drive_mode == KBO_DRIVE_MIXED && !st->bot_mode) return -1;
    derived_t d; derive(cfg, &d);
    int nt = num_threads > 1 ? num_threads : 1;
    int err = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        work_t w;
        int ok = work_alloc(&w, cfg, &d) == 0;
        if (!ok) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = -2;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int e = 0; e < cfg->num_envs; ++e) {
            if (!ok) continue;
            w.status = 0;
            for (int s = 0; s < n_substeps; ++s) substep_env(cfg, &d, st, light_action, flags, e, &w);
        }
        work_free(&w);
    }
    (void)nt;
    return err;
}

int kbo_set_actions(const kbo_config *cfg, kbo_state *st, const float *actions) {
    const size_t T = (size_t)cfg->num_envs * cfg->num_bots;
    const float mw = 0.5f * 3.14159265358979323846f;
    for (size_t i = 0; i < T; ++i) {
        float a0 = actions ? actions[2 * i] : 0.0f, a1 = actions ? actions[2 * i + 1] : 0.0f;
        const int law = cfg->drive_mode == KBO_DRIVE_MIXED ? st->bot_mode[i] : cfg->drive_mode;
        if (law == KBO_DRIVE_VELOCITY) {
            /* kilobot.py:216-218, 235-241 */
            st->v[i] = fmaxf(fminf(a0, 0.01f), 0.0f);
            st->w[i] = fmaxf(fminf(a1, mw), -mw);
        } else if (law == KBO_DRIVE_ACCEL) {
            /* kilobot.py:269, 283-289 */
            const float aw = 0.2f * 3.14159265358979323846f;
            st->acc_v[i] = fmaxf(fminf(a0, 0.005f), -0.005f);
            st->acc_w[i] = fmaxf(fminf(a1, aw), -aw);
        } else if (cfg->drive_mode != KBO_DRIVE_MIXED) return -1;      /* (mixed env: kilobots of the other laws take no action) */
    }
    return 0;
}

int kbo_count_contacts(const kbo_config *cfg, const kbo_state *st, int env, int *n_botbot, int *n_wall, int *n_obj) {
    derived_t d; derive(cfg, &d);
    work_t w;
    if (work_alloc(&w, cfg, &d) != 0) return -2;
    const size_t o = (size_t)env * cfg->num_bots;
    for (int b = 0; b < cfg->num_bots; ++b) { w.px[b] = st->x[o + b]; w.py[b] = st->y[o + b]; }
    for (int m = 0; m < cfg->num_objects; ++m) {
        w.px[cfg->num_bots + m] = st->ox[(size_t)env * cfg->num_objects + m];
        w.py[cfg->num_bots + m] = st->oy[(size_t)env * cfg->num_objects + m];
        w.ang[cfg->num_bots + m] = st->otheta[(size_t)env * cfg->num_objects + m];
        if (d.lc_obj[m].x != 0.0f || d.lc_obj[m].y != 0.0f) {
            xf_t t = xf_make(w.px[cfg->num_bots + m], w.py[cfg->num_bots + m], w.ang[cfg->num_bots + m]);
            v2 cm = xf_mul(&t, d.lc_obj[m]);
            w.px[cfg->num_bots + m] = cm.x; w.py[cfg->num_bots + m] = cm.y;
        }
    }
    detect_env(cfg, &d, st, env, &w);
    int nb = 0, nw = 0;
    for (int i = 0; i < w.ncon; ++i) { if (w.con[i].cls >= CLS_BOT_OBJ) continue; if (w.con[i].a < 0) nw++; else nb++; }
    int no = 0;
    for (int i = 0; i < w.ncon; ++i) if (w.con[i].cls >= CLS_BOT_OBJ) no++;
    no += w.nmc;
    if (n_obj) *n_obj = no;
    if (n_botbot) *n_botbot = nb;
    if (n_wall) *n_wall = nw;
    int r = w.status;
    work_free(&w);
    return r;
}
