// kb_abi.hip -- the C ABI of libkilobots_hip.so (include/kilobots_hip.h) and two small elementwise kernels.
//
// The hot kernel is kb_step_kernel (kb_step_kernel.h, instantiated per drive law in kb_inst_d*.hip):
// one workgroup owns one env for the whole launch: poses are loaded once from HBM into LDS, `n_substeps` iterations
// of the reference substep loop (gym_kilobots/envs/kilobots_env.py:168-190) run out of LDS / registers, poses are
// written back once.  Per substep:
//   drive law (kilobot.py:86-127,191-203,253-258,294-300,318-333) + light (light.py:59-75,99-148,176-189,218-319)
//   -> broadphase: uniform grid of per-cell linked lists in LDS (one atomic exchange per bot)
//   -> narrowphase: circle-circle / circle-wall / circle-polygon (Box2D b2CollideCircles, b2CollideEdgeAndCircle,
//      b2CollidePolygonAndCircle), 5-cell half stencil, warm-start impulses matched from the previous substep;
//      manifolds of the object-object and object-wall contacts (kb_objects.h)
//   -> islands: lock-free union-find in LDS; islands placed on wavefronts in order of size
//   -> solver (b2ContactSolver semantics): warm start + 10 sequential-impulse velocity sweeps, symplectic Euler,
//      <= 10 position sweeps with Box2D's per-island early out, continuous step against the walls.
// Gauss-Seidel order.  Every contact gets a key (class, rank): class from the relative grid position of the two
// bodies and the parity of the base cell, rank from its position inside its cell-pair group.  Two contacts with the
// same key never share a body, so all contacts of one key can be solved concurrently and the result equals the
// sequential sweep in (class, group, A, B) order that DESIGN.md specifies.  Islands are independent, so each island
// is bound to ONE wavefront, which sweeps its contacts level by level of their dependency depth with no workgroup
// barrier at all (LDS operations of one wave execute in order).  Only when one island is very large does the whole
// workgroup cooperate on the sweep with s_barrier between keys.
// No MFMA anywhere: this is LDS-, issue- and HBM-bound integer/float work.
//
// Arithmetic: fp32, compiled with -ffp-contract=off; every expression is written in the operation order of the
// specification so results do not depend on launch fusion, workgroup size or sharding.
#include <cstdlib>
#include <new>

#include "kb_common.h"
#include "kb_objects.h"

using namespace kb;

namespace {

// set_action alone (kilobot.py:235-241, 283-289)
__global__ void kb_set_actions_kernel(const Params p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t T = (size_t)p.E * p.N;
    if (i >= T) return;
    float a0 = 0.0f, a1 = 0.0f;
    if (p.actions) { const float2 a = reinterpret_cast<const float2 *>(p.actions)[i]; a0 = a.x; a1 = a.y; }
    const int law = p.drive_mode == KB_DRIVE_MIXED ? (int)p.buf.bot_mode[i] : p.drive_mode;
    if (law == KB_DRIVE_VELOCITY) {
        const float mw = 0.5f * 3.14159265358979323846f;
        p.buf.v[i] = fmaxf(fminf(a0, 0.01f), 0.0f);
        p.buf.w[i] = fmaxf(fminf(a1, mw), -mw);
    } else if (law == KB_DRIVE_ACCEL) {
        const float aw = 0.2f * 3.14159265358979323846f;
        p.buf.acc_v[i] = fmaxf(fminf(a0, 0.005f), -0.005f);
        p.buf.acc_w[i] = fmaxf(fminf(a1, aw), -aw);
    }
}

// Body.get_pose for every kilobot (body.py:63-65): metres, radians
__global__ void kb_get_poses_kernel(const float *x, const float *y, const float *th, float *out, size_t T) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    out[3 * i + 0] = x[i] / WORLD_SCALE;
    out[3 * i + 1] = y[i] / WORLD_SCALE;
    out[3 * i + 2] = th[i];
}

// KilobotsEnv.get_state() in one read (kilobots_env.py:115-118): per env 3 N kilobot words, 3 M object words, the status word
__global__ void kb_get_state_kernel(const Params p, float *out) {
    const int per = p.N + p.M + 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)p.E * per) return;
    const size_t e = i / per;
    const int k = (int)(i - e * per);
    float *row = out + e * (size_t)(3 * per - 2);
    if (k < p.N) {
        const size_t j = e * p.N + k;
        row[3 * k + 0] = p.buf.x[j] / WORLD_SCALE;
        row[3 * k + 1] = p.buf.y[j] / WORLD_SCALE;
        row[3 * k + 2] = p.buf.theta[j];
    } else if (k < p.N + p.M) {
        const size_t j = e * p.M + (k - p.N);
        row[3 * k + 0] = p.buf.ox[j] / WORLD_SCALE;
        row[3 * k + 1] = p.buf.oy[j] / WORLD_SCALE;
        row[3 * k + 2] = p.buf.otheta[j];
    } else {
        row[3 * k] = __int_as_float(p.buf.status[e]);
    }
}


// IR-range neighbour sensing on the current poses (kb_sense): one workgroup per env builds the cell lists of the
// broadphase grid in LDS and runs the sensing pass of the step kernel on them
__global__ void __launch_bounds__(256) kb_sense_kernel(const Params p, const int s, const float R2, unsigned *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int e = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, N = p.N;
    float2 *pos = reinterpret_cast<float2 *>(smem);
    unsigned *cnt16 = reinterpret_cast<unsigned *>(smem + 8 * p.NP);
    unsigned short *nextb = reinterpret_cast<unsigned short *>(smem + 10 * p.NP);
    unsigned short *cellOf = nextb + p.NP;
    unsigned short *head = cellOf + p.NP;
    const size_t o = (size_t)e * N;
    for (int c = tid; c < p.ncell; c += nt) head[c] = EMPTY16;
    for (int b = tid; b < p.NP / 2; b += nt) cnt16[b] = 0;
    __syncthreads();
    for (int b = tid; b < N; b += nt) {
        const float bx = p.buf.x[o + b], by = p.buf.y[o + b];
        pos[b].x = bx; pos[b].y = by;
        int cx = (int)floorf((bx - p.xmin) * p.inv_cell);
        int cy = (int)floorf((by - p.ymin) * p.inv_cell);
        cx = cx < 0 ? 0 : (cx >= p.gw ? p.gw - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= p.gh ? p.gh - 1 : cy);
        const int cell = cy * p.gw + cx;
        cellOf[b] = (unsigned short)cell;
        nextb[b] = (unsigned short)kb_exch16(head, cell, (unsigned)b);
    }
    __syncthreads();
    kb_sense_pass(pos, head, nextb, cellOf, cnt16, N, nt, tid, p.gw, p.gh, s, R2, 0);
    __syncthreads();
    for (int b = tid; b < N; b += nt) out[o + b] = (unsigned)reinterpret_cast<unsigned short *>(cnt16)[b];
}

// The sensing point of a substep on its own (kb_light_sense): light.step + value_and_gradients at every kilobot's sensor
// (kilobots_env.py:171-180), with the arithmetic of the step kernel (shared functions of kb_common.h), for kilobots whose
// _loop runs on the host between the sensing and the motor law.  One workgroup per env.
__global__ void __launch_bounds__(256) kb_light_sense_kernel(const Params p, const float *light_action) {
    const int e = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, N = p.N;
    const kb_buffers &g = p.buf;
    const size_t o = (size_t)e * N;
    const bool general = p.light_type != KB_LIGHT_CIRCULAR;
    float lx = 0.0f, ly = 0.0f;
    float glx[KB_MAX_LIGHTS], gly[KB_MAX_LIGHTS], glvx[KB_MAX_LIGHTS], glvy[KB_MAX_LIGHTS];
#pragma unroll
    for (int i = 0; i < KB_MAX_LIGHTS; ++i) { glx[i] = 0.0f; gly[i] = 0.0f; glvx[i] = 0.0f; glvy[i] = 0.0f; }
    if (!general) { lx = g.light_x[e]; ly = g.light_y[e]; }
    else {
#pragma unroll
        for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
            if (i >= p.lcount) break;
            glx[i] = g.light_x[(size_t)e * p.lcount + i];
            if (p.light_type != KB_LIGHT_GRADIENT) gly[i] = g.light_y[(size_t)e * p.lcount + i];
            if (p.lkind[i] == KB_LIGHT_MOMENTUM) { glvx[i] = g.light_vx[(size_t)e * p.lcount + i]; glvy[i] = g.light_vy[(size_t)e * p.lcount + i]; }
        }
    }
    __syncthreads();      // every thread holds the old light state before thread 0 stores the new one
    if (light_action) {
        if (general) kb_light_general_step(p, light_action + (size_t)e * p.ladim, p.h, glx, gly, glvx, glvy);
        else kb_light_single_step(p, light_action + 2 * e, p.h, lx, ly);
        if (tid == 0) {
            if (!general) { g.light_x[e] = lx; g.light_y[e] = ly; }
            else {
#pragma unroll
                for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
                    if (i >= p.lcount) break;
                    g.light_x[(size_t)e * p.lcount + i] = glx[i];
                    if (p.light_type != KB_LIGHT_GRADIENT) g.light_y[(size_t)e * p.lcount + i] = gly[i];
                    if (p.lkind[i] == KB_LIGHT_MOMENTUM) { g.light_vx[(size_t)e * p.lcount + i] = glvx[i]; g.light_vy[(size_t)e * p.lcount + i] = glvy[i]; }
                }
            }
        }
    }
    for (int b = tid; b < N; b += nt) {
        const float bx = g.x[o + b], by = g.y[o + b];
        float sx = bx, sy = by;
        if ((p.drive_mode == KB_DRIVE_MIXED ? (int)g.bot_mode[o + b] : p.drive_mode) != KB_DRIVE_SIMPLE_PHOTOTAXIS) {  // kilobot.py:54-55: world point of (0, -r)
            float s, c;
            kb_sincosf(g.theta[o + b], s, c);
            const float lx0 = 0.0f, ly0 = -p.r_bot;
            sx = (c * lx0 - s * ly0) + bx;
            sy = (s * lx0 + c * ly0) + by;
        }
        float lval, lgx, lgy;
        if (general) kb_light_general_sense(p, glx, gly, sx / WORLD_SCALE, sy / WORLD_SCALE, lval, lgx, lgy);
        else kb_light_circular(sx / WORLD_SCALE, sy / WORLD_SCALE, lx, ly, p.light_radius, lval, lgx, lgy);
        g.light_value[o + b] = lval; g.light_gx[o + b] = lgx; g.light_gy[o + b] = lgy;
    }
}

// KilobotsEnv.reset spawn (kb_reset): one thread per kilobot, Philox4x32-10 keyed by the seed, counter (global env, bot)
struct ResetArgs {
    unsigned k0, k1;
    int env_offset, random_theta, random_velocity;
    float mean_x, mean_y, std, lo_x, lo_y, hi_x, hi_y;
};
__global__ void kb_reset_kernel(const Params p, const ResetArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t T = (size_t)p.E * p.N;
    if (i >= T) return;
    const int e = (int)(i / p.N), b = (int)(i % p.N);
    U4 c;
    c.x = (unsigned)(a.env_offset + e); c.y = (unsigned)b; c.z = 0u; c.w = 0u;
    const U4 r = kb_philox4x32_10(c, a.k0, a.k1);
    // Box-Muller on u1 in (0, 1], u2 in [0, 1)
    const float u1 = (float)((r.x >> 8) + 1u) * (1.0f / 16777216.0f);
    const float u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * kb_logf(u1));
    float sn, cs;
    kb_sincosf(6.28318530717958647692f * u2, sn, cs);
    float xm = a.mean_x + a.std * (rad * cs), ym = a.mean_y + a.std * (rad * sn);
    xm = fminf(fmaxf(xm, a.lo_x), a.hi_x); ym = fminf(fmaxf(ym, a.lo_y), a.hi_y);   // yaml_kilobots_env.py:350-351
    p.buf.x[i] = xm * WORLD_SCALE; p.buf.y[i] = ym * WORLD_SCALE;
    float th = 0.0f;                                                                   // body.py:28-29
    if (a.random_theta) th = ((float)(r.z >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * 3.14159265358979323846f;
    p.buf.theta[i] = th;
    p.buf.ws_cnt[i] = 0;
    const int law = p.drive_mode == KB_DRIVE_MIXED ? (int)p.buf.bot_mode[i] : p.drive_mode;
    if (law == KB_DRIVE_VELOCITY || law == KB_DRIVE_ACCEL) {
        float v = 0.0f, w = 0.0f;
        if (a.random_velocity) {                                                       // kilobot.py:225-229
            v = (float)(r.w & 0xFFFFu) * (1.0f / 65536.0f) * 0.01f;
            w = ((float)(r.w >> 16) * (1.0f / 65536.0f) * 2.0f - 1.0f) * (0.5f * 3.14159265358979323846f);
        }
        p.buf.v[i] = v; p.buf.w[i] = w;
    }
    if (law == KB_DRIVE_ACCEL) { p.buf.acc_v[i] = 0.0f; p.buf.acc_w[i] = 0.0f; }
    if (law == KB_DRIVE_MOTORS || law == KB_DRIVE_PHOTOTAXIS) { p.buf.motor_l[i] = 255; p.buf.motor_r[i] = 0; }   // turn_left
    if (law == KB_DRIVE_PHOTOTAXIS) {
        p.buf.pt_threshold[i] = -INFINITY; p.buf.pt_update[i] = 0; p.buf.pt_nochange[i] = 0; p.buf.pt_dir[i] = 0;
    }
    if (b == 0) p.buf.status[e] = 0;
    if (p.buf.sleep_time) p.buf.sleep_time[i] = 0.0f;      // new bodies are awake (b2BodyDef::awake)
    if (p.buf.nbr_count) p.buf.nbr_count[i] = 0u;          // (the resolve step does not sense: no counts of the previous episode)
    if (p.M > 0) {      // forget the objects' manifold impulses as well
        float *ows = p.buf.ows_acc + (size_t)e * (MAXOBJ * KB_OWS_COLS * KB_OWS_WORDS);
        for (int k = b; k < MAXOBJ * KB_OWS_COLS * KB_OWS_WORDS; k += p.N) ows[k] = -1.0f;
    }
}

thread_local char g_err[512] = "";

float kb_clampf_host(float a, float lo, float hi) { return fmaxf(lo, fminf(a, hi)); }
// reach of the sensing stencil in cells: cell indices are monotone in the coordinate, and two kilobots within Rw differ
// by at most floor(Rw / cell) + 1 cells (the small margin covers the rounding of the cell computation)
int sense_reach(float Rw, float inv_cell) { return (int)floorf(Rw * inv_cell + 1e-3f) + 1; }

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}


}  // namespace

struct kb_sim {
    kb_config cfg;
    Params p;
    bool bound;
    const void *attr_fn;   // kernel whose dynamic-LDS limit has been raised
    int threads;
    int capL_regular;      // LDS staging entries of the regular image (the fixed-size sorted-bin image has its own: ldsb::CAPL)
    int tier;              // register budget of the kernels without objects: 0 = 128 VGPRs, 2 = 80 VGPRs (6 waves per SIMD)
};

// dynamic LDS of one env: the bucket tables scale with the waves of the workgroup, the object tables exist only in
// scenes with objects (namespace lds, kb_common.h)
static bool uses_fixed_1024(const kb::Params &p, int threads) {   // the instantiation kb_step picks: the ONE place that decides
    const long cap1024 = (4L * 1024 + 64 + 7) & ~7L;
    if (p.allow_sleep && p.M > 0) return false;            // (the fixed-size instantiations with objects do not carry the sleep state)
    return p.drive_mode == KB_DRIVE_VELOCITY && p.N == 1024 && p.light_type == KB_LIGHT_NONE && threads == 64 * kb::MAX_WAVES &&
           kb::BPT * 64 * kb::MAX_WAVES == 1024 && p.NP == 1024 && p.NB == 1024 + KB_MAX_OBJECTS + 4 &&
           (p.M > 0 || p.cap == (int)cap1024);      // (without objects the contact capacity is a compile-time constant of the kernel)
}
static int lds_image_bytes(const kb::Params &p, int threads, int capL);
// sorted-bin image: where the per-island minimum of the sleep times (NB words, SLEEP kernels, idle arrays between the position
// sweeps and the continuous step) lies -- over the bin boundaries, else over the staged pairs, else behind the image; *total
// receives the bytes of the image including that area
static int bins_islmin_offset(const kb::Params &p, int threads, int capL, int *total) {
    const bool hc = p.hmask != 0 && !uses_fixed_1024(p, threads);
    const int nw = threads / 64;
    int tot = kb::ldsb::total(p.NB, p.NP, hc, capL, p.nhead, nw), off;
    if (kb::ldsb::binE_size(p.nhead, nw) >= 4 * p.NB) off = kb::ldsb::binE(p.NB, p.NP, hc, capL);
    else if (capL >= p.NB) off = kb::ldsb::con32(p.NB, p.NP, hc, capL, 0);
    else { off = tot; if (p.allow_sleep) tot += (4 * p.NB + 15) & ~15; }
    if (total) *total = tot;
    return off;
}
static int lds_bytes_for(const kb::Params &p, int threads, int capL) {
    const int img = lds_image_bytes(p, threads, capL);
    // KB_DRIVE_MIXED: the drive law of every kilobot (one byte each) lies behind the image (Params.botlaw_off)
    return p.drive_mode == KB_DRIVE_MIXED ? ((img + 15) & ~15) + ((p.NP + 15) & ~15) : img;
}
static int lds_image_bytes(const kb::Params &p, int threads, int capL) {
    if (p.M == 0 && p.drive_mode != KB_DRIVE_MIXED) {      // kernels without objects: the sorted-bin image (namespace ldsb)
        const bool fixed = uses_fixed_1024(p, threads);
        int total = 0;
        bins_islmin_offset(p, threads, fixed ? kb::ldsb::CAPL : capL, &total);
        return total;
    }
    const bool objarea = p.M > 0 || p.drive_mode == KB_DRIVE_MIXED || uses_fixed_1024(p, threads);
    return kb::lds::total(kb::lds::fixed(objarea, threads / 64), p.NB, capL, p.NP, p.nhead, p.nmc);
}

// workgroups of `threads` threads and `lds` bytes that one CU holds at a register budget of `wps` waves per SIMD
// (LDS is handed out in granules: an image of 54 544 B got two workgroups per CU, one of 53 168 B three)
static int resident_envs(int lds, int threads, int wps) {
    const int LDS_CU = 160 * 1024, GRANULE = 1280;
    const int byLds = LDS_CU / (((lds + GRANULE - 1) / GRANULE) * GRANULE);
    const int byWaves = (4 * wps) / (threads / 64);
    const int n = byLds < byWaves ? byLds : byWaves;
    return n < 1 ? (byLds >= 1 ? 1 : 0) : n;
}
// register budget of a kernel without objects: 80 VGPRs (tier 2) where that holds more envs than 128 VGPRs
static int pick_tier(const kb::Params &p, int threads, int lds) {
    if (p.M > 0 || p.drive_mode == KB_DRIVE_MIXED) return 0;
    if (const char *t = getenv("KB_TIER")) return atoi(t) == 2 ? 2 : 0;      // experiment knob (A/B of the register budgets)
    return resident_envs(lds, threads, KB_COMPACT_WAVES_PER_SIMD) > resident_envs(lds, threads, KB_MIN_WAVES_PER_SIMD) ? 2 : 0;
}

extern "C" {

const char *kb_last_error(void) { return g_err; }
const char *kb_version(void) { return "kilobots_hip 0.1 (gfx950)"; }

int kb_create(const kb_config *cfg, kb_sim **out) {
    if (!cfg || !out) return fail(KB_EINVAL, "kb_create: NULL argument");
    if (cfg->num_envs < 1 || cfg->num_bots < 1 || cfg->num_bots > KB_MAX_BOTS)
        return fail(KB_EINVAL, "kb_create: num_envs >= 1 and 1 <= num_bots <= 1024 required");
    if (cfg->num_objects < 0 || cfg->num_objects > KB_MAX_OBJECTS) return fail(KB_EINVAL, "kb_create: 0 <= num_objects <= 8 required");
    const int nfix = cfg->num_fixtures > 0 ? cfg->num_fixtures : cfg->num_objects;
    if (cfg->num_fixtures != 0 && (cfg->num_fixtures < cfg->num_objects || cfg->num_fixtures > KB_MAX_OBJECTS))
        return fail(KB_EINVAL, "kb_create: num_fixtures must be 0 or num_objects..8");
    {
        int per_body[KB_MAX_OBJECTS] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = 0; f < nfix; ++f) {
            const int b = cfg->num_fixtures > 0 ? cfg->obj_fixture_body[f] : f;
            if (b < 0 || b >= cfg->num_objects) return fail(KB_EINVAL, "kb_create: obj_fixture_body out of range");
            per_body[b]++;
            if (cfg->obj_shape[f] == KB_SHAPE_CIRCLE && per_body[b] > 1) return fail(KB_EINVAL, "kb_create: a circle must be the only fixture of its object");
        }
        for (int b = 0; b < cfg->num_objects; ++b) {
            if (per_body[b] == 0) return fail(KB_EINVAL, "kb_create: every object needs a fixture");
            if (per_body[b] > 1)
                for (int f = 0; f < nfix; ++f)
                    if (cfg->obj_fixture_body[f] == b && cfg->obj_shape[f] == KB_SHAPE_CIRCLE)
                        return fail(KB_EINVAL, "kb_create: a circle must be the only fixture of its object");
        }
    }
    for (int m = 0; m < nfix; ++m) {
        const int sh = cfg->obj_shape[m];
        if (sh < KB_SHAPE_CIRCLE || sh > KB_SHAPE_POLYGON) return fail(KB_EINVAL, "kb_create: bad obj_shape");
        if (sh == KB_SHAPE_CIRCLE && !(cfg->obj_radius[m] > 0.0f)) return fail(KB_EINVAL, "kb_create: obj_radius must be positive");
        if (sh == KB_SHAPE_BOX && !(cfg->obj_verts[m][0][0] > 0.0f && cfg->obj_verts[m][0][1] > 0.0f))
            return fail(KB_EINVAL, "kb_create: box half extents (obj_verts[m][0]) must be positive");
        if (sh == KB_SHAPE_POLYGON) {
            const int n = cfg->obj_nverts[m];
            if (n < 3 || n > KB_MAX_POLY_VERTS) return fail(KB_EINVAL, "kb_create: polygons need 3..4 vertices");
            for (int i = 0; i < n; ++i) {       // counter-clockwise and convex
                const float *p0 = cfg->obj_verts[m][i], *p1 = cfg->obj_verts[m][(i + 1) % n], *p2 = cfg->obj_verts[m][(i + 2) % n];
                if (!((p1[0] - p0[0]) * (p2[1] - p1[1]) - (p1[1] - p0[1]) * (p2[0] - p1[0]) > 0.0f))
                    return fail(KB_EINVAL, "kb_create: polygon vertices must form a counter-clockwise convex hull");
            }
        }
    }
    if (cfg->num_objects > 0 && (!(cfg->obj_friction >= 0.0f) || !(cfg->wall_friction >= 0.0f)))
        return fail(KB_EINVAL, "kb_create: friction coefficients must be non-negative");
    if (cfg->num_objects > 0 && !(cfg->obj_density > 0.0f)) return fail(KB_EINVAL, "kb_create: obj_density must be positive");
    if (cfg->drive_mode < 0 || cfg->drive_mode > KB_DRIVE_MIXED) return fail(KB_EINVAL, "kb_create: bad drive_mode");
    if (cfg->light_type < KB_LIGHT_NONE || cfg->light_type > KB_LIGHT_COMPOSITE)
        return fail(KB_EINVAL, "kb_create: unsupported light_type");
    if (cfg->light_type == KB_LIGHT_COMPOSITE) {
        if (cfg->light_count < 1 || cfg->light_count > KB_MAX_LIGHTS) return fail(KB_EINVAL, "kb_create: 1 <= light_count <= 4 required");
        for (int i = 0; i < cfg->light_count; ++i)
            if (cfg->light_kind[i] != KB_LIGHT_CIRCULAR && cfg->light_kind[i] != KB_LIGHT_MOMENTUM)
                return fail(KB_EINVAL, "kb_create: composite components must be circular or momentum lights");
    }
    if ((cfg->drive_mode == KB_DRIVE_SIMPLE_PHOTOTAXIS || cfg->drive_mode == KB_DRIVE_PHOTOTAXIS) &&
        cfg->light_type == KB_LIGHT_NONE)
        return fail(KB_EINVAL, "kb_create: phototaxis drive modes need a light");
    if (cfg->ws_slots < 1 || cfg->ws_slots > 64) return fail(KB_EINVAL, "kb_create: 1 <= ws_slots <= 64 required");
    if (cfg->solver_mode < 0 || cfg->solver_mode > 4) return fail(KB_EINVAL, "kb_create: solver_mode must be 0..4");
    if (cfg->damping_model != KB_DAMPING_PADE && cfg->damping_model != KB_DAMPING_LINEAR) return fail(KB_EINVAL, "kb_create: damping_model must be KB_DAMPING_PADE or KB_DAMPING_LINEAR");
    if (!(cfg->sense_radius >= 0.0f)) return fail(KB_EINVAL, "kb_create: sense_radius must be >= 0");
    if (cfg->contact_capacity < 0 || cfg->contact_capacity > 65528) return fail(KB_EINVAL, "kb_create: 0 <= contact_capacity <= 65528 required");
    if (!(cfg->dt > 0.0f) || cfg->vel_iters < 0 || cfg->pos_iters < 0 || !(cfg->world_width > 0.0f) ||
        !(cfg->world_height > 0.0f) || !(cfg->bot_radius > 0.0f) || !(cfg->bot_density > 0.0f))
        return fail(KB_EINVAL, "kb_create: non-positive dt / size / radius / density");
    kb_sim *s = new (std::nothrow) kb_sim();
    if (!s) return fail(KB_EINVAL, "kb_create: out of host memory");
    s->cfg = *cfg;
    s->bound = false;
    s->attr_fn = nullptr;
    Params &p = s->p;
    memset(&p, 0, sizeof(p));
    p.N = cfg->num_bots; p.E = cfg->num_envs; p.S = cfg->ws_slots;
    p.allow_sleep = cfg->allow_sleep != 0;
    p.drive_mode = cfg->drive_mode; p.light_type = cfg->light_type;
    p.vel_iters = cfg->vel_iters; p.pos_iters = cfg->pos_iters;
    const float W = cfg->world_width * WORLD_SCALE, H = cfg->world_height * WORLD_SCALE;
    p.xmin = -0.5f * W; p.xmax = 0.5f * W; p.ymin = -0.5f * H; p.ymax = 0.5f * H;
    float cell = CELL_SIZE;
    const float dmin = 2.0f * cfg->bot_radius * WORLD_SCALE;
    while (cell < dmin) cell *= 2.0f;
    for (;;) {
        p.inv_cell = 1.0f / cell;
        p.gw = (int)ceilf(W * p.inv_cell); if (p.gw < 1) p.gw = 1;
        p.gh = (int)ceilf(H * p.inv_cell); if (p.gh < 1) p.gh = 1;
        if ((long)p.gw * p.gh <= MAX_CELLS) break;
        cell *= 2.0f;
    }
    p.ncell = p.gw * p.gh;
    {   // sparse swarms: a hash table of the cells instead of one list head per cell (kb_step_kernel.h, `hashed`); only where
        // it at least halves the table, and never for the fixed-size kernel
        int H = 64;
        while (H < 2 * cfg->num_bots) H <<= 1;
        if (2 * H <= p.ncell && cfg->num_bots != 1024) { p.nhead = H; p.hmask = H - 1; }
        else { p.nhead = p.ncell; p.hmask = 0; }
    }
    p.h = cfg->dt;
    p.r_bot = cfg->bot_radius * WORLD_SCALE;
    const float m = cfg->bot_density * B2_PI * p.r_bot * p.r_bot;  // b2CircleShape::ComputeMass
    p.im_bot = m > 0.0f ? 1.0f / m : 0.0f;
    {
        const float rr = p.r_bot + p.r_bot, rw = B2_POLYGON_RADIUS + p.r_bot;
        p.rr2 = rr * rr; p.rw2 = rw * rw; p.rw_tot = p.r_bot + B2_POLYGON_RADIUS;
        const float kbb = p.im_bot + p.im_bot, kwb = 0.0f + p.im_bot;
        p.nm_bb = kbb > 0.0f ? 1.0f / kbb : 0.0f; p.nm_wb = kwb > 0.0f ? 1.0f / kwb : 0.0f;
    }
    for (int k = 0; k < 5; ++k) {       // KB_DRIVE_MIXED: the classes have different fixture densities (kilobot.py:25 / :214)
        p.im_mode[k] = p.im_bot;
        if (cfg->drive_mode == KB_DRIVE_MIXED && cfg->mode_density[k] > 0.0f) {
            const float mk = cfg->mode_density[k] * B2_PI * p.r_bot * p.r_bot;
            p.im_mode[k] = mk > 0.0f ? 1.0f / mk : 0.0f;
        }
    }
    p.botlaw_off = 0;
    // b2Island::Solve damping factor per step: Pade (Box2D >= 2.3.1) or the older clamped linear form
    auto damp = [&](float c) -> float {
        if (cfg->damping_model == KB_DAMPING_LINEAR) return kb_clampf_host(1.0f - p.h * c, 0.0f, 1.0f);
        return 1.0f / (1.0f + p.h * c);
    };
    p.kl_bot = damp(cfg->bot_linear_damping);
    p.ka_bot = damp(cfg->bot_angular_damping);
    p.light_radius = cfg->light_radius;
    p.lcount = cfg->light_type == KB_LIGHT_COMPOSITE ? cfg->light_count : 1;
    p.ladim = cfg->light_type == KB_LIGHT_NONE ? 0 : (cfg->light_type == KB_LIGHT_GRADIENT ? 1 : 2 * p.lcount);
    for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
        const bool comp = cfg->light_type == KB_LIGHT_COMPOSITE;
        p.lkind[i] = comp ? cfg->light_kind[i] : cfg->light_type;
        p.lradius[i] = comp ? cfg->lightc_radius[i] : cfg->light_radius;
        p.lmaxv[i] = comp ? cfg->lightc_max_velocity[i] : cfg->light_max_velocity;
        for (int k = 0; k < 2; ++k) {
            p.llo[i][k] = comp ? cfg->lightc_lo[i][k] : cfg->light_lo[k];
            p.lhi[i][k] = comp ? cfg->lightc_hi[i][k] : cfg->light_hi[k];
            p.lalo[i][k] = comp ? cfg->lightc_act_lo[i][k] : cfg->light_act_lo[k];
            p.lahi[i][k] = comp ? cfg->lightc_act_hi[i][k] : cfg->light_act_hi[k];
        }
    }
    for (int i = 0; i < 2; ++i) {
        p.light_lo[i] = cfg->light_lo[i]; p.light_hi[i] = cfg->light_hi[i];
        p.act_lo[i] = cfg->light_act_lo[i]; p.act_hi[i] = cfg->light_act_hi[i];
    }
    long cap = (long)p.N * (p.N - 1) / 2 + 4L * p.N;
    if (cap > 2304) cap = 2304;
    if (cap < 4L * p.N + 64) cap = 4L * p.N + 64;
    cap += 40L * cfg->num_objects;
    if (cfg->contact_capacity > 0) cap = cfg->contact_capacity;
    cap = (cap + 7) & ~7L;
    p.cap = (int)cap;
    // with objects the LDS staging area gives up a few entries to the manifold-constraint records, so that two envs
    // of 1024 kilobots still share a CU
    int capLmax = cfg->num_objects > 0 ? CAP_LDS - 8 * mc_candidates(nfix) : CAP_LDS;
    // small swarms: stage what a packed cluster can produce (a hexagonal packing has < 3 contacts per kilobot, + walls) rather than all pairs, so
    // that more one-wave envs fit a CU; a spawn that overlaps more than that takes the global staging slice
    const int typical = 3 * p.N + 64 > 256 ? 3 * p.N + 64 : 256;
    if (capLmax > typical) capLmax = typical;
    p.capL = p.cap < capLmax ? p.cap : capLmax;
    if (cfg->num_objects > 0 || cfg->drive_mode == KB_DRIVE_MIXED) {
        // kernels with objects: the per-island minimum of the sleep times (N + M words) lies over the pair and info arrays of
        // the LDS staging area, whatever contact_capacity says
        int lo = ((((p.N + 3) & ~3) + KB_MAX_OBJECTS + 4 + 1) / 2 + 7) & ~7;
        if (lo < 64) lo = 64;
        if (p.capL < lo) p.capL = lo;
    }
    if (cfg->num_objects == 0 && cfg->drive_mode != KB_DRIVE_MIXED) {
        // sorted-bin image: scratch arrays of the sort lie over the staging area (2 B per kilobot over 4 B per entry); a
        // smaller contact_capacity still bounds what is staged (kernel: min(capL, cap))
        int lo = ((((p.N + 3) & ~3) / 2) + 7) & ~7;
        if (lo < 64) lo = 64;
        if (p.capL < lo) p.capL = lo;
    }
    p.NP = (p.N + 3) & ~3;
    p.NB = p.NP + KB_MAX_OBJECTS + 4;
    p.M = cfg->num_objects;
    p.F = nfix;
    float bm[KB_MAX_OBJECTS], bi[KB_MAX_OBJECTS];
    V2 bc[KB_MAX_OBJECTS];
    for (int m = 0; m < KB_MAX_OBJECTS; ++m) { bm[m] = 0.0f; bi[m] = 0.0f; bc[m] = mk2(0.0f, 0.0f); }
    for (int f = 0; f < KB_MAX_OBJECTS; ++f) {
        float *T = p.otab[f];
        for (int k = 0; k < OT_WORDS; ++k) T[k] = 0.0f;
        if (f >= nfix) continue;
        const int body = cfg->num_fixtures > 0 ? cfg->obj_fixture_body[f] : f;
        const int kind = cfg->obj_shape[f];
        float mo, io;
        V2 ce = mk2(0.0f, 0.0f);
        T[OT_KIND] = (float)kind; T[OT_N] = 0.0f; T[OT_BODY] = (float)body;
        if (kind == KB_SHAPE_CIRCLE) {
            const float r = cfg->obj_radius[f] * WORLD_SCALE;
            T[OT_RADIUS] = r;
            mo = cfg->obj_density * B2_PI * r * r;       // b2CircleShape::ComputeMass
            io = mo * (0.5f * r * r);                     // I = mass * (0.5 r^2 + |p|^2), p = 0
        } else {
            int n;
            if (kind == KB_SHAPE_BOX) {                   // b2PolygonShape::SetAsBox
                const float hx = cfg->obj_verts[f][0][0], hy = cfg->obj_verts[f][0][1];
                n = 4;
                const float vx[4] = {-hx, hx, hx, -hx}, vy[4] = {-hy, -hy, hy, hy};
                const float nx[4] = {0.0f, 1.0f, 0.0f, -1.0f}, ny[4] = {-1.0f, 0.0f, 1.0f, 0.0f};
                for (int i = 0; i < 4; ++i) {
                    T[OT_VERTS + 2 * i] = vx[i]; T[OT_VERTS + 2 * i + 1] = vy[i];
                    T[OT_NORMALS + 2 * i] = nx[i]; T[OT_NORMALS + 2 * i + 1] = ny[i];
                }
            } else {                                      // b2PolygonShape::Set on an ordered hull
                n = cfg->obj_nverts[f];
                for (int i = 0; i < n; ++i) {
                    T[OT_VERTS + 2 * i] = cfg->obj_verts[f][i][0];
                    T[OT_VERTS + 2 * i + 1] = cfg->obj_verts[f][i][1];
                }
                for (int i = 0; i < n; ++i) {
                    const V2 edge = v_sub(ot_v(T, i + 1 < n ? i + 1 : 0), ot_v(T, i));
                    const V2 nr = v_normalize(v_cross_vs(edge, 1.0f));
                    T[OT_NORMALS + 2 * i] = nr.x; T[OT_NORMALS + 2 * i + 1] = nr.y;
                }
            }
            T[OT_N] = (float)n;
            T[OT_RADIUS] = B2_POLYGON_RADIUS;
            polygon_mass(T, cfg->obj_density, mo, ce, io);
        }
        bm[body] += mo; bc[body] = v_add(bc[body], v_scale(mo, ce)); bi[body] += io;      // b2Body::ResetMassData
    }
    for (int m = 0; m < KB_MAX_OBJECTS; ++m) {
        float *B = p.obody[m];
        for (int k = 0; k < BT_WORDS; ++k) B[k] = 0.0f;
        V2 lc = mk2(0.0f, 0.0f);
        if (bm[m] > 0.0f) { B[BT_IM] = 1.0f / bm[m]; lc = v_scale(B[BT_IM], bc[m]); }
        const float io = bi[m] - bm[m] * v_dot(lc, lc);     // inertia about the centre of mass
        B[BT_II] = io > 0.0f ? 1.0f / io : 0.0f;
        B[BT_LCX] = lc.x; B[BT_LCY] = lc.y;
        B[BT_RADIUS] = B2_POLYGON_RADIUS;
    }
    for (int f = 0; f < nfix; ++f) {
        float *T = p.otab[f];
        const int body = ot_body(T);
        float *B = p.obody[body];
        T[OT_IM] = B[BT_IM]; T[OT_II] = B[BT_II];
        B[BT_KIND] = T[OT_KIND];
        if (ot_kind(T) == KB_SHAPE_CIRCLE) { T[OT_BOUND] = T[OT_RADIUS]; B[BT_RADIUS] = T[OT_RADIUS]; continue; }
        float far2 = 0.0f;                                // bounding radius about the centre of mass of the body
        for (int i = 0; i < ot_n(T); ++i) { const V2 q = v_sub(ot_v(T, i), mk2(B[BT_LCX], B[BT_LCY])); far2 = fmaxf(far2, v_dot(q, q)); }
        T[OT_BOUND] = sqrtf(far2) + B2_POLYGON_RADIUS;
    }
    p.mu_oo = sqrtf(cfg->obj_friction * cfg->obj_friction);
    p.mu_ow = sqrtf(cfg->obj_friction * cfg->wall_friction);
    p.nmc = mc_candidates(p.F);
    p.kl_obj = damp(cfg->obj_linear_damping);
    p.ka_obj = damp(cfg->obj_angular_damping);
    p.sense_s = 0; p.sense_r2 = 0.0f;
    if (cfg->sense_radius > 0.0f) {
        const float Rw = cfg->sense_radius * WORLD_SCALE;
        p.sense_s = sense_reach(Rw, p.inv_cell);
        p.sense_r2 = Rw * Rw;
    }
    p.solver_mode = cfg->solver_mode;
    p.toi_walls = cfg->toi_walls;
    if (p.N > BPT * 64 * MAX_WAVES) { delete s; return fail(KB_EINVAL, "kb_create: num_bots exceeds bots-per-thread x workgroup size of this build"); }
    const int LDS_CU = 160 * 1024;
    {   // Workgroup size (measured on MI355X, 64 ... 1024 kilobots): one kilobot per thread and a power-of-two wave count
        // (3 / 5 / 6 / 7-wave workgroups spread unevenly over the four SIMDs and lose a resident env), unless that costs
        // a resident env (LDS fit vs 16 waves per CU) while many lanes would idle.  kb_set_block_threads overrides.
        int T = 64;
        while (T < p.N && T < 64 * MAX_WAVES) T <<= 1;
        const int ldsT = lds_bytes_for(p, T, p.capL);
        const int fit = LDS_CU / ldsT;
        int resident = resident_envs(ldsT, T, KB_MIN_WAVES_PER_SIMD);
        if (p.M == 0 && resident_envs(ldsT, T, KB_COMPACT_WAVES_PER_SIMD) > resident) resident = resident_envs(ldsT, T, KB_COMPACT_WAVES_PER_SIMD);
        if (T > 64 && resident < fit && p.N * 100 < T * 85) T >>= 1;
        // 257 ... 512 kilobots without objects: two per thread in a 4-wave workgroup (measured at 512: 1.06e10 against 9.3e9 with
        // eight waves -- an 8-wave env takes a third of the CU's wave slots whatever its LDS image is)
        if (p.M == 0 && p.drive_mode != KB_DRIVE_MIXED && p.N > 256 && p.N <= 512) T = 256;
        // with objects, up to 128 kilobots run as one wave: that selects the spill-free 256-VGPR instantiation
        // (kb_step), measured + 6 ... 9 % at 100 kilobots and - 3 % at 128 against two-wave workgroups
        if (p.M > 0 && p.N <= BPT * 64) T = 64;
        if (p.drive_mode == KB_DRIVE_MIXED) T = p.N <= BPT * 64 ? 64 : 64 * MAX_WAVES;       // the two spill-free instantiations (256 VGPRs)
        s->threads = T;
    }
    if (p.M == 0 && p.drive_mode != KB_DRIVE_MIXED) {
        // sorted-bin image (16 B per staged contact): give up staging entries where that lets the CU hold one more env -- a
        // 1024-kilobot swarm down to the 688 of the fixed-size kernel (a settled swarm has ~ 0.55 contacts per kilobot; a
        // jammed one is staged in the global slice whatever the LDS holds, and what it needs is resident envs)
        // (smaller swarms: never below one contact per kilobot + 64 -- a settled lattice has ~ 0.5, a hexagonal cluster up to 3;
        //  the largest staging area that reaches the best residency is taken)
        const int lo = p.N > 512 ? ldsb::CAPL : (p.N + 64 > 128 ? (p.N + 64 + 7) & ~7 : 128);
        auto res = [&](int c) {
            const int l = lds_bytes_for(p, s->threads, c);
            const int a = resident_envs(l, s->threads, KB_MIN_WAVES_PER_SIMD), b = resident_envs(l, s->threads, KB_COMPACT_WAVES_PER_SIMD);
            return a > b ? a : b;
        };
        int best = res(p.capL), bestc = p.capL;
        for (int c = p.capL - 8; c >= lo; c -= 8)
            if (res(c) > best) { best = res(c); bestc = c; }
        p.capL = bestc;
    } else {   // trade a few staging entries for one more env per CU when the LDS footprint is just above a divisor of 160 KiB
        const int fit = LDS_CU / lds_bytes_for(p, s->threads, p.capL);
        const int lo = 5 * p.N / 2 + 64 > 256 ? 5 * p.N / 2 + 64 : 256;
        int c = p.capL;
        while (c - 8 >= lo && lds_bytes_for(p, s->threads, c) > LDS_CU / (fit + 1)) c -= 8;
        if (fit >= 1 && lds_bytes_for(p, s->threads, c) <= LDS_CU / (fit + 1)) p.capL = c;
    }
    s->capL_regular = p.capL;
    if (p.M == 0 && uses_fixed_1024(p, s->threads)) p.capL = ldsb::CAPL;
    p.lds_total = lds_bytes_for(p, s->threads, p.capL);
    p.islmin_off = bins_islmin_offset(p, s->threads, p.capL, nullptr);
    p.botlaw_off = (lds_image_bytes(p, s->threads, p.capL) + 15) & ~15;
    s->tier = pick_tier(p, s->threads, p.lds_total);
    if (p.lds_total > LDS_CU) {
        delete s;
        return fail(KB_ELDS, "kb_create: configuration needs more than 160 KiB of LDS per env");
    }
    *out = s;
    return KB_OK;
}

void kb_destroy(kb_sim *sim) { delete sim; }

int kb_bind(kb_sim *sim, const kb_buffers *b) {
    if (!sim || !b) return fail(KB_EINVAL, "kb_bind: NULL argument");
    if (!b->x || !b->y || !b->theta || !b->ws_key || !b->ws_acc || !b->ws_cnt || !b->status || !b->scratch)
        return fail(KB_ENOTBOUND, "kb_bind: x, y, theta, ws_key, ws_acc, ws_cnt, status and scratch are required");
    const int m = sim->cfg.drive_mode;
    if (m == KB_DRIVE_MIXED && (!b->bot_mode || !b->v || !b->w || !b->acc_v || !b->acc_w || !b->motor_l || !b->motor_r ||
                                !b->pt_threshold || !b->pt_update || !b->pt_nochange || !b->pt_dir))
        return fail(KB_ENOTBOUND, "kb_bind: KB_DRIVE_MIXED needs bot_mode and the state buffers of every drive law");
    if ((m == KB_DRIVE_VELOCITY || m == KB_DRIVE_ACCEL) && (!b->v || !b->w))
        return fail(KB_ENOTBOUND, "kb_bind: v and w are required in the velocity / acceleration modes");
    if (m == KB_DRIVE_ACCEL && (!b->acc_v || !b->acc_w)) return fail(KB_ENOTBOUND, "kb_bind: acc_v, acc_w required");
    if ((m == KB_DRIVE_MOTORS || m == KB_DRIVE_PHOTOTAXIS) && (!b->motor_l || !b->motor_r))
        return fail(KB_ENOTBOUND, "kb_bind: motor_l, motor_r required");
    if (m == KB_DRIVE_PHOTOTAXIS && (!b->pt_threshold || !b->pt_update || !b->pt_nochange || !b->pt_dir))
        return fail(KB_ENOTBOUND, "kb_bind: pt_* buffers required in the phototaxis mode");
    if (sim->cfg.num_objects > 0 && (!b->ox || !b->oy || !b->otheta || !b->ovx || !b->ovy || !b->ow || !b->ows_acc))
        return fail(KB_ENOTBOUND, "kb_bind: ox, oy, otheta, ovx, ovy, ow and ows_acc are required when num_objects > 0");
    if (sim->cfg.light_type != KB_LIGHT_NONE && (!b->light_x || (!b->light_y && sim->cfg.light_type != KB_LIGHT_GRADIENT)))
        return fail(KB_ENOTBOUND, "kb_bind: light_x, light_y required when a light is configured");
    {
        bool momentum = sim->cfg.light_type == KB_LIGHT_MOMENTUM;
        if (sim->cfg.light_type == KB_LIGHT_COMPOSITE)
            for (int i = 0; i < sim->cfg.light_count; ++i) momentum |= sim->cfg.light_kind[i] == KB_LIGHT_MOMENTUM;
        if (momentum && (!b->light_vx || !b->light_vy)) return fail(KB_ENOTBOUND, "kb_bind: light_vx, light_vy required by a MomentumLight");
    }
    if ((b->light_value != nullptr) != (b->light_gx != nullptr) || (b->light_value != nullptr) != (b->light_gy != nullptr))
        return fail(KB_EINVAL, "kb_bind: light_value, light_gx, light_gy must be given together");
    if ((b->cmd_vx != nullptr) != (b->cmd_vy != nullptr) || (b->cmd_vx != nullptr) != (b->cmd_w != nullptr))
        return fail(KB_EINVAL, "kb_bind: cmd_vx, cmd_vy, cmd_w must be given together");
    if (sim->cfg.sense_radius > 0.0f && !b->nbr_count) return fail(KB_ENOTBOUND, "kb_bind: nbr_count is required when sense_radius > 0");
    if (sim->cfg.allow_sleep && (!b->sleep_time || (sim->cfg.num_objects > 0 && !b->osleep)))
        return fail(KB_ENOTBOUND, "kb_bind: sleep_time (and osleep with objects) are required when allow_sleep is set");
    sim->p.buf = *b;
    sim->bound = true;
    return KB_OK;
}

int kb_set_actions(kb_sim *sim, const float *d_actions, void *stream) {
    if (!sim) return fail(KB_EINVAL, "kb_set_actions: NULL handle");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_set_actions: kb_bind() first");
    if (sim->cfg.drive_mode != KB_DRIVE_VELOCITY && sim->cfg.drive_mode != KB_DRIVE_ACCEL && sim->cfg.drive_mode != KB_DRIVE_MIXED)
        return fail(KB_EINVAL, "kb_set_actions: only the velocity / acceleration drive modes take actions");
    Params p = sim->p;
    p.actions = d_actions;
    const size_t T = (size_t)p.E * p.N;
    hipLaunchKernelGGL(kb_set_actions_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_set_actions: %s", hipGetErrorString(err));
    return KB_OK;
}

// the kernel instantiation of a handle (drive law, light model, objects, workgroup size)
static kb_step_fn select_kernel(const kb_sim *sim, const kb::Params &p) {
    const bool obj = p.M > 0;
    int objsel = obj ? (sim->threads <= 64 ? 2 : 1) : (sim->tier == 2 ? 3 : 0);     // one-wave workgroups with objects: the 256-VGPR instantiation; no objects: 128 or 80 VGPRs
    // scenes whose objects are all discs: instantiations without the kilobot - polygon contact code (5 / 6)
    bool discs = obj;
    for (int f = 0; f < p.F; ++f) discs = discs && ot_kind(p.otab[f]) == KB_SHAPE_CIRCLE;
    if (discs) objsel += 4;
    if (p.allow_sleep) objsel |= KB_PICK_SLEEP;      // instantiations with the sleep state
    switch (p.drive_mode) {
    case KB_DRIVE_VELOCITY: {
        // the flagship size has its own instantiation with a compile-time LDS layout
        const bool fixed = uses_fixed_1024(p, sim->threads);
        if (fixed && !obj && p.capL != ldsb::CAPL) return nullptr;      // (cannot happen: kb_create / kb_set_block_threads keep them in step)
        return kb_pick_velocity(fixed ? (p.sense_s > 0 ? KB_PICK_FIXED_1024_SENSE : KB_PICK_FIXED_1024) : p.light_type,
                                fixed ? ((discs ? 5 : (int)obj) | (p.allow_sleep ? KB_PICK_SLEEP : 0)) : objsel);
    }
    case KB_DRIVE_ACCEL: return kb_pick_accel(p.light_type, objsel);
    case KB_DRIVE_MOTORS: return kb_pick_motors(p.light_type, objsel);
    case KB_DRIVE_SIMPLE_PHOTOTAXIS: return kb_pick_simple_phototaxis(p.light_type, objsel);
    case KB_DRIVE_PHOTOTAXIS: return kb_pick_phototaxis(p.light_type, objsel);
    case KB_DRIVE_MIXED: return sim->threads == 64 ? kb_pick_mixed(p.light_type, p.allow_sleep)
                                                   : (sim->threads == 64 * MAX_WAVES ? kb_pick_mixed_large(p.light_type, p.allow_sleep) : nullptr);
    default: return nullptr;
    }
}

int kb_resident_envs_per_cu(kb_sim *sim) {
    if (!sim) return fail(KB_EINVAL, "kb_resident_envs_per_cu: NULL handle");
    kb_step_fn fn = select_kernel(sim, sim->p);
    if (!fn) return fail(KB_EINVAL, "kb_resident_envs_per_cu: no kernel for this configuration");
    if (sim->p.lds_total > 64 * 1024) {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e2 != hipSuccess) return fail(KB_EHIP, "kb_resident_envs_per_cu: hipFuncSetAttribute: %s", hipGetErrorString(e2));
    }
    int n = 0;
    hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(fn), sim->threads, (size_t)sim->p.lds_total);
    if (err != hipSuccess) return fail(KB_EHIP, "kb_resident_envs_per_cu: %s", hipGetErrorString(err));
    return n;
}

int kb_step(kb_sim *sim, const float *d_actions, const float *d_light_action, int n_substeps, int flags, void *stream) {
    if (!sim) return fail(KB_EINVAL, "kb_step: NULL handle");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_step: kb_bind() first");
    if (n_substeps < 0) return fail(KB_EINVAL, "kb_step: n_substeps < 0");
    if (d_actions && sim->cfg.drive_mode != KB_DRIVE_VELOCITY && sim->cfg.drive_mode != KB_DRIVE_ACCEL && sim->cfg.drive_mode != KB_DRIVE_MIXED)
        return fail(KB_EINVAL, "kb_step: only the velocity / acceleration drive modes take actions");
    if (n_substeps == 0 && !d_actions) return KB_OK;
    Params p = sim->p;
    p.actions = d_actions;
    p.light_action = d_light_action;
    p.n_substeps = n_substeps;
    p.flags = flags;
    kb_step_fn fn = select_kernel(sim, p);
    if (!fn) return fail(KB_EINVAL, "kb_step: no kernel for this drive mode / light type");
    if (p.lds_total > 64 * 1024 && sim->attr_fn != reinterpret_cast<const void *>(fn)) {
        // the attribute belongs to the kernel, not to this sim: raise it to the hardware limit, so that sims of
        // different sizes sharing one instantiation never lower it under each other
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e2 != hipSuccess) return fail(KB_EHIP, "kb_step: hipFuncSetAttribute: %s", hipGetErrorString(e2));
        sim->attr_fn = reinterpret_cast<const void *>(fn);
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)p.E), dim3((unsigned)sim->threads), (size_t)p.lds_total,
                       (hipStream_t)stream, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_step: %s", hipGetErrorString(err));
    return KB_OK;
}

int kb_sense(kb_sim *sim, float radius_m, uint32_t *d_count, void *stream) {
    if (!sim || !d_count) return fail(KB_EINVAL, "kb_sense: NULL argument");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_sense: kb_bind() first");
    if (!(radius_m > 0.0f)) return fail(KB_EINVAL, "kb_sense: radius must be positive");
    const Params &p = sim->p;
    const float Rw = radius_m * WORLD_SCALE;
    const size_t lds = (size_t)14 * p.NP + 2 * (size_t)p.ncell + 16;
    hipLaunchKernelGGL(kb_sense_kernel, dim3((unsigned)p.E), dim3(256), lds, (hipStream_t)stream, p, sense_reach(Rw, p.inv_cell), Rw * Rw, d_count);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_sense: %s", hipGetErrorString(err));
    return KB_OK;
}

int kb_light_sense(kb_sim *sim, const float *d_light_action, void *stream) {
    if (!sim) return fail(KB_EINVAL, "kb_light_sense: NULL handle");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_light_sense: kb_bind() first");
    if (sim->cfg.light_type == KB_LIGHT_NONE) return fail(KB_EINVAL, "kb_light_sense: the handle has no light");
    const Params &p = sim->p;
    if (!p.buf.light_value || !p.buf.light_gx || !p.buf.light_gy) return fail(KB_EINVAL, "kb_light_sense: kb_buffers.light_value / light_gx / light_gy are not bound");
    hipLaunchKernelGGL(kb_light_sense_kernel, dim3((unsigned)p.E), dim3(256), 0, (hipStream_t)stream, p, d_light_action);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_light_sense: %s", hipGetErrorString(err));
    return KB_OK;
}

int kb_reset(kb_sim *sim, const kb_reset_params *rp, void *stream) {
    if (!sim || !rp) return fail(KB_EINVAL, "kb_reset: NULL argument");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_reset: kb_bind() first");
    if (!(rp->std >= 0.0f)) return fail(KB_EINVAL, "kb_reset: std must be >= 0");
    const Params &p = sim->p;
    ResetArgs a;
    a.k0 = (unsigned)(rp->seed & 0xFFFFFFFFull); a.k1 = (unsigned)(rp->seed >> 32);
    a.env_offset = rp->env_offset; a.random_theta = rp->random_theta; a.random_velocity = rp->random_velocity;
    a.mean_x = rp->mean[0]; a.mean_y = rp->mean[1]; a.std = rp->std;
    // world_bounds -/+ 0.02 (yaml_kilobots_env.py:350-351), metres
    a.lo_x = -0.5f * sim->cfg.world_width + 0.02f; a.hi_x = 0.5f * sim->cfg.world_width - 0.02f;
    a.lo_y = -0.5f * sim->cfg.world_height + 0.02f; a.hi_y = 0.5f * sim->cfg.world_height - 0.02f;
    const size_t T = (size_t)p.E * p.N;
    hipLaunchKernelGGL(kb_reset_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, a);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_reset: %s", hipGetErrorString(err));
    if (rp->resolve) return kb_step(sim, nullptr, nullptr, 1, KB_STEP_NO_DRIVE, stream);
    return KB_OK;
}

int kb_get_poses(kb_sim *sim, float *d_out, void *stream) {
    if (!sim || !d_out) return fail(KB_EINVAL, "kb_get_poses: NULL argument");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_get_poses: kb_bind() first");
    const size_t T = (size_t)sim->p.E * sim->p.N;
    hipLaunchKernelGGL(kb_get_poses_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sim->p.buf.x, sim->p.buf.y, sim->p.buf.theta, d_out, T);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_get_poses: %s", hipGetErrorString(err));
    return KB_OK;
}

int kb_get_state(kb_sim *sim, float *d_out, void *stream) {
    if (!sim || !d_out) return fail(KB_EINVAL, "kb_get_state: NULL argument");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_get_state: kb_bind() first");
    const size_t T = (size_t)sim->p.E * (size_t)(sim->p.N + sim->p.M + 1);
    hipLaunchKernelGGL(kb_get_state_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sim->p, d_out);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_get_state: %s", hipGetErrorString(err));
    return KB_OK;
}

int kb_lds_bytes(const kb_sim *sim) { return sim ? sim->p.lds_total : KB_EINVAL; }
int kb_light_action_dim(const kb_sim *sim) { return sim ? sim->p.ladim : KB_EINVAL; }
int kb_light_count(const kb_sim *sim) { return sim ? (sim->cfg.light_type == KB_LIGHT_NONE ? 0 : sim->p.lcount) : KB_EINVAL; }
// per contact: the 16-byte staging record + the 16-byte level-sorted record (pair, normal, impulse) of the cooperative sweeps (kernels with objects: normal and effective mass, 12 B)
size_t kb_scratch_bytes(const kb_sim *sim) { return sim ? (size_t)sim->p.E * (size_t)sim->p.cap * 32u : 0; }
int kb_contact_capacity(const kb_sim *sim) { return sim ? sim->p.cap : KB_EINVAL; }
int kb_lds_staging_entries(const kb_sim *sim) { return sim ? sim->p.capL : KB_EINVAL; }
int kb_block_threads(const kb_sim *sim) { return sim ? sim->threads : KB_EINVAL; }
int kb_set_block_threads(kb_sim *sim, int threads) {
    if (!sim || threads < 64 || threads > 64 * MAX_WAVES || (threads & 63)) return fail(KB_EINVAL, "kb_set_block_threads: multiple of 64 up to the build maximum");
    if (sim->p.N > BPT * threads) return fail(KB_EINVAL, "kb_set_block_threads: need num_bots <= bots-per-thread x threads");
    const int capL = (sim->p.M == 0 && uses_fixed_1024(sim->p, threads)) ? ldsb::CAPL : sim->capL_regular;
    const int need = lds_bytes_for(sim->p, threads, capL);
    if (need > 160 * 1024) return fail(KB_ELDS, "kb_set_block_threads: more than 160 KiB of LDS per env at this workgroup size");
    if (sim->p.drive_mode == KB_DRIVE_MIXED && threads != 64 && threads != 64 * MAX_WAVES)
        return fail(KB_EINVAL, "kb_set_block_threads: KB_DRIVE_MIXED runs as one-wave or full workgroups");
    sim->threads = threads;
    sim->p.capL = capL;
    sim->p.lds_total = need;
    sim->p.islmin_off = bins_islmin_offset(sim->p, threads, capL, nullptr);
    sim->p.botlaw_off = (lds_image_bytes(sim->p, threads, capL) + 15) & ~15;
    sim->tier = pick_tier(sim->p, threads, need);
    return KB_OK;
}

}  // extern "C"
