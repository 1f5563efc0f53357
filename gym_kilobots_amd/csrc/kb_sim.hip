// kb_sim.hip -- batched Kilobot world step for MI355X (gfx950 / CDNA4) + its C ABI.
//
// One workgroup owns one env for the whole launch: positions are loaded once from HBM into LDS,
// `n_substeps` iterations of the reference substep loop
// (gym_kilobots/envs/kilobots_env.py:168-190) run out of LDS / registers, poses are written back once.
// Per substep:
//   drive law (kilobot.py:86-127,191-203,253-258,294-300,318-333) + light (light.py:59-75,176-189)
//   -> broadphase: uniform grid of per-cell linked lists in LDS (one atomic exchange per bot)
//   -> narrowphase: circle-circle / circle-wall (Box2D b2CollideCircles, b2CollideEdgeAndCircle),
//      5-cell half stencil, warm-start impulses matched from the previous substep
//   -> islands: lock-free union-find in LDS
//   -> solver (b2ContactSolver semantics): warm start + 10 sequential-impulse velocity sweeps,
//      symplectic Euler, <= 10 position sweeps with Box2D's per-island early out.
// Gauss-Seidel order.  Every contact gets a key (class, rank): class from the relative grid position
// of the two bodies and the parity of the base cell, rank from its position inside its cell-pair
// group.  Two contacts with the same key never share a body, so all contacts of one key can be
// solved concurrently and the result equals the sequential sweep in (class, group, A, B) order that
// DESIGN.md specifies.  Islands are independent, so each island is bound to ONE wavefront
// (root id mod #waves): a wave walks its own contacts key by key with no workgroup barrier at all
// (LDS operations of one wave execute in order).  Only when one island is very large does the whole
// workgroup cooperate on the sweep with s_barrier between keys.
// No MFMA anywhere: this is LDS/latency- and HBM-bound integer/float work.
//
// Arithmetic: fp32, compiled with -ffp-contract=off; every expression is written in the operation
// order of the specification so results do not depend on launch fusion, workgroup size or sharding.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "kilobots_hip.h"

namespace {

// ---- Box2D 2.3.1 constants (b2Settings.h) ---------------------------------------------------
constexpr float B2_PI = 3.14159265359f;
constexpr float B2_LINEAR_SLOP = 0.005f;
constexpr float B2_POLYGON_RADIUS = 2.0f * B2_LINEAR_SLOP;
constexpr float B2_BAUMGARTE = 0.2f;
constexpr float B2_MAX_LINEAR_CORRECTION = 0.2f;
constexpr float B2_MAX_TRANSLATION = 2.0f;
constexpr float B2_MAX_TRANSLATION_SQ = B2_MAX_TRANSLATION * B2_MAX_TRANSLATION;
constexpr float B2_MAX_ROTATION = 0.5f * B2_PI;
constexpr float B2_MAX_ROTATION_SQ = B2_MAX_ROTATION * B2_MAX_ROTATION;
constexpr float B2_EPSILON = 1.19209290e-07f;
constexpr float WORLD_SCALE = 25.0f;  // body.py:7

constexpr float CELL_SIZE = 0.875f;   // world units, >= 2 * bot radius
constexpr int MAX_CELLS = 8192;

constexpr unsigned KEY_WALL = 0x10000u;
constexpr int WALL_CODE = 0xFFF0;     // ca[] value of wall w is WALL_CODE + w
constexpr unsigned EMPTY32 = 0xFFFFFFFFu;
constexpr unsigned short EMPTY16 = 0xFFFFu;

// contact classes in canonical order; +1 on E/N/NE/NW for odd base-cell parity
constexpr int CLS_SAME = 0, CLS_E = 1, CLS_N = 3, CLS_NE = 5, CLS_NW = 7, CLS_WALL = 9, NUM_CLS = 10;
constexpr int RK = 4;                 // rank buckets per class; the last one holds every rank >= RK-1
constexpr int MAX_WAVES = 4;       // workgroups are at most 256 threads
constexpr int BK_PER_WAVE = NUM_CLS * RK;
constexpr int MAX_BUCKETS = MAX_WAVES * BK_PER_WAVE;
constexpr int BPT = 4;                // bots per thread (max): N <= BPT * blockDim.x
constexpr int GIANT_ISLAND = 256;     // contacts; larger islands are swept by the whole workgroup

enum { M_NCON = 0, M_TOTAL = 1, M_ANY = 2, M_STATUS = 3, M_MAXISL = 4, M_COUNT = 8 };

struct Layout {  // byte offsets into dynamic LDS
    int px, py, vx, vy;
    int head, dirCnt, parent, bkStart, bkFill, bkMaxRank, misc, wsum;
    int cacc;
    int ca, cb, cbk, order, next, cellOf, bkList;
    int ccls, crank, cslot, wsCnt, wsCntNew, active, nList;
    int total;
};

struct Params {
    kb_buffers buf;
    const float *actions;
    const float *light_action;
    int N, E, S, gw, gh, ncell, cap, n_substeps, flags, drive_mode, light_type, vel_iters, pos_iters;
    float xmin, ymin, xmax, ymax, inv_cell, r_bot, im_bot, kl_bot, ka_bot, h;
    float light_radius, light_lo[2], light_hi[2], act_lo[2], act_hi[2];
    Layout L;
};

Layout make_layout(int N, int ncell, int cap) {
    Layout L;
    int o = 0;
    auto take = [&](int bytes) { int r = o; o += (bytes + 15) & ~15; return r; };
    L.px = take(4 * N); L.py = take(4 * N); L.vx = take(4 * N); L.vy = take(4 * N);
    L.head = take(4 * ncell); L.dirCnt = take(4 * N); L.parent = take(4 * N);
    L.bkStart = take(4 * (MAX_BUCKETS + 1)); L.bkFill = take(4 * MAX_BUCKETS);
    L.bkMaxRank = take(4 * MAX_WAVES * NUM_CLS);
    L.misc = take(4 * M_COUNT); L.wsum = take(4 * 16);
    L.cacc = take(4 * cap);
    L.ca = take(2 * cap); L.cb = take(2 * cap); L.cbk = take(2 * cap); L.order = take(2 * cap);
    L.next = take(2 * N); L.cellOf = take(2 * N); L.bkList = take(2 * MAX_BUCKETS);
    L.ccls = take(cap); L.crank = take(cap); L.cslot = take(cap);
    L.wsCnt = take(N); L.wsCntNew = take(N); L.active = take(2 * N); L.nList = take(MAX_WAVES);
    L.total = o;
    return L;
}

// ---- device math ----------------------------------------------------------------------------
// sin/cos: Cephes single-precision algorithm (argument reduction by pi/4 in three parts, degree-3
// minimax polynomials in x^2).  Own implementation so that results are identical wherever the
// same specification is evaluated in IEEE fp32.
__device__ __forceinline__ void kb_sincosf(float xx, float &sn, float &cs) {
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
    if (j == 1 || j == 2) { sn = ssign * pc; cs = csign * ps; }
    else { sn = ssign * ps; cs = csign * pc; }
}

// CircularGradientLight.value_and_gradients, light.py:176-189 (zero gradient instead of NaN at distance 0)
__device__ __forceinline__ void kb_light_circular(float sx, float sy, float lx, float ly, float R,
                                                  float &val, float &gx, float &gy) {
    float dx = -1.0f * (sx - lx), dy = -1.0f * (sy - ly);
    float n = sqrtf(dx * dx + dy * dy);
    float v = 1.0f - n / R;
    v = fmaxf(fminf(v, 1.0f), 0.0f);
    val = v * 255.0f;
    if (n > 0.0f) { dx = dx / n; dy = dy / n; }
    else { dx = 0.0f; dy = 0.0f; }
    if (n > R) { dx *= 0.0f; dy *= 0.0f; }
    gx = dx; gy = dy;
}

// Kilobot.step motor law, kilobot.py:86-127; body velocity in world units
__device__ __forceinline__ void kb_motor_law(int ml, int mr, float th, float h, float &vx, float &vy, float &w) {
    const float max_lin = 0.01f, max_ang = 0.5f * 3.14159265358979323846f;
    float s, c;
    kb_sincosf(th, s, c);
    if (ml && mr) {  // kilobot.py:97-101 (intended meaning; the reference raises TypeError at :127)
        float lin = (float)(mr + ml) / 510.0f * max_lin;
        vx = (s * lin) * WORLD_SCALE; vy = (c * lin) * WORLD_SCALE;
        w = (float)(mr - ml) / 510.0f * max_ang;
    } else if (mr || ml) {  // kilobot.py:103-121: pivot about the opposite leg
        float av, lx, ly = -0.009f;
        if (mr) { av = (float)mr / 255.0f * max_ang; lx = -0.013f; }
        else { av = -(float)ml / 255.0f * max_ang; lx = 0.013f; }
        float ds, dc;
        kb_sincosf(av * h, ds, dc);
        float tx = lx - (dc * lx - ds * ly), ty = ly - (ds * lx + dc * ly);
        tx *= WORLD_SCALE; ty *= WORLD_SCALE;
        float wx = c * tx - s * ty, wy = s * tx + c * ty;  // b2Body::GetWorldVector
        wx = wx / WORLD_SCALE / h; wy = wy / WORLD_SCALE / h;
        vx = wx * WORLD_SCALE; vy = wy * WORLD_SCALE; w = av;
    } else {
        vx = 0.0f; vy = 0.0f; w = 0.0f;
    }
}

__device__ __forceinline__ float kb_clampf(float a, float lo, float hi) { return fmaxf(lo, fminf(a, hi)); }

// LDS operations of one wave execute in program order; this only stops the compiler from moving
// LDS accesses across the point where other lanes' results are consumed.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void wall_geom(const Params &p, int wl, float x, float y, float &dist, float &nx, float &ny) {
    switch (wl) {
    case 0: nx = 1.0f; ny = 0.0f; dist = x - p.xmin; break;
    case 1: nx = 0.0f; ny = 1.0f; dist = y - p.ymin; break;
    case 2: nx = -1.0f; ny = 0.0f; dist = p.xmax - x; break;
    default: nx = 0.0f; ny = -1.0f; dist = p.ymax - y; break;
    }
}

#ifdef KB_PROFILE
// diagnostic build: wave 0 / lane 0 accumulates shader cycles per phase into g.status[E + 8*e + phase]
#define KB_STAMP(ph) do { if (tid == 0) { long long t_ = clock64(); prof_acc[ph] += t_ - prof_t; prof_t = t_; } } while (0)
#else
#define KB_STAMP(ph) do { } while (0)
#endif

__device__ __forceinline__ int dir_dx(int k) { return (k == 1 || k == 3) ? 1 : (k == 4 ? -1 : 0); }
__device__ __forceinline__ int dir_dy(int k) { return (k >= 2) ? 1 : 0; }

__global__ void __launch_bounds__(256, 2) kb_step_kernel(const Params p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int e = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int N = p.N, S = p.S;
    const size_t o = (size_t)e * N;
    const float h = p.h;

    float *px = (float *)(smem + p.L.px), *py = (float *)(smem + p.L.py);
    float *vx = (float *)(smem + p.L.vx), *vy = (float *)(smem + p.L.vy);
    unsigned *head = (unsigned *)(smem + p.L.head), *dirCnt = (unsigned *)(smem + p.L.dirCnt);
    unsigned *islCnt = dirCnt;  // alias: dirCnt is dead once the contacts are emitted
    unsigned *parent = (unsigned *)(smem + p.L.parent);
    unsigned *bkStart = (unsigned *)(smem + p.L.bkStart), *bkFill = (unsigned *)(smem + p.L.bkFill);
    unsigned *bkMaxRank = (unsigned *)(smem + p.L.bkMaxRank);
    unsigned *misc = (unsigned *)(smem + p.L.misc);
    float *cacc = (float *)(smem + p.L.cacc);
    unsigned short *ca = (unsigned short *)(smem + p.L.ca), *cb = (unsigned short *)(smem + p.L.cb);
    unsigned short *cbk = (unsigned short *)(smem + p.L.cbk), *order = (unsigned short *)(smem + p.L.order);
    unsigned short *nextb = (unsigned short *)(smem + p.L.next), *cellOf = (unsigned short *)(smem + p.L.cellOf);
    unsigned short *bkList = (unsigned short *)(smem + p.L.bkList);
    unsigned char *ccls = smem + p.L.ccls, *crank = smem + p.L.crank, *cslot = smem + p.L.cslot;
    unsigned char *wsCnt = smem + p.L.wsCnt, *wsCntNew = smem + p.L.wsCntNew;
    unsigned char *active = smem + p.L.active, *nList = smem + p.L.nList;

    const kb_buffers &g = p.buf;
#ifdef KB_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
#endif

    // ---- load state; optional fused set_action (kilobot.py:235-241, 283-289) ----
    float th[BPT], bw[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = tid + q * nt;
        th[q] = 0.0f; bw[q] = 0.0f;
        if (b < N) {
            px[b] = g.x[o + b]; py[b] = g.y[o + b]; th[q] = g.theta[o + b];
            if (p.actions) {
                const float2 a = reinterpret_cast<const float2 *>(p.actions)[o + b];
                const float mw = 0.5f * 3.14159265358979323846f;
                if (p.drive_mode == KB_DRIVE_VELOCITY) {
                    g.v[o + b] = fmaxf(fminf(a.x, 0.01f), 0.0f);
                    g.w[o + b] = fmaxf(fminf(a.y, mw), -mw);
                } else if (p.drive_mode == KB_DRIVE_ACCEL) {
                    const float aw = 0.2f * 3.14159265358979323846f;
                    g.acc_v[o + b] = fmaxf(fminf(a.x, 0.005f), -0.005f);
                    g.acc_w[o + b] = fmaxf(fminf(a.y, aw), -aw);
                }
            }
        }
    }
    for (int c = tid; c < p.ncell; c += nt) head[c] = EMPTY32;
    if (tid == 0) misc[M_STATUS] = 0;
    float lx = 0.0f, ly = 0.0f;
    if (p.light_type == KB_LIGHT_CIRCULAR) { lx = g.light_x[e]; ly = g.light_y[e]; }
    const bool drive = !(p.flags & KB_STEP_NO_DRIVE);
    const float rr = p.r_bot + p.r_bot, rr2 = rr * rr;
    const float rw = B2_POLYGON_RADIUS + p.r_bot, rw2 = rw * rw;
    __syncthreads();

    for (int sub = 0; sub < p.n_substeps; ++sub) {
        // ---- light.step: SinglePositionLight.step, light.py:59-75 (uniform per env) ----
        if (p.light_action && p.light_type == KB_LIGHT_CIRCULAR && drive) {
            float ax = fminf(fmaxf(p.light_action[2 * e + 0], p.act_lo[0]), p.act_hi[0]);
            float ay = fminf(fmaxf(p.light_action[2 * e + 1], p.act_lo[1]), p.act_hi[1]);
            float nlx = lx + ax * h, nly = ly + ay * h;
            lx = fminf(fmaxf(nlx, p.light_lo[0]), p.light_hi[0]);
            ly = fminf(fmaxf(nly, p.light_lo[1]), p.light_hi[1]);
        }
        // ---- sensing + drive law + damping; grid insertion; reset per-substep scratch ----
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const int b = tid + q * nt;
            if (b >= N) continue;
            float bvx = 0.0f, bvy = 0.0f, bww = 0.0f;
            const float bx = px[b], by = py[b];
            if (drive) {
                const float t = th[q];
                float lval = 0.0f, lgx = 0.0f, lgy = 0.0f;
                if (p.light_type == KB_LIGHT_CIRCULAR) {
                    float sx = bx, sy = by;
                    if (p.drive_mode != KB_DRIVE_SIMPLE_PHOTOTAXIS) {  // kilobot.py:54-55: world point of (0, -r)
                        float s, c;
                        kb_sincosf(t, s, c);
                        const float lx0 = 0.0f, ly0 = -p.r_bot;
                        sx = (c * lx0 - s * ly0) + bx;
                        sy = (s * lx0 + c * ly0) + by;
                    }
                    kb_light_circular(sx / WORLD_SCALE, sy / WORLD_SCALE, lx, ly, p.light_radius, lval, lgx, lgy);
                    if (g.light_value) { g.light_value[o + b] = lval; g.light_gx[o + b] = lgx; g.light_gy[o + b] = lgy; }
                }
                switch (p.drive_mode) {
                case KB_DRIVE_ACCEL: {  // kilobot.py:294-300
                    float v = g.v[o + b] + g.acc_v[o + b] * h, ww = g.w[o + b] + g.acc_w[o + b] * h;
                    const float mw = 0.5f * 3.14159265358979323846f;
                    v = fminf(fmaxf(v, 0.0f), 0.01f); ww = fminf(fmaxf(ww, -mw), mw);
                    g.v[o + b] = v; g.w[o + b] = ww;
                    float s, c;
                    kb_sincosf(t, s, c);
                    float sp = v * WORLD_SCALE;
                    bvx = c * sp; bvy = s * sp; bww = ww;
                } break;
                case KB_DRIVE_VELOCITY: {  // kilobot.py:253-258
                    float s, c;
                    kb_sincosf(t, s, c);
                    float sp = g.v[o + b] * WORLD_SCALE;
                    bvx = c * sp; bvy = s * sp; bww = g.w[o + b];
                } break;
                case KB_DRIVE_PHOTOTAXIS: {  // kilobot.py:318-333
                    int upd = g.pt_update[o + b];
                    if (upd % 6 == 0) {
                        float meas = lval;
                        if (meas > g.pt_threshold[o + b] || g.pt_nochange[o + b] >= 15) {
                            g.pt_threshold[o + b] = meas + 0.01f;
                            if (g.pt_dir[o + b] == 0) { g.pt_dir[o + b] = 1; g.motor_l[o + b] = 0; g.motor_r[o + b] = 255; }
                            else { g.pt_dir[o + b] = 0; g.motor_l[o + b] = 255; g.motor_r[o + b] = 0; }
                            g.pt_nochange[o + b] = 0;
                        } else g.pt_nochange[o + b] += 1;
                    }
                    g.pt_update[o + b] = upd + 1;
                    kb_motor_law(g.motor_l[o + b], g.motor_r[o + b], t, h, bvx, bvy, bww);
                } break;
                case KB_DRIVE_MOTORS:
                    kb_motor_law(g.motor_l[o + b], g.motor_r[o + b], t, h, bvx, bvy, bww);
                    break;
                case KB_DRIVE_SIMPLE_PHOTOTAXIS: {  // kilobot.py:191-203
                    float n = sqrtf(lgx * lgx + lgy * lgy);
                    float mx = lgx, my = lgy;
                    if (n > 0.01f) { mx = lgx / n * 0.01f; my = lgy / n * 0.01f; }
                    bvx = mx * WORLD_SCALE; bvy = my * WORLD_SCALE; bww = 0.0f;
                } break;
                default: break;
                }
            }
            if (g.cmd_vx) { g.cmd_vx[o + b] = bvx; g.cmd_vy[o + b] = bvy; g.cmd_w[o + b] = bww; }
            // b2Island::Solve: v *= 1/(1 + h c)  (SimplePhototaxisKilobot sets linearDamping = 0, kilobot.py:203)
            const float kl = (p.drive_mode == KB_DRIVE_SIMPLE_PHOTOTAXIS) ? 1.0f / (1.0f + h * 0.0f) : p.kl_bot;
            vx[b] = bvx * kl; vy[b] = bvy * kl; bw[q] = bww * p.ka_bot;
            wsCnt[b] = g.ws_cnt[o + b];
            parent[b] = b;
            // broadphase: push the bot on its cell's list
            int cx = (int)floorf((bx - p.xmin) * p.inv_cell);
            int cy = (int)floorf((by - p.ymin) * p.inv_cell);
            cx = cx < 0 ? 0 : (cx >= p.gw ? p.gw - 1 : cx);
            cy = cy < 0 ? 0 : (cy >= p.gh ? p.gh - 1 : cy);
            const int cell = cy * p.gw + cx;
            cellOf[b] = (unsigned short)cell;
            nextb[b] = (unsigned short)atomicExch(&head[cell], (unsigned)b);
        }
        for (int k = tid; k < nw * BK_PER_WAVE; k += nt) { bkStart[k] = 0; bkFill[k] = 0; }
        for (int k = tid; k < nw * NUM_CLS; k += nt) bkMaxRank[k] = 0;
        if (tid == 0) { bkStart[nw * BK_PER_WAVE] = 0; misc[M_NCON] = 0; misc[M_TOTAL] = 0; misc[M_ANY] = 0; misc[M_MAXISL] = 0; }
        __syncthreads();
        KB_STAMP(0);

        // ---- narrowphase pass 1: per bot, number of contacts it owns per direction ----
#pragma unroll 1
        for (int a = tid; a < N; a += nt) {
            const int cell = cellOf[a];
            const int cx = cell % p.gw, cy = cell / p.gw;
            const float ax = px[a], ay = py[a];
            unsigned cnt = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int ox = cx + dir_dx(k), oy = cy + dir_dy(k);
                if (ox < 0 || ox >= p.gw || oy >= p.gh) continue;
                unsigned ck = 0;
                for (unsigned b = head[oy * p.gw + ox]; b != EMPTY32; b = (nextb[b] == EMPTY16 ? EMPTY32 : nextb[b])) {
                    if (k == 0 && (int)b <= a) continue;
                    const float dx = px[b] - ax, dy = py[b] - ay;
                    const float dd = dx * dx + dy * dy;
                    if (dd > rr2) continue;  // b2CollideCircles
                    ck++;
                }
                if (ck > 63u) { ck = 63u; atomicOr(&misc[M_STATUS], 4u); }
                cnt |= ck << (6 * k);
            }
            dirCnt[a] = cnt;
        }
        __syncthreads();
        KB_STAMP(1);

        // ---- narrowphase pass 2: emit contacts (class, rank), warm-start impulses, hook islands ----
#pragma unroll 1
        for (int a = tid; a < N; a += nt) {
            const int cell = cellOf[a];
            const int cx = cell % p.gw, cy = cell / p.gw;
            const float ax = px[a], ay = py[a];
            const unsigned mycnt = dirCnt[a];
            int nslot = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int nk = (int)((mycnt >> (6 * k)) & 63u);
                if (nk == 0) continue;
                const int oc = (cy + dir_dy(k)) * p.gw + (cx + dir_dx(k));
                int cls;
                if (k == 0) cls = CLS_SAME;
                else if (k == 1) cls = CLS_E + (cx & 1);
                else if (k == 2) cls = CLS_N + (cy & 1);
                else if (k == 3) cls = CLS_NE + (cx & 1);
                else cls = CLS_NW + (cx & 1);
                // rank base: contacts of this (cell, direction) group owned by lower-id bots of the cell
                int rbase = 0;
                for (unsigned a2 = head[cell]; a2 != EMPTY32; a2 = (nextb[a2] == EMPTY16 ? EMPTY32 : nextb[a2]))
                    if ((int)a2 < a) rbase += (int)((dirCnt[a2] >> (6 * k)) & 63u);
                for (unsigned b = head[oc]; b != EMPTY32; b = (nextb[b] == EMPTY16 ? EMPTY32 : nextb[b])) {
                    if (k == 0 && (int)b <= a) continue;
                    const float dx = px[b] - ax, dy = py[b] - ay;
                    const float dd = dx * dx + dy * dy;
                    if (dd > rr2) continue;
                    // position of b among a's touching partners of this direction, in ascending id order
                    int j = 0;
                    if (nk > 1) {
                        for (unsigned b2 = head[oc]; b2 != EMPTY32; b2 = (nextb[b2] == EMPTY16 ? EMPTY32 : nextb[b2])) {
                            if (b2 >= b || (k == 0 && (int)b2 <= a)) continue;
                            const float ex = px[b2] - ax, ey = py[b2] - ay;
                            if (!(ex * ex + ey * ey > rr2)) j++;
                        }
                    }
                    // warm start: impulse of the same pair in the previous substep (b2Contact::Update id match)
                    float acc = 0.0f;
                    bool found = false;
                    for (int sl = 0; sl < (int)wsCnt[a] && sl < S; ++sl) {
                        const size_t idx = ((size_t)e * S + sl) * N + a;
                        if (g.ws_key[idx] == (unsigned)b) { acc = g.ws_acc[idx]; found = true; break; }
                    }
                    if (!found) {
                        for (int sl = 0; sl < (int)wsCnt[b] && sl < S; ++sl) {
                            const size_t idx = ((size_t)e * S + sl) * N + b;
                            if (g.ws_key[idx] == (unsigned)a) { acc = g.ws_acc[idx]; break; }
                        }
                    }
                    int slot = nslot + j;
                    if (slot >= S) { slot = 255; atomicOr(&misc[M_STATUS], 2u); }
                    int r = rbase + j;
                    if (r > 255) { r = 255; atomicOr(&misc[M_STATUS], 4u); }
                    const unsigned c = atomicAdd(&misc[M_NCON], 1u);
                    if (c >= (unsigned)p.cap) { atomicOr(&misc[M_STATUS], 1u); continue; }
                    ca[c] = (unsigned short)a; cb[c] = (unsigned short)b;
                    ccls[c] = (unsigned char)cls; crank[c] = (unsigned char)r; cslot[c] = (unsigned char)slot;
                    cacc[c] = acc;
                    // island hooking: larger root goes under the smaller one
                    unsigned ra = a, rb = b;
                    for (;;) {
                        while (true) { unsigned t = ((volatile unsigned *)parent)[ra]; if (t == ra) break; ra = t; }
                        while (true) { unsigned t = ((volatile unsigned *)parent)[rb]; if (t == rb) break; rb = t; }
                        if (ra == rb) break;
                        if (ra < rb) { unsigned t = ra; ra = rb; rb = t; }
                        if (atomicCAS(&parent[ra], ra, rb) == ra) break;
                    }
                }
                nslot += nk;
            }
            // walls: b2CollideEdgeAndCircle (region AB) against the chain loop of kilobots_env.py:48-51
            int wrank = 0;
#pragma unroll
            for (int wl = 0; wl < 4; ++wl) {
                float dist, nx, ny;
                wall_geom(p, wl, ax, ay, dist, nx, ny);
                if (dist * dist > rw2) continue;
                float acc = 0.0f;
                for (int sl = 0; sl < (int)wsCnt[a] && sl < S; ++sl) {
                    const size_t idx = ((size_t)e * S + sl) * N + a;
                    if (g.ws_key[idx] == KEY_WALL + (unsigned)wl) { acc = g.ws_acc[idx]; break; }
                }
                int slot = nslot;
                if (slot >= S) { slot = 255; atomicOr(&misc[M_STATUS], 2u); }
                nslot++;
                const int r = wrank++;
                const unsigned c = atomicAdd(&misc[M_NCON], 1u);
                if (c >= (unsigned)p.cap) { atomicOr(&misc[M_STATUS], 1u); continue; }
                ca[c] = (unsigned short)(WALL_CODE + wl); cb[c] = (unsigned short)a;
                // bit 7: the centre is outside the wall line, so the manifold normal points outwards
                ccls[c] = (unsigned char)(CLS_WALL | (dist < 0.0f ? 0x80 : 0)); crank[c] = (unsigned char)r; cslot[c] = (unsigned char)slot;
                cacc[c] = acc;
            }
            wsCntNew[a] = (unsigned char)(nslot < S ? nslot : S);
        }
        __syncthreads();
        KB_STAMP(2);
        const int ncon = min((int)misc[M_NCON], p.cap);

        // ---- islands: flatten roots; empty the grid for the next substep ----
        for (int b = tid; b < N; b += nt) {
            unsigned r = b;
            while (true) { unsigned t = ((volatile unsigned *)parent)[r]; if (t == r) break; r = t; }
            parent[b] = r;   // only ever replaces an ancestor by an older ancestor: concurrent walks stay valid
            islCnt[b] = 0;
            head[cellOf[b]] = EMPTY32;
            active[b] = 1; active[N + b] = 0;
        }
        __syncthreads();
        for (int c = tid; c < ncon; c += nt) {
            const unsigned root = parent[cb[c]];
            const unsigned n = atomicAdd(&islCnt[root], 1u) + 1u;
            if (n > (unsigned)GIANT_ISLAND) atomicMax(&misc[M_MAXISL], n);
        }
        __syncthreads();
        // coop: one island is so large that a single wave would serialise the env -> whole-workgroup sweeps
        const bool coop = (misc[M_MAXISL] > (unsigned)GIANT_ISLAND) || nw == 1;
        const int W = coop ? 1 : nw;
        for (int c = tid; c < ncon; c += nt) {
            const unsigned root = parent[cb[c]];
            const int w = coop ? 0 : (int)(root % (unsigned)nw);
            const int cls = ccls[c] & 0x7F, r = crank[c];
            const int bk = (w * NUM_CLS + cls) * RK + (r < RK - 1 ? r : RK - 1);
            cbk[c] = (unsigned short)bk;
            atomicAdd(&bkStart[bk], 1u);
            if (r >= RK - 1) atomicMax(&bkMaxRank[w * NUM_CLS + cls], (unsigned)r);
        }
        __syncthreads();
        // exclusive scan of the bucket counts (<= 640 entries) by wave 0
        if (wave == 0) {
            const int nb = W * BK_PER_WAVE;
            const int chunk = (nb + 63) / 64;
            const int s0 = lane * chunk, e0 = min(nb, s0 + chunk);
            unsigned sum = 0;
            for (int i = s0; i < e0; ++i) sum += bkStart[i];
            unsigned incl = sum;
            for (int d = 1; d < 64; d <<= 1) {
                unsigned t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            unsigned run = incl - sum;
            for (int i = s0; i < e0; ++i) { unsigned v = bkStart[i]; bkStart[i] = run; run += v; }
            if (lane == 63) bkStart[nb] = incl;
        }
        __syncthreads();
        for (int c = tid; c < ncon; c += nt) {
            const int bk = cbk[c];
            const unsigned pos = bkStart[bk] + atomicAdd(&bkFill[bk], 1u);
            order[pos] = (unsigned short)c;
        }
        // every (virtual) wave compacts the list of its non-empty buckets, in key order
        if (wave < W) {
            const int base = wave * BK_PER_WAVE;
            const bool ne = lane < BK_PER_WAVE && bkStart[base + lane + 1] > bkStart[base + lane];
            const unsigned long long m = __ballot(ne);
            if (ne) bkList[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(base + lane);
            if (lane == 0) nList[wave] = (unsigned char)__popcll(m);
        }
        __syncthreads();
        KB_STAMP(3);

        // ---- solver ----
        // A "round" = all contacts of one key owned by this (virtual) wave.  coop: the workgroup is one
        // virtual wave and rounds are separated by s_barrier; otherwise each wave runs alone.
        const int myw = coop ? 0 : wave;
        const int lid = coop ? tid : lane;
        const int stride = coop ? nt : 64;
        const int nl = nList[myw];
#define KB_ROUND_SYNC() do { if (coop) __syncthreads(); else wave_sync(); } while (0)
#define KB_FOR_ROUNDS(...)                                                                          \
        for (int li = 0; li < nl; ++li) {                                                           \
            const int bk = bkList[myw * BK_PER_WAVE + li];                                          \
            const int s_ = (int)bkStart[bk], e_ = (int)bkStart[bk + 1];                             \
            if ((bk % RK) < RK - 1) {                                                               \
                for (int i_ = s_ + lid; i_ < e_; i_ += stride) { const int c = order[i_]; __VA_ARGS__ }    \
                KB_ROUND_SYNC();                                                                    \
            } else {                                                                                \
                const int maxr_ = (int)bkMaxRank[bk / RK];                                          \
                for (int r_ = RK - 1; r_ <= maxr_; ++r_) {                                          \
                    for (int i_ = s_ + lid; i_ < e_; i_ += stride) {                                \
                        const int c = order[i_];                                                    \
                        if ((int)crank[c] == r_) { __VA_ARGS__ }                                           \
                    }                                                                               \
                    KB_ROUND_SYNC();                                                                \
                }                                                                                   \
            }                                                                                       \
        }

        // velocity-phase normal of contact c from the start-of-step positions (b2WorldManifold::Initialize)
#define KB_VEL_NORMAL(a, b, nx, ny)                                                                 \
        float nx, ny;                                                                               \
        if (a >= WALL_CODE) {                                                                       \
            float dist_;                                                                            \
            wall_geom(p, a - WALL_CODE, px[b], py[b], dist_, nx, ny);                               \
            if (dist_ < 0.0f) { nx = -nx; ny = -ny; }                                               \
        } else {                                                                                    \
            const float dx_ = px[b] - px[a], dy_ = py[b] - py[a];                                   \
            const float dd_ = dx_ * dx_ + dy_ * dy_;                                                \
            nx = 1.0f; ny = 0.0f;                                                                   \
            if (dd_ > B2_EPSILON * B2_EPSILON) {                                                    \
                const float len_ = sqrtf(dd_);                                                      \
                const float inv_ = 1.0f / len_;                                                     \
                nx = dx_ * inv_; ny = dy_ * inv_;                                                   \
            }                                                                                       \
        }

        // b2ContactSolver::WarmStart
        KB_FOR_ROUNDS({
            const int a = ca[c], b = cb[c];
            KB_VEL_NORMAL(a, b, nx, ny)
            const float acc = cacc[c];
            const float Px = acc * nx, Py = acc * ny;
            if (a < WALL_CODE) { vx[a] -= p.im_bot * Px; vy[a] -= p.im_bot * Py; }
            vx[b] += p.im_bot * Px; vy[b] += p.im_bot * Py;
        })
        // SolveVelocityConstraints: friction 0, restitution 0, one manifold point
        for (int it = 0; it < p.vel_iters; ++it) {
            KB_FOR_ROUNDS({
                const int a = ca[c], b = cb[c];
                KB_VEL_NORMAL(a, b, nx, ny)
                float vax = 0.0f, vay = 0.0f, ima = 0.0f;
                if (a < WALL_CODE) { vax = vx[a]; vay = vy[a]; ima = p.im_bot; }
                const float vbx = vx[b], vby = vy[b];
                const float dvx = vbx - vax, dvy = vby - vay;
                const float vn = dvx * nx + dvy * ny;
                const float k = ima + p.im_bot;
                const float nm = k > 0.0f ? 1.0f / k : 0.0f;
                float lambda = -(nm * vn);
                const float accOld = cacc[c];
                const float newimp = fmaxf(accOld + lambda, 0.0f);
                lambda = newimp - accOld;
                cacc[c] = newimp;
                const float Px = lambda * nx, Py = lambda * ny;
                if (a < WALL_CODE) { vx[a] = vax - ima * Px; vy[a] = vay - ima * Py; }
                vx[b] = vbx + p.im_bot * Px; vy[b] = vby + p.im_bot * Py;
            })
        }
        __syncthreads();
        KB_STAMP(4);
        // ---- StoreImpulses -> warm-start cache of the next substep ----
        for (int c = tid; c < ncon; c += nt) {
            const int sl = cslot[c];
            if (sl == 255) continue;
            const int a = ca[c], b = cb[c];
            const int owner = a < WALL_CODE ? a : b;
            const unsigned key = a < WALL_CODE ? (unsigned)b : KEY_WALL + (unsigned)(a - WALL_CODE);
            const size_t idx = ((size_t)e * S + sl) * N + owner;
            g.ws_key[idx] = key; g.ws_acc[idx] = cacc[c];
        }
        // ---- integrate positions (b2Island::Solve) ----
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const int b = tid + q * nt;
            if (b >= N) continue;
            g.ws_cnt[o + b] = wsCntNew[b];
            float vxx = vx[b], vyy = vy[b], ww = bw[q];
            const float tx = h * vxx, ty = h * vyy;
            if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
                const float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
                vxx *= ratio; vyy *= ratio;
            }
            const float rot = h * ww;
            if (rot * rot > B2_MAX_ROTATION_SQ) {
                const float ratio = B2_MAX_ROTATION / fabsf(rot);
                ww *= ratio;
            }
            px[b] += h * vxx; py[b] += h * vyy;
            th[q] += h * ww;
        }
        __syncthreads();
        KB_STAMP(5);
        // ---- SolvePositionConstraints; an island stops once its minSeparation >= -3 slop ----
        for (int it = 0; it < p.pos_iters; ++it) {
            unsigned char *act = active + (it & 1) * N, *nxt = active + ((it + 1) & 1) * N;
            bool viol = false;
            KB_FOR_ROUNDS({
                const int a = ca[c], b = cb[c];
                const int isl = (int)parent[b];
                if (act[isl]) {
                    float nx, ny, sep, ima = 0.0f;
                    const float bx = px[b], by = py[b];
                    float axx = 0.0f, ayy = 0.0f;
                    if (a >= WALL_CODE) {
                        float dist, wx, wy;
                        wall_geom(p, a - WALL_CODE, bx, by, dist, wx, wy);
                        // manifold normal fixed at detection: flipped iff the centre was outside then
                        const bool flipped = (ccls[c] & 0x80) != 0;
                        nx = flipped ? -wx : wx; ny = flipped ? -wy : wy;
                        const float along = flipped ? -dist : dist;
                        sep = along - B2_POLYGON_RADIUS - p.r_bot;
                    } else {
                        axx = px[a]; ayy = py[a]; ima = p.im_bot;
                        const float dx = bx - axx, dy = by - ayy;
                        const float len = sqrtf(dx * dx + dy * dy);
                        nx = dx; ny = dy;
                        if (!(len < B2_EPSILON)) { const float inv = 1.0f / len; nx = dx * inv; ny = dy * inv; }
                        sep = (dx * nx + dy * ny) - p.r_bot - p.r_bot;
                    }
                    if (sep < -3.0f * B2_LINEAR_SLOP) { nxt[isl] = 1; viol = true; }
                    const float C = kb_clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                    const float K = ima + p.im_bot;
                    const float imp = K > 0.0f ? -C / K : 0.0f;
                    const float Px = imp * nx, Py = imp * ny;
                    if (a < WALL_CODE) { px[a] = axx - ima * Px; py[a] = ayy - ima * Py; }
                    px[b] = bx + p.im_bot * Px; py[b] = by + p.im_bot * Py;
                }
            })
            bool any;
            if (coop) {
                if (viol) misc[M_ANY] = 1u;
                __syncthreads();
                any = misc[M_ANY] != 0u;
                __syncthreads();
                if (tid == 0) misc[M_ANY] = 0u;
            } else {
                any = __any(viol);
            }
            if (!any) break;
            // the flags the next sweep sets must start cleared (only islands of this virtual wave)
            if (myw < W) {
                const int s_ = (int)bkStart[myw * BK_PER_WAVE], e_ = (int)bkStart[(myw + 1) * BK_PER_WAVE];
                for (int i_ = s_ + lid; i_ < e_; i_ += stride) act[parent[cb[order[i_]]]] = 0;
            }
            KB_ROUND_SYNC();
        }
        __syncthreads();
        KB_STAMP(6);
#undef KB_FOR_ROUNDS
#undef KB_VEL_NORMAL
#undef KB_ROUND_SYNC
    }

    // ---- write back ----
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = tid + q * nt;
        if (b < N) { g.x[o + b] = px[b]; g.y[o + b] = py[b]; g.theta[o + b] = th[q]; }
    }
    if (tid == 0) {
        if (p.light_type == KB_LIGHT_CIRCULAR && p.light_action && drive) { g.light_x[e] = lx; g.light_y[e] = ly; }
        if (misc[M_STATUS]) atomicOr(&g.status[e], (int)misc[M_STATUS]);
#ifdef KB_PROFILE
        prof_acc[7] += clock64() - prof_t;
        for (int k = 0; k < 8; ++k) g.status[p.E + 8 * e + k] += (int)(prof_acc[k] >> 4);   // units of 16 cycles
#endif
    }
}

// set_action alone (kilobot.py:235-241, 283-289)
__global__ void kb_set_actions_kernel(const Params p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t T = (size_t)p.E * p.N;
    if (i >= T) return;
    float a0 = 0.0f, a1 = 0.0f;
    if (p.actions) { const float2 a = reinterpret_cast<const float2 *>(p.actions)[i]; a0 = a.x; a1 = a.y; }
    if (p.drive_mode == KB_DRIVE_VELOCITY) {
        const float mw = 0.5f * 3.14159265358979323846f;
        p.buf.v[i] = fmaxf(fminf(a0, 0.01f), 0.0f);
        p.buf.w[i] = fmaxf(fminf(a1, mw), -mw);
    } else {
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

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

}  // namespace

struct kb_sim {
    kb_config cfg;
    Params p;
    bool bound;
    int threads;
};

extern "C" {

const char *kb_last_error(void) { return g_err; }
const char *kb_version(void) { return "kilobots_hip 0.1 (gfx950)"; }

int kb_create(const kb_config *cfg, kb_sim **out) {
    if (!cfg || !out) return fail(KB_EINVAL, "kb_create: NULL argument");
    if (cfg->num_envs < 1 || cfg->num_bots < 1 || cfg->num_bots > KB_MAX_BOTS)
        return fail(KB_EINVAL, "kb_create: num_envs >= 1 and 1 <= num_bots <= 1024 required");
    if (cfg->num_objects != 0) return fail(KB_EINVAL, "kb_create: objects are not supported by this version");
    if (cfg->drive_mode < 0 || cfg->drive_mode > KB_DRIVE_PHOTOTAXIS) return fail(KB_EINVAL, "kb_create: bad drive_mode");
    if (cfg->light_type != KB_LIGHT_NONE && cfg->light_type != KB_LIGHT_CIRCULAR)
        return fail(KB_EINVAL, "kb_create: unsupported light_type");
    if ((cfg->drive_mode == KB_DRIVE_SIMPLE_PHOTOTAXIS || cfg->drive_mode == KB_DRIVE_PHOTOTAXIS) &&
        cfg->light_type == KB_LIGHT_NONE)
        return fail(KB_EINVAL, "kb_create: phototaxis drive modes need a light");
    if (cfg->ws_slots < 1 || cfg->ws_slots > 64) return fail(KB_EINVAL, "kb_create: 1 <= ws_slots <= 64 required");
    if (!(cfg->dt > 0.0f) || cfg->vel_iters < 0 || cfg->pos_iters < 0 || !(cfg->world_width > 0.0f) ||
        !(cfg->world_height > 0.0f) || !(cfg->bot_radius > 0.0f) || !(cfg->bot_density > 0.0f))
        return fail(KB_EINVAL, "kb_create: non-positive dt / size / radius / density");
    kb_sim *s = new (std::nothrow) kb_sim();
    if (!s) return fail(KB_EINVAL, "kb_create: out of host memory");
    s->cfg = *cfg;
    s->bound = false;
    Params &p = s->p;
    memset(&p, 0, sizeof(p));
    p.N = cfg->num_bots; p.E = cfg->num_envs; p.S = cfg->ws_slots;
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
    p.h = cfg->dt;
    p.r_bot = cfg->bot_radius * WORLD_SCALE;
    const float m = cfg->bot_density * B2_PI * p.r_bot * p.r_bot;  // b2CircleShape::ComputeMass
    p.im_bot = m > 0.0f ? 1.0f / m : 0.0f;
    p.kl_bot = 1.0f / (1.0f + p.h * cfg->bot_linear_damping);
    p.ka_bot = 1.0f / (1.0f + p.h * cfg->bot_angular_damping);
    p.light_radius = cfg->light_radius;
    for (int i = 0; i < 2; ++i) {
        p.light_lo[i] = cfg->light_lo[i]; p.light_hi[i] = cfg->light_hi[i];
        p.act_lo[i] = cfg->light_act_lo[i]; p.act_hi[i] = cfg->light_act_hi[i];
    }
    long cap = (long)p.N * (p.N - 1) / 2 + 4L * p.N;
    if (cap > 2304) cap = 2304;
    if (cap < 4L * p.N + 64) cap = 4L * p.N + 64;
    p.cap = (int)cap;
    p.L = make_layout(p.N, p.ncell, p.cap);
    if (p.L.total > 160 * 1024) {
        delete s;
        return fail(KB_ELDS, "kb_create: configuration needs more than 160 KiB of LDS per env");
    }
    s->threads = p.N <= 64 ? 64 : (p.N <= 128 ? 128 : 256);   // N <= BPT * threads
    *out = s;
    return KB_OK;
}

void kb_destroy(kb_sim *sim) { delete sim; }

int kb_bind(kb_sim *sim, const kb_buffers *b) {
    if (!sim || !b) return fail(KB_EINVAL, "kb_bind: NULL argument");
    if (!b->x || !b->y || !b->theta || !b->ws_key || !b->ws_acc || !b->ws_cnt || !b->status)
        return fail(KB_ENOTBOUND, "kb_bind: x, y, theta, ws_key, ws_acc, ws_cnt and status are required");
    const int m = sim->cfg.drive_mode;
    if ((m == KB_DRIVE_VELOCITY || m == KB_DRIVE_ACCEL) && (!b->v || !b->w))
        return fail(KB_ENOTBOUND, "kb_bind: v and w are required in the velocity / acceleration modes");
    if (m == KB_DRIVE_ACCEL && (!b->acc_v || !b->acc_w)) return fail(KB_ENOTBOUND, "kb_bind: acc_v, acc_w required");
    if ((m == KB_DRIVE_MOTORS || m == KB_DRIVE_PHOTOTAXIS) && (!b->motor_l || !b->motor_r))
        return fail(KB_ENOTBOUND, "kb_bind: motor_l, motor_r required");
    if (m == KB_DRIVE_PHOTOTAXIS && (!b->pt_threshold || !b->pt_update || !b->pt_nochange || !b->pt_dir))
        return fail(KB_ENOTBOUND, "kb_bind: pt_* buffers required in the phototaxis mode");
    if (sim->cfg.light_type != KB_LIGHT_NONE && (!b->light_x || !b->light_y))
        return fail(KB_ENOTBOUND, "kb_bind: light_x, light_y required when a light is configured");
    if ((b->light_value != nullptr) != (b->light_gx != nullptr) || (b->light_value != nullptr) != (b->light_gy != nullptr))
        return fail(KB_EINVAL, "kb_bind: light_value, light_gx, light_gy must be given together");
    if ((b->cmd_vx != nullptr) != (b->cmd_vy != nullptr) || (b->cmd_vx != nullptr) != (b->cmd_w != nullptr))
        return fail(KB_EINVAL, "kb_bind: cmd_vx, cmd_vy, cmd_w must be given together");
    sim->p.buf = *b;
    sim->bound = true;
    return KB_OK;
}

int kb_set_actions(kb_sim *sim, const float *d_actions, void *stream) {
    if (!sim) return fail(KB_EINVAL, "kb_set_actions: NULL handle");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_set_actions: kb_bind() first");
    if (sim->cfg.drive_mode != KB_DRIVE_VELOCITY && sim->cfg.drive_mode != KB_DRIVE_ACCEL)
        return fail(KB_EINVAL, "kb_set_actions: only the velocity / acceleration drive modes take actions");
    Params p = sim->p;
    p.actions = d_actions;
    const size_t T = (size_t)p.E * p.N;
    hipLaunchKernelGGL(kb_set_actions_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_set_actions: %s", hipGetErrorString(err));
    return KB_OK;
}

int kb_step(kb_sim *sim, const float *d_actions, const float *d_light_action, int n_substeps, int flags, void *stream) {
    if (!sim) return fail(KB_EINVAL, "kb_step: NULL handle");
    if (!sim->bound) return fail(KB_ENOTBOUND, "kb_step: kb_bind() first");
    if (n_substeps < 0) return fail(KB_EINVAL, "kb_step: n_substeps < 0");
    if (d_actions && sim->cfg.drive_mode != KB_DRIVE_VELOCITY && sim->cfg.drive_mode != KB_DRIVE_ACCEL)
        return fail(KB_EINVAL, "kb_step: only the velocity / acceleration drive modes take actions");
    if (n_substeps == 0 && !d_actions) return KB_OK;
    Params p = sim->p;
    p.actions = d_actions;
    p.light_action = d_light_action;
    p.n_substeps = n_substeps;
    p.flags = flags;
    static thread_local int attr_set_for = -1;
    if (p.L.total > 64 * 1024 && attr_set_for != p.L.total) {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kb_step_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, p.L.total);
        if (e2 != hipSuccess) return fail(KB_EHIP, "kb_step: hipFuncSetAttribute: %s", hipGetErrorString(e2));
        attr_set_for = p.L.total;
    }
    hipLaunchKernelGGL(kb_step_kernel, dim3((unsigned)p.E), dim3((unsigned)sim->threads), (size_t)p.L.total,
                       (hipStream_t)stream, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(KB_EHIP, "kb_step: %s", hipGetErrorString(err));
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

int kb_lds_bytes(const kb_sim *sim) { return sim ? sim->p.L.total : KB_EINVAL; }
int kb_contact_capacity(const kb_sim *sim) { return sim ? sim->p.cap : KB_EINVAL; }
int kb_block_threads(const kb_sim *sim) { return sim ? sim->threads : KB_EINVAL; }
int kb_set_block_threads(kb_sim *sim, int threads) {
    if (!sim || threads < 64 || threads > 64 * MAX_WAVES || (threads & 63)) return fail(KB_EINVAL, "kb_set_block_threads: multiple of 64 in [64, 256]");
    if (sim->p.N > BPT * threads) return fail(KB_EINVAL, "kb_set_block_threads: need num_bots <= 4 * threads");
    sim->threads = threads;
    return KB_OK;
}

}  // extern "C"
