// kb_common.h -- constants, kernel parameters, LDS layout and device helpers shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "kilobots_hip.h"

namespace kb {

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
constexpr float B2_TOI_BAUMGARTE = 0.75f;
constexpr int B2_MAX_SUBSTEPS = 8;     // b2World::SolveTOI k_maxSubSteps
constexpr float B2_TIME_TO_SLEEP = 0.5f;                                  // b2_timeToSleep
constexpr float B2_LINEAR_SLEEP_TOL = 0.01f;                              // b2_linearSleepTolerance
constexpr float B2_ANGULAR_SLEEP_TOL = 2.0f / 180.0f * B2_PI;             // b2_angularSleepTolerance
constexpr float WORLD_SCALE = 25.0f;  // body.py:7

constexpr float CELL_SIZE = 0.875f;   // world units, >= 2 * bot radius
constexpr int MAX_CELLS = 8192;

constexpr unsigned KEY_WALL = 0x10000u, KEY_OBJ = 0x20000u;
constexpr int WALL_CODE = 0xFFF0;     // body id of wall w is WALL_CODE + w
constexpr int OBJ_CODE = 0xFFE0;      // 16-bit warm-start key of object m (its body id is N + m)
constexpr int MAXOBJ = KB_MAX_OBJECTS, OBJ_LIST = 64;   // OBJ_LIST: kilobots that may touch one fixture at a time
constexpr int OT_WORDS_C = 7 + 4 * KB_MAX_POLY_VERTS;   // floats per fixture in the fixture table (kb_objects.h)
constexpr int BT_WORDS_C = 6;                          // floats per object in the body table
constexpr int MC_FIELDS_C = 37;                        // words per manifold-constraint record (kb_objects.h)
constexpr unsigned short EMPTY16 = 0xFFFFu;

// contact classes in canonical order; +1 on E/N/NE/NW for odd base-cell parity
constexpr int CLS_SAME = 0, CLS_E = 1, CLS_N = 3, CLS_NE = 5, CLS_NW = 7, CLS_WALL = 9,
              CLS_BOT_OBJ = 10, NUM_CLS = 11;
constexpr int RK = 4;                 // rank buckets per class; the last one holds every rank >= RK-1
#ifndef KB_MAX_WAVES
#define KB_MAX_WAVES 8
#endif
#ifndef KB_BPT
#define KB_BPT 2
#endif
#ifndef KB_KREG
#define KB_KREG 2
#endif
#ifndef KB_MIN_WAVES_PER_SIMD
#define KB_MIN_WAVES_PER_SIMD 4
#endif
#ifndef KB_COMPACT_WAVES_PER_SIMD
#define KB_COMPACT_WAVES_PER_SIMD 6     // the compact fixed-size kernel: 80 VGPRs, three 8-wave workgroups per CU
#endif
constexpr int MAX_WAVES = KB_MAX_WAVES;   // waves per workgroup
constexpr int BK_PER_WAVE = NUM_CLS * RK;
constexpr int MAX_BUCKETS = MAX_WAVES * BK_PER_WAVE;
constexpr int BPT = KB_BPT;                // bots per thread (max): N <= BPT * blockDim.x
constexpr int KB_LIGHT_GENERAL = 99;   // kernel template value: any light model other than NONE / single CIRCULAR
constexpr int GIANT_ISLAND = 256;     // contacts; larger islands are swept by the whole workgroup
constexpr int BIG_ISLAND = 16;        // contacts; islands from this size on are placed on waves one by one when the hash placement overloads a wave
constexpr int KREG = KB_KREG;               // contacts a lane can keep in registers (register-resident solver)
#ifndef KB_NSOLVE_DIV
#define KB_NSOLVE_DIV 1                 // sweeping waves = waves of the workgroup / KB_NSOLVE_DIV (A/B knob, see kb_regsolve_bins.inc)
#endif
#ifndef KB_KREG_BINS
#define KB_KREG_BINS (2 * KB_NSOLVE_DIV)                  // ... in the kernels without objects, where only half the waves of a workgroup sweep (kb_regsolve_bins.inc)
#endif
constexpr int CAP_LDS = 1024;         // contacts staged in LDS; denser envs stage in the global scratch slice

enum { M_NCON = 0, M_TOTAL = 1, M_ANY = 2, M_STATUS = 3, M_MAXISL = 4, M_PROF = 5, M_XTRA = 6, M_XFILL = 7, M_WCNT = 8, M_WAKE = 32 /* .. 63: one bit per kilobot (sleeping; the -DKB_PROFILE build keeps its stamps there and does not implement the wake rule) */, M_WFILL = 8 + MAX_WAVES, M_COUNT = 8 + 2 * MAX_WAVES };
static_assert(M_COUNT <= 64, "misc area");

// ---- LDS layout ----------------------------------------------------------------------------------
// The small arrays come first: a few of fixed size, the bucket tables of the contact sort (they scale with the waves of
// the workgroup) and -- only in kernels with objects -- the object tables; `fixed(obj, nw)` is where the arrays that
// scale with the scene start.  Those are grouped by stride, so that every offset is (base + k * stride) of four scene
// sizes.  The kernel recomputes offsets where they are used instead of keeping ~35 of them alive in scalar registers
// for the whole launch; in the fixed-size instantiations all of this folds to constants.
namespace lds {
constexpr int A16(int x) { return (x + 15) & ~15; }
constexpr int MISC = 0;
constexpr int WSUM = MISC + A16(4 * 64);
constexpr int BKSTART = WSUM + A16(4 * 16);
// buckets of a workgroup of nw waves (the tables double as scratch of the island placement: >= 64 entries)
__host__ __device__ constexpr int nbk(int nw) { return nw * BK_PER_WAVE < 64 ? 64 : nw * BK_PER_WAVE; }
__host__ __device__ constexpr int bkfill(int nw) { return BKSTART + A16(4 * (nbk(nw) + 1)); }
__host__ __device__ constexpr int bkmaxrank(int nw) { return bkfill(nw) + A16(4 * nbk(nw)); }
__host__ __device__ constexpr int bklist(int nw) { return bkmaxrank(nw) + A16(4 * MAX_WAVES * NUM_CLS); }
__host__ __device__ constexpr int nlist(int nw) { return bklist(nw) + A16(2 * nbk(nw)); }
__host__ __device__ constexpr int objtab(int nw) { return nlist(nw) + 16; }   // object table: mass, shape
// object areas, relative to objtab(nw)
constexpr int OBJBODY = A16(4 * OT_WORDS_C * KB_MAX_OBJECTS);            // body table
constexpr int OBJCNT = OBJBODY + A16(4 * BT_WORDS_C * KB_MAX_OBJECTS);
constexpr int OBJLIST = OBJCNT + A16(4 * KB_MAX_OBJECTS);
constexpr int OBJW = OBJLIST + A16(2 * KB_MAX_OBJECTS * OBJ_LIST);       // angular velocity, angle, angle at the start of the substep
constexpr int OBJA = OBJW + A16(4 * KB_MAX_OBJECTS);
constexpr int OBJA0 = OBJA + A16(4 * KB_MAX_OBJECTS);
constexpr int MCMASK = OBJA0 + A16(4 * KB_MAX_OBJECTS);                  // per wave: which manifold constraints it owns (u64)
constexpr int OBJSLP = MCMASK + A16(8 * (MAX_WAVES + 1));                // sleeping: b2Body::m_sleepTime of the objects (< 0: asleep)
constexpr int OBJ_AREA = OBJSLP + A16(4 * KB_MAX_OBJECTS);
__host__ __device__ constexpr int fixed(bool obj, int nw) { return objtab(nw) + (obj ? OBJ_AREA : 0); }
// per-body 32-bit arrays (stride 4 * NB): px py vx vy x0 y0 dirCnt parent
constexpr int BODY32_COUNT = 8;
// per-contact 32-bit arrays (stride 4 * capL): sPair sInfo sAcc oldAcc;  16-bit (stride 2 * capL): cbk order oldKey
constexpr int CON32_COUNT = 4, CON16_COUNT = 3;
// per-bot 16-bit arrays (stride 2 * NP): wsOff newOff next cellOf;  8-bit (stride NP): wsCnt wsCntNew
constexpr int BOT16_COUNT = 4, BOT8_COUNT = 2;
// (fx = fixed(obj, nw))
__host__ __device__ inline int body32(int fx, int NB, int k) { return fx + 4 * NB * k; }
__host__ __device__ inline int con32(int fx, int NB, int capL, int k) { return body32(fx, NB, BODY32_COUNT) + 4 * capL * k; }
__host__ __device__ inline int con16(int fx, int NB, int capL, int k) { return con32(fx, NB, capL, CON32_COUNT) + 2 * capL * k; }
__host__ __device__ inline int bot16(int fx, int NB, int capL, int NP, int k) { return con16(fx, NB, capL, CON16_COUNT) + 2 * NP * k; }
__host__ __device__ inline int bot8(int fx, int NB, int capL, int NP, int k) { return bot16(fx, NB, capL, NP, BOT16_COUNT) + NP * k; }
__host__ __device__ inline int active(int fx, int NB, int capL, int NP) { return bot8(fx, NB, capL, NP, BOT8_COUNT); }
__host__ __device__ inline int islwave(int fx, int NB, int capL, int NP) { return active(fx, NB, capL, NP) + 2 * NB; }   // u8 per body: wave that sweeps its island
__host__ __device__ inline int head(int fx, int NB, int capL, int NP) { return (islwave(fx, NB, capL, NP) + NB + 15) & ~15; }
__host__ __device__ inline int mcarea(int fx, int NB, int capL, int NP, int ncell) { return (head(fx, NB, capL, NP) + 2 * ncell + 4 + 15) & ~15; }
// manifold-constraint records (objects only): MC_FIELDS words x nmc candidates, field-major
__host__ __device__ inline int total(int fx, int NB, int capL, int NP, int ncell, int nmc) { return (mcarea(fx, NB, capL, NP, ncell) + 4 * MC_FIELDS_C * nmc + 15) & ~15; }
}  // namespace lds

// ---- compact LDS image: three envs of 1024 kilobots per CU --------------------------------------------------------------
// What decides the throughput of the latency-bound step is how many envs a CU holds (profiles/: waves wait 60 % of their
// cycles, no pipe is more than a third busy; 3 resident envs instead of 2 gave + 37 % at 480 kilobots).  Every kernel
// without objects uses this image; the fixed-size one runs at 80 VGPRs (6 waves per SIMD = three 8-wave workgroups) with
// 53 168 B instead of 80 416 B, the others pick the register budget that holds more envs (kb_create):
//   - no image of the previous substep's warm-start list (6 B per entry): the label pass reads the packed list in HBM / L2;
//   - the poses at the start of the substep (continuous step) are saved at integration time -- they do not change earlier --
//     into arrays that are dead by then: x over [nextb | cellOf], y over dirCnt;
//   - fixed-size kernel: the per-contact scratch of the contact sort / slot dealing (lCbk) lies over nextb (dead once the
//     label pass is done and needed only before the integration; needs capL <= NP);
//   - the bucket tables of the contact sort / island placement live in the cell-head area, which is dead once the label
//     pass is done; the heads are cleared as a whole at the end of the substep;
//   - 688 staged contacts (the settled benchmark scene has 550 +- 20 per env, at most 630 in 4096 envs; envs beyond take
//     the global staging slice as before); no object tables.
namespace ldsc {
constexpr int CAPL = 688;                                                   // staged contacts of the fixed-size kernel
__host__ __device__ constexpr int tables(int nw) { return lds::nlist(nw) + 16 - lds::BKSTART; }   // bucket tables, relative to the head area
__host__ __device__ constexpr int pos(int) { return 320; }                 // (MISC 256 + WSUM 64 in front)
__host__ __device__ constexpr int vel(int NB) { return pos(NB) + 8 * NB; }
__host__ __device__ constexpr int dircnt(int NB) { return vel(NB) + 8 * NB; }
__host__ __device__ constexpr int parent(int NB) { return dircnt(NB) + 4 * NB; }
__host__ __device__ constexpr int con32(int NB, int capL, int k) { return parent(NB) + 4 * NB + 4 * capL * k; }      // lPair lInfo lAcc
// 16-bit per-contact arrays: lOrder, and lCbk unless it lies over nextb (`fold`: only where capL <= NP, the fixed-size kernel)
__host__ __device__ constexpr int con16(int NB, int capL, int k) { return con32(NB, capL, 3) + ((2 * capL + 3) & ~3) * k; }
__host__ __device__ constexpr int bot16(int NB, int capL, int NP, bool fold, int k) { return con16(NB, capL, fold ? 1 : 2) + 2 * NP * k; }   // wsOff newOff nextb cellOf
__host__ __device__ constexpr int bot8(int NB, int capL, int NP, bool fold, int k) { return bot16(NB, capL, NP, fold, 4) + NP * k; }        // wsCnt wsCntNew
__host__ __device__ constexpr int active(int NB, int capL, int NP, bool fold) { return bot8(NB, capL, NP, fold, 2); }
__host__ __device__ constexpr int islwave(int NB, int capL, int NP, bool fold) { return active(NB, capL, NP, fold) + 2 * NB; }
__host__ __device__ constexpr int head(int NB, int capL, int NP, bool fold) { return (islwave(NB, capL, NP, fold) + NB + 15) & ~15; }
__host__ __device__ inline int total(int NB, int capL, int NP, bool fold, int ncell, int nw) {
    const int h = (2 * ncell + 4 + 15) & ~15;
    return head(NB, capL, NP, fold) + (h > tables(nw) ? h : tables(nw));
}
}  // namespace ldsc

// ---- sorted-bin LDS image (round 3): every kernel without objects ------------------------------------------------------
// The broadphase is a counting sort of the kilobots by grid cell ("bins"; a hash of the cells for sparse swarms), and the
// bodies LIVE in that order for the length of a substep: pos / vel / parent / ... are indexed by the kilobot's SLOT in the
// sorted order, so the candidates of a stencil row are one contiguous run of slots (no list heads, no next pointers, no
// indirection through an id).  A contact is stored at its position in the packed warm-start list of the substep
// (newOff[owner] + slot of the owner's list), so StoreImpulses is a copy, and the previous substep's list is kept as an
// LDS image (6 B per entry) that aliases arrays which only live between the island phase and the end of the substep.
//   always            : misc | pos | vel
//   parent            : union-find (emit .. sleep bookkeeping); start-of-substep angle of the TOI candidates behind it
//   dircnt            : per-direction contact counts (find .. emit) -> island census / body depth -> startY (integrate .. TOI)
//   botA              : [wsOff | wsCnt | idOf | (cellOfSlot: hashed bins)]; startX over its front (integrate .. TOI)
//   botB              : [newOff | wsCntNew]
//   con32 x 3         : lPair lInfo lAcc   (the arrival-order scratch of the sort lies over lInfo, the neighbour counters of
//                                           the sensing pass over lAcc: both dead before the emit pass writes the records)
//   act0              : position-solver island flags of even iterations (also "island has an awake body": cleared in the
//                       drive phase, hence outside the aliased zone)
//   zone              : [act1 | islWave | lOrder | lCbk]  aliased by the warm-start image [oldAcc | oldKey]
//   binE              : bin boundaries (u16, entry i = kilobots in bins < i) -> bucket tables of the contact sort -> unused
namespace ldsb {
__host__ __device__ constexpr int A16(int x) { return (x + 15) & ~15; }
__host__ __device__ constexpr int A4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int pos() { return 320; }                       // (MISC 256 + WSUM 64 in front)
__host__ __device__ constexpr int vel(int NB) { return pos() + 8 * NB; }
__host__ __device__ constexpr int parent(int NB) { return vel(NB) + 8 * NB; }
__host__ __device__ constexpr int dircnt(int NB) { return parent(NB) + 4 * NB; }
__host__ __device__ constexpr int botA(int NB) { return dircnt(NB) + 4 * NB; }
__host__ __device__ constexpr int botA_size(int NB, int NP, bool hc) { return A16((5 + (hc ? 2 : 0)) * NP > 4 * NB ? (5 + (hc ? 2 : 0)) * NP : 4 * NB); }
__host__ __device__ constexpr int wsoff(int NB) { return botA(NB); }
__host__ __device__ constexpr int wscnt(int NB, int NP) { return botA(NB) + 2 * NP; }
__host__ __device__ constexpr int idof(int NB, int NP) { return botA(NB) + 3 * NP; }
__host__ __device__ constexpr int cellofslot(int NB, int NP) { return botA(NB) + 5 * NP; }
__host__ __device__ constexpr int botB(int NB, int NP, bool hc) { return botA(NB) + botA_size(NB, NP, hc); }
__host__ __device__ constexpr int newoff(int NB, int NP, bool hc) { return botB(NB, NP, hc); }
__host__ __device__ constexpr int wscntnew(int NB, int NP, bool hc) { return botB(NB, NP, hc) + 2 * NP; }
__host__ __device__ constexpr int con32(int NB, int NP, bool hc, int capL, int k) { return A16(botB(NB, NP, hc) + 3 * NP) + 4 * capL * k; }
__host__ __device__ constexpr int act0(int NB, int NP, bool hc, int capL) { return con32(NB, NP, hc, capL, 3); }
__host__ __device__ constexpr int zone(int NB, int NP, bool hc, int capL) { return act0(NB, NP, hc, capL) + NB; }      // act1 follows act0 directly
__host__ __device__ constexpr int islwave(int NB, int NP, bool hc, int capL) { return zone(NB, NP, hc, capL) + NB; }
__host__ __device__ constexpr int order(int NB, int NP, bool hc, int capL) { return A4(islwave(NB, NP, hc, capL) + NB); }
__host__ __device__ constexpr int cbk(int NB, int NP, bool hc, int capL) { return order(NB, NP, hc, capL) + A4(2 * capL); }
__host__ __device__ constexpr int oldacc(int NB, int NP, bool hc, int capL) { return A4(zone(NB, NP, hc, capL)); }
__host__ __device__ constexpr int oldkey(int NB, int NP, bool hc, int capL) { return oldacc(NB, NP, hc, capL) + 4 * capL; }
__host__ __device__ constexpr int zone_end(int NB, int NP, bool hc, int capL) {
    return cbk(NB, NP, hc, capL) + A4(2 * capL) > oldkey(NB, NP, hc, capL) + 2 * capL ? cbk(NB, NP, hc, capL) + A4(2 * capL) : oldkey(NB, NP, hc, capL) + 2 * capL;
}
__host__ __device__ constexpr int binE(int NB, int NP, bool hc, int capL) { return A16(zone_end(NB, NP, hc, capL)); }
__host__ __device__ constexpr int bin_entries(int nbin) { return (nbin + 1 + 7) & ~7; }                 // u16 entries, whole 16-byte chunks
__host__ __device__ constexpr int tables(int nw) { return lds::nlist(nw) + 16 - lds::BKSTART; }       // bucket tables, relative to binE
__host__ __device__ inline int binE_size(int nbin, int nw) { return A16(2 * bin_entries(nbin) > tables(nw) ? 2 * bin_entries(nbin) : tables(nw)); }
__host__ __device__ inline int total(int NB, int NP, bool hc, int capL, int nbin, int nw) { return binE(NB, NP, hc, capL) + binE_size(nbin, nw); }
constexpr int CAPL = 688;                                                     // staged contacts of the fixed-size kernel
}  // namespace ldsb

struct Params {
    kb_buffers buf;
    const float *actions;
    const float *light_action;
    int N, NP, NB, M, E, S, gw, gh, ncell, cap, capL, n_substeps, flags, drive_mode, light_type, vel_iters, pos_iters;
    int solver_mode, toi_walls;
    float xmin, ymin, xmax, ymax, inv_cell, r_bot, im_bot, kl_bot, ka_bot, h;
    float light_radius, light_lo[2], light_hi[2], act_lo[2], act_hi[2];
    // general light model (GradientLight / MomentumLight / CompositeLight): per component
    int lcount, ladim, lkind[KB_MAX_LIGHTS];
    float lradius[KB_MAX_LIGHTS], lmaxv[KB_MAX_LIGHTS];
    float llo[KB_MAX_LIGHTS][2], lhi[KB_MAX_LIGHTS][2], lalo[KB_MAX_LIGHTS][2], lahi[KB_MAX_LIGHTS][2];
    float kl_obj, ka_obj;
    float otab[KB_MAX_OBJECTS][OT_WORDS_C];   // fixture table (kb_objects.h: OT_*)
    float obody[KB_MAX_OBJECTS][BT_WORDS_C];  // body table (kb_objects.h: BT_*)
    int F;                                    // fixtures (>= M)
    float mu_oo, mu_ow;                       // b2MixFriction: object-object, object-wall
    int nmc;                                  // manifold-constraint candidates: pairs + 4 walls per object
    int lds_total;
    int nhead, hmask;                         // cell heads: nhead entries; hmask != 0: a hash table of the cells (slot = cell & hmask)
    int allow_sleep;                          // kb_config.allow_sleep: the SLEEP instantiations are launched
    float im_mode[5];                         // KB_DRIVE_MIXED: inverse mass of a kilobot by drive law
    int botlaw_off;                           // ... and the LDS offset of the per-kilobot law bytes (behind the image)
    int islmin_off;                           // sorted-bin image, SLEEP: LDS offset of the per-island minimum of the sleep times (NB words)
    int sense_s;                              // IR neighbour sensing: reach of the stencil in cells (0 = off)
    float sense_r2;                           // ... and the squared radius in world units
    // uniform constants of the contact search and the solver, evaluated once on the host (same fp32 expressions): as kernel
    // arguments they live in scalar registers -- computed in the kernel they would hold a vector register each
    float rr2, rw2, rw_tot;                   // (r + r)^2;  (polygonRadius + r)^2;  r + polygonRadius
    float nm_bb, nm_wb;                       // effective mass of a kilobot-kilobot and of a wall-kilobot contact
};


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

// ---- light models shared by the step kernel and kb_light_sense (same operations in the same order: same bits) ----
// SinglePositionLight.step of the single circular light, light.py:59-75; la = this env's action (2 floats)
__device__ __forceinline__ void kb_light_single_step(const Params &p, const float *la, float h, float &lx, float &ly) {
    float ax = fminf(fmaxf(la[0], p.act_lo[0]), p.act_hi[0]);
    float ay = fminf(fmaxf(la[1], p.act_lo[1]), p.act_hi[1]);
    float nlx = lx + ax * h, nly = ly + ay * h;
    lx = fminf(fmaxf(nlx, p.light_lo[0]), p.light_hi[0]);
    ly = fminf(fmaxf(nly, p.light_lo[1]), p.light_hi[1]);
}
// Light.step of every component of the general model: light.py:59-75 (positional), 300-316 (momentum), 237-253 (gradient);
// la = this env's action (ladim floats)
__device__ __forceinline__ void kb_light_general_step(const Params &p, const float *la, float h, float (&glx)[KB_MAX_LIGHTS],
                                                      float (&gly)[KB_MAX_LIGHTS], float (&glvx)[KB_MAX_LIGHTS], float (&glvy)[KB_MAX_LIGHTS]) {
    if (p.light_type == KB_LIGHT_GRADIENT) {
        const float pi = 3.14159265358979323846f;
        float ang = fminf(fmaxf(la[0], -2.0f * pi), 2.0f * pi);
        if (ang < -pi) ang += 2.0f * pi;
        if (ang > pi) ang -= 2.0f * pi;
        glx[0] = ang;
    } else {
#pragma unroll
        for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
            if (i >= p.lcount) break;
            const float ax = fminf(fmaxf(la[2 * i + 0], p.lalo[i][0]), p.lahi[i][0]);
            const float ay = fminf(fmaxf(la[2 * i + 1], p.lalo[i][1]), p.lahi[i][1]);
            float nlx, nly;
            if (p.lkind[i] == KB_LIGHT_MOMENTUM) {
                float mvx = glvx[i] + ax * h, mvy = glvy[i] + ay * h;
                const float nv = sqrtf(mvx * mvx + mvy * mvy);
                if (nv > p.lmaxv[i]) { const float sc = p.lmaxv[i] / nv; mvx *= sc; mvy *= sc; }
                glvx[i] = mvx; glvy[i] = mvy;
                nlx = glx[i] + mvx * h; nly = gly[i] + mvy * h;
            } else {
                nlx = glx[i] + ax * h; nly = gly[i] + ay * h;
            }
            glx[i] = fminf(fmaxf(nlx, p.llo[i][0]), p.lhi[i][0]);
            gly[i] = fminf(fmaxf(nly, p.llo[i][1]), p.lhi[i][1]);
        }
    }
}
// value_and_gradients of the general model at one sensor position (metres): light.py:176-189, 137-141;
// GradientLight: projection on the gradient direction (intent of light.py:255-260)
__device__ __forceinline__ void kb_light_general_sense(const Params &p, const float (&glx)[KB_MAX_LIGHTS], const float (&gly)[KB_MAX_LIGHTS],
                                                       float sx, float sy, float &val, float &gx, float &gy) {
    if (p.light_type == KB_LIGHT_GRADIENT) {
        float s_, c_;
        kb_sincosf(glx[0], s_, c_);
        val = c_ * sx + s_ * sy; gx = c_; gy = s_;
        return;
    }
    float vsum = 0.0f, vbest = 0.0f, bgx = 0.0f, bgy = 0.0f;
#pragma unroll
    for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
        if (i >= p.lcount) break;
        float v, x, y;
        kb_light_circular(sx, sy, glx[i], gly[i], p.lradius[i], v, x, y);
        vsum = i == 0 ? v : vsum + v;
        if (i == 0 || v > vbest) { vbest = v; bgx = x; bgy = y; }   // np.argmax: first maximum
    }
    val = vsum; gx = bgx; gy = bgy;
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

// Read of an LDS word that other waves update with atomics (the union-find parents).  NOTE: through a generic pointer a `volatile`
// access is a FLAT load with system-coherence bits and a full `s_waitcnt vmcnt(0) lgkmcnt(0)` behind it, not a ds_read_b32 (seen
// in the label and flatten passes).  Three ways to get the ds_read were tried in round 3 (a relaxed __hip_atomic_load, a volatile
// access through an LDS-qualified pointer, the same via the low half of the address): each made hipcc 7.2 fail in some OTHER
// instantiation with "Illegal instruction detected: Operand has incorrect register class. V_CMP_NE_U32_e32 0, $src_shared_base",
// so the flat form stays; it was worth ~ 0.5 % of a cfg3 launch.
__device__ __forceinline__ unsigned lds_load_relaxed(const unsigned *p_) {
    return *reinterpret_cast<const volatile unsigned *>(p_);
}

// Workgroup barrier for phases that only exchange LDS data: __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), i.e. it waits for every global load a thread has in flight -- which is exactly what a sweep that
// requests its records several rounds ahead must not do.  No memory instruction inside: the compiler's own s_waitcnt
// bookkeeping for the registers of outstanding loads stays valid.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- wave-wide reductions / scan over all 64 lanes with DPP (data-parallel primitives: row shifts inside the four rows of
// 16 lanes, then the row_bcast steps across rows): six VALU instructions with no trip to the LDS crossbar, where a
// __shfl_xor / __shfl_up ladder is six dependent ds_bpermute round trips.  All 64 lanes must be active.
#define KB_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, true)
__device__ __forceinline__ unsigned wave_or(unsigned v) {       // OR of v over the wave, the same value in every lane
    v |= (unsigned)KB_DPP(v, 0x111, 0xf);      // row_shr:1
    v |= (unsigned)KB_DPP(v, 0x112, 0xf);      // row_shr:2
    v |= (unsigned)KB_DPP(v, 0x114, 0xf);      // row_shr:4
    v |= (unsigned)KB_DPP(v, 0x118, 0xf);      // row_shr:8   -> lane 15 of each row holds its row
    v |= (unsigned)KB_DPP(v, 0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    v |= (unsigned)KB_DPP(v, 0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, (unsigned)KB_DPP(v, 0x111, 0xf));
    v = max(v, (unsigned)KB_DPP(v, 0x112, 0xf));
    v = max(v, (unsigned)KB_DPP(v, 0x114, 0xf));
    v = max(v, (unsigned)KB_DPP(v, 0x118, 0xf));
    v = max(v, (unsigned)KB_DPP(v, 0x142, 0xa));
    v = max(v, (unsigned)KB_DPP(v, 0x143, 0xc));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {   // inclusive prefix sum over the lanes
    v += (unsigned)KB_DPP(v, 0x111, 0xf);
    v += (unsigned)KB_DPP(v, 0x112, 0xf);
    v += (unsigned)KB_DPP(v, 0x114, 0xf);
    v += (unsigned)KB_DPP(v, 0x118, 0xf);
    v += (unsigned)KB_DPP(v, 0x142, 0xa);
    v += (unsigned)KB_DPP(v, 0x143, 0xc);
    return v;
}

template <class AR>
__device__ __forceinline__ void wall_geom(const AR &p, int wl, float x, float y, float &dist, float &nx, float &ny) {
    switch (wl) {
    case 0: nx = 1.0f; ny = 0.0f; dist = x - p.xmin; break;
    case 1: nx = 0.0f; ny = 1.0f; dist = y - p.ymin; break;
    case 2: nx = -1.0f; ny = 0.0f; dist = p.xmax - x; break;
    default: nx = 0.0f; ny = -1.0f; dist = p.ymax - y; break;
    }
}

#ifdef KB_PROFILE
// diagnostic build: thread 0 accumulates shader cycles per phase into g.status[E + 8*e + phase]
// (light-weight: one 32-bit time register and fire-and-forget LDS atomics on the free words 24.. of the misc area; the first
//  version kept 24 64-bit accumulators in registers, which the allocator spilled: every stamp was a scratch round trip and a
//  fifth of the profiled time was the profile itself)
#define KB_PROF(k) misc[24 + (k)]
#define KB_STAMP(ph) do { if (tid == 0) { const unsigned t_ = (unsigned)clock64(); atomicAdd(&KB_PROF(ph), t_ - prof_t); prof_t = t_; } } while (0)
// time since the last stamp without closing the interval: wave 0's own work before it waits at the barrier
#define KB_STAMP_PRE(ph) do { if (tid == 0) { atomicAdd(&KB_PROF(ph), (unsigned)clock64() - prof_t); } } while (0)
#else
#define KB_STAMP(ph) do { } while (0)
#define KB_STAMP_PRE(ph) do { } while (0)
#endif

// b2TimeOfImpact for a circle (radius R) whose centre moves linearly from (x0,y0) to (x1,y1) against wall wl:
// Box2D's control flow (conservative advancement + bisection / secant root finder) on the closed-form distance of
// the centre to an axis-aligned wall.  True and t in [0,1] when the state is e_touching.
template <class AR>
__device__ __forceinline__ bool kb_toi_wall(const AR &p, int wl, float R, float x0, float y0, float x1, float y1, float &tout) {
    const float total = R + B2_POLYGON_RADIUS;
    const float target = fmaxf(B2_LINEAR_SLOP, total - 3.0f * B2_LINEAR_SLOP);
    const float tol = 0.25f * B2_LINEAR_SLOP;
    float nx, ny;
    auto dist_at = [&](float t) __attribute__((always_inline)) -> float {
        float dv;
        wall_geom(p, wl, (1.0f - t) * x0 + t * x1, (1.0f - t) * y0 + t * y1, dv, nx, ny);
        return dv;
    };
    float t1 = 0.0f;
    for (int iter = 0; iter < 20; ++iter) {
        const float dist = dist_at(t1);
        if (fabsf(dist) <= 0.0f) return false;                        // overlapped
        if (fabsf(dist) < target + tol) { tout = t1; return true; }   // touching
        float t2 = 1.0f;
        for (int push = 0; push < 8; ++push) {
            float s2 = dist_at(t2);
            if (s2 > target + tol) return false;                      // separated at the end of the step
            if (s2 > target - tol) { t1 = t2; break; }
            float s1 = dist_at(t1);
            if (s1 < target - tol) return false;                      // failed
            if (s1 <= target + tol) { tout = t1; return true; }
            float a1 = t1, a2 = t2;
            for (int root = 0; root < 50; ++root) {                   // bisection / secant alternating
                const float t = (root & 1) ? a1 + (target - s1) * (a2 - a1) / (s2 - s1) : 0.5f * (a1 + a2);
                const float sv = dist_at(t);
                if (fabsf(sv - target) < tol) { t2 = t; break; }
                if (sv > target) { a1 = t; s1 = sv; } else { a2 = t; s2 = sv; }
            }
        }
    }
    return false;                                                     // failed (iteration cap)
}

// b2World::SolveTOI + b2Island::SolveTOI for one circular body against the arena walls (the only TOI events Box2D
// computes for non-bullet bodies).  (x0,y0,a0): pose at the start of the step; (x,y,a): pose after b2Island::Solve;
// (vx,vy,w): velocity after the solve; updated in place.
__device__ __forceinline__ void kb_toi_walls_body(const Params &p, float R, float im, float x0, float y0, float a0,
                                                  float &x, float &y, float &a, float &vx, float &vy, float &w) {
    const float h = p.h;
    const float total = R + B2_POLYGON_RADIUS;
    // quick reject: a body that stays clear of every wall by more than its contact radius has no TOI event
    {
        const float m0 = fminf(fminf(x0 - p.xmin, p.xmax - x0), fminf(y0 - p.ymin, p.ymax - y0));
        const float m1 = fminf(fminf(x - p.xmin, p.xmax - x), fminf(y - p.ymin, p.ymax - y));
        if (m0 > total && m1 > total) return;
    }
    float alpha0 = 0.0f;
    float c0x = x0, c0y = y0, ca0 = a0, cx = x, cy = y, ca = a;
    for (int ev = 0; ev < B2_MAX_SUBSTEPS; ++ev) {
        float minAlpha = 1.0f;
#pragma unroll
        for (int wl = 0; wl < 4; ++wl) {
            float t;
            float alpha = 1.0f;
            if (kb_toi_wall(p, wl, R, c0x, c0y, cx, cy, t)) alpha = fminf(alpha0 + (1.0f - alpha0) * t, 1.0f);
            if (alpha < minAlpha) minAlpha = alpha;
        }
        if (1.0f - 10.0f * B2_EPSILON < minAlpha) break;
        const float beta = (minAlpha - alpha0) / (1.0f - alpha0);      // b2Body::Advance
        c0x += beta * (cx - c0x); c0y += beta * (cy - c0y); ca0 += beta * (ca - ca0);
        alpha0 = minAlpha;
        cx = c0x; cy = c0y; ca = ca0;
        bool touch[4];
        float wnx[4], wny[4];
#pragma unroll
        for (int wl = 0; wl < 4; ++wl) {                               // manifolds of the static contacts at the TOI pose
            float dist;
            wall_geom(p, wl, cx, cy, dist, wnx[wl], wny[wl]);
            touch[wl] = !(dist * dist > total * total);
            if (dist < 0.0f) { wnx[wl] = -wnx[wl]; wny[wl] = -wny[wl]; }
        }
        for (int it = 0; it < 20; ++it) {                              // SolveTOIPositionConstraints
            float minSep = 0.0f;
#pragma unroll
            for (int wl = 0; wl < 4; ++wl) {
                if (!touch[wl]) continue;
                float dist, bx, by;
                wall_geom(p, wl, cx, cy, dist, bx, by);
                const float along = (wnx[wl] == bx && wny[wl] == by) ? dist : -dist;
                const float sep = along - B2_POLYGON_RADIUS - R;
                minSep = fminf(minSep, sep);
                const float C = kb_clampf(B2_TOI_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                const float K = 0.0f + im;
                const float imp = K > 0.0f ? -C / K : 0.0f;
                cx += im * (imp * wnx[wl]); cy += im * (imp * wny[wl]);
            }
            if (minSep >= -1.5f * B2_LINEAR_SLOP) break;
        }
        c0x = cx; c0y = cy; ca0 = ca;                                   // leap of faith to the new safe state
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < p.vel_iters; ++it) {                      // velocity constraints, no warm starting
#pragma unroll
            for (int wl = 0; wl < 4; ++wl) {
                if (!touch[wl]) continue;
                const float vn = vx * wnx[wl] + vy * wny[wl];
                const float k = 0.0f + im;
                const float nm = k > 0.0f ? 1.0f / k : 0.0f;
                float lambda = -(nm * vn);
                const float newimp = fmaxf(acc[wl] + lambda, 0.0f);
                lambda = newimp - acc[wl];
                acc[wl] = newimp;
                vx += im * (lambda * wnx[wl]); vy += im * (lambda * wny[wl]);
            }
        }
        const float hh = (1.0f - minAlpha) * h;                         // integrate the rest of the step
        const float tx = hh * vx, ty = hh * vy;
        if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
            const float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
            vx *= ratio; vy *= ratio;
        }
        const float rot = hh * w;
        if (rot * rot > B2_MAX_ROTATION_SQ) w *= B2_MAX_ROTATION / fabsf(rot);
        cx += hh * vx; cy += hh * vy; ca += hh * w;
    }
    x = cx; y = cy; a = ca;
}

// atomic exchange on a 16-bit LDS cell (LDS atomics are 32-bit: compare-and-swap on the containing word)
__device__ __forceinline__ unsigned kb_exch16(unsigned short *base, int idx, unsigned val) {
    unsigned *word = reinterpret_cast<unsigned *>(base) + (idx >> 1);
    const int sh = (idx & 1) * 16;
    unsigned old = lds_load_relaxed(word);
    for (;;) {
        const unsigned want = (old & ~(0xFFFFu << sh)) | (val << sh);
        const unsigned seen = atomicCAS(word, old, want);
        if (seen == old) break;
        old = seen;
    }
    return (old >> sh) & 0xFFFFu;
}

// IR-range neighbour sensing off the cell lists of the broadphase: kilobot a walks the half stencil of reach s
// (own cell: partners with a higher id; the cells to the east in its row; every cell of the s rows above), so that every
// pair is met exactly once, and both ends of a pair within range are counted.  cnt16: u16 counters packed in pairs
// (LDS atomics are 32-bit; a count is at most N - 1 < 65536, so no carry crosses the halves); zeroed by the caller.
// Out of line: the kernels that never sense do not pay registers for it.
__device__ __noinline__ void kb_sense_pass(const float2 *pos, const unsigned short *head, const unsigned short *nextb,
                                           const unsigned short *cellOf, unsigned *cnt16, int N, int nt, int tid,
                                           int gw, int gh, int s, float R2, int hmask) {
    constexpr int W = 8;      // list heads fetched together (one LDS round trip); most cells are empty
    for (int a = tid; a < N; a += nt) {
        const int cell = cellOf[a];
        const int cx = cell % gw, cy = cell / gw;
        const float2 pa = pos[a];
        unsigned mine = 0;
        for (int dy = 0; dy <= s; ++dy) {
            const int oy = cy + dy;
            if (oy >= gh) break;
            const int x0 = dy == 0 ? cx : max(cx - s, 0), x1 = min(cx + s, gw - 1);
            for (int xb = x0; xb <= x1; xb += W) {
                unsigned cur[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {      // (hmask: the heads are a hash table of the cells, the walk checks the candidate's cell)
                    const int c_ = oy * gw + xb + i;
                    cur[i] = xb + i <= x1 ? (unsigned)head[hmask ? (c_ & hmask) : c_] : (unsigned)EMPTY16;
                }
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const bool own = dy == 0 && xb + i == cx;
                    for (unsigned b = cur[i]; b != (unsigned)EMPTY16;) {
                        const float2 pb = pos[b];
                        const unsigned nb = nextb[b];
                        if (!(own && (int)b <= a) && (!hmask || (int)cellOf[b] == oy * gw + xb + i)) {
                            const float ex = pb.x - pa.x, ey = pb.y - pa.y;
                            const float dd = ex * ex + ey * ey;
                            if (!(dd > R2)) {
                                mine++;
                                atomicAdd(&cnt16[b >> 1], 1u << (16 * (b & 1u)));
                            }
                        }
                        b = nb;
                    }
                }
            }
        }
        if (mine) atomicAdd(&cnt16[a >> 1], mine << (16 * (a & 1)));
    }
}

// ---- counter-based random numbers for kb_reset: Philox4x32-10 (Salmon et al., SC'11), Cephes logf ------------------
struct U4 { unsigned x, y, z, w; };
__host__ __device__ inline unsigned kb_mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32); }
__host__ __device__ inline U4 kb_philox4x32_10(U4 c, unsigned k0, unsigned k1) {
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = kb_mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = kb_mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        U4 n;
        n.x = hi1 ^ c.y ^ k0; n.y = lo1; n.z = hi0 ^ c.w ^ k1; n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
// natural logarithm of a positive normal float: Cephes single-precision algorithm (frexp by bit manipulation, degree-8
// minimax polynomial); own implementation so that the device and the oracle give the same bits.
__device__ __forceinline__ float kb_logf(float xx) {
    unsigned bits = __float_as_uint(xx);
    int e = (int)((bits >> 23) & 255u) - 126;
    float x = __uint_as_float((bits & 0x807FFFFFu) | 0x3F000000u);     // mantissa in [0.5, 1)
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

__device__ __forceinline__ int dir_dx(int k) { return (k == 1 || k == 3) ? 1 : (k == 4 ? -1 : 0); }
__device__ __forceinline__ int dir_dy(int k) { return (k >= 2) ? 1 : 0; }

// exclusive scan of NP (multiple of 4, <= 4 * blockDim.x) u8 counts into u16 offsets; returns the total.
// All threads must call; contains two workgroup barriers.
// (tid: the caller's copy of threadIdx.x -- inside the substep loop an opaque one, so that nothing derived from it is hoisted
//  out of the loop and kept in spilled registers)
__device__ __forceinline__ unsigned block_scan_u8(const unsigned char *cnt, unsigned short *off, int NP, unsigned *wsum, int tid) {
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const bool in = 4 * tid < NP;
    const unsigned c4 = in ? *reinterpret_cast<const unsigned *>(cnt + 4 * tid) : 0u;
    const unsigned c0 = c4 & 255u, c1 = (c4 >> 8) & 255u, c2 = (c4 >> 16) & 255u, c3 = c4 >> 24;
    const unsigned sum = c0 + c1 + c2 + c3;
    const unsigned incl = wave_incl_scan(sum);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < nw; ++w) { const unsigned s = wsum[w]; if (w < wave) base += s; total += s; }
    if (in) {
        const unsigned r0 = base + incl - sum, r1 = r0 + c0, r2 = r1 + c1, r3 = r2 + c2;
        reinterpret_cast<unsigned *>(off + 4 * tid)[0] = r0 | (r1 << 16);
        reinterpret_cast<unsigned *>(off + 4 * tid)[1] = r2 | (r3 << 16);
    }
    __syncthreads();
    return total;
}



// ---- sorted bins: inclusive scan of the bin counters ------------------------------------------------------------------
// E1[0] = 0, E1[i] (i >= 1) = kilobots counted into bin i - 1; afterwards E1[i] = kilobots in bins < i, i.e. bin b occupies the
// slots [E1[b], E1[b + 1]).  nchunks chunks of 8 u16 entries (16 bytes, padding zero); thread t scans `per` consecutive chunks.
// All threads must call; two workgroup barriers inside.
template <bool ONE>
__device__ __forceinline__ void block_scan_bins(unsigned short *E1, int nchunks, int per, unsigned *wsum, int tid) {
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int c0 = ONE ? tid : tid * per, c1 = ONE ? min(tid + 1, nchunks) : min(c0 + per, nchunks);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    unsigned sum = 0;
    for (int c = c0; c < c1; ++c) {
        v = reinterpret_cast<const uint4 *>(E1)[c];
        sum += (v.x & 0xFFFFu) + (v.x >> 16) + (v.y & 0xFFFFu) + (v.y >> 16) + (v.z & 0xFFFFu) + (v.z >> 16) + (v.w & 0xFFFFu) + (v.w >> 16);
    }
    const unsigned incl = wave_incl_scan(sum);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int w = 0; w < nw; ++w) { const unsigned s = wsum[w]; if (w < wave) run += s; }
    for (int c = c0; c < c1; ++c) {
        if (!ONE) v = reinterpret_cast<const uint4 *>(E1)[c];
        uint4 o;
        unsigned a, b;
        a = run + (v.x & 0xFFFFu); b = a + (v.x >> 16); o.x = a | (b << 16);
        a = b + (v.y & 0xFFFFu); b = a + (v.y >> 16); o.y = a | (b << 16);
        a = b + (v.z & 0xFFFFu); b = a + (v.z >> 16); o.z = a | (b << 16);
        a = b + (v.w & 0xFFFFu); b = a + (v.w >> 16); o.w = a | (b << 16);
        run = b;
        reinterpret_cast<uint4 *>(E1)[c] = o;
    }
    __syncthreads();
}

// IR-range neighbour sensing off the sorted bins (arena-sized grid, no hashing): the kilobots of the cells x0 .. x1 of a
// grid row are ONE run of slots, two loads per row.  Slot sa walks the half stencil of reach s (its own row: the slots
// behind it up to the end of cell cx + s; the s rows above: cells cx - s .. cx + s), meets every pair once and counts
// both ends.  cnt16: u16 counters per slot packed in pairs (zeroed by the caller).
__device__ __noinline__ void kb_sense_bins(const float2 *pos, const unsigned short *E1, unsigned *cnt16, int sa, int cx, int cy,
                                           int gw, int gh, int s, float R2) {
    const float2 pa = pos[sa];
    unsigned mine = 0;
    for (int dy = 0; dy <= s; ++dy) {
        const int oy = cy + dy;
        if (oy >= gh) break;
        const int x0 = max(cx - s, 0), x1 = min(cx + s, gw - 1);
        int lo = dy == 0 ? sa + 1 : (int)E1[oy * gw + x0];
        const int hi = (int)E1[oy * gw + x1 + 1];
        for (int b = lo; b < hi; ++b) {
            const float2 pb = pos[b];
            const float ex = pb.x - pa.x, ey = pb.y - pa.y;
            const float dd = ex * ex + ey * ey;
            if (!(dd > R2)) {
                mine++;
                atomicAdd(&cnt16[b >> 1], 1u << (16 * (b & 1)));
            }
        }
    }
    if (mine) atomicAdd(&cnt16[sa >> 1], mine << (16 * (sa & 1)));
}
// the same over hashed bins (sparse swarms): cell by cell, every candidate checked against the cell it must be in
__device__ __noinline__ void kb_sense_bins_hashed(const float2 *pos, const unsigned short *E1, const unsigned short *cellOfSlot, unsigned *cnt16,
                                                  int sa, int cx, int cy, int gw, int gh, int s, float R2, int hmask) {
    const float2 pa = pos[sa];
    unsigned mine = 0;
    for (int dy = 0; dy <= s; ++dy) {
        const int oy = cy + dy;
        if (oy >= gh) break;
        const int x0 = dy == 0 ? cx : max(cx - s, 0), x1 = min(cx + s, gw - 1);
        for (int ox = x0; ox <= x1; ++ox) {
            const int cell = oy * gw + ox, bin = cell & hmask;
            const bool own = dy == 0 && ox == cx;
            const int lo = own ? sa + 1 : (int)E1[bin], hi = (int)E1[bin + 1];
            for (int b = lo; b < hi; ++b) {
                if ((int)cellOfSlot[b] != cell) continue;
                const float2 pb = pos[b];
                const float ex = pb.x - pa.x, ey = pb.y - pa.y;
                const float dd = ex * ex + ey * ey;
                if (!(dd > R2)) {
                    mine++;
                    atomicAdd(&cnt16[b >> 1], 1u << (16 * (b & 1)));
                }
            }
        }
    }
    if (mine) atomicAdd(&cnt16[sa >> 1], mine << (16 * (sa & 1)));
}

typedef void (*kb_step_fn)(const Params);
constexpr int KB_PICK_SLEEP = 16;      // kb_pick_*(..., objects | KB_PICK_SLEEP): the instantiation with the sleep state
constexpr int KB_PICK_FIXED_1024 = -1024, KB_PICK_FIXED_1024_SENSE = -1025;   // ... the same with the neighbour-sensing hook   // kb_pick_velocity: the num_bots == 1024 specialisations (no light; without / with objects)
// one translation unit per drive law (kb_inst_d*.hip) instantiates its kernels and hands out the right one
kb_step_fn kb_pick_velocity(int light_type, int objects);   // objects: 0 none, 1 yes, 2 yes + one-wave workgroup
kb_step_fn kb_pick_velocity_discs(int light_type, int objects);   // objects: 5 discs, 6 discs + one-wave workgroup
kb_step_fn kb_pick_accel(int light_type, int objects);
kb_step_fn kb_pick_accel_discs(int light_type, int objects);   // objects: 0 none, 1 yes, 2 yes + one-wave workgroup
kb_step_fn kb_pick_motors(int light_type, int objects);
kb_step_fn kb_pick_motors_discs(int light_type, int objects);   // objects: 0 none, 1 yes, 2 yes + one-wave workgroup
kb_step_fn kb_pick_simple_phototaxis(int light_type, int objects);
kb_step_fn kb_pick_simple_phototaxis_discs(int light_type, int objects);   // objects: 0 none, 1 yes, 2 yes + one-wave workgroup
kb_step_fn kb_pick_phototaxis(int light_type, int objects);
kb_step_fn kb_pick_phototaxis_discs(int light_type, int objects);   // objects: 0 none, 1 yes, 2 yes + one-wave workgroup
kb_step_fn kb_pick_mixed(int light_type, int sleep);                // KB_DRIVE_MIXED: one-wave workgroups (kb_inst_d5.hip)
kb_step_fn kb_pick_mixed_large(int light_type, int sleep);          // ... beyond 128 kilobots: the full workgroup at 256 VGPRs (kb_inst_d5w.hip)

}  // namespace kb
