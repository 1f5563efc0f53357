// kb_step_kernel.h -- the world-step kernel template (see kb_common.h for the overview).
#pragma once
#include "kb_common.h"
#include "kb_objects.h"

namespace kb {

// One instantiation per (drive law, light model): keeps only that law's code (and registers) in the kernel.
// FN > 0: specialisation for num_bots == FN and the full workgroup (64 * KB_MAX_WAVES threads): every LDS offset, array
// size and trip count is a compile-time constant instead of a value kept in scalar registers (with objects the size
// of the contact staging area, and so the offsets behind it, stay run-time values: they depend on the fixture count).
// WIDE: instantiation for one-wave workgroups (swarms of up to 64 kilobots) in scenes with objects: launch bounds of two
// waves per SIMD give it 256 VGPRs, i.e. no register spills at all, where the LDS footprint of such scenes allows at most
// 7 - 8 envs per CU anyway (and spill code is where the toolchain bug of DESIGN.md "Robustness" lives).  Measured: + 13 %
// at 64 kilobots + 4 boxes; two-wave workgroups would lose a third of their residency, so they keep the 128-register
// instantiation.
// POLY = false: instantiation for scenes whose objects are all discs (BASELINE config 4): the kilobot - polygon contact
// code and its per-slot registers fold away (half the register spills, + 9 % at cfg4).
// SENSE = false: instantiation without the IR-range neighbour sensing hook (the fixed-size kernels are at their register
// budget: the hook costs them 2 more spilled VGPRs, 20 B/lane of scratch traffic per launch); kb_step picks it when
// kb_config.sense_radius == 0.
// Kernels without objects use the compact LDS image (namespace ldsc).  TIER picks the register budget:
//   0: 128 VGPRs (launch bounds of 4 waves per SIMD);
//   1: "WIDE", one-wave workgroups in scenes with objects: 256 VGPRs (2 waves per SIMD), no spills;
//   2: 80 VGPRs (6 waves per SIMD): three 8-wave / six 4-wave workgroups per CU where the LDS image admits them;
//   3: 256 VGPRs for the full workgroup (2 waves per SIMD = one 8-wave workgroup per CU), no spills: mixed drive laws beyond 128 kilobots.
// The fixed-size kernel without objects is always tier 2.
// SLEEP = true: b2World(doSleep=True) of kilobots_env.py:45 -- bodies carry b2Body::m_sleepTime, islands without an awake
// body are not solved (b2World::Solve), islands at rest for b2_timeToSleep whose position constraints converged fall asleep
// (b2Island::Solve).  Generic kernels and the fixed-size one without objects; kb_step picks it when kb_config.allow_sleep != 0.
template <int DRIVE_MODE, int LIGHT_TYPE, bool OBJ, int FN = 0, int TIER = 0, bool POLY = true, bool SENSE = true, bool SLEEP = false>
__global__ void __launch_bounds__(TIER == 1 ? 64 : 64 * KB_MAX_WAVES, (TIER == 1 || TIER == 3) ? 2 : ((TIER == 2 || (FN != 0 && !OBJ)) ? KB_COMPACT_WAVES_PER_SIMD : KB_MIN_WAVES_PER_SIMD)) kb_step_kernel(const Params p) {
    constexpr bool WIDE = TIER == 1 || TIER == 3;       // (256 VGPRs: no register spills)
    constexpr bool BINS = !OBJ;          // sorted-bin broadphase, bodies in slot order, contacts at their warm-start position (namespace ldsb)
    constexpr int KRX = BINS ? KB_KREG_BINS : KREG;      // contacts per lane of the register-resident solver
    static_assert(!SLEEP || FN == 0 || !OBJ, "the fixed-size instantiations with objects do not carry the sleep state");
    extern __shared__ __align__(16) unsigned char smem[];
    int e = blockIdx.x;
    int tid = threadIdx.x;
    const int nt = FN ? 64 * KB_MAX_WAVES : (int)blockDim.x;
    int lane = tid & 63, wave = tid >> 6;
    // the wave's index inside the workgroup in a scalar register: with it the thread index can be re-made at every phase boundary
    // (KB_RETID) from the lane count, without the hardware's copy in v0 -- which otherwise lives across the whole kernel, in the
    // kernels with little room in scratch memory (one exposed reload per phase)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int nw = nt >> 6;
    const int nsolve = BINS ? (nw >= KB_NSOLVE_DIV ? nw / KB_NSOLVE_DIV : 1) : nw;      // waves that sweep the contacts (kb_regsolve_bins.inc: the solver is issue-bound, fewer waves fill their lanes better)
    const int N = FN ? FN : p.N, NP = FN ? ((FN + 3) & ~3) : p.NP, S = p.S;
    size_t o = (size_t)e * N;
    size_t wo = (size_t)e * p.cap;       // this env's slice of the packed warm-start / scratch arrays
    const float h = p.h;

    const int NB = FN ? ((FN + 3) & ~3) + KB_MAX_OBJECTS + 4 : p.NB;
    const int capL_ = (BINS && FN != 0) ? ldsb::CAPL : p.capL;
    // sparse swarms: the bins are a hash of the cells (bin = cell & hmask, kb_create: a power of two >= 2 N) instead of one
    // bin per cell of the arena; a bin may then hold kilobots of several cells and every candidate's cell is checked.
    // The set of contacts and their canonical order (which use the cell coordinates) do not change.
    const bool hashed = FN == 0 && p.hmask != 0;
    // LDS arrays (offsets: namespace lds / ldsb in kb_common.h)
    // (the fixed-size instantiations with objects keep room for the object tables: all their offsets are compile-time constants)
    const int fx = lds::fixed(OBJ || FN != 0, nw), ot_ = lds::objtab(nw);
    // positions, velocities and start-of-substep positions as (x, y) pairs: one 8-byte LDS access per body
    float2 *pos = (float2 *)(smem + (BINS ? ldsb::pos() : lds::body32(fx, NB, 0))), *vel = (float2 *)(smem + (BINS ? ldsb::vel(NB) : lds::body32(fx, NB, 2)));
    float2 *start = (float2 *)(smem + lds::body32(fx, NB, 4));      // (not in the sorted-bin image: startX / startY below)
    unsigned *dirCnt = (unsigned *)(smem + (BINS ? ldsb::dircnt(NB) : lds::body32(fx, NB, 6))), *parent = (unsigned *)(smem + (BINS ? ldsb::parent(NB) : lds::body32(fx, NB, 7)));
    unsigned *islCnt = dirCnt;  // alias: dirCnt is dead once the contacts are emitted
    unsigned *lPair = (unsigned *)(smem + (BINS ? ldsb::con32(NB, NP, hashed, capL_, 0) : lds::con32(fx, NB, capL_, 0))), *lInfo = (unsigned *)(smem + (BINS ? ldsb::con32(NB, NP, hashed, capL_, 1) : lds::con32(fx, NB, capL_, 1)));
    float *lAcc = (float *)(smem + (BINS ? ldsb::con32(NB, NP, hashed, capL_, 2) : lds::con32(fx, NB, capL_, 2)));
    float *oldAcc = (float *)(smem + (BINS ? ldsb::oldacc(NB, NP, hashed, capL_) : lds::con32(fx, NB, capL_, 3)));
    unsigned short *lCbk = (unsigned short *)(smem + (BINS ? ldsb::cbk(NB, NP, hashed, capL_) : lds::con16(fx, NB, capL_, 0)));
    unsigned short *lOrder = (unsigned short *)(smem + (BINS ? ldsb::order(NB, NP, hashed, capL_) : lds::con16(fx, NB, capL_, 1)));
    unsigned short *oldKey = (unsigned short *)(smem + (BINS ? ldsb::oldkey(NB, NP, hashed, capL_) : lds::con16(fx, NB, capL_, 2)));
    unsigned short *wsOff = (unsigned short *)(smem + (BINS ? ldsb::wsoff(NB) : lds::bot16(fx, NB, capL_, NP, 0))), *newOff = (unsigned short *)(smem + (BINS ? ldsb::newoff(NB, NP, hashed) : lds::bot16(fx, NB, capL_, NP, 1)));
    unsigned short *nextb = (unsigned short *)(smem + lds::bot16(fx, NB, capL_, NP, 2)), *cellOf = (unsigned short *)(smem + lds::bot16(fx, NB, capL_, NP, 3));   // (cell lists: kernels with objects only)
    unsigned char *wsCnt = smem + (BINS ? ldsb::wscnt(NB, NP) : lds::bot8(fx, NB, capL_, NP, 0)), *wsCntNew = smem + (BINS ? ldsb::wscntnew(NB, NP, hashed) : lds::bot8(fx, NB, capL_, NP, 1));
    unsigned char *active = smem + (BINS ? ldsb::act0(NB, NP, hashed, capL_) : lds::active(fx, NB, capL_, NP));
    unsigned char *islWave = smem + (BINS ? ldsb::islwave(NB, NP, hashed, capL_) : lds::islwave(fx, NB, capL_, NP));   // wave that sweeps the island rooted at body b
    unsigned short *head = (unsigned short *)(smem + lds::head(fx, NB, capL_, NP));   // per-cell list heads (EMPTY16 = empty; kernels with objects only)
    // sorted-bin image: slot -> kilobot id, cell of a slot (hashed bins), bin boundaries, arrival-order scratch of the sort
    unsigned short *idOf = (unsigned short *)(smem + ldsb::idof(NB, NP)), *cellOfSlot = (unsigned short *)(smem + ldsb::cellofslot(NB, NP));
    unsigned short *E1 = (unsigned short *)(smem + ldsb::binE(NB, NP, hashed, capL_));
    unsigned short *tmpSort = reinterpret_cast<unsigned short *>(lInfo);
    // ... start-of-substep positions over arrays that are dead from the integration on
    float *startX = reinterpret_cast<float *>(smem + ldsb::botA(NB)), *startY = reinterpret_cast<float *>(dirCnt);
    unsigned *misc = (unsigned *)(smem + lds::MISC), *wsum = (unsigned *)(smem + lds::WSUM);
    // (sorted-bin image: the bucket tables lie over the bin boundaries, which are dead between the emit pass and the next substep)
    const int tb_ = BINS ? ldsb::binE(NB, NP, hashed, capL_) - lds::BKSTART : 0;
    unsigned *bkStart = (unsigned *)(smem + tb_ + lds::BKSTART), *bkFill = (unsigned *)(smem + tb_ + lds::bkfill(nw));
    unsigned *bkMaxRank = (unsigned *)(smem + tb_ + lds::bkmaxrank(nw));
    unsigned short *bkList = (unsigned short *)(smem + tb_ + lds::bklist(nw));
    unsigned char *nList = smem + tb_ + lds::nlist(nw);
    float *objTab = (float *)(smem + ot_);                       // fixture table (kb_objects.h: OT_*)
    float *objBody = (float *)(smem + ot_ + lds::OBJBODY);                     // body table (kb_objects.h: BT_*)
    unsigned *objCnt = (unsigned *)(smem + ot_ + lds::OBJCNT);                 // kilobots touching object m
    unsigned short *objList = (unsigned short *)(smem + ot_ + lds::OBJLIST);   // ... and who they are
    float *objW = (float *)(smem + ot_ + lds::OBJW), *objA = (float *)(smem + ot_ + lds::OBJA), *objA0 = (float *)(smem + ot_ + lds::OBJA0);
    unsigned long long *mcMask = (unsigned long long *)(smem + ot_ + lds::MCMASK);   // manifold constraints owned by wave w
    float *objSlp = (float *)(smem + ot_ + lds::OBJSLP);                              // sleep time of object m (< 0: asleep)
    const int M = OBJ ? p.M : 0;   // OBJ = false: every object loop below folds away
    // MIX: KB_DRIVE_MIXED, any mix of the five drive laws in one env (the reference steps whatever is in _kilobots,
    // kilobots_env.py:183-184): the law of every kilobot comes from kb_buffers.bot_mode, its inverse mass (the classes have
    // different densities, kilobot.py:25 / :214) from p.im_mode[law].  Instantiated on the code path with objects (per-body
    // masses in the contacts' registers), one-wave workgroups.
    constexpr bool MIX = DRIVE_MODE == KB_DRIVE_MIXED;
    static_assert(!MIX || (OBJ && FN == 0), "mixed drive laws run on the generic code path with per-body masses");
    unsigned char *botLaw = smem + (MIX ? p.botlaw_off : 0);     // MIX: [NP] law of every kilobot of the env (masses of contact partners)
#define KB_IM_BOT(a_) (MIX ? p.im_mode[botLaw[a_]] : p.im_bot)     // inverse mass of kilobot a_
    // inverse mass / radius of a body id: kilobot < N, object N + m, wall >= WALL_CODE (static, edge skin radius)
    auto bim = [&](int id) __attribute__((always_inline)) -> float {
        return id >= WALL_CODE ? 0.0f : ((!OBJ || id < N) ? KB_IM_BOT(id) : objBody[(id - N) * BT_WORDS + BT_IM]);
    };
    auto brad = [&](int id) __attribute__((always_inline)) -> float {
        return id >= WALL_CODE ? B2_POLYGON_RADIUS : ((!OBJ || id < N) ? p.r_bot : objBody[(id - N) * BT_WORDS + BT_RADIUS]);
    };
    // is body id a polygon object (kilobot - polygon contacts carry a lever arm on the object)
    auto bpoly = [&](int id) __attribute__((always_inline)) -> bool {
        return OBJ && POLY && id >= N && id < WALL_CODE && objBody[(id - N) * BT_WORDS + BT_KIND] != 0.0f;
    };
    ObjCtx ox;
    ox.pos = pos; ox.vel = vel; ox.objW = objW; ox.objA = objA; ox.objTab = objTab; ox.objBody = objBody;
    ox.mc = (float *)(smem + lds::mcarea(fx, NB, capL_, NP, p.nhead));
    ox.N = N; ox.MCN = p.nmc; ox.mu_oo = p.mu_oo; ox.mu_ow = p.mu_ow;
    const int NMC = OBJ ? p.nmc : 0;      // manifold-constraint candidates (fixture pairs, fixture-wall)
    const int F = OBJ ? p.F : 0;          // fixtures of the objects (>= M)

    const kb_buffers &g = p.buf;
    // (kernels with objects: the cell heads are a hash table of the cells for sparse swarms, like the bins)
    auto hix = [&](int cell_) __attribute__((always_inline)) -> int { return hashed ? (cell_ & p.hmask) : cell_; };
    // contact staging in global scratch, used when an env has more contacts than fit the LDS staging area
    unsigned *gPair, *gInfo;
    float *gAcc;
    unsigned short *gCbk, *gOrder;
    // Global addresses are re-derived from an opaque copy of the block index at the top of every substep and in
    // the epilogue: otherwise the compiler keeps the prologue's per-lane 64-bit addresses alive (and spilled to
    // scratch memory) across the whole substep loop just to reuse them for the final stores.
#define KB_ENV_ADDRESSES()                                                                          \
    do {                                                                                            \
        int e_ = blockIdx.x;                                                                        \
        asm volatile("" : "+s"(e_));                                                                \
        e = e_; o = (size_t)e * N; wo = (size_t)e * p.cap;                                          \
        gPair = reinterpret_cast<unsigned *>(g.scratch) + wo * 4;                                   \
        gInfo = gPair + p.cap;                                                                      \
        gAcc = reinterpret_cast<float *>(gInfo + p.cap);                                            \
        gCbk = reinterpret_cast<unsigned short *>(gAcc + p.cap); gOrder = gCbk + p.cap;             \
    } while (0)
    KB_ENV_ADDRESSES();

#ifdef KB_PROFILE
    static_assert(M_COUNT <= 24, "profile words start at misc[24]");
    constexpr int M_PROF_DEPTH = 12;
    if (tid < 40) misc[24 + tid] = 0;
    __syncthreads();
    unsigned prof_t = (unsigned)clock64();
#endif

    // ---- load state; optional fused set_action (kilobot.py:235-241, 283-289) ----
    float th[BPT], bw[BPT], cv[BPT], cw[BPT], av[BPT], aw[BPT];
    float sth0[BPT];     // angle at the start of the substep (continuous step against the walls)
    float slp[BPT];      // SLEEP: b2Body::m_sleepTime; < 0: the kilobot is asleep
    int law[BPT];        // drive law of this thread's kilobots
    int ms[BPT];         // sorted-bin image: the slot this thread's kilobots live in (the kilobot's id until the first sort)
#define KB_LAW(q) (MIX ? law[q] : DRIVE_MODE)
    // Velocity control, sorted-bin image: the command (v, omega) of a kilobot waits for the drive phase in the LDS velocity array
    // (idle between the substeps, indexed by id) instead of in two registers that the allocator would keep in scratch memory
    // from the kernel start on; between fused substeps it is read again from g.v / g.w, where the kernel start left it.
    constexpr bool PARK = BINS && !MIX && DRIVE_MODE == KB_DRIVE_VELOCITY;
#define KB_SLOT(q, b_) (BINS ? ms[q] : (b_))     // index of kilobot b_ (= tid + q * nt) in the per-body LDS arrays
    // Every load of the kernel start is issued before the first store: the set_action stores (g.v / g.w) may alias anything as far
    // as the compiler can tell, and loads that follow them in program order would wait for the previous kilobot's round trip to
    // HBM -- four dependent trips per thread (state, action, state, action) instead of one.
    float ldx[BPT], ldy[BPT];
    unsigned ldc[BPT];
    float2 lda[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = tid + q * nt;
        th[q] = 0.0f; bw[q] = 0.0f; cv[q] = 0.0f; cw[q] = 0.0f; av[q] = 0.0f; aw[q] = 0.0f; slp[q] = 0.0f; law[q] = DRIVE_MODE;
        float z_ = 0.0f;
        asm volatile("" : "+v"(z_));     // (opaque: keeps the clamping of the action behind all loads instead of behind its own)
        ldx[q] = 0.0f; ldy[q] = 0.0f; ldc[q] = 0u; lda[q] = make_float2(z_, z_);
        ms[q] = b < N ? b : NB - 1;
        if (b < N) {
            ldx[q] = g.x[o + b]; ldy[q] = g.y[o + b]; th[q] = g.theta[o + b];
            if (MIX) law[q] = min((int)g.bot_mode[o + b], 4);
            if (SLEEP) slp[q] = g.sleep_time[o + b];
            ldc[q] = g.ws_cnt[o + b];
            // (a velocity action replaces the stored command: nothing to load then)
            const bool velmode = KB_LAW(q) == KB_DRIVE_VELOCITY || KB_LAW(q) == KB_DRIVE_ACCEL;
            if (velmode && !(KB_LAW(q) == KB_DRIVE_VELOCITY && p.actions)) { cv[q] = g.v[o + b]; cw[q] = g.w[o + b]; }
            if (KB_LAW(q) == KB_DRIVE_ACCEL) { av[q] = g.acc_v[o + b]; aw[q] = g.acc_w[o + b]; }
            if (p.actions && velmode) lda[q] = reinterpret_cast<const float2 *>(p.actions)[o + b];
        }
    }
    // Sorted-bin image: the packed warm-start list of the previous launch is requested together with the state (the same trip
    // to HBM); how many of its entries exist is only known behind the offset scan below.
    unsigned pfKey[2] = {0u, 0u};
    float pfAcc[2] = {0.0f, 0.0f};
    if (BINS && p.n_substeps > 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + q * nt;
            if (i < capL_ && i < p.cap) { pfKey[q] = g.ws_key[wo + i]; pfAcc[q] = g.ws_acc[wo + i]; }
        }
    }
    int hasAct = __builtin_amdgcn_readfirstlane((int)(p.actions != nullptr));
    asm volatile("" : "+s"(hasAct));      // (opaque: the two "is there an action" branches must not be merged again)
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = tid + q * nt;
        if (b < N) {
            pos[b] = make_float2(ldx[q], ldy[q]);
            if (MIX) botLaw[b] = (unsigned char)law[q];
            wsCnt[b] = (unsigned char)ldc[q];
            if (hasAct) {
                const float2 a = lda[q];
                const float mw = 0.5f * 3.14159265358979323846f;
                if (KB_LAW(q) == KB_DRIVE_VELOCITY) {
                    cv[q] = fmaxf(fminf(a.x, 0.01f), 0.0f);
                    cw[q] = fmaxf(fminf(a.y, mw), -mw);
                    g.v[o + b] = cv[q]; g.w[o + b] = cw[q];
                } else if (KB_LAW(q) == KB_DRIVE_ACCEL) {
                    const float aw_ = 0.2f * 3.14159265358979323846f;
                    av[q] = fmaxf(fminf(a.x, 0.005f), -0.005f);
                    aw[q] = fmaxf(fminf(a.y, aw_), -aw_);
                    g.acc_v[o + b] = av[q]; g.acc_w[o + b] = aw[q];
                }
            }
            if (PARK) vel[b] = make_float2(cv[q], cw[q]);
        }
    }
    for (int b = N + tid; b < NP; b += nt) { wsCnt[b] = 0; wsCntNew[b] = 0; }
    // pushable objects: pose and velocity live in LDS (pos / vel / objA / objW), thread m integrates object m
    if (tid < M) {
        const size_t oi = (size_t)e * M + tid;
        pos[N + tid].x = g.ox[oi]; pos[N + tid].y = g.oy[oi]; vel[N + tid].x = g.ovx[oi]; vel[N + tid].y = g.ovy[oi];
        objA[tid] = g.otheta[oi]; objW[tid] = g.ow[oi];
        if (SLEEP) objSlp[tid] = g.osleep[oi];
        const float lx = p.obody[tid][BT_LCX], ly = p.obody[tid][BT_LCY];
        if (lx != 0.0f || ly != 0.0f) {    // the state holds the body origin, the solver works on the centre of mass
            const XF t = xf_make(g.ox[oi], g.oy[oi], g.otheta[oi]);
            const V2 cm = xf_mul(t, mk2(lx, ly));
            pos[N + tid].x = cm.x; pos[N + tid].y = cm.y;
        }
    }
    for (int k = tid; k < F * OT_WORDS; k += nt) objTab[k] = p.otab[k / OT_WORDS][k % OT_WORDS];
    for (int k = tid; k < M * BT_WORDS; k += nt) objBody[k] = p.obody[k / BT_WORDS][k % BT_WORDS];
    // manifold-constraint candidate t (object pair / object-wall) is looked after by lane t of wave 0
    bool mcTouch = false;
    float *owsMine = nullptr;
    if (BINS) { for (int c = tid; c < ldsb::bin_entries(p.nhead) / 2; c += nt) reinterpret_cast<unsigned *>(E1)[c] = 0u; }
    else for (int c = tid; c < p.nhead; c += nt) head[c] = EMPTY16;
    if (tid == 0) misc[M_STATUS] = 0;
    float lx = 0.0f, ly = 0.0f;
    if (LIGHT_TYPE == KB_LIGHT_CIRCULAR) { lx = g.light_x[e]; ly = g.light_y[e]; }
    // general light model: state of every component in registers (wave-uniform values)
    constexpr bool LGEN = LIGHT_TYPE == KB_LIGHT_GENERAL;
    float glx[KB_MAX_LIGHTS], gly[KB_MAX_LIGHTS], glvx[KB_MAX_LIGHTS], glvy[KB_MAX_LIGHTS];
#pragma unroll
    for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
        glx[i] = 0.0f; gly[i] = 0.0f; glvx[i] = 0.0f; glvy[i] = 0.0f;
        if (LGEN && i < p.lcount) {
            glx[i] = g.light_x[(size_t)e * p.lcount + i];
            if (p.light_type != KB_LIGHT_GRADIENT) gly[i] = g.light_y[(size_t)e * p.lcount + i];
            if (p.lkind[i] == KB_LIGHT_MOMENTUM) { glvx[i] = g.light_vx[(size_t)e * p.lcount + i]; glvy[i] = g.light_vy[(size_t)e * p.lcount + i]; }
        }
    }
    // value_and_gradients of the general model at one sensor position (metres): kb_common.h
    auto sense_general = [&](float sx, float sy, float &val, float &gx, float &gy) __attribute__((always_inline)) {
        kb_light_general_sense(p, glx, gly, sx, sy, val, gx, gy);
    };
    const bool drive = !(p.flags & KB_STEP_NO_DRIVE);
    const float rr2 = p.rr2, rw2 = p.rw2;     // (r + r)^2, (polygonRadius + r)^2: host-evaluated, scalar registers
    __syncthreads();
    // warm-start list of the previous substep: offsets, and an LDS image of the packed entries if it fits
    unsigned oldTotal = block_scan_u8(wsCnt, wsOff, NP, wsum, tid);
    bool oldInLds = oldTotal <= (unsigned)capL_;
    if (BINS && oldInLds) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned i = tid + q * nt;
            if (i < oldTotal) {
                oldKey[i] = (unsigned short)(pfKey[q] >= KEY_WALL ? WALL_CODE + (pfKey[q] - KEY_WALL) : pfKey[q]);
                oldAcc[i] = pfAcc[q];
            }
        }
    }
    if (oldInLds) {
        for (unsigned i = tid + (BINS ? 2 * nt : 0); i < oldTotal; i += nt) {
            const unsigned k = g.ws_key[wo + i];
            oldKey[i] = (unsigned short)(k >= KEY_OBJ ? OBJ_CODE + (k - KEY_OBJ) : (k >= KEY_WALL ? WALL_CODE + (k - KEY_WALL) : k));
            oldAcc[i] = g.ws_acc[wo + i];
        }
    }
    __syncthreads();

    KB_STAMP(13);    // kernel start: state loads, warm-start scan and list load
    for (int sub = 0; sub < p.n_substeps; ++sub) {
        KB_ENV_ADDRESSES();
        // The thread index is re-read (through an opaque copy) at the top of every substep and of every phase: addresses
        // and predicates derived from it are then computed where they are used instead of being kept alive -- in spilled
        // registers -- across the phases.  (In the 80-VGPR kernel this alone took the spills from 35 to 11 VGPRs.)
#ifdef KB_ABLATE      // measurement builds (tools/phase_ablation.py): the kernel ends behind phase KB_ABLATE without writing anything back
#define KB_ABLATE_EXIT(k_) do { if (KB_ABLATE == (k_)) return; } while (0)
#else
#define KB_ABLATE_EXIT(k_) do { } while (0)
#endif
#define KB_RETID() do { int w_ = __builtin_amdgcn_readfirstlane(wave_s); asm volatile("" : "+s"(w_));                                                                  \
                        int l_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(l_));    \
                        wave = w_; lane = l_; tid = (w_ << 6) | l_; } while (0)
        KB_RETID();
        // ---- light.step: SinglePositionLight.step, light.py:59-75 (uniform per env) ----
        if (p.light_action && LGEN && drive)
            kb_light_general_step(p, p.light_action + (size_t)e * p.ladim, h, glx, gly, glvx, glvy);
        if (p.light_action && LIGHT_TYPE == KB_LIGHT_CIRCULAR && drive)
            kb_light_single_step(p, p.light_action + 2 * e, h, lx, ly);
        // ---- sensing + drive law + damping; grid insertion; reset per-substep scratch ----
        // sorted bins: position, damped velocity, cell and arrival index in its bin of this thread's kilobots, kept in
        // registers until the kilobot's slot of this substep is known
        float sbx[BPT], sby[BPT], svx[BPT], svy[BPT];
        int scell[BPT], sarr[BPT];       // scell: cx | cy << 12 | walls touched << 24 | outside the wall line << 28
#define SC_CX(s_) ((s_) & 0xFFF)
#define SC_CY(s_) (((s_) >> 12) & 0xFFF)
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const int b = tid + q * nt;
            sbx[q] = 0.0f; sby[q] = 0.0f; svx[q] = 0.0f; svy[q] = 0.0f; scell[q] = 0; sarr[q] = 0;
            if (b >= N) continue;
            float bvx = 0.0f, bvy = 0.0f, bww = 0.0f;
            const float bx = pos[KB_SLOT(q, b)].x, by = pos[KB_SLOT(q, b)].y;
            if (!BINS) { start[b].x = bx; start[b].y = by; }
            sth0[q] = th[q];
            if (drive) {
                const float t = th[q];
                float lval = 0.0f, lgx = 0.0f, lgy = 0.0f;
                if (LIGHT_TYPE != KB_LIGHT_NONE) {
                    float sx = bx, sy = by;
                    if (KB_LAW(q) != KB_DRIVE_SIMPLE_PHOTOTAXIS) {  // kilobot.py:54-55: world point of (0, -r)
                        float s, c;
                        kb_sincosf(t, s, c);
                        const float lx0 = 0.0f, ly0 = -p.r_bot;
                        sx = (c * lx0 - s * ly0) + bx;
                        sy = (s * lx0 + c * ly0) + by;
                    }
                    if (LGEN) sense_general(sx / WORLD_SCALE, sy / WORLD_SCALE, lval, lgx, lgy);
                    else kb_light_circular(sx / WORLD_SCALE, sy / WORLD_SCALE, lx, ly, p.light_radius, lval, lgx, lgy);
                    if (g.light_value) { g.light_value[o + b] = lval; g.light_gx[o + b] = lgx; g.light_gy[o + b] = lgy; }
                }
                switch (KB_LAW(q)) {
                case KB_DRIVE_ACCEL: {  // kilobot.py:294-300
                    float v = cv[q] + av[q] * h, ww = cw[q] + aw[q] * h;
                    const float mw = 0.5f * 3.14159265358979323846f;
                    v = fminf(fmaxf(v, 0.0f), 0.01f); ww = fminf(fmaxf(ww, -mw), mw);
                    cv[q] = v; cw[q] = ww;
                    float s, c;
                    kb_sincosf(t, s, c);
                    float sp = v * WORLD_SCALE;
                    bvx = c * sp; bvy = s * sp; bww = ww;
                } break;
                case KB_DRIVE_VELOCITY: {  // kilobot.py:253-258
                    float s, c;
                    kb_sincosf(t, s, c);
                    float cvq = cv[q], cwq = cw[q];
                    if (PARK) { const float2 c_ = vel[b]; cvq = c_.x; cwq = c_.y; }
                    asm volatile("" : "+v"(cvq));     // (keeps the product below inside the substep loop: hoisted, it is one more register spilled across the launch)
                    float sp = cvq * WORLD_SCALE;
                    bvx = c * sp; bvy = s * sp; bww = cwq;
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
            if (SLEEP) {
                // kilobot.py:123-127 assigns angularVelocity / linearVelocity: b2Body::SetAngularVelocity / SetLinearVelocity
                // wake a sleeping body iff the value is non-zero (w * w > 0, b2Dot(v, v) > 0); one that stays asleep has
                // zero velocity
                if (drive && slp[q] < 0.0f && (bww * bww > 0.0f || bvx * bvx + bvy * bvy > 0.0f)) slp[q] = 0.0f;
                if (slp[q] < 0.0f) { bvx = 0.0f; bvy = 0.0f; bww = 0.0f; }
                active[b] = 0;       // "the island rooted here has an awake body": set in the flatten phase
            }
            // b2Island::Solve: v *= 1/(1 + h c)  (SimplePhototaxisKilobot sets linearDamping = 0, kilobot.py:203)
            const float kl = (KB_LAW(q) == KB_DRIVE_SIMPLE_PHOTOTAXIS) ? 1.0f / (1.0f + h * 0.0f) : p.kl_bot;
            bw[q] = bww * p.ka_bot;
            int cx = (int)floorf((bx - p.xmin) * p.inv_cell);
            int cy = (int)floorf((by - p.ymin) * p.inv_cell);
            cx = cx < 0 ? 0 : (cx >= p.gw ? p.gw - 1 : cx);
            cy = cy < 0 ? 0 : (cy >= p.gh ? p.gh - 1 : cy);
            const int cell = cy * p.gw + cx;
            if (BINS) {
                // broadphase: count the kilobot into its bin (entry bin + 1 of the boundary array; 16-bit counters, two per word)
                sbx[q] = bx; sby[q] = by; svx[q] = bvx * kl; svy[q] = bvy * kl; scell[q] = cx | (cy << 12);
                const int bi = (hashed ? (cell & p.hmask) : cell) + 1;
                const unsigned old = atomicAdd(reinterpret_cast<unsigned *>(E1) + (bi >> 1), 1u << (16 * (bi & 1)));
                sarr[q] = (int)((old >> (16 * (bi & 1))) & 0xFFFFu);
                if (SENSE && p.sense_s > 0) reinterpret_cast<unsigned short *>(lAcc)[b] = 0;   // (dead until the emit pass: the neighbour counters of the sensing pass, by slot)
            } else {
                vel[b].x = bvx * kl; vel[b].y = bvy * kl;
                parent[b] = b;
                // broadphase: push the bot on its cell's list
                cellOf[b] = (unsigned short)cell;
                nextb[b] = (unsigned short)kb_exch16(head, hix(cell), (unsigned)b);
                if (SENSE && p.sense_s > 0) newOff[b] = 0;       // (dead until the scan behind the label pass: the neighbour counters of the sensing pass)
            }
        }
        if (tid < M) { start[N + tid].x = pos[N + tid].x; start[N + tid].y = pos[N + tid].y; objA0[tid] = objA[tid]; }
        if (tid < M) {   // b2Island::Solve damping of the objects; they keep their velocity between substeps
            vel[N + tid].x *= p.kl_obj; vel[N + tid].y *= p.kl_obj; objW[tid] *= p.ka_obj;
            parent[N + tid] = N + tid;
            objCnt[tid] = 0;
            if (SLEEP) active[N + tid] = 0;
        }
        if (OBJ && tid >= M && tid < F) objCnt[tid] = 0;
        if (tid < M_COUNT && tid != M_STATUS) misc[tid] = 0;
        if (SLEEP && tid < 32) misc[M_WAKE + tid] = 0;      // kilobots woken by a contact that stopped touching (one bit each)
        KB_STAMP_PRE(16);
        __syncthreads();
        KB_STAMP(0);
        KB_RETID();
        KB_ABLATE_EXIT(1);     // drive law + bin counters
        // ---- warm start: impulse of the same pair in the previous substep (b2Contact::Update id match) ----
        auto ws_find = [&](int owner, unsigned key16) __attribute__((always_inline)) -> float {
            const int cnt = wsCnt[owner], off = wsOff[owner];
            if (oldInLds) {
                for (int s = 0; s < cnt; ++s)
                    if (oldKey[off + s] == (unsigned short)key16) {
                        const float a_ = oldAcc[off + s];
                        if (SLEEP) oldAcc[off + s] = -1.0f - a_;      // (matched: b2Contact::Update "was touching and still is"; impulses are >= 0)
                        return a_;
                    }
            } else {
                const unsigned key32 = key16 >= (unsigned)WALL_CODE ? KEY_WALL + (key16 - WALL_CODE)
                                     : (key16 >= (unsigned)OBJ_CODE ? KEY_OBJ + (key16 - OBJ_CODE) : key16);
                // (entries behind the contact capacity were never stored: an env that overflowed lists more per-bot
                //  entries than its slice holds, and the next env's slice starts right behind it)
                // four probes are requested together: one trip to L2 / HBM instead of up to four dependent ones
                for (int s0 = 0; s0 < cnt; s0 += 4) {
                    unsigned k4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = off + s0 + u;
                        k4[u] = (s0 + u < cnt && idx < p.cap) ? g.ws_key[wo + idx] : 0xFFFFFFFFu;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (k4[u] == key32) {
                            const float a_ = g.ws_acc[wo + off + s0 + u];
                            if (SLEEP) g.ws_acc[wo + off + s0 + u] = -1.0f - a_;
                            return a_;
                        }
                }
            }
            return -1.0f;   // accumulated impulses are >= 0
        };
        bool big = false;
        int stageCap = 0, ncon = 0;
        unsigned newTotal = 0;        // entries of the packed warm-start list of this substep
        if constexpr (BINS) {
            // =============== sorted-bin broadphase + contact emission (kernels without objects) ===============
            // 1. boundaries of the bins: inclusive scan of the counters (E1[b] .. E1[b + 1] = slots of bin b)
            {
                const int nchunks = ldsb::bin_entries(p.nhead) >> 3;
                if (nchunks <= nt) block_scan_bins<true>(E1, nchunks, 1, wsum, tid);
                else block_scan_bins<false>(E1, nchunks, (nchunks + nt - 1) / nt, wsum, tid);
            }
            KB_STAMP_PRE(31);    // (profile build, cumulative since the drive barrier) bin boundaries scanned
            // 2. scatter in arrival order ...
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b >= N) continue;
                const int cell = SC_CY(scell[q]) * p.gw + SC_CX(scell[q]);
                const int bin = hashed ? (cell & p.hmask) : cell;
                tmpSort[(int)E1[bin] + sarr[q]] = (unsigned short)b;
            }
            __syncthreads();
            KB_STAMP_PRE(32);    // ... + arrival-order scatter and its barrier
            KB_RETID();
            // 3. ... and settle every kilobot in its slot: inside a bin by ascending id (the canonical order of the contacts
            //    is defined on ids).  From here to the end of the substep the kilobot's body lives at index ms[q].
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b >= N) continue;
                const int cell = SC_CY(scell[q]) * p.gw + SC_CX(scell[q]);
                const int bin = hashed ? (cell & p.hmask) : cell;
                const int s0 = (int)E1[bin], s1 = (int)E1[bin + 1];
                int rank = 0;
                if (s1 - s0 > 1)
                    for (int s_ = s0; s_ < s1; ++s_) rank += ((int)tmpSort[s_] < b) ? 1 : 0;
                const int sl = s0 + rank;
                ms[q] = sl;
                idOf[sl] = (unsigned short)b;
                if (hashed) cellOfSlot[sl] = (unsigned short)cell;
                pos[sl] = make_float2(sbx[q], sby[q]);
                vel[sl] = make_float2(svx[q], svy[q]);
                parent[sl] = (unsigned)sl;
            }
            if (tid < 4) {      // the four walls as bodies at rest (kb_regsolve_bins.inc)
                float z_ = 0.0f;
                asm volatile("" : "+v"(z_));      // (a zero made here, not a register pair kept -- in scratch memory -- from the kernel start on)
                vel[NB - 5 + tid] = make_float2(z_, z_);
            }
            KB_STAMP_PRE(33);    // ... + wave 0's kilobots settled in their slots
            __syncthreads();
            KB_STAMP(0);
            KB_RETID();
            KB_ABLATE_EXIT(2);     // bins sorted
            // ---- IR-range neighbour sensing (kb_config.sense_radius): at the sensing point of the substep, like the light
            //      (kilobots_env.py:174-180), off the bins that the contact search uses ----
            if (SENSE && p.sense_s > 0 && drive) {
#pragma unroll 1
                for (int q = 0; q < BPT; ++q) {
                    const int b = tid + q * nt;
                    if (b >= N) continue;
                    const int sl = q == 0 ? ms[0] : ms[BPT - 1];
                    const int sc = q == 0 ? scell[0] : scell[BPT - 1];
                    if (hashed) kb_sense_bins_hashed(pos, E1, cellOfSlot, reinterpret_cast<unsigned *>(lAcc), sl, SC_CX(sc), SC_CY(sc), p.gw, p.gh, p.sense_s, p.sense_r2, p.hmask);
                    else kb_sense_bins(pos, E1, reinterpret_cast<unsigned *>(lAcc), sl, SC_CX(sc), SC_CY(sc), p.gw, p.gh, p.sense_s, p.sense_r2);
                }
            }
            // ---- narrowphase, pass 1 (thread per kilobot): the contacts the kilobot owns -- touching partners of the half
            //      stencil (same cell behind it, E, N, NE, NW: contiguous runs of slots) and walls ----
            // Candidate runs in enumeration order k = 0 .. 4.  Arena-sized grid: three runs, two cells of a grid row each
            // ([same | E], [N | NE], [NW]); hashed bins: one run per cell.  sp = first slot of the run's second cell.
            // Candidate runs in enumeration order k = 0 .. 4.  Arena-sized grid: three runs, two cells of a grid row each
            // ([same | E], [N | NE], [NW]); hashed bins: one run per cell.  sp = first slot of the run's second cell.
            constexpr int NR = FN ? 3 : 5;
            auto candidate_runs = [&](int sa, int cx, int cy, int (&lo)[NR], int (&hi)[NR], int (&sp)[NR], int (&tc)[5]) __attribute__((always_inline)) {
                const int gw = p.gw, gh = p.gh;
                const int cell = cy * gw + cx;
                const bool e_ok = cx + 1 < gw, n_ok = cy + 1 < gh, w_ok = cx > 0;
                tc[0] = cell; tc[1] = cell + 1; tc[2] = cell + gw; tc[3] = cell + gw + 1; tc[4] = cell + gw - 1;
                if constexpr (NR == 3) {
                    const int up = cell + gw;
                    const int e1 = (int)E1[cell + 1], e2 = (int)E1[e_ok ? cell + 2 : cell + 1];
                    const int u0 = (int)E1[n_ok ? (w_ok ? up - 1 : up) : 0], u1 = (int)E1[n_ok ? up : 0];
                    const int u2 = (int)E1[n_ok ? up + 1 : 0], u3 = (int)E1[n_ok ? (e_ok ? up + 2 : up + 1) : 0];
                    lo[0] = sa + 1; sp[0] = e1; hi[0] = e2;
                    lo[1] = u1; sp[1] = u2; hi[1] = u3;
                    lo[2] = u0; sp[2] = u1; hi[2] = u1;
                } else {
                    const bool ok[5] = {true, e_ok, n_ok, n_ok && e_ok, n_ok && w_ok};
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const int bin = hashed ? (tc[k] & p.hmask) : tc[k];
                        const int l_ = ok[k] ? (int)E1[bin] : 0, h_ = ok[k] ? (int)E1[bin + 1] : 0;
                        lo[k] = k == 0 ? sa + 1 : l_; hi[k] = h_; sp[k] = h_;
                    }
                }
            };
            // every touching partner of the kilobot at slot sa, in enumeration order: cb(slot, k)
            auto for_partners = [&](int sa, int cx, int cy, auto &&cb) __attribute__((always_inline)) {
                int lo[NR], hi[NR], sp[NR], tc[5];
                candidate_runs(sa, cx, cy, lo, hi, sp, tc);
                int pre[NR + 1];
                pre[0] = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r) pre[r + 1] = pre[r] + max(hi[r] - lo[r], 0);
                const int tot = pre[NR];
                const float2 pa = pos[sa];
#ifndef KB_FIND_BATCH
#define KB_FIND_BATCH 2
#endif
                constexpr int BATCH = KB_FIND_BATCH;      // candidates fetched together (one LDS round trip)
                for (int base = 0; base < tot; base += BATCH) {
                    int cs[BATCH], ck[BATCH];
                    float2 cp[BATCH];
                    int cc[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        const int i = base + u;
                        int sl = lo[0] + i, spl = sp[0], kb_ = 0;
#pragma unroll
                        for (int r = 1; r < NR; ++r)
                            if (i >= pre[r]) { sl = lo[r] + (i - pre[r]); spl = sp[r]; kb_ = NR == 3 ? 2 * r : r; }
                        ck[u] = kb_ + ((NR == 3 && sl >= spl) ? 1 : 0);
                        cs[u] = i < tot ? sl : sa;
                        cp[u] = pos[cs[u]];
                        cc[u] = hashed ? (int)cellOfSlot[cs[u]] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        const float dx = cp[u].x - pa.x, dy = cp[u].y - pa.y;
                        const float dd = dx * dx + dy * dy;
                        bool hit = base + u < tot && !(dd > rr2);      // b2CollideCircles
                        if (hashed) {
                            int want = tc[0];
#pragma unroll
                            for (int k = 1; k < 5; ++k) if (ck[u] == k) want = tc[k];
                            hit = hit && cc[u] == want;
                        }
                        if (hit) cb(cs[u], ck[u]);
                    }
                }
            };
            unsigned pl0[BPT], pl1[BPT];      // the first four partners of a kilobot: slot | k << 12, 16 bits each
            unsigned snb[BPT], snm[BPT];      // ... how many it has, and how many entries of the packed list it owns (walls included, capped)
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                pl0[q] = 0u; pl1[q] = 0u; snb[q] = 0u; snm[q] = 0u;
                if (b >= N) continue;
                const int sa = ms[q];
                unsigned cnt = 0, nbb = 0, l0 = 0u, l1 = 0u;
                for_partners(sa, SC_CX(scell[q]), SC_CY(scell[q]), [&](int sl, int k) __attribute__((always_inline)) {
                    const unsigned e16 = (unsigned)sl | ((unsigned)k << 12);
                    if (nbb < 2u) l0 |= e16 << (16u * nbb);
                    else if (nbb < 4u) l1 |= e16 << (16u * (nbb - 2u));
                    nbb++;
                    if (((cnt >> (6 * k)) & 63u) == 63u) atomicOr(&misc[M_STATUS], 4u);
                    else cnt += 1u << (6 * k);
                });
                pl0[q] = l0; pl1[q] = l1; snb[q] = nbb;
                // walls: b2CollideEdgeAndCircle (region AB) against the chain loop of kilobots_env.py:48-51
                unsigned wm = 0u;
                const float ax = pos[sa].x, ay = pos[sa].y;
#pragma unroll
                for (int wl = 0; wl < 4; ++wl) {
                    float dist, nx, ny;
                    wall_geom(p, wl, ax, ay, dist, nx, ny);
                    if (dist * dist > rw2) continue;
                    wm |= (1u << wl) | (dist < 0.0f ? 16u << wl : 0u);     // (bit 4 + wl: centre outside the wall line, the manifold normal points outwards)
                }
                scell[q] |= (int)(wm << 24);
                unsigned mine = nbb + (unsigned)__popc(wm & 15u);
                dirCnt[sa] = cnt;
                if (mine > (unsigned)S) { atomicOr(&misc[M_STATUS], 2u); atomicAdd(&misc[M_XTRA], mine - (unsigned)S); mine = S; }
                wsCntNew[b] = (unsigned char)mine;
                snm[q] = mine;
            }
            KB_STAMP_PRE(17);
            // 4. offsets of the new packed warm-start list (owners in ascending id: thread t owns kilobots t and nt + t) = where
            //    the contacts are staged.  Every thread scans its own counts: one wave scan per kilobot row, one barrier.
            unsigned myOff[BPT];
            {
                static_assert(BPT == 2 && MAX_WAVES <= 8, "two rows of per-wave sums in the 16-word scratch");
                const unsigned inc0 = wave_incl_scan(snm[0]), inc1 = wave_incl_scan(snm[1]);
                if (lane == 63) { wsum[wave] = inc0; wsum[8 + wave] = inc1; }
                __syncthreads();
                KB_ABLATE_EXIT(3);     // owned contacts found
                unsigned b0 = 0, b1 = 0, t0 = 0, t1 = 0;
                for (int w = 0; w < nw; ++w) {
                    const unsigned s0 = wsum[w], s1 = wsum[8 + w];
                    if (w < wave) { b0 += s0; b1 += s1; }
                    t0 += s0; t1 += s1;
                }
                myOff[0] = b0 + inc0 - snm[0]; myOff[1] = t0 + b1 + inc1 - snm[1];
                newTotal = t0 + t1;
            }
            KB_STAMP(1);
            KB_RETID();
            KB_ABLATE_EXIT(4);     // offsets of the packed list
            if (SENSE && p.sense_s > 0 && drive) {      // the neighbour counters lie where the emit pass stages the impulses
#pragma unroll
                for (int q = 0; q < BPT; ++q) {
                    const int b = tid + q * nt;
                    if (b < N) g.nbr_count[o + b] = (unsigned)reinterpret_cast<unsigned short *>(lAcc)[ms[q]];
                }
                __syncthreads();
            }
            const unsigned extras = misc[M_XTRA];      // contacts beyond the warm-start slots of their owner: solved, never stored
            const int capS = FN ? capL_ : min(capL_, p.cap);      // (a contact_capacity below the LDS staging area still bounds the list)
            big = p.solver_mode >= 3 || newTotal + extras > (unsigned)capS;
            stageCap = big ? p.cap : capS;
            if (newTotal + extras > (unsigned)stageCap && tid == 0) atomicOr(&misc[M_STATUS], 1u);
            ncon = (int)min(newTotal + extras, (unsigned)stageCap);
            // ---- narrowphase, pass 2 (the owner's thread, as light as possible: its trip count is the busiest lane's): stage the
            //      owned contacts at their position in the packed list, contact j of owner a at newOff[a] + j.  Raw record:
            //      pair; direction k (walls: 8 + wall) | index inside the direction (walls: rank) << 4 | parity of the base
            //      cell << 12 | "kilobots of the cell in front of the owner" << 14 | wall normal flipped << 15 ----
            // (The staging arrays are LDS or the global slice.  Each pass is instantiated once per address space: through a pointer
            //  that may be either, every access is a FLAT instruction -- slower into LDS than ds_read / ds_write, and counted on
            //  the vector-memory counter as well, so that waiting for one waits for every global load in flight.)
            auto stage_pass = [&](unsigned *sPair, unsigned *sInfo) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b >= N) continue;
                const int sa = ms[q], cx = SC_CX(scell[q]), cy = SC_CY(scell[q]);
                const unsigned nbb = snb[q];
                const unsigned wm = (unsigned)scell[q] >> 24;
                const unsigned ntot = nbb + (unsigned)__popc(wm & 15u);
                newOff[b] = (unsigned short)myOff[q];
                if (ntot == 0u) continue;
                const unsigned base = myOff[q], nst = snm[q];
                const int cell = cy * p.gw + cx;
                const unsigned common = ((unsigned)(cx & 1) << 12) | ((unsigned)(cy & 1) << 13) | (sa > (int)E1[hashed ? (cell & p.hmask) : cell] ? 1u << 14 : 0u);
                int curk = -1;
                unsigned jk = 0;
                unsigned wleft = wm & 15u;
                for (unsigned j = 0; j < ntot; ++j) {
                    unsigned pr, inf;
                    if (j < nbb) {
                        unsigned e16;
                        if (j < 4u) e16 = ((j < 2u ? pl0[q] : pl1[q]) >> (16u * (j & 1u))) & 0xFFFFu;
                        else {          // (a dense pile: walk the candidates again up to partner j)
                            unsigned idx = 0, f16 = 0;
                            for_partners(sa, cx, cy, [&](int sl, int kk) __attribute__((always_inline)) { if (idx == j) f16 = (unsigned)sl | ((unsigned)kk << 12); idx++; });
                            e16 = f16;
                        }
                        const int k = (int)(e16 >> 12);
                        if (k != curk) { curk = k; jk = 0; } else jk++;
                        if (jk > 255u) { jk = 255u; atomicOr(&misc[M_STATUS], 4u); }
                        pr = (unsigned)sa | ((e16 & 0xFFFu) << 16);
                        inf = (unsigned)k | (jk << 4) | common;
                    } else {              // wall contact, owned by the kilobot: rank = lower walls it touches
                        const int wl = __builtin_ctz(wleft);
                        wleft &= wleft - 1u;
                        pr = (unsigned)(WALL_CODE + wl) | ((unsigned)sa << 16);
                        inf = (unsigned)(8 + wl) | ((unsigned)__popc(wm & ((1u << wl) - 1u) & 15u) << 4) | (((wm >> (4 + wl)) & 1u) << 15);
                    }
                    const unsigned cid = j < nst ? base + j : newTotal + atomicAdd(&misc[M_XFILL], 1u);
                    if (cid >= (unsigned)stageCap) continue;
                    sPair[cid] = pr;
                    sInfo[cid] = inf;
                }
            }
            };
            if (big) stage_pass(gPair, gInfo); else stage_pass(lPair, lInfo);
            KB_STAMP_PRE(34);    // (cumulative since the offset scan) wave 0's owned contacts staged
            __syncthreads();
            KB_STAMP_PRE(35);    // ... + barrier
            KB_RETID();
            KB_ABLATE_EXIT(5);     // contacts staged
            // ---- narrowphase, pass 3 (thread per contact): class and rank in the canonical order, impulse of the same pair in
            //      the previous substep, key of the new packed list (bits 16.. of the info word), island hooking ----
            auto label_pass_bins = [&](unsigned *sPair, unsigned *sInfo, float *sAcc) __attribute__((always_inline)) {
            for (int c = tid; c < ncon; c += nt) {
                const unsigned pr = sPair[c], raw = sInfo[c];
                const unsigned a = pr & 0xFFFFu, b = pr >> 16;
                unsigned key16, cls, r = (raw >> 4) & 0xFFu;
                float acc;
                if (a >= (unsigned)WALL_CODE) {
                    key16 = a;
                    cls = (unsigned)CLS_WALL | ((raw >> 15) ? 0x80u : 0u);
                    acc = ws_find((int)idOf[b], key16);
                } else {
                    const unsigned ida = idOf[a], idb = idOf[b];
                    const int k = (int)(raw & 15u);
                    const unsigned px = (raw >> 12) & 1u, py = (raw >> 13) & 1u;
                    cls = k == 0 ? (unsigned)CLS_SAME : (k == 1 ? CLS_E + px : (k == 2 ? CLS_N + py : (k == 3 ? CLS_NE + px : CLS_NW + px)));
                    if ((raw >> 14) & 1u) {
                        // rank base of the (cell, direction) group: its contacts owned by the kilobots of the cell in front of the owner
                        const float2 pa = pos[a];
                        int cx = (int)floorf((pa.x - p.xmin) * p.inv_cell);
                        int cy = (int)floorf((pa.y - p.ymin) * p.inv_cell);
                        cx = cx < 0 ? 0 : (cx >= p.gw ? p.gw - 1 : cx);
                        cy = cy < 0 ? 0 : (cy >= p.gh ? p.gh - 1 : cy);
                        const int cell = cy * p.gw + cx;
                        for (int s_ = (int)E1[hashed ? (cell & p.hmask) : cell]; s_ < (int)a; ++s_) {
                            if (hashed && (int)cellOfSlot[s_] != cell) continue;
                            r += (dirCnt[s_] >> (6 * k)) & 63u;
                        }
                        if (r > 255u) { r = 255u; atomicOr(&misc[M_STATUS], 4u); }
                    }
                    acc = ws_find((int)ida, idb);
                    if (acc < 0.0f) acc = ws_find((int)idb, ida);
                    key16 = idb;
                    // island hooking: larger root goes under the smaller one
                    unsigned ra = a, rb = b;
                    for (;;) {
                        while (true) { unsigned t = lds_load_relaxed(&parent[ra]); if (t == ra) break; ra = t; }
                        while (true) { unsigned t = lds_load_relaxed(&parent[rb]); if (t == rb) break; rb = t; }
                        if (ra == rb) break;
                        if (ra < rb) { unsigned t = ra; ra = rb; rb = t; }
                        if (atomicCAS(&parent[ra], ra, rb) == ra) break;
                    }
                }
                if (acc < 0.0f) acc = 0.0f;
                sInfo[c] = cls | (r << 8) | (key16 << 16);
                sAcc[c] = acc;
            }
            };
            if (big) label_pass_bins(gPair, gInfo, gAcc); else label_pass_bins(lPair, lInfo, lAcc);
            KB_STAMP_PRE(18);
            __syncthreads();
            KB_STAMP(2);
            KB_RETID();
            KB_ABLATE_EXIT(6);     // impulses looked up, islands hooked
        } else {
            // ---- IR-range neighbour sensing (kb_config.sense_radius): at the sensing point of the substep, like the light
            //      (kilobots_env.py:174-180), off the cell lists that the contact search uses ----
            if (SENSE && p.sense_s > 0 && drive)
                kb_sense_pass(pos, head, nextb, cellOf, reinterpret_cast<unsigned *>(newOff), N, nt, tid, p.gw, p.gh, p.sense_s, p.sense_r2, hashed ? p.hmask : 0);
            // object-object / object-wall manifolds (b2Contact::Update) + their velocity-constraint set-up: candidate t
            // is lane t of wave 0; the record lives in LDS, the previous substep's impulses come from g.ows_acc
            if (OBJ && wave == 0) {
                mcTouch = false;
                if (lane < NMC) {
                    owsMine = g.ows_acc + (size_t)e * (MAXOBJ * KB_OWS_COLS * KB_OWS_WORDS);
                    Arena ar;
                    ar.xmin = p.xmin; ar.ymin = p.ymin; ar.xmax = p.xmax; ar.ymax = p.ymax;
                    mcTouch = mc_detect(ox, ar, F, lane, owsMine);
                    if (SLEEP && !mcTouch) {
                        // b2Contact::Update: an object pair that was touching in the previous step (a stored manifold) and is not any
                        // more wakes both bodies (the island seeds are planted two barriers later)
                        int owner_, col_;
                        mc_candidate(F, lane, owner_, col_);
                        if (col_ < 8 && owsMine[(owner_ * KB_OWS_COLS + col_) * KB_OWS_WORDS] >= 0.0f) {
                            const int m1_ = ot_body(objTab + owner_ * OT_WORDS), m2_ = ot_body(objTab + col_ * OT_WORDS);
                            if (m1_ != m2_) {
                                if (objSlp[m1_] < 0.0f) objSlp[m1_] = 0.0f;
                                if (objSlp[m2_] < 0.0f) objSlp[m2_] = 0.0f;
                            }
                        }
                    }
                }
            }

            // ---- narrowphase pass 1 (thread per bot): find the contacts each bot owns (5-cell half stencil + walls),
            //      append them to the staging list, count them per direction ----
            auto find_pass = [&](unsigned *sPair, unsigned *sInfo, int stageCap_) __attribute__((always_inline)) {
#pragma unroll 1
                for (int a = tid; a < N; a += nt) {
                    const int cell = cellOf[a];
                    const int cx = cell % p.gw, cy = cell / p.gw;
                    const float ax = pos[a].x, ay = pos[a].y;
                    unsigned cnt = 0, mine = 0;
                    unsigned hd[5];      // heads of the five cell lists, fetched together (one LDS round trip)
                    int tcell[5];        // (hashed heads: the cell a candidate must be in)
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const int ox = cx + dir_dx(k), oy = cy + dir_dy(k);
                        const bool in = ox >= 0 && ox < p.gw && oy < p.gh;
                        tcell[k] = in ? oy * p.gw + ox : cell;
                        hd[k] = in ? (unsigned)head[hix(tcell[k])] : (unsigned)EMPTY16;
                    }
                    // the five lists are walked in lockstep: one LDS round trip serves the next candidate of every list that
                    // still has one (lists that have ended re-read the kilobot itself), so the trips of this pass are the
                    // length of the longest list, not the sum of the five
                    unsigned ckk[5] = {0u, 0u, 0u, 0u, 0u};
                    while (hd[0] != (unsigned)EMPTY16 || hd[1] != (unsigned)EMPTY16 || hd[2] != (unsigned)EMPTY16 ||
                           hd[3] != (unsigned)EMPTY16 || hd[4] != (unsigned)EMPTY16) {
                        float2 pbk[5];
                        unsigned nbk[5];
                        int cbk_[5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const unsigned bb = hd[k] != (unsigned)EMPTY16 ? hd[k] : (unsigned)a;
                            pbk[k] = pos[bb]; nbk[k] = nextb[bb];
                            cbk_[k] = hashed ? (int)cellOf[bb] : tcell[k];
                        }
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const unsigned b = hd[k];
                            if (b == (unsigned)EMPTY16) continue;
                            if (!(k == 0 && (int)b <= a) && cbk_[k] == tcell[k]) {
                                const float dx = pbk[k].x - ax, dy = pbk[k].y - ay;
                                const float dd = dx * dx + dy * dy;
                                if (!(dd > rr2)) {  // b2CollideCircles
                                    ckk[k]++;
                                    const unsigned c = atomicAdd(&misc[M_NCON], 1u);
                                    if (c < (unsigned)stageCap_) { sPair[c] = (unsigned)a | (b << 16); sInfo[c] = (unsigned)k; }
                                }
                            }
                            hd[k] = nbk[k];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        unsigned ck = ckk[k];
                        mine += ck;
                        if (ck > 63u) { ck = 63u; atomicOr(&misc[M_STATUS], 4u); }
                        cnt |= ck << (6 * k);
                    }
                    // walls: b2CollideEdgeAndCircle (region AB) against the chain loop of kilobots_env.py:48-51
#pragma unroll
                    for (int wl = 0; wl < 4; ++wl) {
                        float dist, nx, ny;
                        wall_geom(p, wl, ax, ay, dist, nx, ny);
                        if (dist * dist > rw2) continue;
                        mine++;
                        const unsigned c = atomicAdd(&misc[M_NCON], 1u);
                        // info bits 0-2: 5 + wall; bit 7: centre outside the wall line (manifold normal points outwards)
                        if (c < (unsigned)stageCap_) {
                            sPair[c] = (unsigned)(WALL_CODE + wl) | ((unsigned)a << 16);
                            sInfo[c] = (unsigned)(5 + wl) | (dist < 0.0f ? 0x80u : 0u);
                        }
                    }
                    // pushable objects: b2CollideCircles / b2CollidePolygonAndCircle kilobot - object m
                    // (info 9 + m; bits 8..: how many lower objects this kilobot touches)
                    unsigned nobj = 0;
                    for (int f = 0; f < F; ++f) {
                        const float *T = objTab + f * OT_WORDS;
                        const int m = ot_body(T);
                        const float dx = pos[N + m].x - ax, dy = pos[N + m].y - ay;
                        const float ro = p.r_bot + T[OT_BOUND];       // circle: contact radius; polygon: bounding radius
                        if (dx * dx + dy * dy > ro * ro) continue;
                        if (T[OT_KIND] != 0.0f) {
                            V2 ln, lp;
                            if (!collide_poly_circle(T, body_xf(ox, N + m), mk2(ax, ay), p.r_bot, ln, lp)) continue;
                        }
                        mine++;
                        const unsigned pos = atomicAdd(&objCnt[f], 1u);
                        if (pos < (unsigned)OBJ_LIST) objList[f * OBJ_LIST + pos] = (unsigned short)a;
                        else atomicOr(&misc[M_STATUS], 4u);
                        const unsigned c = atomicAdd(&misc[M_NCON], 1u);
                        if (c < (unsigned)stageCap_) { sPair[c] = (unsigned)a | ((unsigned)(N + m) << 16); sInfo[c] = (unsigned)(9 + f) | (nobj << 8); }
                        nobj++;
                    }
                    dirCnt[a] = cnt;
                    if (mine > (unsigned)S) { atomicOr(&misc[M_STATUS], 2u); mine = S; }
                    wsCntNew[a] = (unsigned char)mine;
                }
            };
            big = p.solver_mode >= 3;
            if (!big) {
                find_pass(lPair, lInfo, capL_);
                KB_STAMP_PRE(17);
                __syncthreads();
                big = (int)misc[M_NCON] > capL_;    // does not fit the LDS staging area: redo into the global scratch slice
                if (big) {
                    __syncthreads();
                    if (tid == 0) misc[M_NCON] = 0;
                    if (tid < F) objCnt[tid] = 0;
                    __syncthreads();
                }
            }
            if (big) {
                find_pass(gPair, gInfo, p.cap);
                __syncthreads();
            }
            KB_STAMP(1);
            KB_RETID();
            stageCap = big ? p.cap : capL_;
            if ((int)misc[M_NCON] > stageCap && tid == 0) atomicOr(&misc[M_STATUS], 1u);
            ncon = min((int)misc[M_NCON], stageCap);

            // ---- narrowphase pass 2 (thread per contact): class, rank, warm-start slot + impulse, island hooking ----
            auto label_pass = [&](unsigned *sPair, unsigned *sInfo, float *sAcc) __attribute__((always_inline)) {
                if (OBJ && wave == 0 && lane < NMC && mcTouch && mci(ox, MC_A, lane) < WALL_CODE) {   // object - object: one island
                    unsigned ra = (unsigned)mci(ox, MC_A, lane), rb = (unsigned)mci(ox, MC_B, lane);
                    for (;;) {
                        while (true) { unsigned t = lds_load_relaxed(&parent[ra]); if (t == ra) break; ra = t; }
                        while (true) { unsigned t = lds_load_relaxed(&parent[rb]); if (t == rb) break; rb = t; }
                        if (ra == rb) break;
                        if (ra < rb) { unsigned t = ra; ra = rb; rb = t; }
                        if (atomicCAS(&parent[ra], ra, rb) == ra) break;
                    }
                }
                for (int c = tid; c < ncon; c += nt) {
                    const unsigned pr = sPair[c], inf0 = sInfo[c];
                    const int k = inf0 & 31;
                    int cls, r, slot;
                    unsigned fixbits = 0;
                    float acc;
                    if (k >= 9) {    // kilobot - object m, owned by the kilobot; all of them strictly sequential
                        const int a = pr & 0xFFFF, m = k - 9;
                        const float ax = pos[a].x, ay = pos[a].y;
                        r = 0;
                        for (int m2 = 0; m2 < m; ++m2) r += (int)min(objCnt[m2], (unsigned)OBJ_LIST);
                        const int nm_ = (int)min(objCnt[m], (unsigned)OBJ_LIST);
                        for (int i = 0; i < nm_; ++i) r += (objList[m * OBJ_LIST + i] < a) ? 1 : 0;
                        const unsigned dc = dirCnt[a];
                        slot = 0;
#pragma unroll
                        for (int k2 = 0; k2 < 5; ++k2) slot += (int)((dc >> (6 * k2)) & 63u);
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            float dist, nx, ny;
                            wall_geom(p, w2, ax, ay, dist, nx, ny);
                            if (!(dist * dist > rw2)) slot++;
                        }
                        slot += (int)((inf0 >> 8) & 15u);      // lower objects this kilobot touches (counted by the find pass)
                        cls = CLS_BOT_OBJ;
                        acc = ws_find(a, (unsigned)(OBJ_CODE + m));
                        fixbits = (unsigned)m << 24;      // the fixture travels with the contact (bits 24..27)
                        unsigned ra = a, rb = pr >> 16;        // the object the fixture belongs to
                        for (;;) {
                            while (true) { unsigned t = lds_load_relaxed(&parent[ra]); if (t == ra) break; ra = t; }
                            while (true) { unsigned t = lds_load_relaxed(&parent[rb]); if (t == rb) break; rb = t; }
                            if (ra == rb) break;
                            if (ra < rb) { unsigned t = ra; ra = rb; rb = t; }
                            if (atomicCAS(&parent[ra], ra, rb) == ra) break;
                        }
                    } else if (k >= 5) {   // wall contact, owned by the bot
                        const int a = pr >> 16, wl = k - 5;
                        const float ax = pos[a].x, ay = pos[a].y;
                        const unsigned dc = dirCnt[a];
                        int nbots = 0;
#pragma unroll
                        for (int k2 = 0; k2 < 5; ++k2) nbots += (int)((dc >> (6 * k2)) & 63u);
                        r = 0;
#pragma unroll
                        for (int w2 = 0; w2 < 3; ++w2) {
                            float dist, nx, ny;
                            wall_geom(p, w2, ax, ay, dist, nx, ny);
                            if (w2 < wl && !(dist * dist > rw2)) r++;
                        }
                        cls = CLS_WALL | (int)(inf0 & 0x80u);
                        slot = nbots + r;
                        acc = ws_find(a, (unsigned)(WALL_CODE + wl));
                    } else {
                        const int a = pr & 0xFFFF;
                        const unsigned b = pr >> 16;
                        const int cell = cellOf[a];
                        const int cx = cell % p.gw, cy = cell / p.gw;
                        const float ax = pos[a].x, ay = pos[a].y;
                        if (k == 0) cls = CLS_SAME;
                        else if (k == 1) cls = CLS_E + (cx & 1);
                        else if (k == 2) cls = CLS_N + (cy & 1);
                        else if (k == 3) cls = CLS_NE + (cx & 1);
                        else cls = CLS_NW + (cx & 1);
                        const unsigned dc = dirCnt[a];
                        int sbase = 0;
                        for (int k2 = 0; k2 < k; ++k2) sbase += (int)((dc >> (6 * k2)) & 63u);
                        // rank base: contacts of this (cell, direction) group owned by lower-id bots of the cell
                        int rbase = 0;
                        for (unsigned a2 = head[hix(cell)]; a2 != (unsigned)EMPTY16; a2 = nextb[a2])
                            if ((int)a2 < a && (!hashed || (int)cellOf[a2] == cell)) rbase += (int)((dirCnt[a2] >> (6 * k)) & 63u);
                        // position of b among a's touching partners of this direction, in ascending id order
                        int j = 0;
                        if (((dc >> (6 * k)) & 63u) > 1u) {
                            const int oc = (cy + dir_dy(k)) * p.gw + (cx + dir_dx(k));
                            for (unsigned b2 = head[hix(oc)]; b2 != (unsigned)EMPTY16; b2 = nextb[b2]) {
                                if (b2 >= b || (k == 0 && (int)b2 <= a) || (hashed && (int)cellOf[b2] != oc)) continue;
                                const float2 pb2 = pos[b2];
                                const float ex = pb2.x - ax, ey = pb2.y - ay;
                                if (!(ex * ex + ey * ey > rr2)) j++;
                            }
                        }
                        r = rbase + j;
                        slot = sbase + j;
                        // warm start: impulse of the same pair in the previous substep (b2Contact::Update id match)
                        acc = ws_find(a, b);
                        if (acc < 0.0f) acc = ws_find((int)b, (unsigned)a);
                        // island hooking: larger root goes under the smaller one
                        unsigned ra = a, rb = b;
                        for (;;) {
                            while (true) { unsigned t = lds_load_relaxed(&parent[ra]); if (t == ra) break; ra = t; }
                            while (true) { unsigned t = lds_load_relaxed(&parent[rb]); if (t == rb) break; rb = t; }
                            if (ra == rb) break;
                            if (ra < rb) { unsigned t = ra; ra = rb; rb = t; }
                            if (atomicCAS(&parent[ra], ra, rb) == ra) break;
                        }
                    }
                    if (acc < 0.0f) acc = 0.0f;
                    if (slot >= S) slot = 255;
                    if (r > 255) { r = 255; atomicOr(&misc[M_STATUS], 4u); }
                    sInfo[c] = (unsigned)cls | ((unsigned)r << 8) | ((unsigned)slot << 16) | fixbits;
                    sAcc[c] = acc;
                }
            };
            if (big) label_pass(gPair, gInfo, gAcc); else label_pass(lPair, lInfo, lAcc);
            KB_STAMP_PRE(18);
            __syncthreads();
            KB_STAMP(2);
            KB_RETID();

        }

        if (SLEEP) {
            // ---- b2Contact::Update (b2ContactManager::Collide, ahead of b2World::Solve): a contact that was touching in the previous
            // step and is not any more wakes BOTH bodies.  "Was touching" = an entry of the previous substep's packed list that no
            // contact of this substep matched (the lookups above flipped the sign of the ones they found).  The owner wakes itself,
            // its partner through a bit (kilobots) or its sleep time in LDS (objects); walls are static.  The awake seeds of the
            // islands are planted behind the next barrier. ----
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b >= N) continue;
                const int cnt = wsCnt[b], off = wsOff[b];
                for (int s_ = 0; s_ < cnt && off + s_ < p.cap; ++s_) {
                    const float a_ = oldInLds ? oldAcc[off + s_] : g.ws_acc[wo + off + s_];
                    if (a_ < 0.0f) continue;      // matched: still touching
                    unsigned key = oldInLds ? (unsigned)oldKey[off + s_] : g.ws_key[wo + off + s_];
                    if (!oldInLds) key = key >= KEY_OBJ ? OBJ_CODE + (key - KEY_OBJ) : (key >= KEY_WALL ? WALL_CODE + (key - KEY_WALL) : key);
                    if (key >= (unsigned)WALL_CODE) continue;
                    if (slp[q] < 0.0f) slp[q] = 0.0f;
                    if (key >= (unsigned)OBJ_CODE) {
                        if (OBJ) { const int m_ = ot_body(objTab + (key - OBJ_CODE) * OT_WORDS); if (objSlp[m_] < 0.0f) objSlp[m_] = 0.0f; }
                    } else atomicOr(&misc[M_WAKE + (key >> 5)], 1u << (key & 31u));
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b < N && slp[q] < 0.0f && ((misc[M_WAKE + (b >> 5)] >> (b & 31)) & 1u)) slp[q] = 0.0f;
            }
        }
        // ---- islands: flatten roots; empty the grid for the next substep; offsets of the new ws list ----
        for (int b = tid; b < N; b += nt) {
            unsigned r = b;
            while (true) { unsigned t = lds_load_relaxed(&parent[r]); if (t == r) break; r = t; }
            parent[b] = r;   // only ever replaces an ancestor by an older ancestor: concurrent walks stay valid
            islCnt[b] = 0;
            islWave[b] = (unsigned char)((unsigned)b % (unsigned)nsolve);
            if (!BINS) head[hix(cellOf[b])] = EMPTY16;
            if (!SLEEP) active[b] = 1;
            active[NB + b] = 0;
            if (!BINS && SENSE && p.sense_s > 0 && drive) g.nbr_count[o + b] = (unsigned)newOff[b];
        }
        if (tid < 64) bkStart[tid] = 0;     // size-class counters of the island placement
        if (tid < 32) bkFill[tid] = 0;
        if (tid < M) {
            const int b = N + tid;
            unsigned r = b;
            while (true) { unsigned t = lds_load_relaxed(&parent[r]); if (t == r) break; r = t; }
            parent[b] = r;
            islCnt[b] = 0;
            islWave[b] = (unsigned char)((unsigned)b % (unsigned)nw);
            if (!SLEEP) active[b] = 1;
            active[NB + b] = 0;
        }
        if (SLEEP && !BINS) {
            // b2World::Solve: islands grow from awake seeds.  active[root] = 1 iff the island has an awake body (every writer
            // stores the same value; the flags were cleared in the drive phase); islands without one are not solved at all.
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b < N && !(slp[q] < 0.0f)) active[parent[b]] = 1;
            }
            if (tid < M && !(objSlp[tid] < 0.0f)) active[parent[N + tid]] = 1;
        }
        if (BINS) __syncthreads();
        else newTotal = block_scan_u8(wsCntNew, newOff, NP, wsum, tid);   // (barriers inside)
        if (SLEEP && BINS) {
            // (sorted bins: a kilobot's slot is flattened by whichever thread walks that index, so the awake seeds are planted
            //  behind the barrier; nothing reads the flags before the next one)
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b < N && !(slp[q] < 0.0f)) active[parent[ms[q]]] = 1;
            }
        }
        // is the new packed list kept as an LDS image for the next substep's lookups (sorted bins: the staged contacts ARE the list)
        const bool newInLds = BINS ? !big : newTotal <= (unsigned)capL_;
        if (OBJ && wave == 0) {   // island of every manifold constraint
            unsigned root = 0;
            const bool on = lane < NMC && mcTouch;
            if (on) { root = parent[mci(ox, MC_B, lane)]; mci_set(ox, MC_ISL, lane, (int)root); }
        }
        KB_STAMP(14);    // flatten roots + warm-start offset scan
        KB_RETID();
        KB_ABLATE_EXIT(7);     // roots flattened
        // per contact: island size (giant islands force the cooperative sweep) and contacts per wave
        auto census = [&](const unsigned *sPair) __attribute__((always_inline)) {
            for (int c = tid; c < ncon; c += nt) {
                const unsigned root = parent[sPair[c] >> 16];
                const unsigned n = atomicAdd(&islCnt[root], 1u) + 1u;
                if (n > (unsigned)GIANT_ISLAND) atomicMax(&misc[M_MAXISL], n);
                atomicAdd(&misc[M_WCNT + islWave[root]], 1u);
            }
        };
        if (big) census(gPair); else census(lPair);
        __syncthreads();
        unsigned maxw = 0;
        for (int w = 0; w < nw; ++w) maxw = max(maxw, misc[M_WCNT + w]);
        // solver selection (block-uniform):
        //   coop: one island is so large that a single wave would serialise the env
        //   reg : every wave can hold its contacts in registers (<= KREG per lane) -> no contact arrays in the sweeps
        //   list: per-wave sweeps over the staged contact arrays
        // (a one-wave workgroup is its own cooperative group; it may still take the register path)
        bool coopForced = misc[M_MAXISL] > (unsigned)GIANT_ISLAND || p.solver_mode == 2 || p.solver_mode == 4;
        bool coop = coopForced || nw == 1;
        if (!coop && !big && p.solver_mode == 0) {
            // Placement of the islands on the waves in order of size: the largest (= deepest) islands share the first
            // waves, the many tiny ones fill the rest.  A wave sweeps as many rounds as its deepest island has levels
            // and the LDS pipeline is shared, so what counts is the SUM of those depths over the waves -- about a third
            // less than with a hash placement.  Counting sort over 32 size classes; results never depend on it.
            unsigned *szSum = bkStart, *szPre = bkStart + 32, *szFill = bkFill;     // (zeroed in the flatten phase)
            for (int b = tid; b < N + M; b += nt) {
                if (parent[b] != (unsigned)b) continue;
                const unsigned cnt_ = islCnt[b];
                if (cnt_ != 0) atomicAdd(&szSum[32u - min(cnt_, 32u)], cnt_);
            }
            __syncthreads();
            if (wave == 0) {
                const unsigned v_ = lane < 32 ? szSum[lane] : 0u;
                const unsigned incl = wave_incl_scan(v_);
                if (lane < 32) szPre[lane] = incl - v_;
                if (lane < nw) misc[M_WCNT + lane] = 0;
            }
            __syncthreads();
            // the first nw - 1 waves get about 60 contacts each (one register slot: no dealing, one-slot rounds), the
            // last one the remaining small islands (two slots, but only one or two depth levels)
            unsigned chunk = 60u;
            if ((unsigned)ncon > 60u * (unsigned)(nw - 1) + 124u) chunk = ((unsigned)ncon - 124u + (unsigned)nw - 2u) / (unsigned)(nw - 1);
            if (BINS) chunk = ((unsigned)ncon + (unsigned)nsolve - 1u) / (unsigned)nsolve;      // (the sweeping waves share the contacts evenly, the largest islands first)
            for (int b = tid; b < N + M; b += nt) {
                if (parent[b] != (unsigned)b) continue;
                const unsigned cnt_ = islCnt[b];
                if (cnt_ == 0) continue;
                const unsigned q_ = 32u - min(cnt_, 32u);
                const unsigned at = szPre[q_] + atomicAdd(&szFill[q_], cnt_);
                const unsigned w_ = min(at / max(chunk, 1u), (unsigned)nsolve - 1u);
                islWave[b] = (unsigned char)w_;
                atomicAdd(&misc[M_WCNT + w_], cnt_);
            }
            __syncthreads();
            maxw = 0;
            for (int w = 0; w < nw; ++w) maxw = max(maxw, misc[M_WCNT + w]);
        }
        bool reg = !coopForced && !big && maxw <= 64u * KRX && p.solver_mode == 0;
        if (!coop && !big && maxw > 64u * KRX && p.solver_mode == 0) {
            // The default placement (root id mod #waves) overloads a wave.  Place the islands of BIG_ISLAND contacts or
            // more one by one, largest first, each on the wave with the least load (ties: lowest root / lowest wave);
            // the small ones stay where they are.  Results do not depend on the placement, only the time does.
            unsigned short *bigList = bkList;           // (idle until the bucket sort)
            __syncthreads();
            if (tid < nw) misc[M_WCNT + tid] = 0;
            if (tid == 0) misc[M_TOTAL] = 0;
            __syncthreads();
            for (int b = tid; b < N + M; b += nt) {
                if (parent[b] != (unsigned)b) continue;
                const unsigned cnt_ = islCnt[b];
                if (cnt_ == 0) continue;
                unsigned slot_ = 64u;
                if (cnt_ >= (unsigned)BIG_ISLAND) slot_ = atomicAdd(&misc[M_TOTAL], 1u);
                if (slot_ < 64u) bigList[slot_] = (unsigned short)b;
                else atomicAdd(&misc[M_WCNT + islWave[b]], cnt_);
            }
            __syncthreads();
            if (wave == 0) {
                const int nb = (int)min(misc[M_TOTAL], 64u);
                const unsigned root = lane < nb ? (unsigned)bigList[lane] : 0u;
                const unsigned size = lane < nb ? islCnt[root] : 0u;
                int rank = 0;
                for (int j = 0; j < nb; ++j) {
                    const unsigned sj = __shfl(size, j), rj = __shfl(root, j);
                    if (sj > size || (sj == size && rj < root)) rank++;
                }
                unsigned load = lane < nsolve ? misc[M_WCNT + lane] : 0u;
                int myWave = 0;
                for (int r = 0; r < nb; ++r) {
                    const int src = __builtin_ctzll(__ballot(lane < nb && rank == r));
                    const unsigned sz = __shfl(size, src);
                    unsigned key = lane < nsolve ? ((load << 4) | (unsigned)lane) : 0xFFFFFFFFu;
                    for (int d = 8; d >= 1; d >>= 1) key = min(key, __shfl_xor(key, d));
                    const int wmin = (int)(__shfl(key, 0) & 15u);
                    if (lane == wmin) load += sz;
                    if (lane == src) myWave = wmin;
                }
                if (lane < nb) islWave[root] = (unsigned char)myWave;
                if (lane < nsolve) misc[M_WCNT + lane] = load;
            }
            __syncthreads();
            maxw = 0;
            for (int w = 0; w < nw; ++w) maxw = max(maxw, misc[M_WCNT + w]);
            reg = maxw <= 64u * KRX;
        }
        // Kernels without objects: whatever does not fit the registers (an overloaded wave, contacts staged in the global slice) is
        // swept by the whole workgroup level by level as well -- the per-wave list sweeps go key by key (~ 20 rounds per pass where
        // the levels are ~ 12) and are left to the one-wave workgroups and the test knob.
        if (BINS && nw > 1 && !reg && p.solver_mode == 0) { coopForced = true; coop = true; }
        if (OBJ && wave == 0) {   // which wave sweeps which manifold constraint (slot nw: all of them)
            const bool on = lane < NMC && mcTouch && (!SLEEP || active[mci(ox, MC_ISL, lane)] != 0);
            const unsigned w_ = on ? (unsigned)islWave[mci(ox, MC_ISL, lane)] : 0u;
            for (int w = 0; w < nw; ++w) {
                const unsigned long long mk = __ballot(on && (int)w_ == w);
                if (lane == 0) mcMask[w] = mk;
            }
            const unsigned long long all = __ballot(on);
            if (lane == 0) mcMask[nw] = all;
        }
        // manifold constraints of one (virtual) wave, swept by its leader thread after the regular contacts of a sweep
        auto mc_warm_pass = [&](unsigned long long mask, bool leader) __attribute__((always_inline)) {
            if (leader) for (unsigned long long m_ = mask; m_; m_ &= m_ - 1) mc_warm_start(ox, __builtin_ctzll(m_));
        };
        auto mc_velocity_pass = [&](unsigned long long mask, bool leader) __attribute__((always_inline)) {
            if (leader) for (unsigned long long m_ = mask; m_; m_ &= m_ - 1) mc_solve_velocity(ox, __builtin_ctzll(m_));
        };
        auto mc_position_pass = [&](unsigned long long mask, bool leader, const unsigned char *act, unsigned char *nxt, bool lastIt) __attribute__((always_inline)) -> bool {
            bool viol = false;
            if (leader)
                for (unsigned long long m_ = mask; m_; m_ &= m_ - 1) {
                    const int t = __builtin_ctzll(m_);
                    const int isl = mci(ox, MC_ISL, t);
                    if (!act[isl]) continue;
                    const float minSep = mc_solve_position(ox, t, B2_BAUMGARTE);
                    if (minSep < -3.0f * B2_LINEAR_SLOP) { nxt[isl] = 1; viol = true; if (SLEEP && lastIt) islWave[isl] = 3; }
                }
            return viol;
        };
        auto mc_clear_flags = [&](unsigned long long mask, bool leader, unsigned char *act) __attribute__((always_inline)) {
            if (leader) for (unsigned long long m_ = mask; m_; m_ &= m_ - 1) act[mci(ox, MC_ISL, __builtin_ctzll(m_))] = 0;
        };
        // StoreImpulses of the manifold constraints: candidate t owns entry (owner, column) of g.ows_acc
        auto mc_store = [&]() __attribute__((always_inline)) {
            if (OBJ && wave == 0 && lane < NMC) {
                int owner, col;
                mc_candidate(F, lane, owner, col);
                float *dst = owsMine + (owner * KB_OWS_COLS + col) * KB_OWS_WORDS;
                float o[KB_OWS_WORDS] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f, -1.0f};
                if (mcTouch) {
                    const int cnt = (mci(ox, MC_TYPE, lane) >> 2) & 3, ids = mci(ox, MC_ID, lane);
                    o[0] = (float)(ids & 255); o[1] = mcf(ox, MC_NI0, lane); o[2] = mcf(ox, MC_TI0, lane);
                    if (cnt == 2) { o[3] = (float)((ids >> 8) & 255); o[4] = mcf(ox, MC_NI1, lane); o[5] = mcf(ox, MC_TI1, lane); }
                }
#pragma unroll
                for (int k = 0; k < KB_OWS_WORDS; ++k) dst[k] = o[k];
            }
        };
        // ---- counting sort of the contacts by (wave, class, rank bucket): `order` lists every (virtual) wave's
        //      contacts key by key; bkList = its non-empty keys in canonical order ----
        auto bucket_sort = [&](const unsigned *sPair, const unsigned *sInfo, unsigned short *cbk,
                               unsigned short *order) __attribute__((always_inline)) {
            const int W = coop ? 1 : nw;
            for (int b = tid; b < N + M; b += nt) islCnt[b] = 0;   // reused as per-body dependency depth by the register solver
            for (int k = tid; k < W * BK_PER_WAVE; k += nt) { bkStart[k] = 0; bkFill[k] = 0; }
            for (int k = tid; k < W * NUM_CLS; k += nt) bkMaxRank[k] = 0;
            if (tid == 0) bkStart[W * BK_PER_WAVE] = 0;
            __syncthreads();
            for (int c = tid; c < ncon; c += nt) {
                const unsigned root = parent[sPair[c] >> 16];
                if (SLEEP && !active[root]) { cbk[c] = EMPTY16; continue; }      // sleeping island: never swept
                const int w = coop ? 0 : (int)islWave[root];
                const unsigned inf = sInfo[c];
                const int cls = inf & 0x7F, r = (inf >> 8) & 0xFF;
                const int bk = (w * NUM_CLS + cls) * RK + (r < RK - 1 ? r : RK - 1);
                cbk[c] = (unsigned short)bk;
                atomicAdd(&bkStart[bk], 1u);
                if (r >= RK - 1) atomicMax(&bkMaxRank[w * NUM_CLS + cls], (unsigned)r);
            }
            __syncthreads();
            if (wave == 0) {   // exclusive scan of the bucket counts (<= 160 entries)
                const int nb = W * BK_PER_WAVE;
                const int chunk = (nb + 63) / 64;
                const int s0 = lane * chunk, e0 = min(nb, s0 + chunk);
                unsigned sum = 0;
                for (int i = s0; i < e0; ++i) sum += bkStart[i];
                const unsigned incl = wave_incl_scan(sum);
                unsigned run = incl - sum;
                for (int i = s0; i < e0; ++i) { unsigned v = bkStart[i]; bkStart[i] = run; run += v; }
                if (lane == 63) bkStart[nb] = incl;
            }
            __syncthreads();
            for (int c = tid; c < ncon; c += nt) {
                const int bk = cbk[c];
                if (SLEEP && bk == (int)EMPTY16) continue;
                order[bkStart[bk] + atomicAdd(&bkFill[bk], 1u)] = (unsigned short)c;
            }
            if (wave < W) {   // every (virtual) wave compacts the list of its non-empty buckets, in key order
                const int base = wave * BK_PER_WAVE;
                const bool ne = lane < BK_PER_WAVE && bkStart[base + lane + 1] > bkStart[base + lane];
                const unsigned long long m = __ballot(ne);
                if (ne) bkList[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(base + lane);
                if (lane == 0) nList[wave] = (unsigned char)__popcll(m);
            }
            __syncthreads();
        };
        if (reg) {
            // register solver: it only needs each wave's contacts grouped together (any order inside a wave)
            for (int b = tid; b < N + M; b += nt) islCnt[b] = 0;   // reused as per-body dependency depth
            if (tid < nw * NUM_CLS) bkMaxRank[tid] = 0;
            for (int c = tid; c < ncon; c += nt) {
                const unsigned w = islWave[parent[lPair[c] >> 16]];
                unsigned base = 0;
                for (unsigned w2 = 0; w2 < w; ++w2) base += misc[M_WCNT + w2];
                lOrder[base + atomicAdd(&misc[M_WFILL + w], 1u)] = (unsigned short)c;
            }
            __syncthreads();
        } else if (big) bucket_sort(gPair, gInfo, gCbk, gOrder);
        else bucket_sort(lPair, lInfo, lCbk, lOrder);
        KB_STAMP(3);
        KB_RETID();
        KB_ABLATE_EXIT(8);     // census, placement, contacts grouped by wave

        // a giant island (or the test knob): the whole workgroup sweeps, level by level (kernels without objects; one-wave
        // workgroups keep the list sweep, their "barrier" is free)
        const bool coopLevels = BINS && coopForced && nw > 1;
        if constexpr (BINS) {
            if (reg) {
                // =========================== register-resident solver, kernels without objects ===========================
#include "kb_regsolve_bins.inc"
            } else if (coopLevels) {
                // =========================== cooperative solver by dependency depth, kernels without objects ===========================
#include "kb_coopsolve_bins.inc"
            }
        }
        if (BINS && (reg || coopLevels)) {
        } else if (reg) {
            // =========================== register-resident solver ===========================
            // wave w owns the contacts of the islands placed on it (islWave); lane l holds contacts l, l+64, ...
            unsigned mybase = 0;
            for (int w = 0; w < wave; ++w) mybase += misc[M_WCNT + w];
            const unsigned mycnt = misc[M_WCNT + wave];
            int ra[KREG], rb[KREG], rkey[KREG], rrank[KREG], rslot[KREG], risl[KREG], rc[KREG], rdepth[KREG];
            float racc[KREG], rnx[KREG], rny[KREG], rima[KREG], rimb[KREG], rra[KREG], rrb[KREG], rnm[KREG];
            bool rvalid[KREG], rflip[KREG];
            // kilobot - polygon contacts (objects only): lever arm on the polygon, manifold in the polygon's frame
            bool rpoly[KREG];
            float rrAx[KREG], rrAy[KREG], rlnx[KREG], rlny[KREG], rlpx[KREG], rlpy[KREG];
            unsigned mlo = 0, mhi = 0;
            // without objects the masses / radii of a contact follow from "is A a wall": no registers needed
            const float nm_bb = p.nm_bb, nm_wb = p.nm_wb;
#define R_IMA(j) (OBJ ? rima[j] : (ra[j] < WALL_CODE ? p.im_bot : 0.0f))
#define R_IMB(j) (OBJ ? rimb[j] : p.im_bot)
#define R_RA(j) (OBJ ? rra[j] : (ra[j] < WALL_CODE ? p.r_bot : B2_POLYGON_RADIUS))
#define R_RB(j) (OBJ ? rrb[j] : p.r_bot)
#define R_NM(j) (OBJ ? rnm[j] : (ra[j] < WALL_CODE ? nm_bb : nm_wb))
            // ---- light load: bodies and key of the contacts of this wave, in arrival order ----
#pragma unroll
            for (int j = 0; j < KREG; ++j) {
                const unsigned idx = lane + 64u * j;
                rvalid[j] = idx < mycnt;
                ra[j] = 0; rb[j] = 0; rkey[j] = 0; rrank[j] = 0; rc[j] = 0; rdepth[j] = 0;
                if (rvalid[j]) {
                    const int c = lOrder[mybase + idx];
                    const unsigned pr = lPair[c], inf = lInfo[c];
                    const int cls = inf & 0x7F, r = (inf >> 8) & 0xFF;
                    rc[j] = c; ra[j] = pr & 0xFFFF; rb[j] = pr >> 16; rrank[j] = r;
                    const int key = cls * RK + (r < RK - 1 ? r : RK - 1);
                    if (SLEEP && !active[parent[rb[j]]]) {
                        rkey[j] = -1;        // contact of a sleeping island: in no round (depth 0), its impulse is carried over
                    } else {
                        rkey[j] = key;
                        if (key < 32) mlo |= 1u << key; else mhi |= 1u << (key - 32);
                        if (r >= RK - 1) atomicMax(&bkMaxRank[wave * NUM_CLS + cls], (unsigned)r);
                    }
                }
            }
            // keys present in this wave (wave-uniform 52-bit mask)
            mlo = wave_or(mlo); mhi = wave_or(mhi);
            const unsigned long long keymask = ((unsigned long long)mhi << 32) | mlo;
            wave_sync();   // bkMaxRank of this wave
#ifdef KB_PROFILE
            if (tid == 0) { atomicAdd(&KB_PROF(8), (unsigned)__popcll(keymask)); atomicAdd(&KB_PROF(11), 1u); }
#endif

#define KB_REG_KEY_ROUNDS(...)                                                                      \
            for (unsigned long long m_ = keymask; m_; m_ &= m_ - 1) {                               \
                const int key_ = __builtin_ctzll(m_);                                               \
                if ((key_ % RK) < RK - 1) {                                                         \
                    _Pragma("unroll") for (int j = 0; j < KREG; ++j)                                \
                        if (rvalid[j] && rkey[j] == key_) { __VA_ARGS__ }                           \
                    wave_sync();                                                                    \
                } else {                                                                            \
                    const int maxr_ = (int)bkMaxRank[wave * NUM_CLS + key_ / RK];                   \
                    for (int r_ = RK - 1; r_ <= maxr_; ++r_) {                                      \
                        _Pragma("unroll") for (int j = 0; j < KREG; ++j)                            \
                            if (rvalid[j] && rkey[j] == key_ && rrank[j] == r_) { __VA_ARGS__ }     \
                        wave_sync();                                                                \
                    }                                                                               \
                }                                                                                   \
            }

            KB_STAMP_PRE(19);    // (profile build) wave 0: contacts grouped + light load
            // ---- dependency depth of every contact: 1 + the depth of the latest earlier contact (canonical key
            // order) on either of its bodies.  Contacts of equal depth never share a body, and sweeping by
            // increasing depth keeps the relative order of any two contacts that do -- so depth rounds give exactly
            // the result of the key-by-key sweep, in (typically) a third of the rounds. ----
            unsigned *bodyDepth = islCnt;   // zeroed before the wave sort
            {
                // one round per key (contacts of equal key share no body): the reads of all slots go out together,
                // slots that are not part of the round work on the scratch body
                const int DUMMYD = NB - 1;
                auto depth_round = [&](int key_, int r_) __attribute__((always_inline)) {
                    int ia[KREG], ib[KREG];
                    unsigned da[KREG], db[KREG];
                    bool on[KREG];
#pragma unroll
                    for (int j = 0; j < KREG; ++j) {
                        on[j] = rvalid[j] && rkey[j] == key_ && (r_ < 0 || rrank[j] == r_);
                        ia[j] = (on[j] && ra[j] < WALL_CODE) ? ra[j] : DUMMYD;
                        ib[j] = on[j] ? rb[j] : DUMMYD;
                        da[j] = bodyDepth[ia[j]]; db[j] = bodyDepth[ib[j]];
                    }
#pragma unroll
                    for (int j = 0; j < KREG; ++j) {
                        const unsigned d = max((on[j] && ra[j] < WALL_CODE) ? da[j] : 0u, db[j]) + 1u;
                        if (on[j]) {
                            rdepth[j] = (int)d;
                            bodyDepth[ib[j]] = d;
                            if (ra[j] < WALL_CODE) bodyDepth[ia[j]] = d;
                        }
                    }
                    wave_sync();
                };
                for (unsigned long long m_ = keymask; m_; m_ &= m_ - 1) {
                    const int key_ = __builtin_ctzll(m_);
                    if ((key_ % RK) < RK - 1) depth_round(key_, -1);
                    else {
                        const int maxr_ = (int)bkMaxRank[wave * NUM_CLS + key_ / RK];
                        for (int r_ = RK - 1; r_ <= maxr_; ++r_) depth_round(key_, r_);
                    }
                }
            }
            int maxD = 0;
#pragma unroll
            for (int j = 0; j < KREG; ++j) maxD = max(maxD, rdepth[j]);
            maxD = (int)wave_umax((unsigned)maxD);
            KB_STAMP_PRE(20);    // ... + depth pass
            // ---- deal the contacts to (lane, slot) in order of depth: the 64 shallowest go to slot 0, the rest to
            // slot 1, ...  A depth level then lives in one slot (two at a boundary), and a sweep round only pays for
            // the slots that hold contacts of its level. ----
            if (KREG > 1 && mycnt > 64u) {
                unsigned base_ = 0;
                for (int d_ = SLEEP ? 0 : 1; d_ <= maxD; ++d_) {      // (depth 0: contacts of sleeping islands)
#pragma unroll
                    for (int j = 0; j < KREG; ++j) {
                        const bool is = rvalid[j] && rdepth[j] == d_;
                        const unsigned long long m_ = __ballot(is);
                        if (is) {
                            const unsigned at = mybase + base_ + (unsigned)__popcll(m_ & ((1ull << lane) - 1ull));
                            lOrder[at] = (unsigned short)rc[j];
                            lCbk[at] = (unsigned short)d_;
                        }
                        base_ += (unsigned)__popcll(m_);
                    }
                }
                wave_sync();
#pragma unroll
                for (int j = 0; j < KREG; ++j)
                    if (rvalid[j]) { rc[j] = lOrder[mybase + lane + 64u * j]; rdepth[j] = lCbk[mybase + lane + 64u * j]; }
            }
            KB_STAMP_PRE(21);    // ... + dealing
            // depth levels present in every slot (bit min(depth, 63)), wave-uniform
            unsigned long long slotLevels[KREG];
#pragma unroll
            for (int j = 0; j < KREG; ++j) {
                const unsigned long long mine = rvalid[j] ? 1ull << min(rdepth[j], 63) : 0ull;
                const unsigned lo = wave_or((unsigned)mine), hi = wave_or((unsigned)(mine >> 32));
                slotLevels[j] = ((unsigned long long)hi << 32) | (unsigned long long)lo;
            }
            // The levels of a slot are a contiguous range (the dealing above; without it everything sits in slot 0) and the ranges
            // of successive slots follow each other: a sweep is slot 0's levels in order, then slot 1's, ... -- contacts of one
            // level never share a body, so a level split over two slots may run in two rounds.  (kb_regsolve_bins.inc)
            int dlo[KREG], dhi[KREG];
#pragma unroll
            for (int j = 0; j < KREG; ++j) {
                const unsigned long long lv = slotLevels[j] >> 1;      // (level 0: contacts of sleeping islands, in no round)
                dlo[j] = lv ? (int)__builtin_ctzll(lv) + 1 : 1;
                dhi[j] = lv ? ((lv >> 62) ? maxD : 64 - (int)__builtin_clzll(lv)) : 0;
            }
            // ---- full load of the contacts ----
            // Kernels with objects (except the spill-free WIDE ones): lanes without a contact in slot j load a copy of the
            // wave's first contact and drop it again, so that the slot loads run under the full EXEC mask.  (These kernels
            // spill registers; the toolchain
            // was caught storing slot 0's spilled values under the narrowed mask of `if (rvalid[1])` and reloading them
            // under the full one -- DESIGN.md "Robustness".  Without a lane-dependent region around the loads there is no
            // narrowed mask to get wrong here.)
            const bool anyContact = (unsigned)__builtin_amdgcn_readfirstlane((int)mycnt) > 0u;
            constexpr bool FULLMASK = OBJ && !WIDE;
            const int cSafe = (FULLMASK && anyContact) ? (int)lOrder[mybase] : 0;
#pragma unroll
            for (int j = 0; j < KREG; ++j) {
                ra[j] = 0; rb[j] = 0; rslot[j] = 255; risl[j] = 0;
                racc[j] = 0.0f; rnx[j] = 1.0f; rny[j] = 0.0f; rflip[j] = false;
                rima[j] = 0.0f; rimb[j] = 0.0f; rra[j] = 0.0f; rrb[j] = 0.0f; rnm[j] = 0.0f;
                rpoly[j] = false; rrAx[j] = 0.0f; rrAy[j] = 0.0f; rlnx[j] = 0.0f; rlny[j] = 0.0f; rlpx[j] = 0.0f; rlpy[j] = 0.0f;
                if (FULLMASK ? anyContact : rvalid[j]) {
                    const int c = (FULLMASK && !rvalid[j]) ? cSafe : rc[j];
                    const unsigned pr = lPair[c], inf = lInfo[c];
                    const int a = pr & 0xFFFF, b = pr >> 16;
                    ra[j] = a; rb[j] = b; rslot[j] = (inf >> 16) & 0xFF; racc[j] = lAcc[c];
                    rflip[j] = (inf & 0x80) != 0;
                    risl[j] = (int)parent[b];
                    if (OBJ) {
                        rima[j] = bim(a); rimb[j] = bim(b); rra[j] = brad(a); rrb[j] = brad(b);
                        const float k_ = rima[j] + rimb[j];
                        rnm[j] = k_ > 0.0f ? 1.0f / k_ : 0.0f;   // b2ContactSolver normalMass
                    }
                    // velocity-phase normal from the start-of-step positions (b2WorldManifold::Initialize)
                    if (bpoly(b)) {
                        // Box2D's A = the polygon b, B = the kilobot a: manifold, normal polygon -> kilobot, lever arm, normalMass
                        const float *T = objTab + ((inf >> 24) & 15u) * OT_WORDS;     // the fixture that is touched
                        const XF xo = body_xf(ox, b);
                        const V2 bc = mk2(pos[a].x, pos[a].y);
                        V2 ln = mk2(0.0f, 0.0f), lp = mk2(0.0f, 0.0f);
                        collide_poly_circle(T, xo, bc, p.r_bot, ln, lp);
                        const V2 normal = rot_mul(xo, ln);
                        const V2 planePoint = xf_mul(xo, lp);
                        const V2 cA = v_add(bc, v_scale(T[OT_RADIUS] - v_dot(v_sub(bc, planePoint), normal), normal));
                        const V2 cB = v_sub(bc, v_scale(p.r_bot, normal));
                        const V2 point = v_scale(0.5f, v_add(cA, cB));
                        const V2 rA = v_sub(point, mk2(pos[b].x, pos[b].y));
                        const float rnA = v_cross(rA, normal);
                        const float kNormal = T[OT_IM] + KB_IM_BOT(a) + T[OT_II] * rnA * rnA;
                        rpoly[j] = true; rnx[j] = normal.x; rny[j] = normal.y; rrAx[j] = rA.x; rrAy[j] = rA.y;
                        rlnx[j] = ln.x; rlny[j] = ln.y; rlpx[j] = lp.x; rlpy[j] = lp.y;
                        rnm[j] = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
                    } else if (a >= WALL_CODE) {
                        float dist, nx, ny;
                        wall_geom(p, a - WALL_CODE, pos[b].x, pos[b].y, dist, nx, ny);
                        if (rflip[j]) { nx = -nx; ny = -ny; }
                        rnx[j] = nx; rny[j] = ny;
                    } else {
                        const float dx = pos[b].x - pos[a].x, dy = pos[b].y - pos[a].y;
                        const float dd = dx * dx + dy * dy;
                        if (dd > B2_EPSILON * B2_EPSILON) {
                            const float len = sqrtf(dd);
                            const float inv = 1.0f / len;
                            rnx[j] = dx * inv; rny[j] = dy * inv;
                        }
                    }
                    if (FULLMASK && !rvalid[j]) {    // not a contact of this lane: back to the neutral values
                        ra[j] = 0; rb[j] = 0; rslot[j] = 255; risl[j] = 0;
                        racc[j] = 0.0f; rnx[j] = 1.0f; rny[j] = 0.0f; rflip[j] = false;
                        rima[j] = 0.0f; rimb[j] = 0.0f; rra[j] = 0.0f; rrb[j] = 0.0f; rnm[j] = 0.0f;
                        rpoly[j] = false; rrAx[j] = 0.0f; rrAy[j] = 0.0f; rlnx[j] = 0.0f; rlny[j] = 0.0f; rlpx[j] = 0.0f; rlpy[j] = 0.0f;
                    }
                }
            }
#ifdef KB_PROFILE
            if (lane == 0) atomicMax(&misc[M_PROF], (unsigned)maxD);
            if (tid == 0) atomicAdd(&KB_PROF(M_PROF_DEPTH), (unsigned)maxD);
            KB_STAMP(7);     // register load + depth sweep of wave 0 (no barrier: wave-local time)
#endif
#define KB_REG_ROUNDS(...)                                                                          \
            _Pragma("unroll") for (int j = 0; j < KREG; ++j)                                        \
                for (int d_ = dlo[j]; d_ <= dhi[j]; ++d_) {                                         \
                    if (rvalid[j] && rdepth[j] == d_) { __VA_ARGS__ }                               \
                    wave_sync();                                                                    \
                }

            KB_ABLATE_EXIT(9);     // register set-up: light load, depth pass, dealing, full load
            // manifold constraints of this wave's islands (uniform mask)
            unsigned long long myMc = 0ull;
            if (OBJ) {
                const unsigned long long mm = mcMask[wave];
                // (readfirstlane returns int: go through unsigned, or bit 31 smears into the upper half)
                myMc = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(mm >> 32)) << 32) |
                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)mm);
                // only candidates that exist: a stray bit (round 1 met one, a sign-extended bit 31) must never index a
                // record behind the manifold-constraint area
                myMc &= NMC >= 64 ? ~0ull : ((1ull << NMC) - 1ull);
            }
            // b2ContactSolver::WarmStart
            KB_REG_ROUNDS({
                const int a = ra[j], b = rb[j];
                const float Px = racc[j] * rnx[j], Py = racc[j] * rny[j];
                if (OBJ && rpoly[j]) {   // A = polygon b, B = kilobot a
                    const int m = b - N;
                    objW[m] -= objBody[m * BT_WORDS + BT_II] * (rrAx[j] * Py - rrAy[j] * Px);
                    vel[b].x -= R_IMB(j) * Px; vel[b].y -= R_IMB(j) * Py;
                    vel[a].x += R_IMA(j) * Px; vel[a].y += R_IMA(j) * Py;
                } else {
                    if (a < WALL_CODE) { vel[a].x -= R_IMA(j) * Px; vel[a].y -= R_IMA(j) * Py; }
                    vel[b].x += R_IMB(j) * Px; vel[b].y += R_IMB(j) * Py;
                }
            })
            if (OBJ && myMc) { mc_warm_pass(myMc, lane == 0); wave_sync(); }
            KB_STAMP_PRE(29);    // (profile build) warm start of wave 0, since the register set-up
            KB_ABLATE_EXIT(10);    // warm start
            // SolveVelocityConstraints: friction 0, restitution 0, one manifold point.  Slot by slot, level by level; only the
            // lanes whose contact is of the round's level execute (EXEC mask).
            const int DUMMY = NB - 1;
            for (int it = 0; it < p.vel_iters; ++it) {
#pragma unroll
                for (int j = 0; j < KREG; ++j) {
                    for (int d_ = dlo[j]; d_ <= dhi[j]; ++d_) {
                        if (rvalid[j] && rdepth[j] == d_) {
                            if (OBJ && rpoly[j]) {   // kilobot - polygon contact: one point, friction sqrt(0 * f) = 0
                                const int a = ra[j], b = rb[j], m = b - N;
                                const float wA = objW[m];
                                const float dvx = (vel[a].x - vel[b].x) - (-wA * rrAy[j]), dvy = (vel[a].y - vel[b].y) - (wA * rrAx[j]);
                                const float vn = dvx * rnx[j] + dvy * rny[j];
                                float lambda = -(rnm[j] * vn);
                                const float accOld = racc[j];
                                const float newimp = fmaxf(accOld + lambda, 0.0f);
                                lambda = newimp - accOld;
                                racc[j] = newimp;
                                const float Px = lambda * rnx[j], Py = lambda * rny[j];
                                vel[b].x -= rimb[j] * Px; vel[b].y -= rimb[j] * Py;
                                objW[m] = wA - objBody[m * BT_WORDS + BT_II] * (rrAx[j] * Py - rrAy[j] * Px);
                                vel[a].x += rima[j] * Px; vel[a].y += rima[j] * Py;
                            } else {
                                const bool wallA = ra[j] >= WALL_CODE;
                                const int ia = wallA ? DUMMY : ra[j], ib = rb[j];
                                const float2 va_ = vel[ia], vb_ = vel[ib];
                                const float nx = rnx[j], ny = rny[j];
                                const float ima = R_IMA(j), imb = R_IMB(j);
                                const float ax_ = wallA ? 0.0f : va_.x, ay_ = wallA ? 0.0f : va_.y;
                                const float dvx = vb_.x - ax_, dvy = vb_.y - ay_;
                                const float vn = dvx * nx + dvy * ny;
                                float lambda = -(R_NM(j) * vn);
                                const float accOld = racc[j];
                                const float newimp = fmaxf(accOld + lambda, 0.0f);
                                lambda = newimp - accOld;
                                racc[j] = newimp;
                                const float Px = lambda * nx, Py = lambda * ny;
                                vel[ia] = make_float2(ax_ - ima * Px, ay_ - ima * Py);
                                vel[ib] = make_float2(vb_.x + imb * Px, vb_.y + imb * Py);
                            }
                        }
                        wave_sync();
                    }
                }
                if (OBJ && myMc) { mc_velocity_pass(myMc, lane == 0); wave_sync(); }
            }
            KB_STAMP_PRE(30);    // ... + its 10 velocity sweeps, before the barrier
            __syncthreads();
            KB_STAMP(4);
            KB_RETID();
            KB_ABLATE_EXIT(11);    // velocity sweeps
            mc_store();
#ifdef KB_PROFILE
            if (tid == 0) atomicAdd(&KB_PROF(9), misc[M_PROF]);   // deepest wave of the env (replaces the contacts-per-wave slot)
#endif
            // StoreImpulses -> packed warm-start list of the next substep (LDS image and/or global)
            const bool last = sub == p.n_substeps - 1;
#pragma unroll
            for (int j = 0; j < KREG; ++j) {
                if (BINS) {      // the contact is staged at its position in the packed list: the list goes out as a whole at the end of the substep
                    if (rvalid[j]) lAcc[rc[j]] = racc[j];
                    continue;
                }
                if (!rvalid[j] || rslot[j] == 255) continue;
                const int a = ra[j], b = rb[j];
                const int owner = a < WALL_CODE ? a : b;
                const unsigned key16 = a >= WALL_CODE ? (unsigned)a : (b >= N ? (unsigned)OBJ_CODE + ((lInfo[rc[j]] >> 24) & 15u) : (unsigned)b);
                const unsigned pos = (unsigned)newOff[owner] + (unsigned)rslot[j];
                if (pos >= (unsigned)p.cap) continue;
                if (newInLds) { oldKey[pos] = (unsigned short)key16; oldAcc[pos] = racc[j]; }
                if (last || !newInLds) {
                    g.ws_key[wo + pos] = key16 >= (unsigned)WALL_CODE ? KEY_WALL + (key16 - WALL_CODE)
                                       : (key16 >= (unsigned)OBJ_CODE ? KEY_OBJ + (key16 - OBJ_CODE) : key16);
                    g.ws_acc[wo + pos] = racc[j];
                }
            }
            // integrate positions (b2Island::Solve)
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b_ = tid + q * nt;
                if (b_ >= N) continue;
                const int b = KB_SLOT(q, b_);
                float vxx = vel[b].x, vyy = vel[b].y, ww = bw[q];
                if (BINS) { startX[b] = pos[b].x; startY[b] = pos[b].y; }     // pose at the start of the substep (continuous step)
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
                pos[b].x += h * vxx; pos[b].y += h * vyy;
                th[q] += h * ww;
                vel[b].x = vxx; vel[b].y = vyy; bw[q] = ww;
                // (islWave is idle from here on) island state for the sleep bookkeeping: bit 1 = has an awake body,
                // bit 0 = position constraints not solved (set by the last position sweep; always without sweeps)
                if (SLEEP) islWave[b] = (unsigned char)((active[b] ? 2 : 0) | (p.pos_iters <= 0 ? 1 : 0));
            }
            if (tid < M) {   // objects: same integrator (b2Island writes the clamped velocity back to the body)
                const int b = N + tid;
                float vxx = vel[b].x, vyy = vel[b].y, ww = objW[tid];
                const float tx = h * vxx, ty = h * vyy;
                if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
                    const float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
                    vxx *= ratio; vyy *= ratio;
                }
                const float rot = h * ww;
                if (rot * rot > B2_MAX_ROTATION_SQ) ww *= B2_MAX_ROTATION / fabsf(rot);
                vel[b].x = vxx; vel[b].y = vyy; objW[tid] = ww;
                pos[b].x += h * vxx; pos[b].y += h * vyy;
                objA[tid] += h * ww;
                if (SLEEP) islWave[b] = (unsigned char)((active[b] ? 2 : 0) | (p.pos_iters <= 0 ? 1 : 0));
            }
            __syncthreads();
            KB_STAMP(5);
            KB_RETID();
            KB_ABLATE_EXIT(12);    // impulses stored, positions integrated
            // SolvePositionConstraints; an island stops once its minSeparation >= -3 slop.  The island flags are read
            // once per iteration; depth levels without an active contact in this wave are skipped altogether.
            for (int it = 0; it < p.pos_iters; ++it) {
                unsigned char *act = active + (it & 1) * NB, *nxt = active + ((it + 1) & 1) * NB;
                bool viol = false;
                bool ron[KREG];
#pragma unroll
                for (int j = 0; j < KREG; ++j) ron[j] = rvalid[j] && act[risl[j]] != 0;
#pragma unroll
                for (int j = 0; j < KREG; ++j) {
                    for (int d_ = dlo[j]; d_ <= dhi[j]; ++d_) {
                        if (ron[j] && rdepth[j] == d_) {
                        const int a = ra[j], b = rb[j];
                        const int isl = risl[j];
                        if (OBJ && rpoly[j]) {   // b2PositionSolverManifold e_faceA, A = polygon b, B = kilobot a
                            const int m = b - N;
                            const float *T = objBody + m * BT_WORDS;
                            const XF xo = body_xf(ox, b);
                            const V2 normal = rot_mul(xo, mk2(rlnx[j], rlny[j]));
                            const V2 planePoint = xf_mul(xo, mk2(rlpx[j], rlpy[j]));
                            const V2 clipPoint = mk2(pos[a].x, pos[a].y);
                            const float sep = v_dot(v_sub(clipPoint, planePoint), normal) - T[BT_RADIUS] - p.r_bot;
                            const V2 rA = v_sub(clipPoint, mk2(pos[b].x, pos[b].y));
                            if (sep < -3.0f * B2_LINEAR_SLOP) { nxt[isl] = 1; viol = true; if (SLEEP && it == p.pos_iters - 1) islWave[isl] = 3; }
                            const float C = kb_clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                            const float rnA = v_cross(rA, normal);
                            const float K = T[BT_IM] + KB_IM_BOT(a) + T[BT_II] * rnA * rnA;
                            const float imp = K > 0.0f ? -C / K : 0.0f;
                            const V2 P = v_scale(imp, normal);
                            pos[b].x -= T[BT_IM] * P.x; pos[b].y -= T[BT_IM] * P.y;
                            objA[m] -= T[BT_II] * v_cross(rA, P);
                            pos[a].x += KB_IM_BOT(a) * P.x; pos[a].y += KB_IM_BOT(a) * P.y;
                        } else {
                        float nx, ny, sep;
                        const float ima = R_IMA(j), imb = R_IMB(j);
                        const float bx = pos[b].x, by = pos[b].y;
                        float axx = 0.0f, ayy = 0.0f;
                        if (a >= WALL_CODE) {
                            float dist, wx, wy;
                            wall_geom(p, a - WALL_CODE, bx, by, dist, wx, wy);
                            nx = rflip[j] ? -wx : wx; ny = rflip[j] ? -wy : wy;   // manifold normal fixed at detection
                            const float along = rflip[j] ? -dist : dist;
                            sep = along - R_RA(j) - R_RB(j);
                        } else {
                            axx = pos[a].x; ayy = pos[a].y;
                            const float dx = bx - axx, dy = by - ayy;
                            const float len = sqrtf(dx * dx + dy * dy);
                            nx = dx; ny = dy;
                            if (!(len < B2_EPSILON)) { const float inv = 1.0f / len; nx = dx * inv; ny = dy * inv; }
                            sep = (dx * nx + dy * ny) - R_RA(j) - R_RB(j);
                        }
                        if (sep < -3.0f * B2_LINEAR_SLOP) { nxt[isl] = 1; viol = true; if (SLEEP && it == p.pos_iters - 1) islWave[isl] = 3; }
                        const float C = kb_clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                        const float K = ima + imb;
                        const float imp = K > 0.0f ? -C / K : 0.0f;
                        const float Px = imp * nx, Py = imp * ny;
                        if (a < WALL_CODE) { pos[a].x = axx - ima * Px; pos[a].y = ayy - ima * Py; }
                        pos[b].x = bx + imb * Px; pos[b].y = by + imb * Py;
                        }
                        }
                        wave_sync();
                    }
                }
                if (OBJ && myMc) { viol |= mc_position_pass(myMc, lane == 0, act, nxt, it == p.pos_iters - 1); wave_sync(); }
#ifdef KB_PROFILE
                if (tid == 0) atomicAdd(&KB_PROF(10), 1u);
#endif
                if (!__any(viol)) break;
                // the flags the next sweep sets must start cleared (only this wave's islands)
#pragma unroll
                for (int j = 0; j < KREG; ++j) if (rvalid[j]) act[risl[j]] = 0;
                if (OBJ && myMc) mc_clear_flags(myMc, lane == 0, act);
                wave_sync();
            }
            KB_STAMP_PRE(24);    // (profile build) wave 0's own position sweeps, before it waits for the slowest wave
#undef KB_REG_ROUNDS
#undef KB_REG_KEY_ROUNDS
#undef R_IMA
#undef R_IMB
#undef R_RA
#undef R_RB
#undef R_NM
        } else {
            // =========================== list solver (staged contact arrays) ===========================
            // (the instantiations at the 80-VGPR budget and the object kernels keep the plain walk: the requests in flight
            // cost ~ 15 VGPRs, which there end in spills of the shape tools/lint_spills.py rejects)
            constexpr bool PF = !OBJ && FN == 0 && (TIER == 0 || WIDE);
            // sN: normal (x, y) and effective mass of every record, written by the warm-start pass and read by the velocity
            // sweeps instead of an IEEE sqrt and two divisions per record and sweep (global staging of the PF kernels only:
            // the LDS staging has no room for them); haveN is a literal at both call sites
            auto solve_list = [&](unsigned *sPair, unsigned *sInfo, float *sAcc, unsigned short *cbk,
                                  unsigned short *order, float *sN, const bool haveN) __attribute__((always_inline)) {
                const bool HN = PF && haveN;
                float *sNx = sN, *sNy = sN + p.cap, *sNm = sN + 2 * p.cap;
                // a "round" = all contacts of one key owned by this (virtual) wave.  coop: the workgroup is one
                // virtual wave and rounds are separated by s_barrier; otherwise every wave runs alone.
                const int myw = coop ? 0 : wave;
                const int lid = coop ? tid : lane;
                const int stride = coop ? nt : 64;
                const int nl = nList[myw];
                // manifold constraints swept by this (virtual) wave; its leader thread runs them
                unsigned long long myMc = 0ull;
                if (OBJ) {
                    const unsigned long long mm = mcMask[coop ? nw : wave];
                    myMc = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(mm >> 32)) << 32) |
                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)mm);
                    myMc &= NMC >= 64 ? ~0ull : ((1ull << NMC) - 1ull);      // only candidates that exist
                }
                const bool leader = lid == 0;
#ifdef KB_PROFILE
#define KB_ROUND_SYNC() do { if (tid == 0) atomicAdd(&KB_PROF(8), 1u); if (coop) __syncthreads(); else wave_sync(); } while (0)   // rounds of wave 0 (all passes)
#else
#define KB_ROUND_SYNC() do { if (coop) __syncthreads(); else wave_sync(); } while (0)
#endif
                // The records of a round are staged (global scratch for dense envs) behind dependent indirections (bucket list
                // -> bucket bounds -> order -> record).  None of them changes during the solve and the thread that sweeps a
                // record is the same in every pass, so they run two list entries ahead of the sweep: the bounds and the index of
                // this thread's first record of entry li + 2 are requested behind the sweep of entry li (before its barrier),
                // the record of entry li + 1 (pair, info, impulse) in front of it.  A round then only waits for the body
                // velocities / positions in LDS.  Nothing waits for a request in the round that issues it (the index is kept
                // as loaded and resolved one entry later).  Further records of a thread in one entry (buckets larger than
                // the sweeping group) are read in place.
                auto pf_entry = [&](int li_, int &bk_, int &s_, int &e_, unsigned &raw_) __attribute__((always_inline)) {
                    bk_ = 0; s_ = 0; e_ = 0; raw_ = 0u;
                    if (li_ < nl) {
                        bk_ = bkList[myw * BK_PER_WAVE + li_];
                        s_ = (int)bkStart[bk_]; e_ = (int)bkStart[bk_ + 1];
                        raw_ = order[s_ + lid < e_ ? s_ + lid : 0];
                    }
                };
#define KB_FOR_ROUNDS_PF(WITH_ACC, WITH_N, ...)                                                           \
                {                                                                                   \
                int bk0_, s0_, e0_, bk1_, s1_, e1_; unsigned raw0_, raw1_;                          \
                pf_entry(0, bk0_, s0_, e0_, raw0_);                                                 \
                pf_entry(1, bk1_, s1_, e1_, raw1_);                                                 \
                int c0_ = s0_ + lid < e0_ ? (int)raw0_ : -1;                                        \
                unsigned pr0_ = sPair[c0_ < 0 ? 0 : c0_], inf0_ = sInfo[c0_ < 0 ? 0 : c0_];         \
                float acc0_ = WITH_ACC ? sAcc[c0_ < 0 ? 0 : c0_] : 0.0f;                            \
                float nx0_ = 0.0f, ny0_ = 0.0f, nm0_ = 0.0f;                                        \
                if (WITH_N && HN) { nx0_ = sNx[c0_ < 0 ? 0 : c0_]; ny0_ = sNy[c0_ < 0 ? 0 : c0_]; nm0_ = sNm[c0_ < 0 ? 0 : c0_]; } \
                for (int li = 0; li < nl; ++li) {                                                   \
                    const int c1_ = s1_ + lid < e1_ ? (int)raw1_ : -1;                              \
                    const unsigned pr1_ = sPair[c1_ < 0 ? 0 : c1_], inf1_ = sInfo[c1_ < 0 ? 0 : c1_]; \
                    const float acc1_ = WITH_ACC ? sAcc[c1_ < 0 ? 0 : c1_] : 0.0f;                  \
                    float nx1_ = 0.0f, ny1_ = 0.0f, nm1_ = 0.0f;                                    \
                    if (WITH_N && HN) { nx1_ = sNx[c1_ < 0 ? 0 : c1_]; ny1_ = sNy[c1_ < 0 ? 0 : c1_]; nm1_ = sNm[c1_ < 0 ? 0 : c1_]; } \
                    int bk2_, s2_, e2_; unsigned raw2_;                                             \
                    const int bk = bk0_, s_ = s0_, e_ = e0_;                                        \
                    if ((bk % RK) < RK - 1) {                                                       \
                        if (c0_ >= 0) do {                                                          \
                            const int c = c0_;                                                      \
                            const unsigned KB_PR = pr0_, KB_INF = inf0_;                            \
                            const float KB_ACC = acc0_;                                             \
                            const float KB_NX = nx0_, KB_NY = ny0_, KB_NM = nm0_;                   \
                            __VA_ARGS__                                                             \
                        } while (0);                                                                \
                        for (int i_ = s_ + lid + stride; i_ < e_; i_ += stride) {                   \
                            const int c = (int)order[i_];                                           \
                            const unsigned KB_PR = sPair[c], KB_INF = sInfo[c];                     \
                            const float KB_ACC = WITH_ACC ? sAcc[c] : 0.0f;                         \
                            float KB_NX = 0.0f, KB_NY = 0.0f, KB_NM = 0.0f;                         \
                            if (WITH_N && HN) { KB_NX = sNx[c]; KB_NY = sNy[c]; KB_NM = sNm[c]; }   \
                            __VA_ARGS__                                                             \
                        }                                                                           \
                        pf_entry(li + 2, bk2_, s2_, e2_, raw2_);                                    \
                        KB_ROUND_SYNC();                                                            \
                    } else {                                                                        \
                        pf_entry(li + 2, bk2_, s2_, e2_, raw2_);                                    \
                        const int maxr_ = (int)bkMaxRank[bk / RK];                                  \
                        for (int r_ = RK - 1; r_ <= maxr_; ++r_) {                                  \
                            if (c0_ >= 0 && (int)((inf0_ >> 8) & 0xFF) == r_) do {                  \
                                const int c = c0_;                                                  \
                                const unsigned KB_PR = pr0_, KB_INF = inf0_;                        \
                                const float KB_ACC = acc0_;                                         \
                                const float KB_NX = nx0_, KB_NY = ny0_, KB_NM = nm0_;               \
                                __VA_ARGS__                                                         \
                            } while (0);                                                            \
                            for (int i_ = s_ + lid + stride; i_ < e_; i_ += stride) {               \
                                const int c = (int)order[i_];                                       \
                                const unsigned KB_PR = sPair[c], KB_INF = sInfo[c];                 \
                                if ((int)((KB_INF >> 8) & 0xFF) == r_) {                            \
                                    const float KB_ACC = WITH_ACC ? sAcc[c] : 0.0f;                 \
                                    float KB_NX = 0.0f, KB_NY = 0.0f, KB_NM = 0.0f;                 \
                                    if (WITH_N && HN) { KB_NX = sNx[c]; KB_NY = sNy[c]; KB_NM = sNm[c]; } \
                                    __VA_ARGS__                                                     \
                                }                                                                   \
                            }                                                                       \
                            KB_ROUND_SYNC();                                                        \
                        }                                                                           \
                    }                                                                               \
                    c0_ = c1_; pr0_ = pr1_; inf0_ = inf1_; acc0_ = acc1_; nx0_ = nx1_; ny0_ = ny1_; nm0_ = nm1_; \
                    bk0_ = bk1_; s0_ = s1_; e0_ = e1_;                                              \
                    bk1_ = bk2_; s1_ = s2_; e1_ = e2_; raw1_ = raw2_;                               \
                }                                                                                   \
                }
#define KB_FOR_ROUNDS_PLAIN(WITH_ACC, ...)                                                          \
                for (int li = 0; li < nl; ++li) {                                                   \
                    const int bk = bkList[myw * BK_PER_WAVE + li];                                  \
                    const int s_ = (int)bkStart[bk], e_ = (int)bkStart[bk + 1];                     \
                    if ((bk % RK) < RK - 1) {                                                       \
                        for (int i_ = s_ + lid; i_ < e_; i_ += stride) {                            \
                            const int c = order[i_];                                                \
                            const unsigned KB_PR = sPair[c], KB_INF = sInfo[c];                     \
                            const float KB_ACC = WITH_ACC ? sAcc[c] : 0.0f;                         \
                            const float KB_NX = 0.0f, KB_NY = 0.0f, KB_NM = 0.0f;                   \
                            __VA_ARGS__                                                             \
                        }                                                                           \
                        KB_ROUND_SYNC();                                                            \
                    } else {                                                                        \
                        const int maxr_ = (int)bkMaxRank[bk / RK];                                  \
                        for (int r_ = RK - 1; r_ <= maxr_; ++r_) {                                  \
                            for (int i_ = s_ + lid; i_ < e_; i_ += stride) {                        \
                                const int c = order[i_];                                            \
                                const unsigned KB_PR = sPair[c], KB_INF = sInfo[c];                 \
                                if ((int)((KB_INF >> 8) & 0xFF) == r_) {                            \
                                    const float KB_ACC = WITH_ACC ? sAcc[c] : 0.0f;                 \
                                    const float KB_NX = 0.0f, KB_NY = 0.0f, KB_NM = 0.0f;           \
                                    __VA_ARGS__                                                     \
                                }                                                                   \
                            }                                                                       \
                            KB_ROUND_SYNC();                                                        \
                        }                                                                           \
                    }                                                                               \
                }
#define KB_FOR_ROUNDS(WITH_ACC, WITH_N, ...)                                                        \
                if constexpr (PF) { KB_FOR_ROUNDS_PF(WITH_ACC, WITH_N, __VA_ARGS__) } else { KB_FOR_ROUNDS_PLAIN(WITH_ACC, __VA_ARGS__) }
#define KB_VEL_NORMAL(a, b, flip, nx, ny)                                                           \
                float nx, ny;                                                                       \
                if (a >= WALL_CODE) {                                                               \
                    float dist_;                                                                    \
                    wall_geom(p, a - WALL_CODE, pos[b].x, pos[b].y, dist_, nx, ny);                       \
                    if (flip) { nx = -nx; ny = -ny; }                                               \
                } else {                                                                            \
                    const float dx_ = pos[b].x - pos[a].x, dy_ = pos[b].y - pos[a].y;                           \
                    const float dd_ = dx_ * dx_ + dy_ * dy_;                                        \
                    nx = 1.0f; ny = 0.0f;                                                           \
                    if (dd_ > B2_EPSILON * B2_EPSILON) {                                            \
                        const float len_ = sqrtf(dd_);                                              \
                        const float inv_ = 1.0f / len_;                                             \
                        nx = dx_ * inv_; ny = dy_ * inv_;                                           \
                    }                                                                               \
                }
                // b2ContactSolver::WarmStart
                KB_FOR_ROUNDS(true, false, {
                    const unsigned pr = KB_PR;
                    const int a = pr & 0xFFFF, b = pr >> 16;
                    const bool flip = (KB_INF & 0x80) != 0;
                    if (bpoly(b)) {   // kilobot a - polygon b: Box2D's A = the polygon, B = the kilobot
                        const int m = b - N;
                        const float *T = objTab + ((KB_INF >> 24) & 15u) * OT_WORDS;      // the fixture that is touched
                        PolyCon pc;
                        poly_contact_setup(T, body_xf(ox, b), pos[b].x, pos[b].y, mk2(pos[a].x, pos[a].y), p.r_bot, KB_IM_BOT(a), pc);
                        const float acc = KB_ACC;
                        const float Px = acc * pc.normal.x, Py = acc * pc.normal.y;
                        objW[m] -= T[OT_II] * (pc.rA.x * Py - pc.rA.y * Px);
                        vel[b].x -= T[OT_IM] * Px; vel[b].y -= T[OT_IM] * Py;
                        vel[a].x += KB_IM_BOT(a) * Px; vel[a].y += KB_IM_BOT(a) * Py;
                    } else {
                    KB_VEL_NORMAL(a, b, flip, nx, ny)
                    const float acc = KB_ACC;
                    const float Px = acc * nx, Py = acc * ny;
                    const float ima = bim(a), imb = bim(b);
                    if (HN) {
                        const float k_ = ima + imb;
                        sNx[c] = nx; sNy[c] = ny; sNm[c] = k_ > 0.0f ? 1.0f / k_ : 0.0f;
                    }
                    if (a < WALL_CODE) { vel[a].x -= ima * Px; vel[a].y -= ima * Py; }
                    vel[b].x += imb * Px; vel[b].y += imb * Py;
                    }
                })
                if (OBJ && myMc) { mc_warm_pass(myMc, leader); KB_ROUND_SYNC(); }
                // SolveVelocityConstraints
                for (int it = 0; it < p.vel_iters; ++it) {
                    KB_FOR_ROUNDS(true, true, {
                        const unsigned pr = KB_PR;
                        const int a = pr & 0xFFFF, b = pr >> 16;
                        const bool flip = (KB_INF & 0x80) != 0;
                        if (bpoly(b)) {   // kilobot a - polygon b: one point, friction sqrt(0 * f) = 0
                            const int m = b - N;
                            const float *T = objTab + ((KB_INF >> 24) & 15u) * OT_WORDS;
                            PolyCon pc;
                            poly_contact_setup(T, body_xf(ox, b), pos[b].x, pos[b].y, mk2(pos[a].x, pos[a].y), p.r_bot, KB_IM_BOT(a), pc);
                            const float wA = objW[m];
                            const float dvx = (vel[a].x - vel[b].x) - (-wA * pc.rA.y), dvy = (vel[a].y - vel[b].y) - (wA * pc.rA.x);
                            const float vn = dvx * pc.normal.x + dvy * pc.normal.y;
                            float lambda = -(pc.nmass * vn);
                            const float accOld = KB_ACC;
                            const float newimp = fmaxf(accOld + lambda, 0.0f);
                            lambda = newimp - accOld;
                            sAcc[c] = newimp;
                            const float Px = lambda * pc.normal.x, Py = lambda * pc.normal.y;
                            vel[b].x -= T[OT_IM] * Px; vel[b].y -= T[OT_IM] * Py;
                            objW[m] = wA - T[OT_II] * (pc.rA.x * Py - pc.rA.y * Px);
                            vel[a].x += KB_IM_BOT(a) * Px; vel[a].y += KB_IM_BOT(a) * Py;
                            continue;
                        }
                        float vax = 0.0f, vay = 0.0f;
                        if (a < WALL_CODE) { vax = vel[a].x; vay = vel[a].y; }       // (requested together with the positions)
                        const float vbx = vel[b].x, vby = vel[b].y;
                        const float ima = bim(a), imb = bim(b);
                        float nx = KB_NX, ny = KB_NY, nm = KB_NM;        // (the warm-start pass left them with the record)
                        if (!HN) {
                            KB_VEL_NORMAL(a, b, flip, nx2_, ny2_)
                            const float k = ima + imb;
                            nx = nx2_; ny = ny2_; nm = k > 0.0f ? 1.0f / k : 0.0f;
                        }
                        const float dvx = vbx - vax, dvy = vby - vay;
                        const float vn = dvx * nx + dvy * ny;
                        float lambda = -(nm * vn);
                        const float accOld = KB_ACC;
                        const float newimp = fmaxf(accOld + lambda, 0.0f);
                        lambda = newimp - accOld;
                        sAcc[c] = newimp;
                        const float Px = lambda * nx, Py = lambda * ny;
                        if (a < WALL_CODE) { vel[a].x = vax - ima * Px; vel[a].y = vay - ima * Py; }
                        vel[b].x = vbx + imb * Px; vel[b].y = vby + imb * Py;
                    })
                    if (OBJ && myMc) { mc_velocity_pass(myMc, leader); KB_ROUND_SYNC(); }
                }
                __syncthreads();
                KB_STAMP(4);
                KB_RETID();
                mc_store();
                // StoreImpulses -> packed warm-start list of the next substep
                const bool last = sub == p.n_substeps - 1;
                for (int c = tid; c < (BINS ? 0 : ncon); c += nt) {      // (sorted bins: the staged contacts are the packed list; it goes out at the end of the substep)
                    const unsigned inf = sInfo[c];
                    const int sl = (inf >> 16) & 0xFF;
                    if (sl == 255) continue;
                    const unsigned pr = sPair[c];
                    const int a = pr & 0xFFFF, b = pr >> 16;
                    const int owner = a < WALL_CODE ? a : b;
                    const float acc = sAcc[c];
                    const unsigned key16 = a >= WALL_CODE ? (unsigned)a : (b >= N ? (unsigned)OBJ_CODE + ((inf >> 24) & 15u) : (unsigned)b);
                    const unsigned pos = (unsigned)newOff[owner] + (unsigned)sl;
                    if (pos >= (unsigned)p.cap) continue;
                    if (newInLds) { oldKey[pos] = (unsigned short)key16; oldAcc[pos] = acc; }
                    if (last || !newInLds) {
                        g.ws_key[wo + pos] = key16 >= (unsigned)WALL_CODE ? KEY_WALL + (key16 - WALL_CODE)
                                           : (key16 >= (unsigned)OBJ_CODE ? KEY_OBJ + (key16 - OBJ_CODE) : key16);
                        g.ws_acc[wo + pos] = acc;
                    }
                }
                // integrate positions (b2Island::Solve)
#pragma unroll
                for (int q = 0; q < BPT; ++q) {
                    const int b_ = tid + q * nt;
                    if (b_ >= N) continue;
                    const int b = KB_SLOT(q, b_);
                    float vxx = vel[b].x, vyy = vel[b].y, ww = bw[q];
                    if (BINS) { startX[b] = pos[b].x; startY[b] = pos[b].y; }
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
                    pos[b].x += h * vxx; pos[b].y += h * vyy;
                    th[q] += h * ww;
                    vel[b].x = vxx; vel[b].y = vyy; bw[q] = ww;
                    if (SLEEP) islWave[b] = (unsigned char)((active[b] ? 2 : 0) | (p.pos_iters <= 0 ? 1 : 0));
                }
                if (tid < M) {   // objects
                    const int b = N + tid;
                    float vxx = vel[b].x, vyy = vel[b].y, ww = objW[tid];
                    const float tx = h * vxx, ty = h * vyy;
                    if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
                        const float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
                        vxx *= ratio; vyy *= ratio;
                    }
                    const float rot = h * ww;
                    if (rot * rot > B2_MAX_ROTATION_SQ) ww *= B2_MAX_ROTATION / fabsf(rot);
                    vel[b].x = vxx; vel[b].y = vyy; objW[tid] = ww;
                    pos[b].x += h * vxx; pos[b].y += h * vyy;
                    objA[tid] += h * ww;
                    if (SLEEP) islWave[b] = (unsigned char)((active[b] ? 2 : 0) | (p.pos_iters <= 0 ? 1 : 0));
                }
                __syncthreads();
                KB_STAMP(5);
                KB_RETID();
                // SolvePositionConstraints; an island stops once its minSeparation >= -3 slop
                for (int it = 0; it < p.pos_iters; ++it) {
                    unsigned char *act = active + (it & 1) * NB, *nxt = active + ((it + 1) & 1) * NB;
                    bool viol = false;
                    KB_FOR_ROUNDS(false, false, {
                        const unsigned pr = KB_PR;
                        const int a = pr & 0xFFFF, b = pr >> 16;
                        const int isl = (int)parent[b];
                        if (act[isl] && bpoly(b)) {
                            // b2PositionSolverManifold e_faceA, A = polygon b, B = kilobot a; the manifold is the one of
                            // the start-of-substep poses
                            const int m = b - N;
                            const float *T = objTab + ((KB_INF >> 24) & 15u) * OT_WORDS;
                            V2 ln, lp;
                            collide_poly_circle(T, xf_of_body(objBody + m * BT_WORDS, start[b].x, start[b].y, objA0[m]), mk2(start[a].x, start[a].y), p.r_bot, ln, lp);
                            const XF xo = body_xf(ox, b);
                            const V2 normal = rot_mul(xo, ln);
                            const V2 planePoint = xf_mul(xo, lp);
                            const V2 clipPoint = mk2(pos[a].x, pos[a].y);
                            const float sep = v_dot(v_sub(clipPoint, planePoint), normal) - T[OT_RADIUS] - p.r_bot;
                            const V2 rA = v_sub(clipPoint, mk2(pos[b].x, pos[b].y));
                            if (sep < -3.0f * B2_LINEAR_SLOP) { nxt[isl] = 1; viol = true; if (SLEEP && it == p.pos_iters - 1) islWave[isl] = 3; }
                            const float C = kb_clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                            const float rnA = v_cross(rA, normal);
                            const float K = T[OT_IM] + KB_IM_BOT(a) + T[OT_II] * rnA * rnA;
                            const float imp = K > 0.0f ? -C / K : 0.0f;
                            const V2 P = v_scale(imp, normal);
                            pos[b].x -= T[OT_IM] * P.x; pos[b].y -= T[OT_IM] * P.y;
                            objA[m] -= T[OT_II] * v_cross(rA, P);
                            pos[a].x += KB_IM_BOT(a) * P.x; pos[a].y += KB_IM_BOT(a) * P.y;
                        } else if (act[isl]) {
                            float nx, ny, sep;
                            const float ima = bim(a), imb = bim(b), rda = brad(a), rdb = brad(b);
                            const float bx = pos[b].x, by = pos[b].y;
                            float axx = 0.0f, ayy = 0.0f;
                            if (a >= WALL_CODE) {
                                float dist, wx, wy;
                                wall_geom(p, a - WALL_CODE, bx, by, dist, wx, wy);
                                const bool flipped = (KB_INF & 0x80) != 0;   // manifold normal fixed at detection
                                nx = flipped ? -wx : wx; ny = flipped ? -wy : wy;
                                const float along = flipped ? -dist : dist;
                                sep = along - rda - rdb;
                            } else {
                                axx = pos[a].x; ayy = pos[a].y;
                                const float dx = bx - axx, dy = by - ayy;
                                const float len = sqrtf(dx * dx + dy * dy);
                                nx = dx; ny = dy;
                                if (!(len < B2_EPSILON)) { const float inv = 1.0f / len; nx = dx * inv; ny = dy * inv; }
                                sep = (dx * nx + dy * ny) - rda - rdb;
                            }
                            if (sep < -3.0f * B2_LINEAR_SLOP) { nxt[isl] = 1; viol = true; if (SLEEP && it == p.pos_iters - 1) islWave[isl] = 3; }
                            const float C = kb_clampf(B2_BAUMGARTE * (sep + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                            const float K = ima + imb;
                            const float imp = K > 0.0f ? -C / K : 0.0f;
                            const float Px = imp * nx, Py = imp * ny;
                            if (a < WALL_CODE) { pos[a].x = axx - ima * Px; pos[a].y = ayy - ima * Py; }
                            pos[b].x = bx + imb * Px; pos[b].y = by + imb * Py;
                        }
                    })
                    if (OBJ && myMc) { viol |= mc_position_pass(myMc, leader, act, nxt, it == p.pos_iters - 1); KB_ROUND_SYNC(); }
#ifdef KB_PROFILE
                    if (tid == 0) atomicAdd(&KB_PROF(10), 1u);
#endif
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
                    {
                        const int s_ = (int)bkStart[myw * BK_PER_WAVE], e_ = (int)bkStart[(myw + 1) * BK_PER_WAVE];
                        for (int i_ = s_ + lid; i_ < e_; i_ += stride) act[parent[sPair[order[i_]] >> 16]] = 0;
                        if (OBJ && myMc) mc_clear_flags(myMc, leader, act);
                    }
                    KB_ROUND_SYNC();
                }
#undef KB_FOR_ROUNDS
#undef KB_FOR_ROUNDS_PF
#undef KB_FOR_ROUNDS_PLAIN
#undef KB_VEL_NORMAL
#undef KB_ROUND_SYNC
            };
            if (big) {
                // normals + effective masses of the records: 12 B per contact behind the 16-byte records of all envs
                float *gN = reinterpret_cast<float *>(g.scratch) + (size_t)p.E * p.cap * 4 + wo * 3;
                solve_list(gPair, gInfo, gAcc, gCbk, gOrder, gN, true);
            } else solve_list(lPair, lInfo, lAcc, lCbk, lOrder, nullptr, false);
        }
        __syncthreads();
        KB_STAMP_PRE(25);        // ... + waiting for the wave with the most position sweeps
        KB_RETID();
        KB_ABLATE_EXIT(13);    // position sweeps
        if (SLEEP) {
            // ---- b2Island::Solve, the allowSleep block: a body slower than the sleep tolerances accumulates sleep time; an
            // island (awake ones only: bit 1 of islWave[root]) whose bodies have all rested for b2_timeToSleep and whose
            // position constraints converged (bit 0 clear) is put to sleep: b2Body::SetAwake(false) zeroes the sleep time and
            // the velocities.  Bodies of an awake island that were asleep have been woken by it (b2World::Solve).
            // Per island: minimum of the sleep times (non-negative floats order like their bit patterns) in the contact
            // staging area, which is idle between the position sweeps and the continuous step (2 capL >= NB words).
            // (sorted-bin image: wherever NB words are idle by now -- the bin boundaries or the staged pairs, kb_create decides)
            unsigned *islMin = BINS ? reinterpret_cast<unsigned *>(smem + (FN ? ldsb::binE(NB, NP, false, capL_) : p.islmin_off)) : lPair;
            const float linTolSqr = B2_LINEAR_SLEEP_TOL * B2_LINEAR_SLEEP_TOL, angTolSqr = B2_ANGULAR_SLEEP_TOL * B2_ANGULAR_SLEEP_TOL;
            for (int b = tid; b < N + M; b += nt) islMin[b] = 0x7F7FFFFFu;      // b2_maxFloat
            __syncthreads();
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b_ = tid + q * nt;
                if (b_ >= N) continue;
                const int b = KB_SLOT(q, b_);
                const unsigned r = parent[b];
                if (!(islWave[r] & 2)) continue;
                if (slp[q] < 0.0f) slp[q] = 0.0f;
                const float2 v_ = vel[b];
                if (bw[q] * bw[q] > angTolSqr || v_.x * v_.x + v_.y * v_.y > linTolSqr) { slp[q] = 0.0f; atomicMin(&islMin[r], 0u); }
                else { slp[q] += h; atomicMin(&islMin[r], __float_as_uint(slp[q])); }
            }
            if (tid < M) {
                const int b = N + tid;
                const unsigned r = parent[b];
                if (islWave[r] & 2) {
                    float sl = objSlp[tid];
                    if (sl < 0.0f) sl = 0.0f;
                    const float2 v_ = vel[b];
                    const float w_ = objW[tid];
                    if (w_ * w_ > angTolSqr || v_.x * v_.x + v_.y * v_.y > linTolSqr) { sl = 0.0f; atomicMin(&islMin[r], 0u); }
                    else { sl += h; atomicMin(&islMin[r], __float_as_uint(sl)); }
                    objSlp[tid] = sl;
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b_ = tid + q * nt;
                if (b_ >= N) continue;
                const int b = KB_SLOT(q, b_);
                const unsigned r = parent[b];
                const unsigned st_ = islWave[r];
                if ((st_ & 2) && !(st_ & 1) && __uint_as_float(islMin[r]) >= B2_TIME_TO_SLEEP) {
                    slp[q] = -1.0f; vel[b].x = 0.0f; vel[b].y = 0.0f; bw[q] = 0.0f;
                }
            }
            if (tid < M) {
                const int b = N + tid;
                const unsigned r = parent[b];
                const unsigned st_ = islWave[r];
                if ((st_ & 2) && !(st_ & 1) && __uint_as_float(islMin[r]) >= B2_TIME_TO_SLEEP) {
                    objSlp[tid] = -1.0f; vel[b].x = 0.0f; vel[b].y = 0.0f; objW[tid] = 0.0f;
                }
            }
            __syncthreads();       // islMin is read; the continuous step reuses the staging area
        }
        // ---- b2World::SolveTOI: continuous step of every dynamic body against the static walls ----
        // Only bodies that come within their contact radius of a wall can have a TOI event.  They are collected in
        // a candidate list (in the staging area, idle after the solve) and processed one per thread, so that
        // the event logic exists once in the kernel instead of once per unrolled bot slot.
        if (p.toi_walls) {
            // candidate records (body, angle at the start of the substep, angle, angular velocity) in arrays that are idle by now;
            // sorted-bin image: the staged impulses and keys must survive to the end of the substep (they are the packed list)
            float *cTh0 = reinterpret_cast<float *>(BINS ? parent : lInfo), *cTh = BINS ? reinterpret_cast<float *>(lPair) : lAcc;
            float *cW = reinterpret_cast<float *>(BINS ? lOrder : lCbk);
            unsigned short *cSlot = reinterpret_cast<unsigned short *>(active);      // (sorted-bin image)
            const int candMax = BINS ? min(capL_, N) : capL_ / 2;
            int cand[BPT];
            // (sorted bins: nothing has counted on misc[M_NCON] since the drive phase zeroed it, and the arrays the candidates
            //  go to have been idle since the barrier behind the position sweeps)
            if (!BINS) {
                if (tid == 0) misc[M_NCON] = 0;
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b_ = tid + q * nt;
                cand[q] = -1;
                if (b_ >= N) continue;
                const int b = KB_SLOT(q, b_);
                if (SLEEP && slp[q] < 0.0f) continue;           // b2World::SolveTOI skips contacts without an awake dynamic body
                const float total = p.rw_tot;       // r + polygonRadius
                const float xa = BINS ? startX[b] : start[b].x, ya = BINS ? startY[b] : start[b].y, xb = pos[b].x, yb = pos[b].y;
                const float m0 = fminf(fminf(xa - p.xmin, p.xmax - xa), fminf(ya - p.ymin, p.ymax - ya));
                const float m1 = fminf(fminf(xb - p.xmin, p.xmax - xb), fminf(yb - p.ymin, p.ymax - yb));
                if (m0 > total && m1 > total) continue;          // stays clear of every wall: no event possible
                const int i = (int)atomicAdd(&misc[M_NCON], 1u);
                if (i >= candMax) { atomicOr(&misc[M_STATUS], 8u); continue; }
                cand[q] = i;
                if (BINS) cSlot[i] = (unsigned short)b; else lPair[i] = (unsigned)b;
                cTh0[i] = sth0[q]; cTh[i] = th[q]; cW[i] = bw[q];
            }
            __syncthreads();
            KB_STAMP_PRE(26);    // ... + candidates of the continuous step collected
            const int ncand = min((int)misc[M_NCON], candMax);
            for (int i = tid; i < ncand; i += nt) {
                const int b = BINS ? (int)cSlot[i] : (int)lPair[i];
                float R = p.r_bot, im = KB_IM_BOT(b);
                asm volatile("" : "+v"(R), "+v"(im));   // per-lane copies: keeps the two constants out of the scalar file over the event loop
                float x_ = pos[b].x, y_ = pos[b].y, a_ = cTh[i], vx_ = vel[b].x, vy_ = vel[b].y, w_ = cW[i];
                kb_toi_walls_body(p, R, im, BINS ? startX[b] : start[b].x, BINS ? startY[b] : start[b].y, cTh0[i], x_, y_, a_, vx_, vy_, w_);
                pos[b].x = x_; pos[b].y = y_; vel[b].x = vx_; vel[b].y = vy_; cTh[i] = a_; cW[i] = w_;
            }
            if (OBJ && tid < M && !(SLEEP && objSlp[tid] < 0.0f)) {   // objects: the TOI sub-solve runs on the manifold-constraint records of their wall contacts
                Arena ar;
                ar.xmin = p.xmin; ar.ymin = p.ymin; ar.xmax = p.xmax; ar.ymax = p.ymax;
                toi_walls_object(ox, ar, F, tid, start[N + tid].x, start[N + tid].y, objA0[tid], p.h, p.vel_iters,
                                 g.ows_acc + (size_t)e * (MAXOBJ * KB_OWS_COLS * KB_OWS_WORDS));
            }
            KB_STAMP_PRE(27);    // ... + wave 0's own candidates processed
            __syncthreads();
            KB_STAMP_PRE(28);    // ... + every wave's
#pragma unroll
            for (int q = 0; q < BPT; ++q)
                if (cand[q] >= 0) { th[q] = cTh[cand[q]]; bw[q] = cW[cand[q]]; }
        }
        KB_ABLATE_EXIT(14);    // continuous step
        // The state of an object between substeps is its body origin (what one-substep launches store and load, and
        // what the specification's substep does): bodies whose centre of mass is off the origin go through the same
        // origin -> centre of mass conversion inside a fused launch, so that fusing never changes a bit.
        if (OBJ && tid < M && sub + 1 < p.n_substeps &&
            (objBody[tid * BT_WORDS + BT_LCX] != 0.0f || objBody[tid * BT_WORDS + BT_LCY] != 0.0f)) {
            const XF t0 = body_xf(ox, N + tid);                                   // b2Body::SynchronizeTransform
            const XF t1 = xf_make(t0.p.x, t0.p.y, objA[tid]);
            const V2 cm = xf_mul(t1, mk2(objBody[tid * BT_WORDS + BT_LCX], objBody[tid * BT_WORDS + BT_LCY]));
            pos[N + tid].x = cm.x; pos[N + tid].y = cm.y;
        }
        // the new warm-start list becomes the old one (nothing of this is read again behind the last substep of a launch)
        const bool lastSub = sub == p.n_substeps - 1;
        if (PARK && !lastSub) {      // the commands of the next substep's drive phase (the velocity array is idle behind the last barrier)
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int b = tid + q * nt;
                if (b < N) vel[b] = make_float2(g.v[o + b], g.w[o + b]);
            }
        }
        if (!lastSub) for (int b = tid; b < NP; b += nt) { wsCnt[b] = wsCntNew[b]; wsOff[b] = newOff[b]; }
        if (BINS) {
            if (p.toi_walls && !lastSub) __syncthreads();      // (the candidate records of the continuous step lie where the image goes)
            // sorted-bin image: the staged contacts 0 .. newTotal - 1 are the packed list (key = bits 16.. of the info word).
            // It becomes the LDS image of the next substep's lookups, or goes out to the global list (last substep of the
            // launch; always when the contacts were staged in the global scratch slice).
            const int nlist = (int)min(newTotal, (unsigned)stageCap);
            auto list_out = [&](const unsigned *sInfo, const float *sAcc) __attribute__((always_inline)) {      // (one address space per instantiation, as above)
                for (int i = tid; i < nlist; i += nt) {
                    const unsigned key16 = sInfo[i] >> 16;
                    const float acc = sAcc[i];
                    if (newInLds && !lastSub) { oldKey[i] = (unsigned short)key16; oldAcc[i] = acc; }
                    if (lastSub || !newInLds) {
                        g.ws_key[wo + i] = key16 >= (unsigned)WALL_CODE ? KEY_WALL + (key16 - WALL_CODE) : key16;
                        g.ws_acc[wo + i] = acc;
                    }
                }
            };
            if (big) list_out(gInfo, gAcc); else list_out(lInfo, lAcc);
            if (!lastSub) for (int c = tid; c < ldsb::bin_entries(p.nhead) / 2; c += nt) reinterpret_cast<unsigned *>(E1)[c] = 0u;     // (the bucket tables lay over the bin boundaries)
        }
        oldInLds = newInLds;
        oldTotal = newTotal;
        if (!lastSub) __syncthreads();
        KB_STAMP(6);
    }

    // ---- write back ----
    KB_ENV_ADDRESSES();
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = tid + q * nt;
        if (b < N) {
            g.x[o + b] = pos[KB_SLOT(q, b)].x; g.y[o + b] = pos[KB_SLOT(q, b)].y; g.theta[o + b] = th[q];
            if (SLEEP && p.n_substeps > 0) g.sleep_time[o + b] = slp[q];
            if (p.n_substeps > 0) g.ws_cnt[o + b] = wsCntNew[b];      // (the counts of the last substep)
            if (KB_LAW(q) == KB_DRIVE_ACCEL && p.n_substeps > 0 && drive) { g.v[o + b] = cv[q]; g.w[o + b] = cw[q]; }
        }
    }
    if (tid < M && p.n_substeps > 0) {
        const size_t oi = (size_t)e * M + tid;
        g.ox[oi] = pos[N + tid].x; g.oy[oi] = pos[N + tid].y; g.otheta[oi] = objA[tid];
        if (objBody[tid * BT_WORDS + BT_LCX] != 0.0f || objBody[tid * BT_WORDS + BT_LCY] != 0.0f) {   // b2Body::SynchronizeTransform
            const XF t = body_xf(ox, N + tid);
            g.ox[oi] = t.p.x; g.oy[oi] = t.p.y;
        }
        g.ovx[oi] = vel[N + tid].x; g.ovy[oi] = vel[N + tid].y; g.ow[oi] = objW[tid];
        if (SLEEP) g.osleep[oi] = objSlp[tid];
    }
    if (tid == 0) {
        if (LIGHT_TYPE == KB_LIGHT_CIRCULAR && p.light_action && drive) { g.light_x[e] = lx; g.light_y[e] = ly; }
        if (LGEN && p.light_action && drive) {
#pragma unroll
            for (int i = 0; i < KB_MAX_LIGHTS; ++i) {
                if (i >= p.lcount) break;
                g.light_x[(size_t)e * p.lcount + i] = glx[i];
                if (p.light_type != KB_LIGHT_GRADIENT) g.light_y[(size_t)e * p.lcount + i] = gly[i];
                if (p.lkind[i] == KB_LIGHT_MOMENTUM) { g.light_vx[(size_t)e * p.lcount + i] = glvx[i]; g.light_vy[(size_t)e * p.lcount + i] = glvy[i]; }
            }
        }
        if (misc[M_STATUS]) atomicOr(&g.status[e], (int)misc[M_STATUS]);
#ifdef KB_PROFILE
        atomicAdd(&KB_PROF(22), (unsigned)clock64() - prof_t);     // write-back (thread 0's own stores)
        for (int k = 0; k < 8; ++k) g.status[p.E + 40 * e + k] += (int)(KB_PROF(k) >> 4);   // units of 16 cycles
        for (int k = 8; k < 13; ++k) g.status[p.E + 40 * e + k] += (int)KB_PROF(k);
        for (int k = 13; k < 40; ++k) g.status[p.E + 40 * e + k] += (int)(KB_PROF(k) >> 4);
#endif
    }
}


// instantiation chooser of the kb_inst_d*.hip units: objects = 0 none (128 VGPRs), 1 with objects, 2 with objects and a
// one-wave workgroup (the WIDE instantiation), 3 none at 80 VGPRs (six waves per SIMD)
template <int DRIVE_MODE, int LIGHT_TYPE>
static kb_step_fn kb_pick_obj(int objects) {
    if (objects & KB_PICK_SLEEP) {     // kb_config.allow_sleep: generic instantiations with the sleep state (none at 128 / 80 VGPRs, objects, objects + one wave)
        const int o_ = objects & ~KB_PICK_SLEEP;
        if (o_ == 2) return kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, true, 0, 1, true, true, true>;
        if (o_ == 1) return kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, true, 0, 0, true, true, true>;
        if (o_ == 3) return kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, false, 0, 2, true, true, true>;
        return kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, false, 0, 0, true, true, true>;
    }
    if (objects == 2) return kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, true, 0, 1>;
    if (objects == 3) return kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, false, 0, 2>;
    return objects ? kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, true> : kb_step_kernel<DRIVE_MODE, LIGHT_TYPE, false>;
}

}  // namespace kb
