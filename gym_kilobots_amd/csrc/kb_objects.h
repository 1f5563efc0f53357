// kb_objects.h -- pushable objects with Box2D's full contact model: shapes (circle / box / convex polygon),
// narrowphase manifolds (b2CollidePolygonAndCircle, b2CollidePolygons, b2CollideEdgeAndPolygon for a long edge),
// and the manifold constraints (b2ContactSolver with friction and the two-point block solver) that the step kernel
// runs for object-object and object-wall contacts.  Reference surface: gym_kilobots/lib/body.py:129-192, 217-262.
//
// Arithmetic follows b2Math.h operation by operation (-ffp-contract=off), so that the specification evaluates to
// the same bits wherever it runs.
#pragma once
#include "kb_common.h"

namespace kb {

struct V2 { float x, y; };
struct XF { V2 p; float s, c; };   // b2Transform: position + b2Rot (sin, cos)

#define KB_HD __host__ __device__ __forceinline__
KB_HD V2 mk2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
KB_HD V2 v_add(V2 a, V2 b) { return mk2(a.x + b.x, a.y + b.y); }
KB_HD V2 v_sub(V2 a, V2 b) { return mk2(a.x - b.x, a.y - b.y); }
KB_HD V2 v_scale(float s, V2 a) { return mk2(s * a.x, s * a.y); }
KB_HD V2 v_neg(V2 a) { return mk2(-a.x, -a.y); }
KB_HD float v_dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
KB_HD float v_cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
KB_HD V2 v_cross_vs(V2 a, float s) { return mk2(s * a.y, -s * a.x); }   // b2Cross(vector, scalar)
KB_HD V2 v_cross_sv(float s, V2 a) { return mk2(-s * a.y, s * a.x); }   // b2Cross(scalar, vector)
KB_HD V2 v_normalize(V2 a) {                                            // b2Vec2::Normalize
    const float len = sqrtf(a.x * a.x + a.y * a.y);
    if (len < B2_EPSILON) return a;
    const float inv = 1.0f / len;
    return mk2(a.x * inv, a.y * inv);
}
KB_HD V2 rot_mul(const XF &t, V2 v) { return mk2(t.c * v.x - t.s * v.y, t.s * v.x + t.c * v.y); }
KB_HD V2 rot_mulT(const XF &t, V2 v) { return mk2(t.c * v.x + t.s * v.y, -t.s * v.x + t.c * v.y); }
KB_HD V2 xf_mul(const XF &t, V2 v) { return mk2((t.c * v.x - t.s * v.y) + t.p.x, (t.s * v.x + t.c * v.y) + t.p.y); }
KB_HD V2 xf_mulT(const XF &t, V2 v) {
    const float px = v.x - t.p.x, py = v.y - t.p.y;
    return mk2(t.c * px + t.s * py, -t.s * px + t.c * py);
}
KB_HD XF xf_mulT_xf(const XF &A, const XF &B) {                         // b2MulT(A, B) = inv(A) B
    XF C;
    C.s = A.c * B.s - A.s * B.c; C.c = A.c * B.c + A.s * B.s;
    C.p = rot_mulT(A, v_sub(B.p, A.p));
    return C;
}
__device__ __forceinline__ XF xf_make(float px, float py, float ang) {
    XF t;
    t.p = mk2(px, py);
    kb_sincosf(ang, t.s, t.c);
    return t;
}
KB_HD XF xf_identity() { XF t; t.p = mk2(0.0f, 0.0f); t.s = 0.0f; t.c = 1.0f; return t; }

// ---- fixture table: OT_WORDS floats per fixture (kernel parameter -> LDS).  OT_IM / OT_II are those of the body the
//      fixture belongs to (OT_BODY); vertices and normals are in the body frame (origin, not centre of mass) ----
enum { OT_IM = 0, OT_II = 1, OT_RADIUS = 2, OT_BOUND = 3, OT_KIND = 4, OT_N = 5, OT_VERTS = 6, OT_NORMALS = 6 + 2 * KB_MAX_POLY_VERTS,
       OT_BODY = 6 + 4 * KB_MAX_POLY_VERTS, OT_WORDS = 7 + 4 * KB_MAX_POLY_VERTS };
// ---- body table: BT_WORDS floats per object: inverse mass / inertia, b2Sweep::localCenter, skin or circle radius, kind ----
enum { BT_IM = 0, BT_II = 1, BT_LCX = 2, BT_LCY = 3, BT_RADIUS = 4, BT_KIND = 5, BT_WORDS = 6 };
static_assert(BT_WORDS == BT_WORDS_C, "body table size");
static_assert(OT_WORDS == OT_WORDS_C, "object table size");
KB_HD int ot_kind(const float *T) { return (int)T[OT_KIND]; }
KB_HD int ot_n(const float *T) { return (int)T[OT_N]; }
KB_HD int ot_body(const float *T) { return (int)T[OT_BODY]; }
KB_HD V2 ot_v(const float *T, int i) { return mk2(T[OT_VERTS + 2 * i], T[OT_VERTS + 2 * i + 1]); }
KB_HD V2 ot_nrm(const float *T, int i) { return mk2(T[OT_NORMALS + 2 * i], T[OT_NORMALS + 2 * i + 1]); }

// b2PolygonShape::ComputeMass (triangle fan about the vertex average): mass, centroid, inertia about the body origin.
// Host side (kb_create).
inline void polygon_mass(const float *T, float density, float &mass, V2 &centroid, float &inertia) {
    const int n = ot_n(T);
    V2 center = mk2(0.0f, 0.0f), s = mk2(0.0f, 0.0f);
    float area = 0.0f, I = 0.0f;
    for (int i = 0; i < n; ++i) s = v_add(s, ot_v(T, i));
    s = v_scale(1.0f / (float)n, s);
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < n; ++i) {
        const V2 e1 = v_sub(ot_v(T, i), s), e2 = v_sub(ot_v(T, i + 1 < n ? i + 1 : 0), s);
        const float D = v_cross(e1, e2);
        const float triangleArea = 0.5f * D;
        area += triangleArea;
        center = v_add(center, v_scale(triangleArea * k_inv3, v_add(e1, e2)));
        const float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x;
        const float inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
        I += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    const float m = density * area;
    center = v_scale(1.0f / area, center);
    const V2 c = v_add(center, s);
    float Io = density * I;
    Io += m * (v_dot(c, c) - v_dot(center, center));
    mass = m; centroid = c; inertia = Io;
}

// ---- narrowphase ------------------------------------------------------------------------------------------------
// b2CollidePolygonAndCircle: polygon (table T) at xfA, circle centre c (world), circle radius rc
__device__ __forceinline__ bool collide_poly_circle(const float *T, const XF &xfA, V2 c, float rc, V2 &ln, V2 &lp) {
    const int n = ot_n(T);
    const V2 cLocal = xf_mulT(xfA, c);
    int normalIndex = 0;
    float separation = -3.402823466e+38f;
    const float radius = T[OT_RADIUS] + rc;
    for (int i = 0; i < n; ++i) {
        const float sp = v_dot(ot_nrm(T, i), v_sub(cLocal, ot_v(T, i)));
        if (sp > radius) return false;
        if (sp > separation) { separation = sp; normalIndex = i; }
    }
    const int i1 = normalIndex, i2 = i1 + 1 < n ? i1 + 1 : 0;
    const V2 v1 = ot_v(T, i1), v2 = ot_v(T, i2);
    if (separation < B2_EPSILON) {                       // centre inside the polygon
        ln = ot_nrm(T, normalIndex); lp = v_scale(0.5f, v_add(v1, v2));
        return true;
    }
    const float u1 = v_dot(v_sub(cLocal, v1), v_sub(v2, v1));
    const float u2 = v_dot(v_sub(cLocal, v2), v_sub(v1, v2));
    if (u1 <= 0.0f) {
        const V2 dd = v_sub(cLocal, v1);
        if (v_dot(dd, dd) > radius * radius) return false;
        ln = v_normalize(dd); lp = v1;
    } else if (u2 <= 0.0f) {
        const V2 dd = v_sub(cLocal, v2);
        if (v_dot(dd, dd) > radius * radius) return false;
        ln = v_normalize(dd); lp = v2;
    } else {
        const V2 faceCenter = v_scale(0.5f, v_add(v1, v2));
        const float sp = v_dot(v_sub(cLocal, faceCenter), ot_nrm(T, i1));
        if (sp > radius) return false;
        ln = ot_nrm(T, i1); lp = faceCenter;
    }
    return true;
}

// manifold of one object-object / object-wall contact (b2Manifold)
struct Manifold {
    int type, count;          // e_circles 0 / e_faceA 1 / e_faceB 2
    V2 localNormal, localPoint;
    V2 lp[2];
    int id[2];
};

constexpr int FEAT_VERTEX = 0, FEAT_FACE = 1;
__device__ __forceinline__ int feature_key(int ia, int ib, int ta, int tb) { return ia | (ib << 2) | (ta << 4) | (tb << 5); }
struct ClipV { V2 v; int ia, ib, ta, tb; };

__device__ __forceinline__ float find_max_separation(int &edgeIndex, const float *T1, const XF &xf1, const float *T2, const XF &xf2) {
    const XF xf = xf_mulT_xf(xf2, xf1);
    const int n1 = ot_n(T1), n2 = ot_n(T2);
    int bestIndex = 0;
    float maxSeparation = -3.402823466e+38f;
    for (int i = 0; i < n1; ++i) {
        const V2 n = rot_mul(xf, ot_nrm(T1, i));
        const V2 v1 = xf_mul(xf, ot_v(T1, i));
        float si = 3.402823466e+38f;
        for (int j = 0; j < n2; ++j) {
            const float sij = v_dot(n, v_sub(ot_v(T2, j), v1));
            if (sij < si) si = sij;
        }
        if (si > maxSeparation) { maxSeparation = si; bestIndex = i; }
    }
    edgeIndex = bestIndex;
    return maxSeparation;
}

__device__ __forceinline__ int clip_segment_to_line(ClipV vOut[2], const ClipV vIn[2], V2 normal, float offset, int vertexIndexA) {
    int numOut = 0;
    const float distance0 = v_dot(normal, vIn[0].v) - offset;
    const float distance1 = v_dot(normal, vIn[1].v) - offset;
    if (distance0 <= 0.0f) vOut[numOut++] = vIn[0];
    if (distance1 <= 0.0f) vOut[numOut++] = vIn[1];
    if (distance0 * distance1 < 0.0f) {
        const float interp = distance0 / (distance0 - distance1);
        vOut[numOut].v = v_add(vIn[0].v, v_scale(interp, v_sub(vIn[1].v, vIn[0].v)));
        vOut[numOut].ia = vertexIndexA; vOut[numOut].ib = vIn[0].ib;
        vOut[numOut].ta = FEAT_VERTEX; vOut[numOut].tb = FEAT_FACE;
        ++numOut;
    }
    return numOut;
}

// b2CollidePolygons
__device__ inline void collide_polygons(Manifold &m, const float *TA, const XF &xfA, const float *TB, const XF &xfB) {
    m.count = 0;
    const float totalRadius = TA[OT_RADIUS] + TB[OT_RADIUS];
    int edgeA = 0, edgeB = 0;
    const float separationA = find_max_separation(edgeA, TA, xfA, TB, xfB);
    if (separationA > totalRadius) return;
    const float separationB = find_max_separation(edgeB, TB, xfB, TA, xfA);
    if (separationB > totalRadius) return;
    const float k_tol = 0.1f * B2_LINEAR_SLOP;
    const bool flip = separationB > separationA + k_tol;
    const float *T1 = flip ? TB : TA, *T2 = flip ? TA : TB;
    const XF xf1 = flip ? xfB : xfA, xf2 = flip ? xfA : xfB;
    const int edge1 = flip ? edgeB : edgeA;
    m.type = flip ? 2 : 1;
    const int n1 = ot_n(T1), n2 = ot_n(T2);
    ClipV incident[2];
    {   // b2FindIncidentEdge
        const V2 normal1 = rot_mulT(xf2, rot_mul(xf1, ot_nrm(T1, edge1)));
        int index = 0;
        float minDot = 3.402823466e+38f;
        for (int i = 0; i < n2; ++i) { const float dt = v_dot(normal1, ot_nrm(T2, i)); if (dt < minDot) { minDot = dt; index = i; } }
        const int i1 = index, i2 = i1 + 1 < n2 ? i1 + 1 : 0;
        incident[0].v = xf_mul(xf2, ot_v(T2, i1)); incident[0].ia = edge1; incident[0].ib = i1; incident[0].ta = FEAT_FACE; incident[0].tb = FEAT_VERTEX;
        incident[1].v = xf_mul(xf2, ot_v(T2, i2)); incident[1].ia = edge1; incident[1].ib = i2; incident[1].ta = FEAT_FACE; incident[1].tb = FEAT_VERTEX;
    }
    const int iv1 = edge1, iv2 = edge1 + 1 < n1 ? edge1 + 1 : 0;
    V2 v11 = ot_v(T1, iv1), v12 = ot_v(T1, iv2);
    const V2 localTangent = v_normalize(v_sub(v12, v11));
    const V2 localNormal = v_cross_vs(localTangent, 1.0f);
    const V2 planePoint = v_scale(0.5f, v_add(v11, v12));
    const V2 tangent = rot_mul(xf1, localTangent);
    const V2 normal = v_cross_vs(tangent, 1.0f);
    v11 = xf_mul(xf1, v11); v12 = xf_mul(xf1, v12);
    const float frontOffset = v_dot(normal, v11);
    const float sideOffset1 = -v_dot(tangent, v11) + totalRadius;
    const float sideOffset2 = v_dot(tangent, v12) + totalRadius;
    ClipV clip1[2], clip2[2];
    if (clip_segment_to_line(clip1, incident, v_neg(tangent), sideOffset1, iv1) < 2) return;
    if (clip_segment_to_line(clip2, clip1, tangent, sideOffset2, iv2) < 2) return;
    m.localNormal = localNormal; m.localPoint = planePoint;
    int pc = 0;
    for (int i = 0; i < 2; ++i) {
        const float separation = v_dot(normal, clip2[i].v) - frontOffset;
        if (separation <= totalRadius) {
            m.lp[pc] = xf_mulT(xf2, clip2[i].v);
            m.id[pc] = flip ? feature_key(clip2[i].ib, clip2[i].ia, clip2[i].tb, clip2[i].ta)
                            : feature_key(clip2[i].ia, clip2[i].ib, clip2[i].ta, clip2[i].tb);
            ++pc;
        }
    }
    m.count = pc;
}

// wall w: a point on it (vertex v1 of its chain edge, kilobots_env.py:48-51) and its normal into the arena
struct Arena { float xmin, ymin, xmax, ymax; };
__device__ __forceinline__ V2 wall_point(const Arena &p, int wl) {
    switch (wl) {
    case 0: return mk2(p.xmin, p.ymax);
    case 1: return mk2(p.xmin, p.ymin);
    case 2: return mk2(p.xmax, p.ymin);
    default: return mk2(p.xmax, p.ymax);
    }
}
__device__ __forceinline__ V2 wall_normal(int wl) {
    switch (wl) {
    case 0: return mk2(1.0f, 0.0f);
    case 1: return mk2(0.0f, 1.0f);
    case 2: return mk2(-1.0f, 0.0f);
    default: return mk2(0.0f, -1.0f);
    }
}

// b2CollideEdgeAndPolygon for a long two-sided edge with the polygon on its inner side: the edge is the reference
// face, the incident edge is the polygon edge most anti-parallel to the wall normal; both of its vertices are
// candidates (the side planes of a long edge never clip), kept when within 2 * polygonRadius
__device__ inline void collide_wall_poly(Manifold &m, const Arena &p, int wl, const float *T, const XF &xfB) {
    m.count = 0;
    const int n = ot_n(T);
    const V2 nw = wall_normal(wl), v1 = wall_point(p, wl);
    const float radius = 2.0f * B2_POLYGON_RADIUS;
    float edgeSep = 3.402823466e+38f;
    int bestIndex = 0;
    float bestValue = 0.0f;
    for (int i = 0; i < n; ++i) {
        const V2 wv = xf_mul(xfB, ot_v(T, i)), wn = rot_mul(xfB, ot_nrm(T, i));
        const float sp = v_dot(nw, v_sub(wv, v1));
        if (sp < edgeSep) edgeSep = sp;
        const float value = v_dot(nw, wn);
        if (i == 0 || value < bestValue) { bestValue = value; bestIndex = i; }
    }
    if (edgeSep > radius) return;
    const int i1 = bestIndex, i2 = i1 + 1 < n ? i1 + 1 : 0;
    m.type = 1; m.localNormal = nw; m.localPoint = v1;
    int pc = 0;
    for (int k = 0; k < 2; ++k) {
        const int idx = k == 0 ? i1 : i2;
        const float separation = v_dot(nw, v_sub(xf_mul(xfB, ot_v(T, idx)), v1));
        if (separation <= radius) { m.lp[pc] = ot_v(T, idx); m.id[pc] = feature_key(0, idx, FEAT_FACE, FEAT_VERTEX); ++pc; }
    }
    m.count = pc;
}

// ---- manifold constraints in LDS: field-major records, MCN records per env -------------------------------------
enum { MC_A = 0, MC_B, MC_TYPE /* type | count << 2 | vcount << 4 */, MC_ID /* id0 | id1 << 8 */, MC_ISL,
       MC_LNX, MC_LNY, MC_LPX, MC_LPY, MC_P0X, MC_P0Y, MC_P1X, MC_P1Y,
       MC_NI0, MC_NI1, MC_TI0, MC_TI1,
       MC_NX, MC_NY, MC_RA0X, MC_RA0Y, MC_RB0X, MC_RB0Y, MC_RA1X, MC_RA1Y, MC_RB1X, MC_RB1Y,
       MC_NM0, MC_NM1, MC_TM0, MC_TM1, MC_K11, MC_K12, MC_K22, MC_N11, MC_N12, MC_N22, MC_FIELDS };

static_assert(MC_FIELDS == MC_FIELDS_C, "manifold-constraint record size");
// number of (object, partner) candidates of M objects: pairs in lexicographic order, then (object, wall)
KB_HD int mc_candidates(int M) { return M * (M - 1) / 2 + 4 * M; }

// body state as the contact solver sees it (b2Position, b2Velocity, inverse mass / inertia)
struct BState { V2 c; float a; V2 v; float w; float m, i; };

// LDS views the manifold-constraint code works on (a bundle of pointers; scalarised after inlining)
struct ObjCtx {
    float2 *pos, *vel;      // bodies: kilobots 0..N-1, objects N + m
    float *objW, *objA;     // angular velocity / angle of object m
    const float *objTab;    // OT_WORDS floats per fixture
    const float *objBody;   // BT_WORDS floats per object
    float *mc;              // field-major manifold-constraint records, MCN per field
    int N, MCN;
    float mu_oo, mu_ow;     // b2MixFriction = sqrt(f1 f2): object-object, object-wall
};
#define KB_DI __device__ __forceinline__
KB_DI float &mcf(const ObjCtx &x, int field, int t) { return x.mc[field * x.MCN + t]; }
KB_DI int mci(const ObjCtx &x, int field, int t) { return __float_as_int(x.mc[field * x.MCN + t]); }
KB_DI void mci_set(const ObjCtx &x, int field, int t, int v) { x.mc[field * x.MCN + t] = __int_as_float(v); }

KB_DI BState body_get(const ObjCtx &x, int id) {
    BState s;
    if (id >= WALL_CODE) {                                  // static arena body
        s.c = mk2(0.0f, 0.0f); s.a = 0.0f; s.v = mk2(0.0f, 0.0f); s.w = 0.0f; s.m = 0.0f; s.i = 0.0f;
        return s;
    }
    const int m = id - x.N;
    const float2 c = x.pos[id], v = x.vel[id];
    s.c = mk2(c.x, c.y); s.a = x.objA[m]; s.v = mk2(v.x, v.y); s.w = x.objW[m];
    s.m = x.objBody[m * BT_WORDS + BT_IM]; s.i = x.objBody[m * BT_WORDS + BT_II];
    return s;
}
KB_DI void body_put_vel(const ObjCtx &x, int id, const BState &s) {
    if (id >= WALL_CODE) return;
    x.vel[id].x = s.v.x; x.vel[id].y = s.v.y; x.objW[id - x.N] = s.w;
}
KB_DI void body_put_pos(const ObjCtx &x, int id, const BState &s) {
    if (id >= WALL_CODE) return;
    x.pos[id].x = s.c.x; x.pos[id].y = s.c.y; x.objA[id - x.N] = s.a;
}
// b2Body::SynchronizeTransform: q from the angle, p = c - q * localCenter (c = centre of mass)
KB_DI XF xf_of_body(const float *B, float cx, float cy, float a) {
    XF t = xf_make(0.0f, 0.0f, a);
    const float lx = B[BT_LCX], ly = B[BT_LCY];
    t.p = mk2(cx - (t.c * lx - t.s * ly), cy - (t.s * lx + t.c * ly));
    return t;
}
KB_DI XF body_xf(const ObjCtx &x, int id) {
    if (id >= WALL_CODE) return xf_identity();
    return xf_of_body(x.objBody + (id - x.N) * BT_WORDS, x.pos[id].x, x.pos[id].y, x.objA[id - x.N]);
}
KB_DI float body_radius(const ObjCtx &x, int id) {
    return id >= WALL_CODE ? B2_POLYGON_RADIUS : x.objBody[(id - x.N) * BT_WORDS + BT_RADIUS];
}

// candidate t -> (owner fixture, column) of the warm-start table; fixture pairs (f1 < f2) first, then (fixture, wall)
KB_DI void mc_candidate(int M, int t, int &owner, int &col) {
    const int npair = M * (M - 1) / 2;
    if (t < npair) {
        int m1 = 0, left = t;
        while (left >= M - 1 - m1) { left -= M - 1 - m1; ++m1; }
        owner = m1; col = m1 + 1 + left;
    } else {
        owner = (t - npair) >> 2; col = 8 + ((t - npair) & 3);
    }
}

// Manifold of candidate t at the current poses + impulses of the same features in the stored manifold
// (b2Contact::Update); writes the manifold part of record t.  False: not touching.
__device__ __noinline__ bool mc_manifold(const ObjCtx &x, const Arena &p, int M, int t, const float *owsOld) {
    int owner, col;
    mc_candidate(M, t, owner, col);
    Manifold mf;
    mf.type = 0; mf.count = 0; mf.localNormal = mk2(0.0f, 0.0f); mf.localPoint = mk2(0.0f, 0.0f);
    mf.lp[0] = mk2(0.0f, 0.0f); mf.lp[1] = mk2(0.0f, 0.0f); mf.id[0] = 0; mf.id[1] = 0;
    int a, b;
    const int N = x.N;
    if (col < 8) {
        const float *T1 = x.objTab + owner * OT_WORDS, *T2 = x.objTab + col * OT_WORDS;
        const int m1 = ot_body(T1), m2 = ot_body(T2);
        if (m1 == m2) return false;                        // fixtures of one body never collide
        const float dx = x.pos[N + m2].x - x.pos[N + m1].x, dy = x.pos[N + m2].y - x.pos[N + m1].y;
        const float rb = T1[OT_BOUND] + T2[OT_BOUND];
        if (dx * dx + dy * dy > rb * rb) return false;      // bounding circles (stand-in for the broadphase)
        const bool c1 = ot_kind(T1) == KB_SHAPE_CIRCLE, c2 = ot_kind(T2) == KB_SHAPE_CIRCLE;
        if (c1 && c2) {                                     // b2CollideCircles
            const float rr = T1[OT_RADIUS] + T2[OT_RADIUS];
            if (dx * dx + dy * dy > rr * rr) return false;
            a = N + m1; b = N + m2; mf.type = 0; mf.count = 1;
        } else if (!c1 && !c2) {
            a = N + m1; b = N + m2;
            collide_polygons(mf, T1, body_xf(x, a), T2, body_xf(x, b));
            if (mf.count == 0) return false;
        } else {                                            // the polygon is fixture A
            const float *Tp = !c1 ? T1 : T2, *Tc = !c1 ? T2 : T1;
            a = N + ot_body(Tp); b = N + ot_body(Tc);
            V2 ln, lp;
            if (!collide_poly_circle(Tp, body_xf(x, a), mk2(x.pos[b].x, x.pos[b].y), Tc[OT_RADIUS], ln, lp)) return false;
            mf.type = 1; mf.count = 1; mf.localNormal = ln; mf.localPoint = lp;
        }
    } else {
        const int wl = col - 8;
        const float *T = x.objTab + owner * OT_WORDS;
        a = WALL_CODE + wl; b = N + ot_body(T);
        if (ot_kind(T) == KB_SHAPE_CIRCLE) {                // b2CollideEdgeAndCircle, region AB
            float dist, nx, ny;
            switch (wl) {       // wall_geom
            case 0: nx = 1.0f; ny = 0.0f; dist = x.pos[b].x - p.xmin; break;
            case 1: nx = 0.0f; ny = 1.0f; dist = x.pos[b].y - p.ymin; break;
            case 2: nx = -1.0f; ny = 0.0f; dist = p.xmax - x.pos[b].x; break;
            default: nx = 0.0f; ny = -1.0f; dist = p.ymax - x.pos[b].y; break;
            }
            const float rwo = B2_POLYGON_RADIUS + T[OT_RADIUS];
            if (dist * dist > rwo * rwo) return false;
            if (dist < 0.0f) { nx = -nx; ny = -ny; }
            mf.type = 1; mf.count = 1; mf.localNormal = mk2(nx, ny); mf.localPoint = wall_point(p, wl);
        } else {
            collide_wall_poly(mf, p, wl, T, body_xf(x, b));
            if (mf.count == 0) return false;
        }
    }
    // warm start: impulses of the manifold points whose feature id is unchanged
    float nimp[2] = {0.0f, 0.0f}, timp[2] = {0.0f, 0.0f};
    {
        const float *old = owsOld + (owner * KB_OWS_COLS + col) * KB_OWS_WORDS;
        float o[KB_OWS_WORDS];
#pragma unroll
        for (int k = 0; k < KB_OWS_WORDS; ++k) o[k] = old[k];
        for (int j = 0; j < mf.count; ++j)
            for (int k = 0; k < 2; ++k)
                if (o[3 * k] >= 0.0f && (int)o[3 * k] == mf.id[j]) { nimp[j] = o[3 * k + 1]; timp[j] = o[3 * k + 2]; break; }
    }
    mci_set(x, MC_A, t, a); mci_set(x, MC_B, t, b);
    mci_set(x, MC_TYPE, t, mf.type | (mf.count << 2));
    mci_set(x, MC_ID, t, mf.id[0] | (mf.id[1] << 8));
    mcf(x, MC_LNX, t) = mf.localNormal.x; mcf(x, MC_LNY, t) = mf.localNormal.y;
    mcf(x, MC_LPX, t) = mf.localPoint.x; mcf(x, MC_LPY, t) = mf.localPoint.y;
    mcf(x, MC_P0X, t) = mf.lp[0].x; mcf(x, MC_P0Y, t) = mf.lp[0].y; mcf(x, MC_P1X, t) = mf.lp[1].x; mcf(x, MC_P1Y, t) = mf.lp[1].y;
    mcf(x, MC_NI0, t) = nimp[0]; mcf(x, MC_NI1, t) = nimp[1]; mcf(x, MC_TI0, t) = timp[0]; mcf(x, MC_TI1, t) = timp[1];
    return true;
}

// b2ContactSolver::InitializeVelocityConstraints (b2WorldManifold::Initialize inside) of record t at the current poses
__device__ __noinline__ void mc_init_velocity(const ObjCtx &x, int t) {
    const int a = mci(x, MC_A, t), b = mci(x, MC_B, t);
    Manifold mf;
    {
        const int tc = mci(x, MC_TYPE, t);
        mf.type = tc & 3; mf.count = (tc >> 2) & 3;
    }
    mf.localNormal = mk2(mcf(x, MC_LNX, t), mcf(x, MC_LNY, t)); mf.localPoint = mk2(mcf(x, MC_LPX, t), mcf(x, MC_LPY, t));
    mf.lp[0] = mk2(mcf(x, MC_P0X, t), mcf(x, MC_P0Y, t)); mf.lp[1] = mk2(mcf(x, MC_P1X, t), mcf(x, MC_P1Y, t));
    const BState A = body_get(x, a), B = body_get(x, b);
    const XF xfA = body_xf(x, a), xfB = body_xf(x, b);
    const float radA = body_radius(x, a), radB = body_radius(x, b);
    V2 pts[2], normal;
    pts[0] = mk2(0.0f, 0.0f); pts[1] = mk2(0.0f, 0.0f);
    if (mf.type == 0) {
        normal = mk2(1.0f, 0.0f);
        const V2 pointA = xf_mul(xfA, mf.localPoint), pointB = xf_mul(xfB, mf.lp[0]);
        const V2 dd = v_sub(pointB, pointA);
        if (v_dot(dd, dd) > B2_EPSILON * B2_EPSILON) normal = v_normalize(dd);
        const V2 cA = v_add(pointA, v_scale(radA, normal)), cB = v_sub(pointB, v_scale(radB, normal));
        pts[0] = v_scale(0.5f, v_add(cA, cB));
    } else if (mf.type == 1) {
        normal = rot_mul(xfA, mf.localNormal);
        const V2 planePoint = xf_mul(xfA, mf.localPoint);
        for (int j = 0; j < mf.count; ++j) {
            const V2 clipPoint = xf_mul(xfB, mf.lp[j]);
            const V2 cA = v_add(clipPoint, v_scale(radA - v_dot(v_sub(clipPoint, planePoint), normal), normal));
            const V2 cB = v_sub(clipPoint, v_scale(radB, normal));
            pts[j] = v_scale(0.5f, v_add(cA, cB));
        }
    } else {
        normal = rot_mul(xfB, mf.localNormal);
        const V2 planePoint = xf_mul(xfB, mf.localPoint);
        for (int j = 0; j < mf.count; ++j) {
            const V2 clipPoint = xf_mul(xfA, mf.lp[j]);
            const V2 cB = v_add(clipPoint, v_scale(radB - v_dot(v_sub(clipPoint, planePoint), normal), normal));
            const V2 cA = v_sub(clipPoint, v_scale(radA, normal));
            pts[j] = v_scale(0.5f, v_add(cA, cB));
        }
        normal = v_neg(normal);
    }
    int vcount = mf.count;
    const V2 tangent = v_cross_vs(normal, 1.0f);
    V2 rA[2], rB[2];
    float nmass[2] = {0.0f, 0.0f}, tmass[2] = {0.0f, 0.0f};
    rA[1] = mk2(0.0f, 0.0f); rB[1] = mk2(0.0f, 0.0f);
    for (int j = 0; j < mf.count; ++j) {
        rA[j] = v_sub(pts[j], A.c); rB[j] = v_sub(pts[j], B.c);
        const float rnA = v_cross(rA[j], normal), rnB = v_cross(rB[j], normal);
        const float kNormal = A.m + B.m + A.i * rnA * rnA + B.i * rnB * rnB;
        nmass[j] = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        const float rtA = v_cross(rA[j], tangent), rtB = v_cross(rB[j], tangent);
        const float kTangent = A.m + B.m + A.i * rtA * rtA + B.i * rtB * rtB;
        tmass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
    }
    float k11 = 0.0f, k12 = 0.0f, k22 = 0.0f, n11 = 0.0f, n12 = 0.0f, n22 = 0.0f;
    if (mf.count == 2) {                                    // block solver set-up
        const float rn1A = v_cross(rA[0], normal), rn1B = v_cross(rB[0], normal);
        const float rn2A = v_cross(rA[1], normal), rn2B = v_cross(rB[1], normal);
        const float q11 = A.m + B.m + A.i * rn1A * rn1A + B.i * rn1B * rn1B;
        const float q22 = A.m + B.m + A.i * rn2A * rn2A + B.i * rn2B * rn2B;
        const float q12 = A.m + B.m + A.i * rn1A * rn2A + B.i * rn1B * rn2B;
        const float k_maxConditionNumber = 1000.0f;
        if (q11 * q11 < k_maxConditionNumber * (q11 * q22 - q12 * q12)) {
            k11 = q11; k12 = q12; k22 = q22;
            float det = q11 * q22 - q12 * q12;              // b2Mat22::GetInverse
            if (det != 0.0f) det = 1.0f / det;
            n11 = det * q22; n12 = -det * q12; n22 = det * q11;
        } else {
            vcount = 1;                                     // the constraints are redundant: use one
        }
    }
    mci_set(x, MC_TYPE, t, mf.type | (mf.count << 2) | (vcount << 4));
    mcf(x, MC_NX, t) = normal.x; mcf(x, MC_NY, t) = normal.y;
    mcf(x, MC_RA0X, t) = rA[0].x; mcf(x, MC_RA0Y, t) = rA[0].y; mcf(x, MC_RB0X, t) = rB[0].x; mcf(x, MC_RB0Y, t) = rB[0].y;
    mcf(x, MC_RA1X, t) = rA[1].x; mcf(x, MC_RA1Y, t) = rA[1].y; mcf(x, MC_RB1X, t) = rB[1].x; mcf(x, MC_RB1Y, t) = rB[1].y;
    mcf(x, MC_NM0, t) = nmass[0]; mcf(x, MC_NM1, t) = nmass[1]; mcf(x, MC_TM0, t) = tmass[0]; mcf(x, MC_TM1, t) = tmass[1];
    mcf(x, MC_K11, t) = k11; mcf(x, MC_K12, t) = k12; mcf(x, MC_K22, t) = k22;
    mcf(x, MC_N11, t) = n11; mcf(x, MC_N12, t) = n12; mcf(x, MC_N22, t) = n22;
}

// b2Contact::Update + InitializeVelocityConstraints of candidate t (the start of a substep).  False: not touching.
KB_DI bool mc_detect(const ObjCtx &x, const Arena &p, int M, int t, const float *owsOld) {
    if (!mc_manifold(x, p, M, t, owsOld)) return false;
    mc_init_velocity(x, t);
    return true;
}

KB_DI void mc_apply(BState &A, BState &B, V2 rA, V2 rB, V2 P) {
    A.v = v_sub(A.v, v_scale(A.m, P)); A.w -= A.i * v_cross(rA, P);
    B.v = v_add(B.v, v_scale(B.m, P)); B.w += B.i * v_cross(rB, P);
}
KB_DI V2 mc_dv(const BState &A, const BState &B, V2 rA, V2 rB) {
    return v_sub(v_sub(v_add(B.v, v_cross_sv(B.w, rB)), A.v), v_cross_sv(A.w, rA));
}

// b2ContactSolver::WarmStart of record t
__device__ __noinline__ void mc_warm_start(const ObjCtx &x, int t) {
    const int a = mci(x, MC_A, t), b = mci(x, MC_B, t), vcount = (mci(x, MC_TYPE, t) >> 4) & 3;
    BState A = body_get(x, a), B = body_get(x, b);
    const V2 normal = mk2(mcf(x, MC_NX, t), mcf(x, MC_NY, t)), tangent = v_cross_vs(normal, 1.0f);
    for (int j = 0; j < vcount; ++j) {
        const V2 rA = mk2(mcf(x, j ? MC_RA1X : MC_RA0X, t), mcf(x, j ? MC_RA1Y : MC_RA0Y, t));
        const V2 rB = mk2(mcf(x, j ? MC_RB1X : MC_RB0X, t), mcf(x, j ? MC_RB1Y : MC_RB0Y, t));
        const V2 P = v_add(v_scale(mcf(x, j ? MC_NI1 : MC_NI0, t), normal), v_scale(mcf(x, j ? MC_TI1 : MC_TI0, t), tangent));
        A.w -= A.i * v_cross(rA, P); A.v = v_sub(A.v, v_scale(A.m, P));
        B.w += B.i * v_cross(rB, P); B.v = v_add(B.v, v_scale(B.m, P));
    }
    body_put_vel(x, a, A); body_put_vel(x, b, B);
}

// b2ContactSolver::SolveVelocityConstraints of record t
__device__ __noinline__ void mc_solve_velocity(const ObjCtx &x, int t) {
    const int a = mci(x, MC_A, t), b = mci(x, MC_B, t), vcount = (mci(x, MC_TYPE, t) >> 4) & 3;
    BState A = body_get(x, a), B = body_get(x, b);
    const V2 normal = mk2(mcf(x, MC_NX, t), mcf(x, MC_NY, t)), tangent = v_cross_vs(normal, 1.0f);
    const float friction = a >= WALL_CODE ? x.mu_ow : x.mu_oo;
    V2 rA[2], rB[2];
    rA[0] = mk2(mcf(x, MC_RA0X, t), mcf(x, MC_RA0Y, t)); rB[0] = mk2(mcf(x, MC_RB0X, t), mcf(x, MC_RB0Y, t));
    rA[1] = mk2(mcf(x, MC_RA1X, t), mcf(x, MC_RA1Y, t)); rB[1] = mk2(mcf(x, MC_RB1X, t), mcf(x, MC_RB1Y, t));
    float nimp[2], timp[2];
    nimp[0] = mcf(x, MC_NI0, t); nimp[1] = mcf(x, MC_NI1, t); timp[0] = mcf(x, MC_TI0, t); timp[1] = mcf(x, MC_TI1, t);
    for (int j = 0; j < vcount; ++j) {                      // friction first
        const V2 dv = mc_dv(A, B, rA[j], rB[j]);
        const float vt = v_dot(dv, tangent) - 0.0f;
        float lambda = mcf(x, j ? MC_TM1 : MC_TM0, t) * (-vt);
        const float maxFriction = friction * nimp[j];
        const float newImpulse = kb_clampf(timp[j] + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - timp[j];
        timp[j] = newImpulse;
        mc_apply(A, B, rA[j], rB[j], v_scale(lambda, tangent));
    }
    if (vcount == 1) {
        const V2 dv = mc_dv(A, B, rA[0], rB[0]);
        const float vn = v_dot(dv, normal);
        float lambda = -mcf(x, MC_NM0, t) * (vn - 0.0f);
        const float newImpulse = fmaxf(nimp[0] + lambda, 0.0f);
        lambda = newImpulse - nimp[0];
        nimp[0] = newImpulse;
        mc_apply(A, B, rA[0], rB[0], v_scale(lambda, normal));
    } else {                                                // block solver (total enumeration of the 2x2 LCP)
        const float k11 = mcf(x, MC_K11, t), k12 = mcf(x, MC_K12, t), k22 = mcf(x, MC_K22, t);
        const float n11 = mcf(x, MC_N11, t), n12 = mcf(x, MC_N12, t), n22 = mcf(x, MC_N22, t);
        const float ax = nimp[0], ay = nimp[1];
        const V2 dv1 = mc_dv(A, B, rA[0], rB[0]), dv2 = mc_dv(A, B, rA[1], rB[1]);
        float vn1 = v_dot(dv1, normal), vn2 = v_dot(dv2, normal);
        float bx = vn1 - 0.0f, by = vn2 - 0.0f;
        bx -= k11 * ax + k12 * ay; by -= k12 * ax + k22 * ay;
        float xx, xy;
        bool solved = false;
        for (;;) {
            xx = -(n11 * bx + n12 * by); xy = -(n12 * bx + n22 * by);
            if (xx >= 0.0f && xy >= 0.0f) { solved = true; break; }
            xx = -mcf(x, MC_NM0, t) * bx; xy = 0.0f; vn1 = 0.0f; vn2 = k12 * xx + by;
            if (xx >= 0.0f && vn2 >= 0.0f) { solved = true; break; }
            xx = 0.0f; xy = -mcf(x, MC_NM1, t) * by; vn1 = k12 * xy + bx; vn2 = 0.0f;
            if (xy >= 0.0f && vn1 >= 0.0f) { solved = true; break; }
            xx = 0.0f; xy = 0.0f; vn1 = bx; vn2 = by;
            if (vn1 >= 0.0f && vn2 >= 0.0f) { solved = true; break; }
            break;                                          // no solution: give up
        }
        if (solved) {
            const float dx = xx - ax, dy = xy - ay;
            const V2 P1 = v_scale(dx, normal), P2 = v_scale(dy, normal);
            A.v = v_sub(A.v, v_scale(A.m, v_add(P1, P2)));
            A.w -= A.i * (v_cross(rA[0], P1) + v_cross(rA[1], P2));
            B.v = v_add(B.v, v_scale(B.m, v_add(P1, P2)));
            B.w += B.i * (v_cross(rB[0], P1) + v_cross(rB[1], P2));
            nimp[0] = xx; nimp[1] = xy;
        }
    }
    mcf(x, MC_NI0, t) = nimp[0]; mcf(x, MC_NI1, t) = nimp[1]; mcf(x, MC_TI0, t) = timp[0]; mcf(x, MC_TI1, t) = timp[1];
    body_put_vel(x, a, A); body_put_vel(x, b, B);
}

// b2ContactSolver::SolvePositionConstraints of record t; returns its minimum separation
__device__ __noinline__ float mc_solve_position(const ObjCtx &x, int t, float baumgarte) {
    const int a = mci(x, MC_A, t), b = mci(x, MC_B, t);
    const int tc = mci(x, MC_TYPE, t), type = tc & 3, count = (tc >> 2) & 3;
    BState A = body_get(x, a), B = body_get(x, b);
    const float radA = body_radius(x, a), radB = body_radius(x, b);
    const V2 localNormal = mk2(mcf(x, MC_LNX, t), mcf(x, MC_LNY, t)), localPoint = mk2(mcf(x, MC_LPX, t), mcf(x, MC_LPY, t));
    float minSeparation = 0.0f;
    for (int j = 0; j < count; ++j) {
        const V2 lpj = mk2(mcf(x, j ? MC_P1X : MC_P0X, t), mcf(x, j ? MC_P1Y : MC_P0Y, t));
        const XF xfA = a >= WALL_CODE ? xf_identity() : xf_of_body(x.objBody + (a - x.N) * BT_WORDS, A.c.x, A.c.y, A.a);
        const XF xfB = xf_of_body(x.objBody + (b - x.N) * BT_WORDS, B.c.x, B.c.y, B.a);
        V2 normal, point;
        float separation;
        if (type == 0) {                                    // b2PositionSolverManifold
            const V2 pointA = xf_mul(xfA, localPoint), pointB = xf_mul(xfB, lpj);
            normal = v_normalize(v_sub(pointB, pointA));
            point = v_scale(0.5f, v_add(pointA, pointB));
            separation = v_dot(v_sub(pointB, pointA), normal) - radA - radB;
        } else if (type == 1) {
            normal = rot_mul(xfA, localNormal);
            const V2 planePoint = xf_mul(xfA, localPoint);
            const V2 clipPoint = xf_mul(xfB, lpj);
            separation = v_dot(v_sub(clipPoint, planePoint), normal) - radA - radB;
            point = clipPoint;
        } else {
            normal = rot_mul(xfB, localNormal);
            const V2 planePoint = xf_mul(xfB, localPoint);
            const V2 clipPoint = xf_mul(xfA, lpj);
            separation = v_dot(v_sub(clipPoint, planePoint), normal) - radA - radB;
            point = clipPoint;
            normal = v_neg(normal);
        }
        const V2 rA = v_sub(point, A.c), rB = v_sub(point, B.c);
        minSeparation = fminf(minSeparation, separation);
        const float C = kb_clampf(baumgarte * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
        const float rnA = v_cross(rA, normal), rnB = v_cross(rB, normal);
        const float K = A.m + B.m + A.i * rnA * rnA + B.i * rnB * rnB;
        const float impulse = K > 0.0f ? -C / K : 0.0f;
        const V2 P = v_scale(impulse, normal);
        A.c = v_sub(A.c, v_scale(A.m, P)); A.a -= A.i * v_cross(rA, P);
        B.c = v_add(B.c, v_scale(B.m, P)); B.a += B.i * v_cross(rB, P);
    }
    body_put_pos(x, a, A); body_put_pos(x, b, B);
    return minSeparation;
}

// ---- continuous step of the objects against the walls (b2World::SolveTOI) ------------------------------------------
KB_DI float wall_sdist(const Arena &p, int wl, V2 v) {
    switch (wl) {
    case 0: return v.x - p.xmin;
    case 1: return v.y - p.ymin;
    case 2: return p.xmax - v.x;
    default: return p.ymax - v.y;
    }
}

// b2TimeOfImpact of polygon fixture T (body table B) sweeping from (c0, a0) to (c1, a1) against wall wl.  The wall
// is proxy A, so the separation function is e_faceA on the wall: the signed distance from the wall line of the
// polygon's support vertex along -normal.  Box2D's control flow (conservative advancement, push-back over the
// vertices, bisection / secant root finder).  True and t when the state is e_touching.
__device__ __noinline__ bool toi_wall_poly(const Arena &p, int wl, const float *T, const float *B, V2 c0, float a0, V2 c1, float a1, float &tout) {
    const float total = B2_POLYGON_RADIUS + B2_POLYGON_RADIUS;
    const float target = fmaxf(B2_LINEAR_SLOP, total - 3.0f * B2_LINEAR_SLOP);
    const float tol = 0.25f * B2_LINEAR_SLOP;
    const int n = ot_n(T);
    {                                                                 // b2Sweep::Normalize
        const float twoPi = 2.0f * B2_PI;
        const float dd = twoPi * floorf(a0 / twoPi);
        a0 -= dd; a1 -= dd;
    }
    const V2 axis = v_neg(wall_normal(wl));
    auto xf_at = [&](float t) __attribute__((always_inline)) -> XF {
        return xf_of_body(B, (1.0f - t) * c0.x + t * c1.x, (1.0f - t) * c0.y + t * c1.y, (1.0f - t) * a0 + t * a1);
    };
    float t1 = 0.0f;
    for (int iter = 0; iter < 20; ++iter) {
        const XF x1 = xf_at(t1);
        float dist = 3.402823466e+38f;
        for (int i = 0; i < n; ++i) dist = fminf(dist, wall_sdist(p, wl, xf_mul(x1, ot_v(T, i))));
        if (dist <= 0.0f) return false;                               // overlapped
        if (dist < target + tol) { tout = t1; return true; }          // touching
        float t2 = 1.0f;
        for (int push = 0; push < 8; ++push) {
            const XF x2 = xf_at(t2);
            const V2 axisB = rot_mulT(x2, axis);                      // FindMinSeparation
            int idx = 0;
            float best = v_dot(ot_v(T, 0), axisB);
            for (int i = 1; i < n; ++i) { const float val = v_dot(ot_v(T, i), axisB); if (val > best) { best = val; idx = i; } }
            float s2 = wall_sdist(p, wl, xf_mul(x2, ot_v(T, idx)));
            if (s2 > target + tol) return false;                      // separated at the end of the step
            if (s2 > target - tol) { t1 = t2; break; }
            float s1 = wall_sdist(p, wl, xf_mul(x1, ot_v(T, idx)));
            if (s1 < target - tol) return false;                      // failed
            if (s1 <= target + tol) { tout = t1; return true; }
            float r1 = t1, r2 = t2;
            for (int root = 0; root < 50; ++root) {
                const float t = (root & 1) ? r1 + (target - s1) * (r2 - r1) / (s2 - s1) : 0.5f * (r1 + r2);
                const XF xt = xf_at(t);
                const float sv = wall_sdist(p, wl, xf_mul(xt, ot_v(T, idx)));
                if (fabsf(sv - target) < tol) { t2 = t; break; }
                if (sv > target) { r1 = t; s1 = sv; } else { r2 = t; s2 = sv; }
            }
        }
    }
    return false;                                                     // failed (iteration cap)
}

// b2World::SolveTOI + b2Island::SolveTOI for object m against the arena walls with the manifold constraints' contact
// model (friction, block solver, rotation), zero initial impulses.  The body's state is the one in LDS (pose after
// b2Island::Solve); (c0, a0) = its pose at the start of the substep.  The wall records of the body's fixtures are
// reused; `ows` = this env's warm-start table (b2Contact::Update at the TOI pose re-matches the stored impulses).
__device__ __noinline__ void toi_walls_object(const ObjCtx &x, const Arena &p, int F, int m, float c0x, float c0y, float a0,
                                              float h, int vel_iters, float *ows) {
    const int N = x.N, b = N + m;
    const float *B = x.objBody + m * BT_WORDS;
    const int npair = F * (F - 1) / 2;
    V2 c0 = mk2(c0x, c0y);
    {   // quick reject: a body that stays clear of every wall by more than its bounding radius has no TOI event
        float bound = 0.0f;
        for (int f = 0; f < F; ++f) { const float *T = x.objTab + f * OT_WORDS; if (ot_body(T) == m) bound = fmaxf(bound, T[OT_BOUND]); }
        bound += 4.0f * B2_POLYGON_RADIUS;
        const float xe = x.pos[b].x, ye = x.pos[b].y;
        const float m0 = fminf(fminf(c0x - p.xmin, p.xmax - c0x), fminf(c0y - p.ymin, p.ymax - c0y));
        const float m1 = fminf(fminf(xe - p.xmin, p.xmax - xe), fminf(ye - p.ymin, p.ymax - ye));
        if (m0 > bound && m1 > bound) return;
    }
    float alpha0 = 0.0f;
    for (int ev = 0; ev < B2_MAX_SUBSTEPS; ++ev) {
        const V2 c1 = mk2(x.pos[b].x, x.pos[b].y);
        const float a1 = x.objA[m];
        float minAlpha = 1.0f;
        for (int f = 0; f < F; ++f) {
            const float *T = x.objTab + f * OT_WORDS;
            if (ot_body(T) != m) continue;
            for (int wl = 0; wl < 4; ++wl) {
                float t = 0.0f, alpha = 1.0f;
                const bool hit = ot_kind(T) == KB_SHAPE_CIRCLE ? kb_toi_wall(p, wl, T[OT_RADIUS], c0.x, c0.y, c1.x, c1.y, t)
                                                               : toi_wall_poly(p, wl, T, B, c0, a0, c1, a1, t);
                if (hit) alpha = fminf(alpha0 + (1.0f - alpha0) * t, 1.0f);
                if (alpha < minAlpha) minAlpha = alpha;
            }
        }
        if (1.0f - 10.0f * B2_EPSILON < minAlpha) break;
        const float beta = (minAlpha - alpha0) / (1.0f - alpha0);      // b2Body::Advance
        c0.x += beta * (c1.x - c0.x); c0.y += beta * (c1.y - c0.y); a0 += beta * (a1 - a0);
        alpha0 = minAlpha;
        x.pos[b].x = c0.x; x.pos[b].y = c0.y; x.objA[m] = a0;
        unsigned touch = 0u;                                            // bit f * 4 + wl
        for (int f = 0; f < F; ++f) {
            if (ot_body(x.objTab + f * OT_WORDS) != m) continue;
            for (int wl = 0; wl < 4; ++wl) {
                const int t = npair + f * 4 + wl;
                float *row = ows + (f * KB_OWS_COLS + 8 + wl) * KB_OWS_WORDS;
                float o[KB_OWS_WORDS] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f, -1.0f};
                if (mc_manifold(x, p, F, t, ows)) {
                    touch |= 1u << (f * 4 + wl);
                    const int cnt = (mci(x, MC_TYPE, t) >> 2) & 3, ids = mci(x, MC_ID, t);
                    o[0] = (float)(ids & 255); o[1] = mcf(x, MC_NI0, t); o[2] = mcf(x, MC_TI0, t);
                    if (cnt == 2) { o[3] = (float)((ids >> 8) & 255); o[4] = mcf(x, MC_NI1, t); o[5] = mcf(x, MC_TI1, t); }
                    mcf(x, MC_NI0, t) = 0.0f; mcf(x, MC_NI1, t) = 0.0f; mcf(x, MC_TI0, t) = 0.0f; mcf(x, MC_TI1, t) = 0.0f;
                }
#pragma unroll
                for (int k = 0; k < KB_OWS_WORDS; ++k) row[k] = o[k];
            }
        }
        for (int it = 0; it < 20; ++it) {                              // SolveTOIPositionConstraints
            float minSep = 0.0f;
            for (unsigned m_ = touch; m_; m_ &= m_ - 1) minSep = fminf(minSep, mc_solve_position(x, npair + __builtin_ctz(m_), B2_TOI_BAUMGARTE));
            if (minSep >= -1.5f * B2_LINEAR_SLOP) break;
        }
        c0 = mk2(x.pos[b].x, x.pos[b].y); a0 = x.objA[m];             // leap of faith to the new safe state
        for (unsigned m_ = touch; m_; m_ &= m_ - 1) mc_init_velocity(x, npair + __builtin_ctz(m_));
        for (int it = 0; it < vel_iters; ++it)
            for (unsigned m_ = touch; m_; m_ &= m_ - 1) mc_solve_velocity(x, npair + __builtin_ctz(m_));
        const float hh = (1.0f - minAlpha) * h;                         // integrate the rest of the step
        float vx = x.vel[b].x, vy = x.vel[b].y, w = x.objW[m];
        const float tx = hh * vx, ty = hh * vy;
        if (tx * tx + ty * ty > B2_MAX_TRANSLATION_SQ) {
            const float ratio = B2_MAX_TRANSLATION / sqrtf(tx * tx + ty * ty);
            vx *= ratio; vy *= ratio;
        }
        const float rot = hh * w;
        if (rot * rot > B2_MAX_ROTATION_SQ) w *= B2_MAX_ROTATION / fabsf(rot);
        x.vel[b].x = vx; x.vel[b].y = vy; x.objW[m] = w;
        x.pos[b].x += hh * vx; x.pos[b].y += hh * vy; x.objA[m] += hh * w;
    }
}

// ---- kilobot - polygon object contact (class 10; Box2D: A = the polygon, B = the kilobot, one point, friction 0) --
struct PolyCon { V2 normal, rA; float nmass; };   // velocity phase: normal polygon -> kilobot, lever arm on the polygon
// manifold at the poses (op, oa) / bot centre bc -> velocity constraint data.  False: not touching.
KB_DI bool poly_contact_setup(const float *T, const XF &xo, float opx, float opy, V2 bc, float r_bot, float im_bot, PolyCon &pc) {
    V2 ln, lp;
    if (!collide_poly_circle(T, xo, bc, r_bot, ln, lp)) return false;
    const V2 normal = rot_mul(xo, ln);
    const V2 planePoint = xf_mul(xo, lp);
    const V2 cA = v_add(bc, v_scale(T[OT_RADIUS] - v_dot(v_sub(bc, planePoint), normal), normal));
    const V2 cB = v_sub(bc, v_scale(r_bot, normal));
    const V2 point = v_scale(0.5f, v_add(cA, cB));
    pc.rA = v_sub(point, mk2(opx, opy));
    pc.normal = normal;
    const float rnA = v_cross(pc.rA, normal);
    const float kNormal = T[OT_IM] + im_bot + T[OT_II] * rnA * rnA;
    pc.nmass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
    return true;
}

}  // namespace kb
