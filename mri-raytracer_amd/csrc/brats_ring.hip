// K1, plane-synchronous LDS ring kernel — BASELINE config 3's march with the volume bricked AND staged through LDS
// (north star), for the VGA layout (three axis-flat copies of the (v, dx, dy, dz) voxels), one modality, no overlays.
//
// Why.  The register-gather kernels (brats_march.hip) hand every tap to the texture path: eight wave-level dwordx4 gathers
// per sample, 1 KiB each through a 64 B/clk return path, 26.7 clk per gather per CU measured at config 3 with TA busy 82 %
// (profiles/r03_c3_experiments).  The 64 rays of an 8 x 8 packet are 0.5-1 voxel apart, so the 512 taps of a wave-step fall
// on ~150 distinct voxels of two or three lattice planes: what the packet needs per plane it crosses is a 16 x 16-voxel
// window, once.  LDS returns 256 B/clk for ds_read_b128 (4 x the texture path) and the window is filled by LDS-DMA in whole
// half-lines.
//
// How it differs from round 2's plane ring (brats_slab.hip: per-lane residency tests, a fall-back gather per step, a ring
// of six windows that had to follow rays which drift apart in plane coordinate) — the march here is synchronous in PLANES,
// not in step index:
//   * A = the packet's dominant axis (index space).  A sample's cell spans lattice planes ia, ia + 1 of A; the ring holds
//     R = 4 consecutive planes [tail, tail + 3] (march order).  Each round a lane takes ITS next sample if that sample's
//     two planes are resident, else it sits the round out; tail = the smallest cell plane any live lane still needs.
//     Every lane still takes its own samples in its own order with the march's own running sum t += dt, so the frame
//     and the counters are the same bits; only the lock-step changes.  Rays advance 1-1.73 planes per step along their
//     dominant axis, so three cell positions per round serve rays of any phase (entry through different faces, different
//     t0) at full rate; a lane idles only to let the others catch up (steady state: never).
//   * the window of a plane is anchored by the packet's HULL: the four corner rays of the 8 x 8 pixels bound every ray
//     of the packet on every plane, and (min, max) over four straight lines are (concave, convex) in the plane index, so
//     their chords over the grid are safe bounds that are LINEAR in the plane: one multiply-add per plane in fixed
//     point on the scalar unit.  If the hull does not fit a window anywhere along the packet's march (wide beams on far
//     planes, rays that graze the planes, an eye inside the box) the WAVE takes the gather march for its whole life —
//     a launch-time-uniform decision per wave, no per-sample residency test.
//   * LDS image of a plane: 16 x 16 voxels, toroidal (voxel (iu, iv) -> row iv & 15, column (iu + 4 iv) & 15), so a
//     reader needs no window origin at all and a moving window re-uses what it already holds; the column skew puts any
//     4 x 4 patch of voxels on 16 different bank groups (conflict-free ds_read_b128 for rays < 1 voxel apart).
//   * fills are four 1-KiB LDS-DMA pieces per plane (global_load_lds_dwordx4; a quad of lanes moves one half-line = 64
//     contiguous bytes), issued right after the round's taps have been read and before they are blended: a plane has
//     the whole blend + composite of the round (and the other waves' rounds) to land.
#include "brats_device.h"

namespace mrirt {

constexpr int kRingPlaneQ = 256;                 // float4 per plane window (16 x 16 voxels)

// smallest value over the wave (every lane takes part), DPP reduction; result uniform
__device__ __forceinline__ int wave_min_i32(int v) {
#define MRIRT_DPP(ctrl, rmask) __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false)
    v = min(v, MRIRT_DPP(0xb1, 0xf));            // quad_perm [1,0,3,2]
    v = min(v, MRIRT_DPP(0x4e, 0xf));            // quad_perm [2,3,0,1]
    v = min(v, MRIRT_DPP(0x114, 0xf));           // row_shr:4
    v = min(v, MRIRT_DPP(0x118, 0xf));           // row_shr:8
    v = min(v, MRIRT_DPP(0x142, 0xa));           // row_bcast:15
    v = min(v, MRIRT_DPP(0x143, 0xc));           // row_bcast:31
#undef MRIRT_DPP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

__device__ __forceinline__ float lane_f32(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// The gather march of a wave the ring cannot serve (the generic kernel's loop on this wave's VGA copy).
template <bool STRICT, bool SHADE, bool GAMMA1>
__device__ __forceinline__ void march_gather(const K1Args& a, const float ro[3], const float rd[3], float t0, float t1,
                                             bool marches, RayState& r) {
    WaveGrid<4> wg;
    wg.f = a.vga.ax[vga_pick_axis(a, ro, rd, marches)];
    if (!marches) return;
    float t = t0;
    const Labels none = { 0u, 0u };
    while (t < t1 && r.T > a.ert) {
        Cell s;
        locate<STRICT>(a, ro, rd, t, s);
        Taps<4, SHADE> taps;
        float sv, gm[3] = { 0.0f, 0.0f, 0.0f };
        taps.template issue<false>(wg.base(a.vol[a.chan[0]]), wg.dims(), s);
        taps.template eval<STRICT>(s, sv, gm);
        float v = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
        v = M<STRICT>::mad(sv, a.weight[a.chan[0]], v);
        if constexpr (SHADE) {
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] = M<STRICT>::mad(gm[k], a.weight[a.chan[0]], g[k]);
        }
        composite<STRICT, SHADE, GAMMA1, false>(a, rd, none, v, g, r);
        t += a.stepSize;
    }
}

// Wave-uniform description of the packet's beam on the planes of axis A, in the copy that is flat along A.
struct RingGeom {
    int loU0, loUs, loV0, loVs;      // lower hull bound of the cells on plane p, 16.16 fixed point: lo0 + p * los
    int hiU0, hiUs, hiV0, hiVs;      // upper bound likewise (what a fill has to bring: voxels floor(lo) .. floor(hi) + 1)
    int lastU, lastV;                // extra widening of plane dimA - 2 (see origin()): the steepest ray's slope, 16.16
    int nbU4, nbV8;                  // largest window origin in lines (lines along U - 4, lines along V - 8)
    int dimA1;                       // last plane
    int dimU2, dimV2;                // last cell along U / V (sampleLinear clamps positions into [0, dim - 1.001])
    uint32_t mulA, mulU, mulV;       // float4 elements per plane / per line along U / per line along V
    bool fwd;                        // planes are visited in ascending order
};

// Window origin (in lines) of plane p.  A sample between the last lattice plane and the box face (plane coordinate in
// [dimA - 1, dimA): sampleLinear clamps it into the last cell) reads planes dimA - 2 and dimA - 1 from up to two planes
// away from dimA - 2: that plane's window is widened by one more slope.
__device__ __forceinline__ void ring_origin(const RingGeom& gm, int p, int& lu0, int& lv0) {
    const int last = p == gm.dimA1 - 1 ? 1 : 0;
    lu0 = min(max((gm.loU0 + p * gm.loUs - last * gm.lastU) >> 18, 0), gm.nbU4);
    lv0 = min(max((gm.loV0 + p * gm.loVs - last * gm.lastV) >> 17, 0), gm.nbV8);
}

// The voxels a fill of plane p has to bring (inclusive ranges): floor(lo) .. floor(hi) + 1 per axis, whole quads of lanes (64
// contiguous bytes) along U.  A beam that is partly outside the grid still reads the border cells: positions are clamped by
// sampleLinear, the bounds are not.
__device__ __forceinline__ void ring_need(const RingGeom& gm, int p, int& uLo, int& uHi, int& vLo, int& vHi) {
    const int last = p == gm.dimA1 - 1 ? 1 : 0;
    uLo = min((gm.loU0 + p * gm.loUs - last * gm.lastU) >> 16, gm.dimU2) & ~3;
    uHi = max(((gm.hiU0 + p * gm.hiUs + last * gm.lastU) >> 16) + 1, 1) | 3;
    vLo = min((gm.loV0 + p * gm.loVs - last * gm.lastV) >> 16, gm.dimV2);
    vHi = max(((gm.hiV0 + p * gm.hiVs + last * gm.lastV) >> 16) + 1, 1);
}

template <int A> struct RingAxes {
    static constexpr int U = A == 0 ? 1 : 0;     // the copy's 4-voxel brick side (first non-flat axis)
    static constexpr int V = A == 2 ? 1 : 2;     // its 2-voxel side
};

template <bool STRICT, bool SHADE, bool GAMMA1, int A, int R>
__device__ __forceinline__ void march_ring(const K1Args& a, const RingGeom& gm, float4* ring, const float ro[3], const float rd[3],
                                           float t0, float t1, bool marches, RayState& r) {
    constexpr int U = RingAxes<A>::U, V = RingAxes<A>::V;
    const uint32_t lane = threadIdx.x & 63u;
    const FlatAxis& f = a.vga.ax[A];
    const char* __restrict__ vbase = static_cast<const char*>(a.vol[a.chan[0]]) + f.baseBytes;
    const char* lds = reinterpret_cast<const char*>(ring);

    // ---- the loader's side: lane L of piece j fills LDS row 4j + (L >> 4), column L & 15 of the plane's slot ----
    // Planes are filled strictly in march order, so everything that depends on the plane is a running scalar value:
    // the hull bounds (fixed point), the plane's byte offset, the ring slot.
    const uint32_t cu = ((lane & 15u) - 4u * (lane >> 4)) & 15u;      // iu mod 16 of the voxel this lane moves (column skew undone)
    int curLu0 = -1, curLv0 = -1;
    uint32_t offU = 0u, offV[4] = { 0u, 0u, 0u, 0u };                 // float4 elements inside the plane
    const int dLoU = gm.fwd ? gm.loUs : -gm.loUs, dLoV = gm.fwd ? gm.loVs : -gm.loVs, dHiV = gm.fwd ? gm.hiVs : -gm.hiVs;
    const int64_t planeBytes = (int64_t)gm.mulA << 4, dPb = gm.fwd ? planeBytes : -planeBytes;
    int qNext = 0, slotNext = 0, loUc = 0, loVc = 0, hiVc = 0;
    int64_t pbOff = 0;
    auto seek = [&](int q) {                                          // the next fill is march plane q, into slot 0
        const int p = gm.fwd ? q : -q;
        qNext = q; slotNext = 0;
        loUc = gm.loU0 + p * gm.loUs;
        loVc = gm.loV0 + p * gm.loVs; hiVc = gm.hiV0 + p * gm.hiVs;
        pbOff = (int64_t)p * planeBytes;
    };
    auto load_next = [&]() {                                          // every lane takes part: EXEC is all ones here
        const int p = gm.fwd ? qNext : -qNext;
        if ((uint32_t)p <= (uint32_t)gm.dimA1 && (a.debugFlags & 2u) == 0u) {   // (planes outside the grid are never read; kernelVariant bit 8: no fills)
            // a sample between the last lattice plane and the box face is clamped into the last cell and reads planes
            // dimA - 2, dimA - 1 from up to two planes away: plane dimA - 2 is widened by one more slope
            const int last = p == gm.dimA1 - 1 ? 1 : 0;
            const int loU = loUc - last * gm.lastU, loV = loVc - last * gm.lastV, hiV = hiVc + last * gm.lastV;
            const int lu0 = min(max(loU >> 18, 0), gm.nbU4), lv0 = min(max(loV >> 17, 0), gm.nbV8);
            if (lu0 != curLu0) {                                      // uniform: every few planes
                curLu0 = lu0;
                const uint32_t iu = 4u * (uint32_t)lu0 + ((cu - 4u * (uint32_t)lu0) & 15u);
                offU = (iu >> 2) * gm.mulU + (iu & 3u);
            }
            if (lv0 != curLv0) {
                curLv0 = lv0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t iv = 2u * (uint32_t)lv0 + ((4u * j + (lane >> 4) - 2u * (uint32_t)lv0) & 15u);
                    offV[j] = (iv >> 1) * gm.mulV + 4u * (iv & 1u);
                }
            }
            // Which of the four pieces (four window rows each) the beam can touch on this plane: rows 2 lv0 .. floor(hi) + 1
            // of the window (the beam's lower edge sits in the first line).  Piece j holds window rows (4 j - 2 lv0) & 15 and the
            // next three, modulo 16: the piece that wraps holds rows 0, 1 and is always wanted.  Whole pieces only (a
            // wave-uniform test; a 16 x 16 window is ~2x what an 8 x 8-pixel beam needs, a quarter of it goes this way).
            const int hRel = max((hiV >> 16) + 1, 1) - 2 * lv0;
            const char* pbase = vbase + pbOff;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int base = (4 * j - 2 * lv0) & 15;
                if (base == 14 || base <= hRel)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(pbase + (size_t)((offU + offV[j]) << 4)),
                        (__attribute__((address_space(3))) void*)(&ring[slotNext * kRingPlaneQ + j * 64]), 16, 0, 0);
            }
        }
        ++qNext; slotNext = slotNext + 1 == R ? 0 : slotNext + 1;
        loUc += dLoU; loVc += dLoV; hiVc += dHiV; pbOff += dPb;
    };
    // march-order index of the lower plane of a cell: planes qa, qa + 1 (ascending: ia, ia + 1; descending: ia + 1, ia)
    auto cell_q = [&](const Cell& s) {
        const uint32_t ia = A == 0 ? s.ix : (A == 1 ? s.iy : s.iz);
        return gm.fwd ? (int)ia : -(int)ia - 1;
    };

    bool alive = marches;                                             // this lane's pending sample exists: t < t1 && T > ert
    float t = t0;
    Cell s;
    locate<STRICT>(a, ro, rd, t, s);
    int qa = cell_q(s);
    int tail = wave_min_i32(alive ? qa : 0x7fffffff);                 // the ring holds march planes [tail, tail + R - 1]
    int tailSlot = 0;                                                 // ... plane tail + i in slot (tailSlot + i) mod R
    seek(tail);
#pragma nounroll
    for (int k = 0; k < R; ++k) load_next();

    const float w = a.weight[a.chan[0]];
    const Labels none = { 0u, 0u };
    while (true) {
        // this round's readers: the sample's planes qa, qa + 1 are resident
        const int dq = qa - tail;                                     // >= 0 for every lane that may still live
        const bool can = alive && dq <= R - 2 && qa + 1 < qNext;      // (qNext: a long step may outrun the two fills a round makes)
        Taps<4, SHADE> taps;
        if (a.debugFlags & 4u) {                                      // diagnostic (kernelVariant bit 9): stats[1] = shaded + lane-rounds spent idle;
            if (a.debugFlags & 1u) r.nShaded += lane == 0u ? 1u : 0u; //   with bit 7 as well: shaded + rounds (one per wave and round)
            else r.nShaded += alive && !can ? 1u : 0u;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every plane requested so far has landed
        if (can) {
            const uint32_t iu = U == 0 ? s.ix : s.iy, iv = V == 1 ? s.iy : s.iz;
            // byte offsets of the slots of march planes tail + dq (the cell's first plane) and tail + dq + 1
            uint32_t sb[R];
#pragma unroll
            for (int i = 0; i < R; ++i) { const int sl = tailSlot + i; sb[i] = (uint32_t)(sl >= R ? sl - R : sl) << 12; }
            uint32_t sFirst = sb[0], sSecond = sb[1];
#pragma unroll
            for (int i = 1; i <= R - 2; ++i) { sFirst = dq == i ? sb[i] : sFirst; sSecond = dq == i ? sb[i + 1] : sSecond; }
            const uint32_t s0 = gm.fwd ? sFirst : sSecond, s1 = gm.fwd ? sSecond : sFirst;     // planes ia, ia + 1
            const uint32_t g0 = (iv << 8) & 0xf00u, g1 = ((iv + 1u) << 8) & 0xf00u;
            const uint32_t cs = (iu + 4u * iv) << 4;                  // column (skewed), in bytes
            const uint32_t f00 = cs & 0xf0u, f10 = (cs + 16u) & 0xf0u, f01 = (cs + 64u) & 0xf0u, f11 = (cs + 80u) & 0xf0u;
            // corner c = (dx, dy, dz) -> (dA, dU, dV) through the axis roles
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int d[3] = { c & 1, (c >> 1) & 1, c >> 2 };
                const uint32_t off = (d[A] ? s1 : s0) | (d[V] ? g1 : g0) | (d[V] ? (d[U] ? f11 : f01) : (d[U] ? f10 : f00));
                taps.c[c] = *reinterpret_cast<const float4*>(lds + off);
            }
            if ((a.debugFlags & 5u) == 1u) {                          // diagnostic (kernelVariant bit 7): stats[1] = shaded + reads outside a window
                const uint32_t ia = A == 0 ? s.ix : (A == 1 ? s.iy : s.iz);
                bool bad = false;
#pragma unroll
                for (int dp = 0; dp < 2; ++dp) {
                    const int p = (int)ia + dp;
                    int lu0, lv0;
                    ring_origin(gm, p, lu0, lv0);
                    bad = bad || (int)iu < 4 * lu0 || (int)iu + 1 > 4 * lu0 + 15 || (int)iv < 2 * lv0 || (int)iv + 1 > 2 * lv0 + 15;
                    int uLo, uHi, vLo, vHi;
                    ring_need(gm, p, uLo, uHi, vLo, vHi);                 // rows beyond vHi may not have been brought (whole pieces are)
                    bad = bad || (int)iv + 1 > vHi;
                }
                r.nShaded += bad ? 1u : 0u;
            }
        }
        // the next sample of the lanes that read (the others keep theirs); whether it exists is known only after this
        // round's compositing (T), so the ring plans with the superset t < t1
        const float tn = can ? t + a.stepSize : t;
        Cell sn;
        locate<STRICT>(a, ro, rd, tn, sn);
        const int qn = cell_q(sn);
        const bool mayLive = alive && tn < t1;
        // the new tail = the smallest qn: nearly always tail, tail + 1 or tail + 2 — three ballots instead of a reduction
        const int dn = mayLive ? qn - tail : 0x7fffffff;
        int adv;
        if (__ballot(dn == 0) != 0) adv = 0;
        else if (__ballot(dn == 1) != 0) adv = 1;
        else if (__ballot(dn == 2) != 0) adv = 2;
        else adv = wave_min_i32(dn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the taps are in registers: their slots may be refilled
        if (adv != 0x7fffffff) {
            tail += adv;
            if (tail > qNext) {                                       // the rays that held the ring back have ended: start over further on
                seek(tail); tailSlot = 0;
            } else {
                tailSlot += adv < R ? adv : adv % R;
                tailSlot = tailSlot >= R ? tailSlot - R : tailSlot;
            }
            if ((a.debugFlags & 2u) != 0u) qNext = tail + R;          // (kernelVariant bit 8: no loader at all — the consumer's side alone, timing only)
            if (qNext < tail + R) load_next();                        // two fills per round at most: rays advance <= 1.73 planes per step
            if (qNext < tail + R) load_next();                        // (longer steps: rounds in which nobody reads catch up)
        }
        if (can) {
            float sv, g[3] = { 0.0f, 0.0f, 0.0f };
            taps.template eval<STRICT>(s, sv, g);
            float v = sv;
            if (w != 1.0f) {                                          // (1 x: see Stage::consume)
                v = M<STRICT>::mad(sv, w, 0.0f);
                if constexpr (SHADE) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) g[k] = M<STRICT>::mad(g[k], w, 0.0f);
                }
            }
            composite<STRICT, SHADE, GAMMA1, false>(a, rd, none, v, g, r);
            alive = tn < t1 && r.T > a.ert;                           // brats_rt.slang:117 for the next sample
        }
        t = tn; s = sn; qa = qn;
        if (__ballot(alive) == 0) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // nothing may still be writing this workgroup's LDS
}

template <bool STRICT, bool SHADE, bool GAMMA1, int R>
__global__ __launch_bounds__(64) void brats_march_ring_kernel(const K1Args a) {
    __shared__ float4 ring[R * kRingPlaneQ];
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    RayState r = { a.bg[0], a.bg[1], a.bg[2], 1.0f, 0u, 0u };
    float ro[3] = { 0.0f, 0.0f, 0.0f }, rd[3] = { 0.0f, 0.0f, 1.0f }, t0 = 0.0f, t1 = 0.0f;
    const bool marches = kind == 1 && setup_ray(a, px, py, ro, rd, t0, t1) && t0 < t1 && 1.0f > a.ert;
    const uint64_t mball = __ballot(marches);
    if (mball == 0) { finish(a, kind, oidx, r); return; }            // uniform

    // ---- wave-uniform geometry: the packet's beam -------------------------------------------------------------
    // corner pixels of the 8 x 8 packet (lanes in Morton or row-major order); their rays exist whatever `kind` says
    float hro[3], hrd[3];
    primary_ray(a.cam, px, py, hro, hrd);
    const int cl[4] = { 0, a.map.laneOrder == 0 ? 7 : 21, a.map.laneOrder == 0 ? 56 : 42, 63 };
    const int ref = (int)__builtin_ctzll(mball);
    // dominant axis in index space of a marching ray
    float best = -1.0f;
    int A = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float m = fabsf(lane_f32(rd[k], ref)) * a.vox[k].r;
        if (m > best) { best = m; A = k; }
    }
    const int U = A == 0 ? 1 : 0, V = A == 2 ? 1 : 2;
    const auto pick = [](const float v[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); };
    const auto pickU = [](const uint32_t v[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); };
    const uint32_t dims[3] = { a.grid.X, a.grid.Y, a.grid.Z };
    const float voxd[3] = { a.vox[0].d, a.vox[1].d, a.vox[2].d };
    const float vA = pick(voxd, A), vU = pick(voxd, U), vV = pick(voxd, V);
    const bool fwd = lane_f32(pick(rd, A), ref) >= 0.0f;
    RingGeom gm;
    gm.fwd = fwd;
    gm.dimA1 = (int)pickU(dims, A) - 1;
    gm.dimU2 = (int)pickU(dims, U) - 2; gm.dimV2 = (int)pickU(dims, V) - 2;
    const FlatAxis& f = a.vga.ax[A];
    gm.mulA = pickU(f.mul, A); gm.mulU = pickU(f.mul, U); gm.mulV = pickU(f.mul, V);
    gm.nbU4 = (int)((pickU(dims, U) + 3u) >> 2) - 4;
    gm.nbV8 = (int)((pickU(dims, V) + 1u) >> 1) - 8;
    bool ok = gm.nbU4 >= 0 && gm.nbV8 >= 0 && gm.dimA1 >= 1;
    // every marching ray crosses the planes in the reference ray's direction, and not at a grazing angle
    ok = ok && __ballot(marches && ((pick(rd, A) >= 0.0f) != fwd || fabsf(pick(rd, A)) < 0.25f)) == 0;
    // hull of the four corner rays on the first and last plane of the grid, index units
    const float P1 = (float)gm.dimA1;
    float loU[2] = { INFINITY, INFINITY }, hiU[2] = { -INFINITY, -INFINITY }, loV[2] = { INFINITY, INFINITY }, hiV[2] = { -INFINITY, -INFINITY };
    float maxKU = 0.0f, maxKV = 0.0f;                                 // steepest ray of the beam, voxels of U / V per plane
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float cro[3], crd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { cro[k] = lane_f32(hro[k], cl[c]); crd[k] = lane_f32(hrd[k], cl[c]); }
        const float dA = pick(crd, A);
        ok = ok && (dA >= 0.0f) == fwd && fabsf(dA) >= 0.25f;
        const float inv = 1.0f / dA;
        const float oA = pick(cro, A) - pick(a.bmin, A);
        const float kU = (pick(crd, U) * inv) * (vA / vU), kV = (pick(crd, V) * inv) * (vA / vV);
        const float u0 = ((pick(cro, U) - pick(a.bmin, U)) - (oA * inv) * pick(crd, U)) / vU;
        const float v0 = ((pick(cro, V) - pick(a.bmin, V)) - (oA * inv) * pick(crd, V)) / vV;
        const float u1 = u0 + kU * P1, v1 = v0 + kV * P1;
        maxKU = fmaxf(maxKU, fabsf(kU)); maxKV = fmaxf(maxKV, fabsf(kV));
        loU[0] = fminf(loU[0], u0); hiU[0] = fmaxf(hiU[0], u0); loU[1] = fminf(loU[1], u1); hiU[1] = fmaxf(hiU[1], u1);
        loV[0] = fminf(loV[0], v0); hiV[0] = fmaxf(hiV[0], v0); loV[1] = fminf(loV[1], v1); hiV[1] = fmaxf(hiV[1], v1);
    }
    // bounds linear in the plane index p (chords of a concave min / convex max: safe), widened to the cells that touch a
    // plane (samples between planes p - 1 and p + 1: the steepest RAY's slope either way — the chord's own slope is not
    // the rays' when their lines cross inside the grid, i.e. with the eye between the first and the last plane) and by
    // the fixed point's and fp32's slack
    const float invP1 = 1.0f / P1;
    const float sLoU = (loU[1] - loU[0]) * invP1, sHiU = (hiU[1] - hiU[0]) * invP1;
    const float sLoV = (loV[1] - loV[0]) * invP1, sHiV = (hiV[1] - hiV[0]) * invP1;
    const float eps = 0.02f + P1 * (1.0f / 65536.0f);
    const float cLoU = loU[0] - maxKU - eps, cHiU = hiU[0] + maxKU + eps;
    const float cLoV = loV[0] - maxKV - eps, cHiV = hiV[0] + maxKV + eps;
    // the planes this packet visits: cells of its rays at t0 and t1
    Cell sa, sb;
    locate<STRICT>(a, ro, rd, t0, sa);
    locate<STRICT>(a, ro, rd, t1, sb);
    const int ia0 = (int)(A == 0 ? sa.ix : (A == 1 ? sa.iy : sa.iz)), ia1 = (int)(A == 0 ? sb.ix : (A == 1 ? sb.iy : sb.iz));
    const float pLo = (float)wave_min_i32(marches ? min(ia0, ia1) : 0x7fffffff);
    const float pHi = (float)(wave_max_i32(marches ? max(ia0, ia1) : -0x7fffffff) + 1);
    // the window: 16 voxels along U from a multiple of 4, along V from a multiple of 2; voxels lo .. hi + 1 are read
    const float extU = fmaxf((cHiU + sHiU * pLo) - (cLoU + sLoU * pLo), (cHiU + sHiU * pHi) - (cLoU + sLoU * pHi));
    const float extV = fmaxf((cHiV + sHiV * pLo) - (cLoV + sLoV * pLo), (cHiV + sHiV * pHi) - (cLoV + sLoV * pHi));
    const float lastK = pHi >= P1 - 1.0f ? 1.0f : 0.0f;              // the packet reaches plane dimA - 2: its wider window
    ok = ok && extU + lastK * maxKU < 10.9f && extV + lastK * maxKV < 12.9f;
    // fixed point: |values| < 2^14 voxels
    ok = ok && fabsf(cLoU) + fabsf(sLoU) * P1 < 16000.0f && fabsf(cLoV) + fabsf(sLoV) * P1 < 16000.0f &&
         fabsf(cHiU) + fabsf(sHiU) * P1 < 16000.0f && fabsf(cHiV) + fabsf(sHiV) * P1 < 16000.0f;
    if (!ok || (a.debugFlags & 8u) != 0u) {                           // (kernelVariant bit 10: every wave takes the gather march)                           // uniform
        march_gather<STRICT, SHADE, GAMMA1>(a, ro, rd, t0, t1, marches, r);
        if (a.debugFlags & 1u) r.nShaded += r.nLive;                  // diagnostic: stats[1] = shaded + samples NOT served by the ring
        finish(a, kind, oidx, r);
        return;
    }
    gm.loU0 = (int)floorf(cLoU * 65536.0f); gm.loUs = (int)floorf(sLoU * 65536.0f);
    gm.loV0 = (int)floorf(cLoV * 65536.0f); gm.loVs = (int)floorf(sLoV * 65536.0f);
    gm.hiU0 = __builtin_amdgcn_readfirstlane((int)ceilf(cHiU * 65536.0f)); gm.hiUs = __builtin_amdgcn_readfirstlane((int)ceilf(sHiU * 65536.0f));
    gm.hiV0 = __builtin_amdgcn_readfirstlane((int)ceilf(cHiV * 65536.0f)); gm.hiVs = __builtin_amdgcn_readfirstlane((int)ceilf(sHiV * 65536.0f));
    gm.lastU = __builtin_amdgcn_readfirstlane((int)ceilf(maxKU * 65536.0f)); gm.lastV = __builtin_amdgcn_readfirstlane((int)ceilf(maxKV * 65536.0f));
    gm.loU0 = __builtin_amdgcn_readfirstlane(gm.loU0); gm.loUs = __builtin_amdgcn_readfirstlane(gm.loUs);
    gm.loV0 = __builtin_amdgcn_readfirstlane(gm.loV0); gm.loVs = __builtin_amdgcn_readfirstlane(gm.loVs);
    if (A == 0)      march_ring<STRICT, SHADE, GAMMA1, 0, R>(a, gm, ring, ro, rd, t0, t1, marches, r);
    else if (A == 1) march_ring<STRICT, SHADE, GAMMA1, 1, R>(a, gm, ring, ro, rd, t0, t1, marches, r);
    else             march_ring<STRICT, SHADE, GAMMA1, 2, R>(a, gm, ring, ro, rd, t0, t1, marches, r);
    finish(a, kind, oidx, r);
}

template <bool STRICT, bool SHADE>
static int launch_ring_t(const K1Args& a, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(64);
    // ring depth: 4 planes serve rays of any phase at full rate; 3 (kernelVariant bit 12) trade that for 13 instead of 10 waves per CU
    const bool r3 = (a.debugFlags & 32u) != 0u;
    if (STRICT && a.gamma == 1.0f) {
        if (r3) hipLaunchKernelGGL((brats_march_ring_kernel<STRICT, SHADE, STRICT, 3>), grid, block, 0, s, a);
        else    hipLaunchKernelGGL((brats_march_ring_kernel<STRICT, SHADE, STRICT, 4>), grid, block, 0, s, a);
    } else {
        if (r3) hipLaunchKernelGGL((brats_march_ring_kernel<STRICT, SHADE, false, 3>), grid, block, 0, s, a);
        else    hipLaunchKernelGGL((brats_march_ring_kernel<STRICT, SHADE, false, 4>), grid, block, 0, s, a);
    }
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

// brats_march.hip calls this for: VGA layout, one modality, no overlays, no skipping, 64-thread workgroups
int launch_ring_march(const K1Args& a, bool strict, bool shade, hipStream_t s) {
    if (a.map.blockPx != 8) return MRIRT_ERR_ARG;
    if (strict) return shade ? launch_ring_t<true, true>(a, s) : launch_ring_t<true, false>(a, s);
    return shade ? launch_ring_t<false, true>(a, s) : launch_ring_t<false, false>(a, s);
}

}  // namespace mrirt
