// K1, LDS-staged ("slab") march kernel — BASELINE config 3's kernel with the volume bricked AND staged through LDS
// (north star), for the VGA layout (three axis-flat copies of the (v, dx, dy, dz) voxels).  EXPERIMENTAL: selected
// with kernelVariant bit 6; bit-identical to the other kernels (tests/test_gpu_random_scenes.py), measured SLOWER
// than the register-gather kernels on the bench geometry (profiles/r02_c3_slab_*: 2.3-3.0 ms against 1.17 ms) —
// kept as the measured answer to "stage the bricks through LDS", see DESIGN.md section 5.
//
// Why it was tried.  The register-gather kernels (brats_march.hip) are bound by the vector L1's tag pipeline, and
// that pipeline charges per 4-lane quad of a gather: a wave-level dwordx4 gather is 16 quads, a quad whose four
// 16-B reads fall in one 128-B line costs one look-up, and with rays 0.65 voxels apart a quad straddles ~1.7-1.9
// lines (30 look-ups per gather with 2x2x2 bricks, 27 with flat bricks, floor 16).  Eight gathers per sample make
// 3.4-3.75 look-ups per sample whatever the brick shape.  What does not have that floor is a line-granular copy: a
// quad that moves 64 CONTIGUOUS bytes is one look-up.
//
// How.  A packet's samples of one march step lie on a sheet normal to the axis of the face the rays entered
// through (t = t0 + k dt with t0 on that face), and consecutive steps move the sheet by ~1.5 voxels.  With the VGA
// copy that is flat along that axis, the voxels of ONE plane that the packet can touch are a window of whole lines
// (WU x WV lines = 4 WU x 2 WV voxels) around the packet's central ray.  Each wave keeps a ring of R such plane
// windows in LDS.  A plane is brought in by LDS-DMA (global_load_lds_dwordx4: lane-linear, every quad = half a line
// = one look-up), once, one step before the first lane needs it; the eight corner fetches of a sample are
// ds_read_b128 from the ring at a separable address (slot table entry + FU(iu) + FV(iv)).  A lane whose cell is not
// covered (plane not resident, cell outside the window, a ray that entered through another face) takes the ordinary
// global gathers for that step: coverage is a performance matter only.  Same arithmetic on the same voxels as the
// other kernels: the frame and the counters are the same bits.
//
// What was measured (512^3, 1024^2, 512 steps): the look-ups do fall — 2.0e8-3.7e8 per frame against 5.5e8-6.1e8 —
// but (i) perspective spreads a packet's footprint on a far plane to ~10 voxels in the oblique direction, so a
// window that covers it is 16 x 16 voxels = 4 KiB per plane and still leaves 8-10 % of the samples to the fall-back,
// which makes almost every wave-step execute BOTH paths; (ii) 18-24 KiB of LDS per wave allow 2 waves per SIMD
// against 4, and the DMA -> wait -> ds_read chain of a step is then exposed (SQ_WAIT_ANY 44-59 % of wave cycles);
// (iii) the window/residency arithmetic adds ~50 % VALU.  One packet (wave) per workgroup, so the ring needs no
// barrier: the wave's own s_waitcnt vmcnt covers its DMA.
#include "brats_device.h"

namespace mrirt {

template <int N> struct IC { static constexpr int value = N; };   // compile-time int as a lambda argument

constexpr int kSlabLoadCap = 2;                  // planes a wave brings in per step at most (stragglers fall back)
constexpr int kSlabAhead = 3;                    // a lane may ask for planes at most this far beyond the ring's head

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// WU x WV lines per plane window ((4 WU) x (2 WV) voxels), R planes in the ring.
template <bool STRICT, bool SHADE, bool GAMMA1, int WU, int WV, int R>
__global__ __launch_bounds__(64) void brats_march_slab_kernel(const K1Args a) {
    constexpr int NDMA = (WU * WV + 7) / 8;                 // 1-KiB LDS-DMA instructions per plane
    constexpr int PLANE_Q = NDMA * 64;                       // float4 per ring slot
    constexpr uint32_t PLANE_B = PLANE_Q * 16;
    __shared__ float4 ring[R * PLANE_Q + (R + 1) / 2];       // + the slot table: {adjusted base, packed origin} per slot
    uint2* const tab = reinterpret_cast<uint2*>(ring + R * PLANE_Q);
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    RayState r = { a.bg[0], a.bg[1], a.bg[2], 1.0f, 0u, 0u };
    float ro[3] = { 0.0f, 0.0f, 0.0f }, rd[3] = { 0.0f, 0.0f, 1.0f }, t0 = 0.0f, t1 = 0.0f;
    const bool marches = kind == 1 && setup_ray(a, px, py, ro, rd, t0, t1) && t0 < t1 && 1.0f > a.ert;
    const uint64_t mball = __ballot(marches);
    if (mball == 0) { finish(a, kind, oidx, r); return; }          // uniform

    // ---- wave-uniform geometry -------------------------------------------------------------------------
    const int A = vga_pick_axis(a, ro, rd, marches);               // sheet normal = flat axis of the copy we read
    const int U = A == 0 ? 1 : 0, V = A == 2 ? 1 : 2;              // in-plane axes: 4-voxel and 2-voxel brick sides
    const FlatAxis f = a.vga.ax[A];
    const char* __restrict__ vbase = static_cast<const char*>(a.vol[a.chan[0]]) + f.baseBytes;
    const int ref = (mball >> 48) & 1u ? 48 : (int)__builtin_ctzll(mball);      // the packet's central pixel if it marches
    float cro[3], crd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { cro[k] = __shfl(ro[k], ref); crd[k] = __shfl(rd[k], ref); }
    const auto pick3 = [](const float v[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); };
    const float dA = pick3(crd, A), dU = pick3(crd, U), dV = pick3(crd, V);
    const float oA = pick3(cro, A) - pick3(a.bmin, A), oU = pick3(cro, U) - pick3(a.bmin, U), oV = pick3(cro, V) - pick3(a.bmin, V);
    const float vA = A == 0 ? a.vox[0].d : (A == 1 ? a.vox[1].d : a.vox[2].d);
    const float vU = U == 0 ? a.vox[0].d : a.vox[1].d, vV = V == 1 ? a.vox[1].d : a.vox[2].d;
    const bool fwd = dA >= 0.0f;                                   // march direction along A (planes visited in this order)
    // central ray in index space as a function of the plane index p:  c(p) = icpt + slope p
    const float invdA = 1.0f / dA;
    const float slopeU = (dU * invdA) * (vA / vU), slopeV = (dV * invdA) * (vA / vV);
    const float icptU = (oU - (oA * invdA) * dU) / vU, icptV = (oV - (oA * invdA) * dV) / vV;
    const uint32_t dimU = U == 0 ? a.grid.X : a.grid.Y, dimV = V == 1 ? a.grid.Y : a.grid.Z;
    const int nbU = (int)((dimU + 3) >> 2), nbV = (int)((dimV + 1) >> 1);       // lines along U / V (>= WU / WV: launch check)
    const int dimA1 = (int)(A == 0 ? a.grid.X : (A == 1 ? a.grid.Y : a.grid.Z)) - 1;
    const uint32_t mulA = A == 0 ? f.mul[0] : (A == 1 ? f.mul[1] : f.mul[2]);
    const uint32_t mulU = U == 0 ? f.mul[0] : f.mul[1], mulV = V == 1 ? f.mul[1] : f.mul[2];
    // this lane's share of a plane load: instruction j moves window lines 8j .. 8j+7, lane L the 16-B slot (L & 7) of
    // line 8j + (L >> 3); the window lies wholly inside the grid (origin clamped), so the address is uniform base + this
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ldOff[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const uint32_t line = min(8u * j + (lane >> 3), (uint32_t)(WU * WV - 1));
        ldOff[j] = ((line % WU) * mulU + (line / WU) * mulV + (lane & 7u)) << 4;
    }
    int qHi = 0, slotHi = 0;                                       // planes [qHi - R, qHi) of the march order are resident; slot of plane qHi
    auto load_plane = [&]() {                                      // every lane takes part: EXEC is all ones here
        const int p = fwd ? qHi : -qHi;
        const float pf = (float)p;
        int lu0 = ((int)floorf(__builtin_fmaf(pf, slopeU, icptU)) - (2 * WU - 2)) >> 2;       // centre the 4 WU voxels on the ray
        int lv0 = ((int)floorf(__builtin_fmaf(pf, slopeV, icptV)) - (WV - 1)) >> 1;
        lu0 = min(max(lu0, 0), nbU - WU);
        lv0 = min(max(lv0, 0), nbV - WV);
        const uint32_t pc = (uint32_t)min(max(p, 0), dimA1);       // (planes outside the grid are never read)
        const uint32_t base = (__umul24(pc, mulA) + __umul24((uint32_t)lu0, mulU) + __umul24((uint32_t)lv0, mulV)) << 4;
#pragma unroll
        for (int j = 0; j < NDMA; ++j)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(vbase + (size_t)(base + ldOff[j])),
                (__attribute__((address_space(3))) void*)(&ring[slotHi * PLANE_Q + j * 64]), 16, 0, 0);
        // what a reader needs to know about this slot: byte address of (line 0, slot 0) of the GRID'S line (0, 0) as if the
        // window went on to the grid origin, and the window origin for the coverage test
        if (lane == 0) tab[slotHi] = make_uint2((uint32_t)slotHi * PLANE_B - (uint32_t)lu0 * 128u - (uint32_t)lv0 * (WU * 128u),
                                                (uint32_t)lu0 | ((uint32_t)lv0 << 16));
        ++qHi;
        slotHi = slotHi + 1 == R ? 0 : slotHi + 1;
    };
    auto cell_q = [&](const Cell& s, uint32_t& iu, uint32_t& iv) {  // planes ia, ia + 1 in march order: [qlo, qlo + 1]
        const uint32_t ia = A == 0 ? s.ix : (A == 1 ? s.iy : s.iz);
        iu = U == 0 ? s.ix : s.iy; iv = V == 1 ? s.iy : s.iz;
        return fwd ? (int)ia : -(int)ia - 1;
    };

    // ---- the march: the planes of step k+1 are requested before step k is read and composited -----------------
    float t = t0;
    Cell sN;
    locate<STRICT>(a, ro, rd, t, sN);
    uint32_t iuN, ivN;
    int qloN = cell_q(sN, iuN, ivN);
    {
        int m = marches ? qloN : 0x7fffffff;                       // the ring starts at the earliest plane any ray needs
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
        qHi = m;
    }
    for (int it = 0; it < 2 * kSlabLoadCap; ++it) {
        if (__ballot(marches && qloN + 1 >= qHi && qloN + 1 < qHi + kSlabAhead) == 0) break;
        load_plane();
    }
    while (true) {
        const bool live = marches && t < t1 && r.T > a.ert;        // brats_rt.slang:117
        if (__ballot(live) == 0) break;
        const Cell s = sN;
        const uint32_t iu = iuN, iv = ivN;
        const int qlo = qloN;
        // next step's cell and planes (speculative: the ray may end at this step; the loads are harmless)
        const float tn = t + a.stepSize;
        locate<STRICT>(a, ro, rd, tn, sN);
        qloN = cell_q(sN, iuN, ivN);
        const bool liveN = live && tn < t1;
        int nNew = 0;
        const int qLanded = qHi;                                   // planes below this were requested in earlier iterations
        for (int it = 0; it < kSlabLoadCap; ++it) {
            // (a ray far ahead of the packet — it entered through another face — does not drag the ring along: it falls back)
            if (__ballot(liveN && qloN + 1 >= qHi && qloN + 1 < qHi + kSlabAhead) == 0) break;
            load_plane();
            ++nNew;
        }
        // everything requested before this iteration has to have landed; what was just requested may still fly
        if (nNew == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nNew == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
        if (live) {
            // residency (in the ring AFTER the requests above: a slot that was just re-used is gone) and slot of plane qlo
            const int back = qHi - qlo;                            // 1 .. R for planes still in the ring; plane qlo + 1: back - 1
            bool fast = back <= R && qlo + 1 < qLanded;             // not overwritten by this iteration's requests, and landed
            const bool dbgRing = !fast;
            int sl0 = slotHi - back, sl1;
            sl0 = sl0 < 0 ? sl0 + R : sl0;
            sl0 = sl0 < 0 ? 0 : sl0;                               // (not resident: any valid slot, the lane falls back)
            sl1 = sl0 + 1 == R ? 0 : sl0 + 1;
            const uint2 e0 = tab[sl0], e1 = tab[sl1];
            // coverage: lines of (iu, iu + 1) x (iv, iv + 1) inside both windows — packed 16-bit compares
            const uint32_t lo = (iu >> 2) | ((iv >> 1) << 16), hi = ((iu + 1) >> 2) | (((iv + 1) >> 1) << 16);
            constexpr uint32_t LIM = (uint32_t)(WU - 1) | ((uint32_t)(WV - 1) << 16);
            const auto pk = [](uint32_t x) { return __builtin_bit_cast(u16x2, x); };
            const auto up = [](u16x2 x) { return __builtin_bit_cast(uint32_t, x); };
            fast = fast && up(__builtin_elementwise_max(pk(lo), pk(e0.y))) == lo && up(__builtin_elementwise_min(pk(hi), pk(e0.y + LIM))) == hi
                        && up(__builtin_elementwise_max(pk(lo), pk(e1.y))) == lo && up(__builtin_elementwise_min(pk(hi), pk(e1.y + LIM))) == hi;
            Taps<4, SHADE> taps;
            if (fast) {
                // separable LDS byte address: adj(plane) + FU(iu + du) + FV(iv + dv)
                const uint32_t fu0 = (iu >> 2) * 128u + (iu & 3u) * 16u, fv0 = (iv >> 1) * (WU * 128u) + (iv & 1u) * 64u;
                const uint32_t du1 = (iu & 3u) == 3u ? 80u : 16u, dv1 = (iv & 1u) ? WU * 128u - 64u : 64u;
                // planes in ASCENDING plane order: march order is descending when the ray goes towards -A
                const uint32_t pA0 = (fwd ? e0.x : e1.x) + fu0 + fv0, pA1 = (fwd ? e1.x : e0.x) + fu0 + fv0;
                const char* lds = reinterpret_cast<const char*>(ring);
                auto at = [&](uint32_t off) { return *reinterpret_cast<const float4*>(lds + off); };
                // corner c = (dx, dy, dz) -> (plane, du, dv) through the axis roles (A uniform: one scalar branch)
                auto gather = [&](auto axC) {
                    constexpr int AX = decltype(axC)::value, UX = AX == 0 ? 1 : 0, VX = AX == 2 ? 1 : 2;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int d[3] = { c & 1, (c >> 1) & 1, c >> 2 };
                        taps.c[c] = at((d[AX] ? pA1 : pA0) + (d[UX] ? du1 : 0u) + (d[VX] ? dv1 : 0u));
                    }
                };
                if (A == 0) gather(IC<0>{}); else if (A == 1) gather(IC<1>{}); else gather(IC<2>{});
            } else {
                taps.template issue<false>(vbase, f, s);            // the ordinary gathers of this layout
            }
            float v, g[3];
            taps.template eval<STRICT>(s, v, g);
            const float w = a.weight[a.chan[0]];
            if (w != 1.0f) {                                        // (1 x: see Stage::consume)
                v = M<STRICT>::mad(v, w, 0.0f);
                if constexpr (SHADE) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) g[k] = M<STRICT>::mad(g[k], w, 0.0f);
                }
            }
            const Labels none = { 0u, 0u };
            composite<STRICT, SHADE, GAMMA1, false>(a, rd, none, v, g, r);
            if (a.debugFlags & 1u) r.nShaded += fast ? 1u : 0u;      // (diagnostic: with bit 0, stats[1] = shaded + LDS-served)
            if (a.debugFlags & 2u) r.nShaded += dbgRing ? 1u : 0u;   // ... + samples whose planes were not (yet / any more) resident
        }
        t = tn;
    }
    finish(a, kind, oidx, r);
}

template <bool STRICT, bool SHADE>
static int launch_slab_t(const K1Args& a, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(64);
#ifndef MRIRT_SLAB_WU
#define MRIRT_SLAB_WU 4
#define MRIRT_SLAB_WV 6
#define MRIRT_SLAB_R 6
#endif
    constexpr int WU = MRIRT_SLAB_WU, WV = MRIRT_SLAB_WV, R = MRIRT_SLAB_R;
    for (int k = 0; k < 3; ++k) {                                    // the plane window has to fit inside the grid
        const uint32_t d = k == 0 ? a.grid.X : (k == 1 ? a.grid.Y : a.grid.Z);
        if (d < 4u * WU || d < 2u * WV) return MRIRT_ERR_DIMS;
    }
    if (STRICT && a.gamma == 1.0f) hipLaunchKernelGGL((brats_march_slab_kernel<STRICT, SHADE, STRICT, WU, WV, R>), grid, block, 0, s, a);
    else                           hipLaunchKernelGGL((brats_march_slab_kernel<STRICT, SHADE, false, WU, WV, R>), grid, block, 0, s, a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

// brats_march.hip calls this for: VGA layout, one modality, no overlays, no skipping, 64-thread workgroups
int launch_slab_march(const K1Args& a, bool strict, bool shade, hipStream_t s) {
    if (a.map.blockPx != 8) return MRIRT_ERR_ARG;
    if (strict) return shade ? launch_slab_t<true, true>(a, s) : launch_slab_t<true, false>(a, s);
    return shade ? launch_slab_t<false, true>(a, s) : launch_slab_t<false, false>(a, s);
}

}  // namespace mrirt
