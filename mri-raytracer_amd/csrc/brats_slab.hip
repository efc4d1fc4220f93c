// K1, LDS-staged ("slab") march kernel — BASELINE config 3's kernel with the volume bricked AND staged through LDS
// (north star), for the VGA layout (three axis-flat copies of the (v, dx, dy, dz) voxels).
//
// Why.  The register-gather kernels (brats_march.hip) are bound by the vector L1's tag pipeline, and that pipeline
// charges per 4-lane quad of a gather: a wave-level dwordx4 gather is 16 quads, a quad whose four 16-B reads fall
// in one 128-B line costs one look-up, and with rays 0.65 voxels apart a quad straddles ~1.7-1.9 lines
// (profiles/r02_c3_*: 30 look-ups per gather with 2x2x2 bricks, 27 with flat bricks, floor 16).  Eight gathers per
// sample make 3.4-3.75 look-ups per sample whatever the brick shape.  What does not have that floor is a
// line-granular copy: a quad that moves 64 CONTIGUOUS bytes is one look-up.
//
// How.  A packet's samples of one march step lie on a sheet normal to the axis of the face the rays entered
// through (t = t0 + k dt with t0 on that face), and consecutive steps move the sheet by ~1.5 voxels.  With the VGA
// copy that is flat along that axis, the voxels of ONE plane that the packet can touch are a small window of whole
// lines: 3 x 5 lines = 12 x 10 voxels around the packet's central ray.  Each wave keeps a ring of R such plane
// windows in LDS (2 KiB each).  A plane is brought in by two LDS-DMA instructions (global_load_lds_dwordx4: 16
// lines, lane-linear, every quad = half a line = one look-up), once, when the first lane needs it; the eight corner
// fetches of a sample are ds_read_b128 from the ring.  Window origins are a function of the plane index alone (the
// central ray is a line in index space), so loader and readers agree without any table.  A lane whose cell is not
// covered (plane fallen out of the ring, cell outside the window, a ray that goes the other way) takes the ordinary
// global gathers for that step: coverage is a performance matter only.  Same arithmetic on the same voxels as the
// other kernels: the frame and the counters are the same bits (tests/test_gpu_parity.py).
//
// One packet (wave) per workgroup, so the ring needs no barrier: the wave's own s_waitcnt vmcnt covers its DMA.
#include "brats_device.h"

namespace mrirt {

template <int N> struct IC { static constexpr int value = N; };   // compile-time int as a lambda argument

constexpr int kSlabWU = 3, kSlabWV = 5;          // window of one plane, in lines: (4 x WU) x (2 x WV) voxels
constexpr int kSlabLines = 16;                   // padded to two 1-KiB DMA instructions
constexpr int kSlabPlaneQ = kSlabLines * 8;      // float4 per ring slot
constexpr int kSlabLoadCap = 4;                  // planes a wave brings in per step at most (stragglers fall back)

template <bool STRICT, bool SHADE, bool GAMMA1, int R>
__global__ __launch_bounds__(64) void brats_march_slab_kernel(const K1Args a) {
    __shared__ float4 ring[R * kSlabPlaneQ];
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
    const uint32_t nbU = (dimU + 3) >> 2, nbV = (dimV + 1) >> 1;
    const uint32_t dimA1 = (A == 0 ? a.grid.X : (A == 1 ? a.grid.Y : a.grid.Z)) - 1;
    const uint32_t mulA = A == 0 ? f.mul[0] : (A == 1 ? f.mul[1] : f.mul[2]);
    const uint32_t mulU = U == 0 ? f.mul[0] : f.mul[1], mulV = V == 1 ? f.mul[1] : f.mul[2];
    // window origin of plane p, in lines: the same expression for the loader and for every reader
    auto origin = [&](int p, uint32_t& lu0, uint32_t& lv0) {
        const float pf = (float)p;
        const int ou = (int)floorf(__builtin_fmaf(pf, slopeU, icptU)) - 4, ov = (int)floorf(__builtin_fmaf(pf, slopeV, icptV)) - 4;
        lu0 = (uint32_t)(ou > 0 ? ou : 0) >> 2;
        lv0 = (uint32_t)(ov > 0 ? ov : 0) >> 1;
    };
    // this lane's share of a plane load: instruction j moves lines 8j .. 8j+7, lane L the 16 B slot (L & 7) of line 8j + (L >> 3)
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ldLu[2], ldLv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const uint32_t line = 8u * j + (lane >> 3); ldLu[j] = line % kSlabWU; ldLv[j] = line / kSlabWU; }
    auto load_plane = [&](int q) {                                 // every lane takes part: EXEC is all ones here
        const int p = fwd ? q : -q;
        uint32_t lu0, lv0;
        origin(p, lu0, lv0);
        const uint32_t slot = (uint32_t)q & (uint32_t)(R - 1);
        const uint32_t pc = (uint32_t)(p < 0 ? 0 : (p > (int)dimA1 ? (int)dimA1 : p));     // (planes outside the grid are never read)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t lu = min(lu0 + ldLu[j], nbU - 1), lv = min(lv0 + ldLv[j], nbV - 1);
            const uint32_t elem = __umul24(pc, mulA) + __umul24(lu, mulU) + __umul24(lv, mulV) + (lane & 7u);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(vbase + ((size_t)elem << 4)),
                (__attribute__((address_space(3))) void*)(&ring[slot * kSlabPlaneQ + j * 64]), 16, 0, 0);
        }
    };

    // ---- the march ---------------------------------------------------------------------------------------
    float t = t0;
    int qHi = 0;                                                   // planes [qHi - R, qHi) of the march order are resident
    bool first = true;
    while (true) {
        const bool live = marches && t < t1 && r.T > a.ert;        // brats_rt.slang:117
        if (__ballot(live) == 0) break;
        Cell s;
        locate<STRICT>(a, ro, rd, t, s);                           // (dead lanes compute a harmless cell)
        const uint32_t ia = A == 0 ? s.ix : (A == 1 ? s.iy : s.iz);
        const uint32_t iu = U == 0 ? s.ix : s.iy, iv = V == 1 ? s.iy : s.iz;
        // planes ia and ia + 1 in march order: [qlo, qlo + 1]
        const int qlo = fwd ? (int)ia : -(int)ia - 1;
        if (first) {                                               // ring starts at the earliest plane any live lane needs
            int m = live ? qlo : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
            qHi = m;
            first = false;
        }
        // bring in what this step needs and the ring does not hold yet
        for (int it = 0; it < kSlabLoadCap; ++it) {
            if (__ballot(live && qlo + 1 >= qHi) == 0) break;
            load_plane(qHi);
            ++qHi;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's DMA has landed (stores of the prologue too)
        if (live) {
            // coverage: both planes resident, the cell inside both windows
            uint32_t lu0[2], lv0[2];
            origin((int)ia, lu0[0], lv0[0]);
            origin((int)ia + 1, lu0[1], lv0[1]);
            const uint32_t luA = iu >> 2, luB = (iu + 1) >> 2, lvA = iv >> 1, lvB = (iv + 1) >> 1;
            bool fast = qlo >= qHi - R && qlo + 1 < qHi;
            bool dbgRing = !fast, dbgU = false, dbgV = false;
            uint32_t rel[2][2][2];                                 // [plane][du][dv] -> float4 index inside the slot
#pragma unroll
            for (int pa = 0; pa < 2; ++pa) {
                const uint32_t ruA = luA - lu0[pa], ruB = luB - lu0[pa], rvA = lvA - lv0[pa], rvB = lvB - lv0[pa];
                fast = fast && max(ruA, ruB) < (uint32_t)kSlabWU && max(rvA, rvB) < (uint32_t)kSlabWV;
                dbgU = dbgU || !(max(ruA, ruB) < (uint32_t)kSlabWU);
                dbgV = dbgV || !(max(rvA, rvB) < (uint32_t)kSlabWV);
                const uint32_t inA = iu & 3u, inB = (iu + 1) & 3u, jnA = (iv & 1u) << 2, jnB = ((iv + 1) & 1u) << 2;
                rel[pa][0][0] = (rvA * kSlabWU + ruA) * 8 + inA + jnA;
                rel[pa][1][0] = (rvA * kSlabWU + ruB) * 8 + inB + jnA;
                rel[pa][0][1] = (rvB * kSlabWU + ruA) * 8 + inA + jnB;
                rel[pa][1][1] = (rvB * kSlabWU + ruB) * 8 + inB + jnB;
            }
            Taps<4, SHADE> taps;
            if (fast) {
                const uint32_t s0 = ((uint32_t)(fwd ? (int)ia : -(int)ia) & (uint32_t)(R - 1)) * kSlabPlaneQ;
                const uint32_t s1 = ((uint32_t)(fwd ? (int)ia + 1 : -(int)ia - 1) & (uint32_t)(R - 1)) * kSlabPlaneQ;
                // corner c = (dx, dy, dz) -> (plane, du, dv) through the axis roles; A is uniform: one scalar branch, constant
                // indices inside (a run-time index into rel[][][] would put it in scratch)
                auto gather = [&](auto axC) {
                    constexpr int AX = decltype(axC)::value, UX = AX == 0 ? 1 : 0, VX = AX == 2 ? 1 : 2;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int d[3] = { c & 1, (c >> 1) & 1, c >> 2 };
                        taps.c[c] = ring[(d[AX] ? s1 : s0) + rel[d[AX]][d[UX]][d[VX]]];
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
            if (a.debugFlags & 2u) r.nShaded += dbgRing ? 1u : 0u;   // ... + samples whose planes were not resident
            if (a.debugFlags & 4u) r.nShaded += dbgU ? 1u : 0u;      // ... + outside the window along U
            if (a.debugFlags & 8u) r.nShaded += dbgV ? 1u : 0u;      // ... + outside the window along V
            t += a.stepSize;
        }
    }
    finish(a, kind, oidx, r);
}

template <bool STRICT, bool SHADE>
static int launch_slab_t(const K1Args& a, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(64);
    constexpr int R = 8;
    if (STRICT && a.gamma == 1.0f) hipLaunchKernelGGL((brats_march_slab_kernel<STRICT, SHADE, STRICT, R>), grid, block, 0, s, a);
    else                           hipLaunchKernelGGL((brats_march_slab_kernel<STRICT, SHADE, false, R>), grid, block, 0, s, a);
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
