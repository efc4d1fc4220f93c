// K1: the BraTS volume ray-marcher — hand-written gfx950 HIP replacement for the Slang compute
// shader `brats_main` (reference: inr/viewer/brats_rt.slang:85-168; helpers :36-83).
//
// One lane = one ray; one wave64 = an 8x8 pixel packet (the reference's numthreads(8,8,1)
// group).  The march loop is a divergent loop: the wave's EXEC mask IS the ballot of live rays,
// and the wave leaves the loop (s_cbranch_execz) when its last lane has terminated (t >= t1 or
// T <= 0.01), so every lane stops accumulating exactly where the scalar shader does.
//
// What bounds it (profiles/r01_*): not HBM bytes.  With dword gathers it was the vector
// L1's tag pipeline (~36 line look-ups per wave-level gather); the shaded float4 kernel still saturates
// it (0.99 look-ups per clock per CU: a sample's cell straddles ~3.4 of the 128-B lines and the kernel makes
// 3.75 look-ups per sample) with VALU issue at 83 %; the unshaded QUAD kernel is VALU-bound (99.8 %) while
// drawing 4-4.6 TB/s from HBM.  How it got there:
//   * layouts where every gather brings 16 useful bytes from a 128-B 2x2x2-voxel brick
//     (VG = value + lattice gradient, QUAD = the four xy neighbours);
//   * a software-pipelined loop (brats_march_pipe_kernel): the gathers of step k+1 are issued before
//     step k is composited.  They are speculative only in that the ray may end at step k; the
//     addresses are clamped into the grid, so the extra fetch is harmless;
//   * one packet per workgroup (64 threads), so a finished packet's wave slot refills at once, and
//     16-pixel bands of packets interleaved over the 8 XCDs (every XCD gets the same mix of long and short rays);
//   * instruction diet for STRICT math: Markstein divisions, a trimmed fp64 exp with SGPR constants,
//     packed-pair trilinear blends, 32-bit cell offsets, specialisations for gamma == 1 / no overlays.
//
// Template axes: STRICT (bit-faithful to the oracle / FAST: FMA + hardware exp2, rcp), LAYOUT
// (0 linear, 1 4x4x2 fp32 bricks, 2 VG, 3 QUAD), SHADE (lattice-gradient Blinn-Phong extension); the
// pipelined kernel adds NCH (modalities), GAMMA1, LABELS, SKIP (exact empty-space skipping).
#include <type_traits>
#include "brats_device.h"

namespace mrirt {

// ---------------------------------------------------------------------------------------
// General kernel: any subset of the four modalities, gathers issued and consumed per step.
// ---------------------------------------------------------------------------------------
template <bool STRICT, int LAYOUT, bool SHADE>
__global__ __launch_bounds__(256) void brats_march_kernel(const K1Args a) {
    using Mm = M<STRICT>;
    __shared__ float4 lutShared[16];
    const float4* lutS = stage_lut(a, lutShared);
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    RayState r = { a.bg[0], a.bg[1], a.bg[2], 1.0f, 0u, 0u };
    float ro[3] = { 0.0f, 0.0f, 0.0f }, rd[3] = { 0.0f, 0.0f, 1.0f }, t0 = 0.0f, t1 = 0.0f;
    const bool marches = kind == 1 && setup_ray(a, px, py, ro, rd, t0, t1);
    WaveGrid<LAYOUT> wg;
    if constexpr (LAYOUT == 4) wg.f = a.vga.ax[vga_pick_axis(a, ro, rd, marches)];      // every lane votes: outside the branch
    else wg.g = &a.grid;
    if (marches) {
        float t = t0;
        const int64_t streamBase = a.classStream != nullptr ? a.rayOffsets[(int64_t)py * a.map.width + px] : 0;
        while (t < t1 && r.T > a.ert) {
            Cell s;
            locate<STRICT>(a, ro, rd, t, s);
            float v = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (a.enabled[m] != 0) {
                    Taps<LAYOUT, SHADE> taps;
                    float sv, gm[3];
                    taps.template issue<true>(wg.base(a.vol[m]), wg.dims(), s);
                    taps.template eval<STRICT>(s, sv, gm);
                    v = Mm::mad(sv, a.weight[m], v);
                    if constexpr (SHADE) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) g[k] = Mm::mad(gm[k], a.weight[m], g[k]);
                    }
                }
            }
            Labels lb;
            fetch_labels_stream(a, s, lb, streamBase + r.nLive); // nLive == index of this step along the ray
            composite<STRICT, SHADE>(a, rd, lb, v, g, r, lutS);
            t += a.stepSize;
        }
    }
    finish(a, kind, oidx, r);
}

// ---------------------------------------------------------------------------------------
// Pipelined kernel: NCH enabled modalities (compacted on the host into a.chan[]).  Two stages
// ping-pong so that while stage A is blended and composited, stage B's gathers (the NEXT step:
// intensities of every enabled modality plus the label fetches) are already in flight; the
// compiler then waits with vmcnt(#gathers of one stage) instead of vmcnt(0).
// ---------------------------------------------------------------------------------------
// LABELS == false: neither overlay is shown, so the stage carries no label words and pIdx dies after
// locate() — about ten VGPRs less across the two stages, which is what lets the kernel fit 4 waves/SIMD.
// SKIP: exact empty-space skipping (march_skip below).  A sample whose 8^3 macro cell is flagged (per launch:
// no enabled modality can lift the transfer function above 0 there and no shown label grid has a label there —
// skip_mask_kernel; a.skipDist holds the flags as an empty-radius map, byte != 0 = flagged) still counts as a
// march step, but composites nothing: the frame and the counters are the same bits as without skipping.
// Second level (a.leap): the map byte r of a sample's macro cell says that every macro cell within Chebyshev distance
// r - 1 is flagged too, so the ray may move 8 (r - 1) voxels along every axis and still be in flagged cells.  When every
// live ray of the packet is in a flagged cell, the packet takes the smallest of its lanes' budgets at
// once: `++nLive; t += stepSize` per step — the march's own running sum and counter, no locate / fetch — and re-primes
// the pipeline where it lands.  Every leapt sample is one the first level would have skipped: same frame, same counters.
constexpr uint32_t kSkipDistCap = 31;  // largest radius the map records: 239 voxels of room

// smallest / largest value (0..255) over the wave: eight ballots each (every lane takes part)
__device__ __forceinline__ uint32_t wave_min8(uint32_t v) {
    uint64_t cand = ~0ull;
    uint32_t m = 0;
#pragma unroll
    for (int b = 7; b >= 0; --b) {
        const uint64_t zero = __ballot(((v >> b) & 1u) == 0u) & cand;
        if (zero != 0) cand = zero; else m |= 1u << b;
    }
    return m;
}
__device__ __forceinline__ uint32_t wave_max8(uint32_t v) { return 255u - wave_min8(255u - v); }

// The packet's window on the distance map: the bytes of a 4 x 4 x 4 block of macro cells, one per lane, read back
// with ds_bpermute — a cross-lane move that does not touch vmcnt, so classifying a sample never waits on (or drains)
// the gathers in flight.  The block is moved (64 byte loads and one full wait) only when a live ray's sample leaves it,
// every ten steps or so; it is placed with the packet's extreme cell at the trailing edge of each axis.  ds_bpermute
// returns 0 from lanes that are switched off, which is why the skipping march keeps every lane of the wave in its
// loops (finished rays ride along as `!alive`) instead of letting them exit.
// MapWindow's index arithmetic (host + device: tests/native/index_harness.hip walks it under ASan / UBSan)
MRIRT_HD uint32_t window_origin(bool towardsPlus, uint32_t lo, uint32_t hi) { return towardsPlus ? lo : max(hi, 3u) - 3u; }
MRIRT_HD bool window_holds(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t ox, uint32_t oy, uint32_t oz) {
    return (cx - ox) < 4u && (cy - oy) < 4u && (cz - oz) < 4u;
}
MRIRT_HD uint32_t window_slot(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t ox, uint32_t oy, uint32_t oz) {
    return ((cx - ox) + 4u * (cy - oy) + 16u * (cz - oz)) & 63u;
}
// the macro cell lane `lane` fetches when the window moves to (ox, oy, oz): clamped into the map
MRIRT_HD uint32_t window_fetch_index(uint32_t ox, uint32_t oy, uint32_t oz, uint32_t lane, uint32_t mX, uint32_t mY, uint32_t mZ, uint32_t mXY) {
    const uint32_t gx = min(ox + (lane & 3u), mX - 1u), gy = min(oy + ((lane >> 2) & 3u), mY - 1u), gz = min(oz + (lane >> 4), mZ - 1u);
    return gx + gy * mX + gz * mXY;
}

struct MapWindow {
    uint32_t bytes;                  // lane l: the map byte of macro cell origin + (l & 3, (l >> 2) & 3, l >> 4)
    uint32_t ox, oy, oz;             // wave-uniform
    __device__ __forceinline__ void reset() { bytes = 0u; ox = oy = oz = 0x40000000u; }
    // the map byte of the sample's macro cell; `alive` = this lane's sample matters
    __device__ __forceinline__ uint32_t lookup(const K1Args& a, const Cell& s, const float rd[3], bool alive) {
        const uint32_t cx = s.ix >> 3, cy = s.iy >> 3, cz = s.iz >> 3;
        bool in = window_holds(cx, cy, cz, ox, oy, oz);
        if (__ballot(alive && !in) != 0) {                           // wave-uniform
            const uint32_t lane = threadIdx.x & 63u;
            // trailing edge per axis, by the first live lane's direction of travel (the packet's rays are near-parallel)
            const int first = __ffsll((long long)__ballot(alive)) - 1;
            const bool px = __shfl(rd[0], first) >= 0.0f, py = __shfl(rd[1], first) >= 0.0f, pz = __shfl(rd[2], first) >= 0.0f;
            // (one reduction per axis, the one its direction needs: eight ballots each — px / py / pz are wave-uniform)
            ox = px ? window_origin(true, wave_min8(alive ? cx : 255u), 0u) : window_origin(false, 0u, wave_max8(alive ? cx : 0u));
            oy = py ? window_origin(true, wave_min8(alive ? cy : 255u), 0u) : window_origin(false, 0u, wave_max8(alive ? cy : 0u));
            oz = pz ? window_origin(true, wave_min8(alive ? cz : 255u), 0u) : window_origin(false, 0u, wave_max8(alive ? cz : 0u));
            uint32_t v = a.skipDist[window_fetch_index(ox, oy, oz, lane, a.mX, a.mY, a.mZ, a.mXY)];
            asm volatile("" : "+v"(v));                             // the wait for this load belongs inside the branch
            bytes = v;
            in = window_holds(cx, cy, cz, ox, oy, oz);
        }
        const uint32_t d = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(window_slot(cx, cy, cz, ox, oy, oz) << 2), (int)bytes);
        // a ray that strays from the packet (more than 4 macro cells wide: a tiny image over a large volume) is simply
        // not skipped: 0 = fetch and composite, always correct
        return in ? d : 0u;
    }
};

// CELLS (QUAD grids): the labels come from the MRIRT_LAYOUT_LABCELL grid — ONE 8-byte gather at the cell's own offset for both
// overlays instead of two nearest-voxel gathers (the plain pipelined kernel counts its loads at compile time, hence a template axis;
// the skipping and generic kernels take the same grid through fetch_labels at run time).
template <int LAYOUT, bool SHADE, int NCH, bool LABELS, bool SKIP, bool CELLS = false>
struct Stage {
    // MOD4: ONE tap set — the VG grid's eight float4 corners carry all four modalities (which of them are enabled is a run-time
    // property of the frame: a.enabled[], one kernel whatever the viewer's check boxes say; NCH is not used)
    static constexpr bool kMod4 = LAYOUT == MRIRT_LAYOUT_MOD4;
    static_assert(!(kMod4 && SHADE), "MOD4 grids carry no gradients");
    using TapSet = typename std::conditional<kMod4, Taps<2, true>, Taps<kMod4 ? 2 : LAYOUT, SHADE>>::type;
    static constexpr int kSets = kMod4 ? 1 : NCH;
    Cell s;
    TapSet taps[kSets];
    Labels lb;
    u32x2 cw;                        // CELLS: the cell's two label words, in flight with the taps
    uint32_t csh;                    // ... and which nibble of them is this sample's
    uint32_t dist;                   // SKIP: the map byte of the sample's macro cell (0 = fetch and composite)
    bool empty;
    __device__ __forceinline__ void classify(uint32_t d) { dist = d; empty = SKIP && d != 0u; }
    // (The class stream of the whole-ray C5 form is NOT read here: its per-lane "does that sample exist" test put the label
    // loads into a divergent branch, and a load the compiler cannot count past turns every vmcnt of the loop into 0 — for the
    // plain kernels too.  mrirt_render_brats_stream takes the generic kernel.)
    // ---- the asynchronous form (plain pipelined kernels: !SKIP): gathers the compiler does not count, one explicit wait ----
    static constexpr int kTapLoads = kSets * (LAYOUT == 3 ? 2 : LAYOUT == 0 ? 4 : 8);
    static constexpr int kLoads = kTapLoads + (LABELS ? (CELLS ? 1 : 2) : 0);   // vector-memory instructions issue_async() emits
    __device__ __forceinline__ void issue_async(const K1Args& a, const WaveGrid<LAYOUT>& wg) {
        CellOffsets k;
        if constexpr (LAYOUT == 4) k = flat_cell(wg.f, s.ix, s.iy, s.iz);
        else if constexpr (LAYOUT == 0) { k.o = s.ix + s.iy * wg.g->sY + s.iz * wg.g->sZ; k.dx = 1u; k.dy = wg.g->sY; k.dz = wg.g->sZ; }
        else k = vec4_cell(*wg.g, s.ix, s.iy, s.iz);
        if constexpr (kMod4) {
            taps[0].issue_async(wg.base(a.vol[0]), k);
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) taps[c].issue_async(wg.base(a.vol[a.chan[c]]), k);
        }
        if constexpr (LABELS && CELLS) {
            static_assert(LAYOUT == 3 || kMod4, "label cells are stored in the QUAD (= VG = MOD4) grid's element order");
            async_load_words2(cw, a.labCell, k.o << 3);
            csh = label_corner_shift(a, s);
        } else if constexpr (LABELS) {
            // both label gathers are ALWAYS issued (a hidden overlay reads word 0 of the first modality and is masked in
            // arrive()): the count the wait rests on must not depend on the overlays
            const uint32_t ix = (uint32_t)roundf(clampf(s.q[0], 0.0f, a.hiLab[0]));
            const uint32_t iy = (uint32_t)roundf(clampf(s.q[1], 0.0f, a.hiLab[1]));
            const uint32_t iz = (uint32_t)roundf(clampf(s.q[2], 0.0f, a.hiLab[2]));
            const uint32_t off = a.lab.off(ix, iy, iz) << 2;               // sampleLabel, brats_rt.slang:78-83 (label grids < 4 GiB: launch())
            const void* dummy = a.vol[a.chan[0]];
            async_load_u32(lb.seg, a.showSeg != 0 ? (const void*)a.labels : dummy, a.showSeg != 0 ? off : 0u);
            async_load_u32(lb.pred, a.showPred != 0 ? (const void*)a.preds : dummy, a.showPred != 0 ? off : 0u);
        }
    }
    // every load of this stage has landed; YOUNGER = vector-memory instructions issued after them (the other stage's)
    template <int YOUNGER>
    __device__ __forceinline__ void arrive(const K1Args& a) {
        if constexpr (LAYOUT == 3) {
            // exactly this stage's registers as in/out operands (an operand listed twice would be COPIED into a second
            // register in front of the asm: a read of a gather destination that has not landed)
            f32x4 q[2 * NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) { q[2 * c] = __builtin_bit_cast(f32x4, taps[c].q0); q[2 * c + 1] = __builtin_bit_cast(f32x4, taps[c].q1); }
            if constexpr (NCH == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q[0]), "+v"(q[1]) : "n"(YOUNGER));
            if constexpr (NCH == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2 % (2 * NCH)]), "+v"(q[3 % (2 * NCH)]) : "n"(YOUNGER));
            if constexpr (NCH == 3) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2 % (2 * NCH)]), "+v"(q[3 % (2 * NCH)]),
                                                 "+v"(q[4 % (2 * NCH)]), "+v"(q[5 % (2 * NCH)]) : "n"(YOUNGER));
            if constexpr (NCH == 4) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2 % (2 * NCH)]), "+v"(q[3 % (2 * NCH)]),
                                                 "+v"(q[4 % (2 * NCH)]), "+v"(q[5 % (2 * NCH)]), "+v"(q[6 % (2 * NCH)]), "+v"(q[7 % (2 * NCH)]) : "n"(YOUNGER));
#pragma unroll
            for (int c = 0; c < NCH; ++c) { taps[c].q0 = __builtin_bit_cast(float4, q[2 * c]); taps[c].q1 = __builtin_bit_cast(float4, q[2 * c + 1]); }
        } else if constexpr (LAYOUT == 0) {
            // four register pairs per modality; an asm statement takes 30 operands, so the wait names the first two modalities'
            // pairs and a second, empty statement (volatile: it stays behind the wait) names the rest
            f32x2 q[4 * NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) q[4 * c + i] = taps[c].p[i];
            constexpr int N = 4 * NCH;
            if constexpr (NCH == 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(YOUNGER));
            else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4 % N]), "+v"(q[5 % N]), "+v"(q[6 % N]), "+v"(q[7 % N]) : "n"(YOUNGER));
            if constexpr (NCH == 3) asm volatile("" : "+v"(q[8 % N]), "+v"(q[9 % N]), "+v"(q[10 % N]), "+v"(q[11 % N]));
            if constexpr (NCH == 4) asm volatile("" : "+v"(q[8 % N]), "+v"(q[9 % N]), "+v"(q[10 % N]), "+v"(q[11 % N]), "+v"(q[12 % N]), "+v"(q[13 % N]), "+v"(q[14 % N]), "+v"(q[15 % N]));
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) taps[c].p[i] = q[4 * c + i];
        } else {
            static_assert(LAYOUT == 3 || NCH == 1 || kMod4, "VG / VGA stages hold one modality");
            taps[0].template arrive<YOUNGER>();
        }
        if constexpr (LABELS && CELLS) {
            u32x2 w = cw;
            asm volatile("" : "+v"(w));                                     // (the label words arrived with the taps: same wait)
            labels_from_cell(a, w.x, w.y, csh, lb);
        } else if constexpr (LABELS) {
            uint32_t ls = lb.seg, lp = lb.pred;
            asm volatile("" : "+v"(ls), "+v"(lp));                          // (the label words arrived with the taps: same wait)
            lb.seg = a.showSeg != 0 ? ls : 0u;
            lb.pred = a.showPred != 0 ? lp : 0u;
        }
    }
    __device__ __forceinline__ void issue(const K1Args& a, const WaveGrid<LAYOUT>& wg) {
        // SKIP: a sample that fetches nothing still ISSUES its gathers, all at cell (0,0,0) — loads inside a branch would
        // make every later s_waitcnt vmcnt conservative (the counter retires in order; a load that may or may not have
        // been issued cannot be counted past), i.e. vmcnt(0) everywhere and no pipelining at all: measured 1.8x on a
        // dense volume.  Lanes on one line cost one tag look-up per quad.
        Cell c0 = s;
        if (SKIP && empty) { c0.ix = 0u; c0.iy = 0u; c0.iz = 0u; }
#pragma unroll
        for (int c = 0; c < kSets; ++c) taps[c].template issue<false>(wg.base(a.vol[kMod4 ? 0 : a.chan[c]]), wg.dims(), c0);   // grid (copy) < 4 GiB (launch())
        if constexpr (LABELS) fetch_labels(a, s, lb);
    }
    template <bool STRICT, bool GAMMA1>
    __device__ __forceinline__ void consume(const K1Args& a, const float rd[3], RayState& r, const float4* lutS = nullptr) const {
        using Mm = M<STRICT>;
        if (SKIP && empty) {
            ++r.nLive;
            if (a.debugFlags & 1u) ++r.nShaded;          // diagnostic (kernelVariant bit 7): stats[1] = shaded + samples NOT fetched
            return;
        }
        float v = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
        if constexpr (kMod4) {
            // the shader's modality loop (brats_rt.slang:121-130) over the four components, ascending, enabled ones only
            float sv[4], rest[3];
            taps[0].template eval<STRICT>(s, sv[0], rest);
            sv[1] = rest[0]; sv[2] = rest[1]; sv[3] = rest[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) if (a.enabled[m] != 0) v = Mm::mad(sv[m], a.weight[m], v);       // uniform branches
        } else
#pragma unroll
        for (int c = 0; c < NCH; ++c) {                      // ascending modality order, as the shader
            float sv, gm[3];
            taps[c].template eval<STRICT>(s, sv, gm);
            const float w = a.weight[a.chan[c]];
            if (NCH == 1 && w == 1.0f) {
                // the viewer's weights are 1 (brats_viewer.py:130): 0 + s*1 == s up to the sign of a zero, and no
                // later step can tell -0 from +0 (val > 0, |g|, g.d are all blind to it) — a uniform branch, no math
                v = sv;
                if constexpr (SHADE) { g[0] = gm[0]; g[1] = gm[1]; g[2] = gm[2]; }
            } else {
                v = Mm::mad(sv, w, v);
                if constexpr (SHADE) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) g[k] = Mm::mad(gm[k], w, g[k]);
                }
            }
        }
        if constexpr (LABELS) {
            composite<STRICT, SHADE, GAMMA1>(a, rd, lb, v, g, r, lutS);
        } else {
            const Labels none = { 0u, 0u };
            composite<STRICT, SHADE, GAMMA1, false>(a, rd, none, v, g, r);
        }
    }
};

// The skipping march (exact empty-space skipping, both levels).  Every lane of the wave stays in the loops; a ray that
// has finished (or never marched) rides along as `!alive`: it counts as empty with unlimited room, issues its gathers
// at cell 0 and composites nothing.  The march alternates between
//   an EMPTY run  — every live ray's sample is flagged: nothing is fetched; with a.leap the packet takes the smallest of
//                   its lanes' budgets (from the map byte: 8 (r - 1) - 1 voxels of room along every axis) in one go; and
//   a DENSE run   — the two-stage pipeline of the plain kernel, gathers always issued (flagged samples at cell 0), left
//                   when the whole packet's next sample is flagged.
// Per step the arithmetic that decides anything (t, the while-condition, the compositing) is the plain kernel's.
// The class stream of C5 is not supported here (the launchers never combine the two).
template <bool STRICT, int LAYOUT, bool SHADE, int NCH, bool GAMMA1, bool LABELS>
__device__ __forceinline__ void march_skip(const K1Args& a, const WaveGrid<LAYOUT>& wg, const float ro[3], const float rd[3],
                                           float t0, float t1, bool marches, RayState& r, const float4* lutS) {
    bool alive = marches;
    if (__ballot(alive) == 0) return;
    float t = t0;
    Stage<LAYOUT, SHADE, NCH, LABELS, true> A, B;
    MapWindow win;
    win.reset();
    // steps per voxel of room along the fastest axis (index-space advance per step = rd / voxelSize * stepSize)
    const float stepsPerVoxel = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(rd[0] * a.vox[0].r), fabsf(rd[1] * a.vox[1].r)),
                                                            fabsf(rd[2] * a.vox[2].r)) * a.stepSize);
    locate<STRICT>(a, ro, rd, t, A.s);
    uint32_t dA = win.lookup(a, A.s, rd, alive);
    while (true) {
        // invariant: A.s / dA describe the sample at t; for live rays (t < t1 && T > ert) holds
        if (__ballot(alive && dA == 0u) == 0) {                      // every live ray's sample is flagged
            uint32_t n = 1u;
            if (a.leap != 0u) {
                // samples 0 (this one) .. m stay within 8 (dA - 1) - 1 voxels of it along every axis: m + 1 steps are free
                const uint32_t nl = !alive ? 255u : dA >= 2u ? min((uint32_t)((float)(8u * (dA - 1u) - 1u) * stepsPerVoxel), 254u) + 1u : 1u;
                n = wave_min8(nl);
            }
            for (uint32_t i = 0; i < n; ++i) {                       // wave-uniform trip count
                const bool go = alive && t < t1;
                r.nLive += go ? 1u : 0u;
                if (a.debugFlags & 1u) r.nShaded += go ? 1u : 0u;    // (diagnostic, as in Stage::consume)
                t = go ? t + a.stepSize : t;
            }
            alive = alive && t < t1;
            if (__ballot(alive) == 0) break;
            locate<STRICT>(a, ro, rd, t, A.s);
            dA = win.lookup(a, A.s, rd, alive);
            continue;
        }
        A.classify(alive ? dA : 1u);
        A.issue(a, wg);
        bool leave = false;
        while (true) {
            float tn = t + a.stepSize;
            locate<STRICT>(a, ro, rd, tn, B.s);
            uint32_t dB = win.lookup(a, B.s, rd, alive);
            B.classify(alive ? dB : 1u);
            leave = __ballot(alive && dB == 0u) == 0;                // the packet's next sample is flagged throughout
            B.issue(a, wg);                                          // (issued regardless: see Stage::issue)
            if (alive) A.template consume<STRICT, GAMMA1>(a, rd, r, lutS);
            t = alive ? tn : t;
            alive = alive && t < t1 && r.T > a.ert;
            if (leave || __ballot(alive) == 0) { A.s = B.s; dA = dB; break; }
            tn = t + a.stepSize;
            locate<STRICT>(a, ro, rd, tn, A.s);
            dA = win.lookup(a, A.s, rd, alive);
            A.classify(alive ? dA : 1u);
            leave = __ballot(alive && dA == 0u) == 0;
            A.issue(a, wg);
            if (alive) B.template consume<STRICT, GAMMA1>(a, rd, r, lutS);
            t = alive ? tn : t;
            alive = alive && t < t1 && r.T > a.ert;
            if (leave || __ballot(alive) == 0) break;                // A.s / dA already describe the sample at t
        }
        if (__ballot(alive) == 0) break;
    }
}

// TAG: the same code under a second symbol (kernelVariant bit 15).  bench.py's side measurements — tile shares, frames in flight —
// launch the kernel it benches at other sizes and overlapped; under the tag a profiler's per-kernel statistics keep them apart
// from the benched launches.  Instantiated for the benched configuration only (launch_pipe).
template <bool STRICT, int LAYOUT, bool SHADE, int NCH, bool GAMMA1, bool LABELS, bool SKIP, bool CELLS = false, bool TAG = false>
__global__ __launch_bounds__(256, (LABELS || SKIP) ? 3 : 4) void brats_march_pipe_kernel(const K1Args a) {
    __shared__ float4 lutShared[LABELS ? 16 : 1];
    const float4* lutS = nullptr;
    if constexpr (LABELS) lutS = stage_lut(a, lutShared);
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    RayState r = { a.bg[0], a.bg[1], a.bg[2], 1.0f, 0u, 0u };
    float ro[3] = { 0.0f, 0.0f, 0.0f }, rd[3] = { 0.0f, 0.0f, 1.0f }, t0 = 0.0f, t1 = 0.0f;
    const bool marches = kind == 1 && setup_ray(a, px, py, ro, rd, t0, t1) && t0 < t1 && 1.0f > a.ert;   // the while-condition at entry
    WaveGrid<LAYOUT> wg;
    if constexpr (LAYOUT == 4) wg.f = a.vga.ax[vga_pick_axis(a, ro, rd, marches)];      // every lane votes: outside the branch
    else wg.g = &a.grid;
    if constexpr (SKIP) {
        march_skip<STRICT, LAYOUT, SHADE, NCH, GAMMA1, LABELS>(a, wg, ro, rd, t0, t1, marches, r, lutS);
    } else if (marches) {
        // Two stages, each consumed and then re-issued TWO steps ahead:   consume A(k); issue A(k+2); consume B(k+1); issue B(k+3).
        // (Round 2 had "issue B(k+1); consume A(k)": hipcc then guarded the address arithmetic that precedes B's gathers —
        // temporaries in registers it also uses as B's load destinations — with s_waitcnt vmcnt(7) .. vmcnt(1) at the top of
        // every trip, i.e. it waited for A's gathers BEFORE issuing B's, and half of the overlap was gone: seen in the ISA of
        // every pipelined kernel.  In this order every wait sits in front of the blend that needs it and nothing else.)
        // t runs exactly as in the shader: t_{k+1} = t_k + stepSize (brats_rt.slang:164), one running fp32 sum.
        Stage<LAYOUT, SHADE, NCH, LABELS, SKIP, CELLS> A, B;
        constexpr int kN = Stage<LAYOUT, SHADE, NCH, LABELS, SKIP, CELLS>::kLoads;
        // ONE static issue site and ONE wait per stage: a second site (a prologue that primes the stages) makes the stage's
        // registers a phi at the loop header, and the copies the allocator resolves phis with would read gather destinations
        // that are still in flight (tools/check_async_loads.py found exactly that).  So the pipeline fills inside the loop:
        // the first trip skips both consumes (n counts the steps issued so far).
        float tI = t0;                                           // time of the next sample to ISSUE (the running sum)
        float tA = t0, tB = t0;                                  // time of the sample each stage holds
        uint32_t n = 0;
        while (true) {
            if (n != 0u) {
                A.template arrive<kN>(a);                        // B's kN loads may still be in flight
                A.template consume<STRICT, GAMMA1>(a, rd, r, lutS);
                if (!(tB < t1 && r.T > a.ert)) break;            // brats_rt.slang:117 for the next sample (held by B)
            }
            tA = tI;
            locate<STRICT>(a, ro, rd, tA, A.s);                  // beyond the ray's end the sample is speculative: clamped addresses
            A.classify(0u);
            A.issue_async(a, wg);
            tI += a.stepSize;
            if (n != 0u) {
                B.template arrive<kN>(a);
                B.template consume<STRICT, GAMMA1>(a, rd, r, lutS);
                if (!(tA < t1 && r.T > a.ert)) break;
            }
            tB = tI;
            locate<STRICT>(a, ro, rd, tB, B.s);
            B.classify(0u);
            B.issue_async(a, wg);
            tI += a.stepSize;
            n = 1u;
        }
        // the speculative gathers of the stage that was not consumed are still in flight: their destinations must stay
        // untouched until they land (the registers are dead to the compiler from here on)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    finish(a, kind, oidx, r);
}

// ---------------------------------------------------------------------------------------
// Rolling pipeline for the float4 voxel layouts (VG / VGA) with 2..4 modalities — the viewer's shaded
// four-modality frame.  A stage of the kernel above holds every modality's gathers of a step (8 float4 = 32 registers
// per modality, two stages), which only fits for one modality; here the unit in flight is one (step, modality) PAIR:
// while pair k is blended and accumulated, pair k+1's eight gathers are out — the next modality of the same cell, or
// modality 0 of the next step's cell (speculative, clamped addresses).  Two tap sets whatever NCH; the label
// fetches travel with modality 0.  Same arithmetic in the same order as the shader's modality loop.
// ---------------------------------------------------------------------------------------
// SKIP (no overlays): exact empty-space skipping at PACKET granularity — while every live ray of the packet sits in a flagged
// macro cell nothing is fetched and the packet leaps (the empty run of march_skip, verbatim); otherwise the rolling pipeline
// runs as usual, flagged samples included (their val <= 0 composites nothing: same bits either way).  Every lane stays in
// the loops (MapWindow's cross-lane reads need the whole wave); finished rays ride along as !alive.
template <bool STRICT, int LAYOUT, bool SHADE, int NCH, bool GAMMA1, bool LABELS, bool SKIP = false>
__global__ __launch_bounds__(256, 2) void brats_march_roll_kernel(const K1Args a) {
    static_assert(!(SKIP && LABELS), "the skipping rolling kernel draws no overlays");
    using Mm = M<STRICT>;
    __shared__ float4 lutShared[LABELS ? 16 : 1];
    const float4* lutS = nullptr;
    if constexpr (LABELS) lutS = stage_lut(a, lutShared);
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    RayState r = { a.bg[0], a.bg[1], a.bg[2], 1.0f, 0u, 0u };
    float ro[3] = { 0.0f, 0.0f, 0.0f }, rd[3] = { 0.0f, 0.0f, 1.0f }, t0 = 0.0f, t1 = 0.0f;
    const bool marches = kind == 1 && setup_ray(a, px, py, ro, rd, t0, t1) && t0 < t1 && 1.0f > a.ert;
    WaveGrid<LAYOUT> wg;
    if constexpr (LAYOUT == 4) wg.f = a.vga.ax[vga_pick_axis(a, ro, rd, marches)];
    else wg.g = &a.grid;
    if constexpr (SKIP) {
        bool alive = marches;
        if (__ballot(alive) != 0) {
            float t = t0;
            MapWindow win;
            win.reset();
            const float stepsPerVoxel = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(rd[0] * a.vox[0].r), fabsf(rd[1] * a.vox[1].r)),
                                                                    fabsf(rd[2] * a.vox[2].r)) * a.stepSize);
            Taps<LAYOUT, SHADE> tp[2];
            Cell cs[2], c0;
            locate<STRICT>(a, ro, rd, t, c0);
            uint32_t d0 = win.lookup(a, c0, rd, alive);
            while (true) {
                // invariant: c0 / d0 describe the sample at t; for live rays (t < t1 && T > ert) holds
                if (__ballot(alive && d0 == 0u) == 0) {                  // every live ray's sample is flagged: the empty run of march_skip
                    uint32_t n = 1u;
                    if (a.leap != 0u) {
                        const uint32_t nl = !alive ? 255u : d0 >= 2u ? min((uint32_t)((float)(8u * (d0 - 1u) - 1u) * stepsPerVoxel), 254u) + 1u : 1u;
                        n = wave_min8(nl);
                    }
                    for (uint32_t i = 0; i < n; ++i) {
                        const bool go = alive && t < t1;
                        r.nLive += go ? 1u : 0u;
                        if (a.debugFlags & 1u) r.nShaded += go ? 1u : 0u;      // (diagnostic: samples not fetched, as Stage::consume)
                        t = go ? t + a.stepSize : t;
                    }
                    alive = alive && t < t1;
                    if (__ballot(alive) == 0) break;
                    locate<STRICT>(a, ro, rd, t, c0);
                    d0 = win.lookup(a, c0, rd, alive);
                    continue;
                }
                // dense run: the rolling pipeline from the sample at t, left when the whole packet's next sample is flagged
                cs[0] = c0;
                tp[0].template issue<false>(wg.base(a.vol[a.chan[0]]), wg.dims(), cs[0]);
                bool out = false;
                while (!out) {
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        if (out) break;
                        float v = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
                        const float tn = t + a.stepSize;
                        uint32_t dN = 0u;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const int k = sp * NCH + c;
                            if (c + 1 < NCH) {
                                tp[(k + 1) & 1].template issue<false>(wg.base(a.vol[a.chan[c + 1]]), wg.dims(), cs[sp]);
                            } else {
                                locate<STRICT>(a, ro, rd, tn, cs[sp ^ 1]);
                                dN = win.lookup(a, cs[sp ^ 1], rd, alive);
                                tp[(k + 1) & 1].template issue<false>(wg.base(a.vol[a.chan[0]]), wg.dims(), cs[sp ^ 1]);   // (speculative: dropped on leaving)
                            }
                            float sv, gm[3];
                            tp[k & 1].template eval<STRICT>(cs[sp], sv, gm);
                            const float w = a.weight[a.chan[c]];
                            v = Mm::mad(sv, w, v);
                            if constexpr (SHADE) {
#pragma unroll
                                for (int q = 0; q < 3; ++q) g[q] = Mm::mad(gm[q], w, g[q]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (alive) { const Labels none = { 0u, 0u }; composite<STRICT, SHADE, GAMMA1, false>(a, rd, none, v, g, r); }
                        t = alive ? tn : t;
                        alive = alive && t < t1 && r.T > a.ert;
                        if (__ballot(alive && dN == 0u) == 0 || __ballot(alive) == 0) { c0 = cs[sp ^ 1]; d0 = dN; out = true; }
                    }
                }
                if (__ballot(alive) == 0) break;
            }
        }
    } else if (marches) {
        float t = t0;
        Taps<LAYOUT, SHADE> tp[2];                                  // pair k lives in tp[k & 1]
        Cell cs[2];                                                 // cell of the step a pair belongs to: step parity
        Labels lb[2];
        locate<STRICT>(a, ro, rd, t, cs[0]);
        tp[0].template issue<false>(wg.base(a.vol[a.chan[0]]), wg.dims(), cs[0]);
        if constexpr (LABELS) fetch_labels(a, cs[0], lb[0]);
        bool done = false;
        while (!done) {
            // two steps per trip: 2 NCH pairs, so the buffer parity of a pair is a compile-time value
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                if (done) break;
                float v = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
                const float tn = t + a.stepSize;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int k = sp * NCH + c;                     // pair index inside the trip
                    // request pair k + 1
                    if (c + 1 < NCH) {
                        tp[(k + 1) & 1].template issue<false>(wg.base(a.vol[a.chan[c + 1]]), wg.dims(), cs[sp]);
                    } else {
                        locate<STRICT>(a, ro, rd, tn, cs[sp ^ 1]);  // the next step's cell (the ray may end here: harmless)
                        tp[(k + 1) & 1].template issue<false>(wg.base(a.vol[a.chan[0]]), wg.dims(), cs[sp ^ 1]);
                        if constexpr (LABELS) fetch_labels(a, cs[sp ^ 1], lb[sp ^ 1]);
                    }
                    // consume pair k: brats_rt.slang:123-130, ascending modality order
                    float sv, gm[3];
                    tp[k & 1].template eval<STRICT>(cs[sp], sv, gm);
                    const float w = a.weight[a.chan[c]];
                    v = Mm::mad(sv, w, v);
                    if constexpr (SHADE) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) g[q] = Mm::mad(gm[q], w, g[q]);
                    }
                    // one pair ahead, no more: left alone the scheduler hoists every pair's gathers of the trip to its top
                    // (8 x 32 registers) and spills
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (LABELS) composite<STRICT, SHADE, GAMMA1>(a, rd, lb[sp], v, g, r, lutS);
                else { const Labels none = { 0u, 0u }; composite<STRICT, SHADE, GAMMA1, false>(a, rd, none, v, g, r); }
                t = tn;
                done = !(t < t1 && r.T > a.ert);
            }
        }
    }
    finish(a, kind, oidx, r);
}

// mrirt_brats_kernel_family: when set, the launchers record which kernel family they WOULD launch and launch nothing
thread_local int* g_family_probe = nullptr;

template <bool STRICT, int LAYOUT, bool SHADE, int NCH>
static int launch_roll(const K1Args& a, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(a.map.blockPx == 8 ? 64 : 256);
    const bool overlays = a.showSeg != 0 || a.showPred != 0;
    const bool skipKernel = a.skipDist != nullptr && !overlays && (!STRICT || a.gamma == 1.0f);      // (mrirt_render_brats_skip builds the map only then)
    if (g_family_probe != nullptr) { *g_family_probe = MRIRT_KERNEL_ROLLING | (skipKernel ? MRIRT_KERNEL_SKIPPING : 0); return MRIRT_OK; }
    if (skipKernel) {
        hipLaunchKernelGGL((brats_march_roll_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, false, true>), grid, block, 0, s, a);
    } else if (STRICT && a.gamma == 1.0f) {
        if (overlays) hipLaunchKernelGGL((brats_march_roll_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, true>), grid, block, 0, s, a);
        else          hipLaunchKernelGGL((brats_march_roll_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, false>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((brats_march_roll_kernel<STRICT, LAYOUT, SHADE, NCH, false, true>), grid, block, 0, s, a);
    }
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <bool STRICT, int LAYOUT, bool SHADE, int NCH>
static int launch_pipe(const K1Args& a, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(a.map.blockPx == 8 ? 64 : 256);
    // the fp64 pow only matters for STRICT (FAST's is two instructions): specialise gamma == 1 there
    // ... and drop the label state when no overlay is shown (STRICT only: FAST already fits)
    const bool overlays = a.showSeg != 0 || a.showPred != 0;
    // SKIP exists for the gamma == 1 STRICT kernels and for FAST; any other launch ignores the mask (still exact)
    // (LINEAR grids have no skipping kernels: mrirt_render_brats_skip never builds a map for them)
    constexpr bool kHasSkip = LAYOUT != 0;
    const bool skip = kHasSkip && a.skipDist != nullptr;
    const bool cellsKernel = (LAYOUT == 3 || LAYOUT == MRIRT_LAYOUT_MOD4) && a.labCell != nullptr && overlays && !skip;      // label cells: the plain pipelined kernel with one label gather
    const bool skipKernel = skip && (!STRICT || a.gamma == 1.0f);
    if (g_family_probe != nullptr) {
        *g_family_probe = MRIRT_KERNEL_PIPELINED | (skipKernel ? MRIRT_KERNEL_SKIPPING : 0) | (cellsKernel ? MRIRT_KERNEL_LABEL_CELLS : 0);
        return MRIRT_OK;
    }
    if constexpr (LAYOUT == 3 || LAYOUT == MRIRT_LAYOUT_MOD4) {
        if (cellsKernel) {
            if (STRICT && a.gamma == 1.0f) hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, true, false, true>), grid, block, 0, s, a);
            else                           hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, false, true, false, true>), grid, block, 0, s, a);
            MRIRT_HIP(hipGetLastError());
            return MRIRT_OK;
        }
    }
    if (STRICT && a.gamma == 1.0f && !overlays) {
        if constexpr (kHasSkip) { if (skip) { hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, !STRICT, true>), grid, block, 0, s, a); MRIRT_HIP(hipGetLastError()); return MRIRT_OK; } }
        if constexpr (STRICT && LAYOUT == 4 && SHADE && NCH == 1) {
            if (a.debugFlags & 256u) {                                  // kernelVariant bit 15: the tagged twin
                hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, !STRICT, false, false, true>), grid, block, 0, s, a);
                MRIRT_HIP(hipGetLastError());
                return MRIRT_OK;
            }
        }
        hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, !STRICT, false>), grid, block, 0, s, a);
    } else if (STRICT && a.gamma == 1.0f) {
        if constexpr (kHasSkip) { if (skip) { hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, true, true>), grid, block, 0, s, a); MRIRT_HIP(hipGetLastError()); return MRIRT_OK; } }
        hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, STRICT, true, false>), grid, block, 0, s, a);
    } else if (!STRICT && skip) {
        if constexpr (kHasSkip) hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, false, true, !STRICT>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((brats_march_pipe_kernel<STRICT, LAYOUT, SHADE, NCH, false, true, false>), grid, block, 0, s, a);
    }
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <bool STRICT, int LAYOUT, bool SHADE>
static int launch(const K1Args& a, bool pipeAsked, hipStream_t s) {
    // the pipelined kernels address label words with 32-bit byte offsets (Stage::issue_async)
    const uint64_t labelElems = (uint64_t)((a.grid.X + 3u) & ~3u) * ((a.grid.Y + 3u) & ~3u) * ((a.grid.Z + 1u) & ~1u);
    const bool pipe = pipeAsked && !((a.showSeg != 0 || a.showPred != 0) && labelElems >= (1ull << 30));
    if constexpr (LAYOUT == 2) {                      // VG: 8 float4 per modality per stage -> one modality; more: rolling pairs
        if (pipe && a.nch == 1 && !a.grid.wide) return launch_pipe<STRICT, 2, SHADE, 1>(a, s);
        if (pipe && !a.grid.wide && a.showSeg == 0 && a.showPred == 0) {   // (with overlays the generic kernel measured faster)
            switch (a.nch) {
                case 2: return launch_roll<STRICT, 2, SHADE, 2>(a, s);
                case 3: return launch_roll<STRICT, 2, SHADE, 3>(a, s);
                case 4: return launch_roll<STRICT, 2, SHADE, 4>(a, s);
                default: break;
            }
        }
    }
    if constexpr (LAYOUT == 4) {                      // VGA: as VG; every copy is < 4 GiB by construction (prepare())
        if (pipe && a.nch == 1) return launch_pipe<STRICT, 4, SHADE, 1>(a, s);
        if (pipe && a.showSeg == 0 && a.showPred == 0) {
            switch (a.nch) {
                case 2: return launch_roll<STRICT, 4, SHADE, 2>(a, s);
                case 3: return launch_roll<STRICT, 4, SHADE, 3>(a, s);
                case 4: return launch_roll<STRICT, 4, SHADE, 4>(a, s);
                default: break;
            }
        }
    }
    if constexpr (LAYOUT == 0 && !SHADE) {            // LINEAR (the plain ABI): 4 register pairs per modality per stage -> up to four
        if (pipe && a.skipDist == nullptr && (uint64_t)a.grid.X * a.grid.Y * a.grid.Z < (1ull << 30)) {    // 32-bit byte offsets
            switch (a.nch) {
                case 1: return launch_pipe<STRICT, 0, false, 1>(a, s);
                case 2: return launch_pipe<STRICT, 0, false, 2>(a, s);
                case 3: return launch_pipe<STRICT, 0, false, 3>(a, s);
                case 4: return launch_pipe<STRICT, 0, false, 4>(a, s);
                default: break;
            }
        }
    }
    if constexpr (LAYOUT == 3) {                      // QUAD: 2 float4 per modality per stage -> up to four
        if (pipe && !a.grid.wide) {
            switch (a.nch) {
                case 1: return launch_pipe<STRICT, 3, false, 1>(a, s);
                case 2: return launch_pipe<STRICT, 3, false, 2>(a, s);
                case 3: return launch_pipe<STRICT, 3, false, 3>(a, s);
                case 4: return launch_pipe<STRICT, 3, false, 4>(a, s);
                default: break;
            }
        }
    }
    if constexpr (LAYOUT == MRIRT_LAYOUT_MOD4) {      // MOD4: the pipelined kernel or nothing (grids < 4 GiB, label grids < 2^30 elements)
        (void)pipe;
        const bool labelsFit = !((a.showSeg != 0 || a.showPred != 0) && labelElems >= (1ull << 30));
        if (!a.grid.wide && labelsFit && a.classStream == nullptr) return launch_pipe<STRICT, MRIRT_LAYOUT_MOD4, false, 4>(a, s);
        return MRIRT_ERR_LAYOUT;
    } else {
        if (g_family_probe != nullptr) { *g_family_probe = MRIRT_KERNEL_GENERIC; return MRIRT_OK; }
        const dim3 grid(a.map.chunk * kXcds), block(a.map.blockPx == 8 ? 64 : 256);
        hipLaunchKernelGGL((brats_march_kernel<STRICT, LAYOUT, SHADE>), grid, block, 0, s, a);
        MRIRT_HIP(hipGetLastError());
        return MRIRT_OK;
    }
}

template <bool STRICT>
static int launch_layout(const K1Args& a, uint32_t layout, bool shade, bool pipe, hipStream_t s) {
    switch (layout) {
        case MRIRT_LAYOUT_LINEAR: return shade ? launch<STRICT, 0, true>(a, false, s) : launch<STRICT, 0, false>(a, pipe, s);
        case MRIRT_LAYOUT_BRICK:  return shade ? launch<STRICT, 1, true>(a, false, s) : launch<STRICT, 1, false>(a, false, s);
        case MRIRT_LAYOUT_VG:     return shade ? launch<STRICT, 2, true>(a, pipe, s) : launch<STRICT, 2, false>(a, pipe, s);
        case MRIRT_LAYOUT_QUAD:   return shade ? (int)MRIRT_ERR_LAYOUT : launch<STRICT, 3, false>(a, pipe, s);
        case MRIRT_LAYOUT_VGA:    return shade ? launch<STRICT, 4, true>(a, pipe, s) : launch<STRICT, 4, false>(a, pipe, s);
        case MRIRT_LAYOUT_MOD4:   return shade ? (int)MRIRT_ERR_LAYOUT : launch<STRICT, MRIRT_LAYOUT_MOD4, false>(a, pipe, s);
        default: return MRIRT_ERR_LAYOUT;
    }
}

// ---------------------------------------------------------------------------------------
// C5 (per-sample INR query): sample counting and MLP-input emission.  Build-defined extension.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_count_kernel(const K1Args a, uint32_t* __restrict__ counts) {
    uint32_t px, py;
    int64_t oidx;
    if (map_pixel(a.map, px, py, oidx) != 1) return;
    float ro[3], rd[3], t0, t1;
    uint32_t n = 0;
    if (setup_ray(a, px, py, ro, rd, t0, t1))
        for (float t = t0; t < t1; t += a.stepSize) ++n;            // the same fp32 accumulation as the march
    counts[(int64_t)py * a.map.width + px] = n;
}

struct EmitArgs {
    UDiv zsigma[4];
    float zmu[4];
    double dimM1[3];             // dim - 1 (fp64 divide: the same expression as predict_volume)
    double rdimM1[3];            // RN(1 / (dim - 1)): Markstein's exact quotient in three fp64 instructions
    const int64_t* offsets;
    float* coords;
    float4* feats;
    float* mix;                  // C5 passes: the sample's weighted intensity (float per row), or with shading
                                 // (intensity, gradient) as float4 per row; and its seg label when showSeg
    uint32_t* seg;
    const float4* geom;          // chunked C5: per pixel (rd, t1), (ro, hit) — written once by the plan kernel
};

// MLP inputs of the sample at index-space cell `s` -> row `row` of coords / feats (shared by the one-pass and
// the chunked emission)
template <bool STRICT, int LAYOUT>
__device__ __forceinline__ void emit_row(const K1Args& a, const EmitArgs& e, const Cell& s, int64_t row, float sv[4]) {
    using Mm = M<STRICT>;
    float z[4];
    if constexpr (LAYOUT == MRIRT_LAYOUT_MOD4) {
        // the four modalities are the four components of ONE float4 grid: the VG grid's eight corner gathers, and the blend
        // its shaded form applies to (v, dx, dy, dz) — each component is sampleLinear's trilinear expression, bit for bit
        Taps<2, true> taps;
        taps.template issue<true>(a.vol[0], a.grid, s);
        float g[3];
        taps.template eval<STRICT>(s, sv[0], g);
        sv[1] = g[0]; sv[2] = g[1]; sv[3] = g[2];
    } else {
        Taps<LAYOUT, false> taps[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) taps[m].template issue<true>(a.vol[m], a.grid, s);      // all gathers in flight first (layouts 0..3)
#pragma unroll
        for (int m = 0; m < 4; ++m) taps[m].template eval<STRICT>(s, sv[m], nullptr);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float v = sv[m];
        z[m] = STRICT ? Mm::divu_data(v - e.zmu[m], e.zsigma[m]) : (v - e.zmu[m]) * e.zsigma[m].r;
    }
    struct __attribute__((packed, aligned(4))) Coord3 { float x, y, z; };      // one 12-byte store (global memory takes it 4-byte aligned)
    float c3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {    // fp64, one rounding: predict_volume's coordinate at lattice points
        // x / (dim - 1) exactly (Markstein: dim - 1 is a small integer, its significand is never all ones)
        const double x = (double)clampf(s.q[k], 0.0f, a.hiLab[k]);
        const double q0 = x * e.rdimM1[k];
        const double q = __builtin_fma(__builtin_fma(-q0, e.dimM1[k], x), e.rdimM1[k], q0);
        c3[k] = (float)(q * 2.0 - 1.0);
    }
    reinterpret_cast<Coord3*>(e.coords)[row] = Coord3{ c3[0], c3[1], c3[2] };
    e.feats[row] = make_float4(z[0], z[1], z[2], z[3]);
}

template <bool STRICT, int LAYOUT>
__global__ __launch_bounds__(256) void emit_samples_kernel(const K1Args a, const EmitArgs e) {
    uint32_t px, py;
    int64_t oidx;
    if (map_pixel(a.map, px, py, oidx) != 1) return;
    float ro[3], rd[3], t0, t1;
    if (!setup_ray(a, px, py, ro, rd, t0, t1)) return;
    int64_t row = e.offsets[(int64_t)py * a.map.width + px];
    for (float t = t0; t < t1; t += a.stepSize, ++row) {
        Cell s;
        locate<STRICT>(a, ro, rd, t, s);
        float sv[4];
        emit_row<STRICT, LAYOUT>(a, e, s, row, sv);
    }
}

// ---------------------------------------------------------------------------------------
// C5, chunked and ERT-aware ("all LIVE sample points", north star): the march advances `chunk` steps per pass.
//   plan:      every ray still alive (t < t1 and T > ert after the previous pass) counts its next <= chunk steps and
//              its wave takes a row range of the pass's batch (one atomic per wave); the batch size stays in device
//              memory (no host round trip);
//   emit:      one thread per row of the batch writes that sample's MLP inputs and what the compositor needs of it
//              besides the class (weighted intensity, gradient when shading, seg label);
//   classify:  mrirt's MFMA forward over that batch (inr_mlp.hip, point count read from the device word);
//   composite: the same rays composite those steps from the per-row records, exactly as brats_main does (ERT tested
//              before every step), park their state (t, T, C) — and plan the NEXT pass (the plan kernel itself only
//              runs for the first one).
// A ray that terminates inside a pass has at most chunk - 1 samples classified in vain; a ray that is dead
// costs nothing in later passes.  The frame is the same bits as the one-pass form (the MLP is batch-position
// invariant), which tests/test_gpu_inr_render.py holds it to.
// ---------------------------------------------------------------------------------------
struct C5Ray { float t, T, C0, C1, C2; uint32_t off, cnt, pad; };      // 32 B per pixel; off = first row of the ray's WAVE

// Row numbering of one wave's samples inside a pass's batch, shared by the plan and composite kernels (which run
// the same pixel -> lane map): step k of the wave's rays occupies consecutive rows, in lane order, after the rows of
// steps 0..k-1.  `step(takes)` must be called by every lane of the wave, once per k: it returns false when no lane
// takes step k, and otherwise leaves this lane's row for that step in `row` (meaningful where takes is true).
struct C5Rows {
    uint32_t next, row;
    __device__ explicit C5Rows(uint32_t waveBase) : next(waveBase), row(0u) {}
    __device__ __forceinline__ bool step(bool takes) {
        const uint64_t m = __ballot(takes);
        if (m == 0) return false;
        row = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        next += (uint32_t)__popcll(m);
        return true;
    }
};

// The planning step of one pass, for the wave's 64 rays at once (shared by the plan kernel — first pass — and the
// composite kernel, which plans the NEXT pass for the rays it has just advanced: one launch and one round trip of the
// ray records less per pass).  `r` holds the ray's (t, T, C); `goes` = it hit the box and is above the ERT threshold.
__device__ __forceinline__ void c5_plan_rows(const K1Args& a, bool mine, int64_t pix, C5Ray r, bool goes, float t1,
                                             C5Ray* __restrict__ rays, uint2* __restrict__ rowOwner,
                                             uint32_t* __restrict__ counter, uint32_t chunk) {
    uint32_t cnt = 0;
    if (mine && goes) {
        float t = r.t;
        for (; cnt < chunk && t < t1; ++cnt) t += a.stepSize;               // the march's own running sum
    }
    // the wave's row range inside the pass's batch: one atomic per wave
    uint32_t total = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0 && total != 0) base = atomicAdd(counter, total);
    base = __shfl(base, 0);
    if (mine) {
        r.off = base;
        r.cnt = cnt;
        rays[pix] = r;
    }
    // rows in (step, lane) order: the wave's rays that take step k are neighbours in the batch, as they are in the volume
    // each row records its pixel and its t (the running sum again: the emission needs neither the ray record nor k)
    C5Rows rows(base);
    float t = r.t;
    for (uint32_t k = 0; rows.step(cnt > k); ++k, t += a.stepSize)
        if (cnt > k) rowOwner[rows.row] = make_uint2((uint32_t)pix, __float_as_uint(t));   // stores only: nothing waits on them
}

// The ray of every pixel is set up ONCE per frame, here: origin, direction, exit distance and whether it marches go to
// `geom` (two float4 per pixel), which the emission reads per row and the composite kernel per ray and pass.  Setting the ray
// up again for each of its samples was a quarter of the emission's instructions (11 IEEE divisions and two square roots
// against the sample's own three).
__global__ __launch_bounds__(256) void c5_plan_kernel(const K1Args a, C5Ray* __restrict__ rays, float4* __restrict__ geom,
                                                      uint2* __restrict__ rowOwner, uint32_t* __restrict__ counter, uint32_t chunk) {
    uint32_t px, py;
    int64_t oidx;
    const bool mine = map_pixel(a.map, px, py, oidx) == 1;
    C5Ray r = { 0.0f, 1.0f, a.bg[0], a.bg[1], a.bg[2], 0u, 0u, 0u };
    const int64_t pix = mine ? (int64_t)py * a.map.width + px : 0;
    float t1 = 0.0f;
    bool goes = false;
    if (mine) {
        float ro[3], rd[3], t0;
        const bool hit = setup_ray(a, px, py, ro, rd, t0, t1);
        goes = hit && r.T > a.ert;
        r.t = t0;
        geom[2 * pix] = make_float4(rd[0], rd[1], rd[2], t1);                     // what every reader needs
        geom[2 * pix + 1] = make_float4(ro[0], ro[1], ro[2], hit ? 1.0f : 0.0f);  // (perspective: the origin is the eye — read only under an orthographic camera)
    }
    c5_plan_rows(a, mine, pix, r, goes, t1, rays, rowOwner, counter, chunk);
}

// One thread per ROW of the pass's batch (not per ray): the emission of a pass is ~10^6 independent samples, each
// of them eight gathers and a few hundred flops, and a per-ray loop serialises a ray's 32 behind one another's
// memory latency with one wave per SIMD slot to hide it (measured: 0.18 ms per pass at 512^2, 1.6 ms of a
// 10.5 ms frame).  The rows of a wave are the same step of neighbouring rays (C5Rows), so the gathers stay as
// coherent as the march's own and the stores are contiguous; the plan recorded each row's pixel and t.
#ifndef MRIRT_EMIT_RUN
#define MRIRT_EMIT_RUN 8
#endif
constexpr uint32_t kEmitRun = MRIRT_EMIT_RUN;
template <bool STRICT, int LAYOUT, bool SHADE>
__global__ __launch_bounds__(256) void c5_emit_kernel(const K1Args a, const EmitArgs e, const uint2* __restrict__ rowOwner,
                                                      const uint32_t* __restrict__ counter) {
    using Mm = M<STRICT>;
    const uint32_t n = *counter, nGroups = (n + 255u) >> 8;
    // a workgroup takes kEmitRun consecutive 256-row groups at a time: the rows of a wave's rays follow one another step by
    // step (C5Rows), so a run is ~kEmitRun * 4 consecutive steps of the same 64 rays on one CU.  (Measured level with a plain
    // grid stride — runs of 1 / 4 / 8 groups 219 / 214 / 219 us per pass, 24: 232 — the emission is bound by the number of
    // its vector-memory instructions, not by where their lines are; kept at 8.)
    for (uint32_t g0 = blockIdx.x * kEmitRun; g0 < nGroups; g0 += gridDim.x * kEmitRun)
    for (uint32_t j = 0; j < kEmitRun && g0 + j < nGroups; ++j) {
        const uint32_t row = ((g0 + j) << 8) + threadIdx.x;
        if (row >= n) break;
        const uint2 own = rowOwner[row];
        const float4 ga = e.geom[2 * (size_t)own.x];
        float ro[3] = { a.cam.eye[0], a.cam.eye[1], a.cam.eye[2] };               // perspective: primary_ray's origin
        if (a.cam.mode != 0) { const float4 gb = e.geom[2 * (size_t)own.x + 1]; ro[0] = gb.x; ro[1] = gb.y; ro[2] = gb.z; }   // uniform
        const float rd[3] = { ga.x, ga.y, ga.z };
        const float t = __uint_as_float(own.y);
        Cell s;
        locate<STRICT>(a, ro, rd, t, s);
        float sv[4];
        emit_row<STRICT, LAYOUT>(a, e, s, (int64_t)row, sv);
        // what the composite pass needs of this sample besides its class: brats_rt.slang:121-141's weighted
        // intensity (and gradient) over the enabled modalities, in the march's own order, and the seg label
        float v = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (a.enabled[m] != 0) {
                if constexpr (SHADE) {
                    Taps<LAYOUT, true> taps;
                    float gm[3];
                    taps.template issue<true>(a.vol[m], a.grid, s);
                    taps.template eval<STRICT>(s, sv[m], gm);
#pragma unroll
                    for (int q = 0; q < 3; ++q) g[q] = Mm::mad(gm[q], a.weight[m], g[q]);
                }
                v = Mm::mad(sv[m], a.weight[m], v);
            }
        }
        if constexpr (SHADE) {
            reinterpret_cast<float4*>(e.mix)[row] = make_float4(v, g[0], g[1], g[2]);
            if (a.showSeg != 0) e.seg[row] = sample_label(a.labels, a.lab, s.q, a.hiLab);
        } else {                     // unshaded: (weighted intensity, seg label) as ONE 8-byte record in the mix array
            const uint32_t l = a.showSeg != 0 ? sample_label(a.labels, a.lab, s.q, a.hiLab) : 0u;
            reinterpret_cast<uint2*>(e.mix)[row] = make_uint2(__float_as_uint(v), l);
        }
    }
}

// The sequential part of a pass: every ray composites its <= chunk samples front to back from the per-row
// records (class from the MLP; intensity, gradient and seg label from the emission) — no volume access, and the
// next step's records are in flight while this step composites.
template <bool STRICT, bool SHADE>
__global__ __launch_bounds__(256) void c5_composite_kernel(const K1Args a, C5Ray* __restrict__ rays, const float4* __restrict__ geom,
                                                           const int16_t* __restrict__ classes,
                                                           const float* __restrict__ mix, const uint32_t* __restrict__ seg,
                                                           uint2* __restrict__ rowOwner, uint32_t* __restrict__ nextCounter,
                                                           uint32_t chunk) {
    __shared__ float4 lutShared[16];
    const float4* lutS = stage_lut(a, lutShared);
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    RayState r = { a.bg[0], a.bg[1], a.bg[2], 1.0f, 0u, 0u };
    const int64_t pix = kind == 1 ? (int64_t)py * a.map.width + px : 0;
    C5Ray st = { 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u };
    float rd[3] = { 0.0f, 0.0f, 1.0f }, t1 = 0.0f;
    bool hit = false;
    if (kind == 1) {
        st = rays[pix];
        r.C0 = st.C0; r.C1 = st.C1; r.C2 = st.C2; r.T = st.T;
        if (nextCounter != nullptr || (SHADE && st.cnt != 0)) {
            const float4 g0 = geom[2 * pix];
            rd[0] = g0.x; rd[1] = g0.y; rd[2] = g0.z; t1 = g0.w;
            hit = geom[2 * pix + 1].w != 0.0f;
        }
    }
    struct Rec { float v, g[3]; Labels lb; };
    auto fetch = [&](uint32_t row, bool takes, Rec& c) {
        c.v = 0.0f; c.g[0] = c.g[1] = c.g[2] = 0.0f; c.lb.seg = 0u; c.lb.pred = 0u;
        if (takes) {
            if constexpr (SHADE) {
                const float4 m = reinterpret_cast<const float4*>(mix)[row];
                c.v = m.x; c.g[0] = m.y; c.g[1] = m.z; c.g[2] = m.w;
                if (a.showSeg != 0) c.lb.seg = seg[row];
            } else {
                const uint2 m = reinterpret_cast<const uint2*>(mix)[row];
                c.v = __uint_as_float(m.x);
                c.lb.seg = m.y;                  // (0 when the overlay is off: the emission wrote it so)
            }
            c.lb.pred = (uint32_t)(uint16_t)classes[row];
        }
    };
    float t = st.t;
    bool alive = true;
    C5Rows rows(st.off);
    Rec cur, nxt;
    bool more = rows.step(st.cnt > 0u);
    fetch(rows.row, st.cnt > 0u, cur);
    for (uint32_t k = 0; more; ++k) {
        more = rows.step(st.cnt > k + 1u);
        fetch(rows.row, more && st.cnt > k + 1u, nxt);
        // brats_rt.slang:117: `while (t < t1 && T > 0.01)` — t < t1 holds for these cnt steps by construction
        alive = alive && st.cnt > k && r.T > a.ert;
        if (__ballot(alive) == 0) break;                 // every ray of the wave is done with this pass
        if (alive) {
            composite<STRICT, SHADE>(a, rd, cur.lb, cur.v, cur.g, r, lutS);
            t += a.stepSize;
        }
        cur = nxt;
    }
    st.t = t; st.T = r.T; st.C0 = r.C0; st.C1 = r.C1; st.C2 = r.C2;
    // the next pass's plan for these rays (nullptr after the last pass: nothing is parked)
    if (nextCounter != nullptr) c5_plan_rows(a, kind == 1, pix, st, hit && r.T > a.ert, t1, rays, rowOwner, nextCounter, chunk);
    finish(a, kind, oidx, r);      // the frame so far (complete after the last pass); live-sample counters
}

struct Prepared { uint32_t layout, math; bool shade, pipe, slab, ring; };

// validate + fill the kernel arguments shared by every K1 entry point
static int prepare(const MrirtBratsParams* p, const MrirtRenderExt* ext, const void* const vol[4],
                   const void* labels, const void* preds, bool needVolumes, int64_t pitch_px,
                   K1Args& a, Prepared& cfg) {
    if (!p || (needVolumes && !vol)) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (p->dims[k] < 2) return MRIRT_ERR_DIMS;
    const uint32_t layout = ext ? ext->layout : (uint32_t)MRIRT_LAYOUT_LINEAR;
    const uint32_t labLayout = ext ? ext->labelLayout : (uint32_t)MRIRT_LAYOUT_LINEAR;
    const uint32_t math = ext ? ext->math : (uint32_t)MRIRT_MATH_STRICT;
    const uint32_t fmt = ext ? ext->outFormat : (uint32_t)MRIRT_OUT_RGBA32F;
    const uint32_t variant = ext ? ext->kernelVariant : 0u;
    const bool labCells = labLayout == MRIRT_LAYOUT_LABCELL;              // both overlays' corner labels per cell: QUAD grids only
    const bool mod4 = layout == MRIRT_LAYOUT_MOD4;                       // all four modalities in one float4 grid (vol[0])
    if ((layout > MRIRT_LAYOUT_VGA && !mod4) || (labLayout > MRIRT_LAYOUT_BRICK && !labCells) || math > MRIRT_MATH_FAST || fmt > MRIRT_OUT_RGBA16F)
        return MRIRT_ERR_LAYOUT;
    if (labCells && layout != MRIRT_LAYOUT_QUAD && !mod4) return MRIRT_ERR_LAYOUT;
    if (labCells && mrirt_vec4_elems(p->dims) >= (int64_t)1 << 29) return MRIRT_ERR_DIMS;      // 32-bit byte offsets of 8-byte elements
    if (layout == MRIRT_LAYOUT_VGA)
        for (int c = 0; c < 3; ++c) if (vga_copy_elems(p->dims, c) >= (1ull << 28)) return MRIRT_ERR_DIMS;   // 32-bit byte offsets per copy
    if (mrirt_brick_elems(p->dims) >= (int64_t)1 << 32 || mrirt_vec4_elems(p->dims) >= (int64_t)1 << 32)
        return MRIRT_ERR_DIMS;                                           // 32-bit element offsets
    if (needVolumes) {
        for (int m = 0; m < (mod4 ? 1 : 4); ++m) if ((mod4 || p->volEnabled[m] != 0) && !vol[m]) return MRIRT_ERR_NULL;
        if ((p->showSeg != 0 || (labCells && p->showPred != 0)) && !labels) return MRIRT_ERR_NULL;
    }
    {
        // The march is `while (t < t1 && T > ert) { ...; t += stepSize; }` (brats_rt.slang:117-165) and the C5
        // count / emit loops have no transmittance exit at all: a step that is not a positive finite number, or
        // too small to move t in fp32 at the far end of the box, would spin a wave forever.  The reference UI
        // clamps its slider to >= 0.001 (brats_viewer.py:168); an API caller gets an error code instead of a
        // hung GPU.  Also bounded: the step count of the box diagonal (kMaxStepsPerRay), so that a frame is a
        // finite amount of work.
        const float h = p->stepSize;
        if (!(h > 0.0f) || !isfinite(h)) return MRIRT_ERR_ARG;
        double diag2 = 0.0, dist2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            if (!(p->voxelSize[k] > 0.0f) || !isfinite(p->voxelSize[k]) || !isfinite(p->volMin[k]) || !isfinite(p->eye[k]))
                return MRIRT_ERR_ARG;
            const double ext = (double)p->voxelSize[k] * (double)p->dims[k];
            const double c = (double)p->volMin[k] + 0.5 * ext - (double)p->eye[k];
            diag2 += ext * ext; dist2 += c * c;
        }
        double tMax = sqrt(dist2) + sqrt(diag2);                     // no sample lies farther along any ray
        if (ext && ext->cameraMode == 1u)                            // orthographic origins are offset from the eye
            tMax += fabs((double)ext->orthoHalfHeight) * (1.0 + (double)p->imageSize[0] / fmax(1.0, (double)p->imageSize[1]));
        if (p->farT > 0.0f && isfinite(p->farT)) tMax = fmin(tMax, (double)p->farT);
        const float tFar = (float)tMax;
        if (!isfinite(tFar) || !(tFar + h > tFar)) return MRIRT_ERR_ARG;          // t += stepSize must advance
        if (sqrt(diag2) / (double)h > (double)kMaxStepsPerRay) return MRIRT_ERR_ARG;
    }
    fill_camera(a.cam, p->eye, p->U, p->V, p->W, p->fovY, p->imageSize[0], p->imageSize[1], ext, false);
    // Workgroup = one 8 x 8 packet (64 threads: a finished packet's wave slot refills at once) — except on VGA grids, where
    // a 16 x 16 block of four packets measured 3.7 % faster (1.145 -> 1.10 ms at C3: the four waves' lines meet in one L1).
    // Variant bit 1 flips the choice; the LDS-staged kernel (bit 6) is written for one packet per workgroup.
    // ... while the launch has enough of them to go round: an eighth of the config-4 frame (one GPU of eight: 2048 workgroups of
    // 16 x 16 for 1024 resident ones) ends in a long tail of half-idle CUs — 0.616 ms against 0.500 ms with one-packet
    // workgroups (tools/tile_share_bench.py, profiles/r03_tile_share.txt); from 4096 upwards the big ones win (0.904 vs 0.918).
    uint64_t blocks16 = (uint64_t)((p->imageSize[0] + 15u) / 16u) * ((p->imageSize[1] + 15u) / 16u);
    if (ext && ext->tileSize != 0 && ext->tileWorld != 0)
        blocks16 = (uint64_t)mrirt_tiles_for_rank(p->imageSize[0], p->imageSize[1], ext->tileSize, ext->tileRank, ext->tileWorld) *
                   ((ext->tileSize + 15u) / 16u) * ((ext->tileSize + 15u) / 16u);
    const bool bigBlocks = ((variant & 2u) != 0u) != (layout == MRIRT_LAYOUT_VGA && (variant & (64u | 2048u)) == 0u && blocks16 >= 4096u);
    // XCD-interleaved bands one workgroup row high by default (8-px bands for 8 x 8 workgroups: config 2 0.606 -> 0.580 ms,
    // K1 at 512^3 level; variant bit 3: contiguous run per XCD; bits 4-5: 16 / 8 / 32 / 64 px)
    const uint32_t bandSel = (variant >> 4) & 3u;
    const uint32_t bandAsked = bandSel == 0 ? (bigBlocks ? 16u : 8u) : bandSel == 1 ? 16u : bandSel == 2 ? 32u : 64u;
    const uint32_t bandPx = (variant & 8u) ? 0u : bandAsked;
    int rc = fill_pixel_map(a.map, p->imageSize[0], p->imageSize[1], pitch_px, ext,
                            bigBlocks ? 16u : 8u, (variant & 1u) ? 0u : 1u, bandPx,
                            // the per-band shift of the workgroup order (PixelMap::bandShift): config 2 0.591 -> 0.457 ms; on the
                            // 16-px workgroups of VGA grids it measured 1.3 % slower (C3), so those keep straight bands.  Bit 9 flips it.
                            ((variant & 512u) == 0u) != bigBlocks);
    if (rc != MRIRT_OK) return rc;
    fill_grid_dims(a.grid, p->dims, (layout == MRIRT_LAYOUT_VGA || mod4) ? (uint32_t)MRIRT_LAYOUT_VG : layout);
    fill_vga_dims(a.vga, p->dims);
    fill_label_addr(a.lab, p->dims, labCells ? (uint32_t)MRIRT_LAYOUT_LINEAR : labLayout);
    for (int k = 0; k < 3; ++k) {
        a.bmin[k] = p->volMin[k];
        a.bmax[k] = p->volMin[k] + p->voxelSize[k] * (float)p->dims[k];
        a.vox[k] = make_udiv(p->voxelSize[k]);
        a.halfInvVoxel[k] = 0.5f / p->voxelSize[k];
        a.hiLin[k] = (float)p->dims[k] - 1.001f;
        a.hiLab[k] = (float)p->dims[k] - 1.0f;
        a.bg[k] = p->bgColor[k];
    }
    a.stepSize = p->stepSize; a.nearT = p->nearT; a.farT = p->farT;
    float wSum = 0.0f;
    a.nch = 0;
    for (int m = 0; m < 4; ++m) a.chan[m] = 0;
    for (int m = 0; m < 4; ++m) {
        a.enabled[m] = p->volEnabled[m];
        a.weight[m] = p->volWeight[m];
        a.vol[m] = vol ? vol[m] : nullptr;
        if (p->volEnabled[m] != 0) { wSum += p->volWeight[m]; a.chan[a.nch++] = (uint32_t)m; }   // shader's order
    }
    a.wsum = make_udiv(wSum);
    a.wwDiv = make_udiv(p->ww);
    a.tfLo = p->wl - p->ww * 0.5f;
    a.intensityAlpha = p->intensityAlpha; a.gamma = p->gamma;
    a.showSeg = p->showSeg; a.showPred = p->showPred;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) a.lut[i][j] = p->lutColorAlpha[i][j];
    for (int i = 0; i < 8; ++i) {
        // brats_rt.slang:147,158 in the oracle's arithmetic: fp32 argument in the written order, exp correctly
        // rounded through fp64, fp32 subtraction
        const float xs = -a.lut[i][3] * a.stepSize;
        const float xp = xs * 1.5f;
        a.segAlpha[i] = 1.0f - (float)exp((double)xs);
        a.predAlpha[i] = 1.0f - (float)exp((double)xp);
    }
    a.ka = ext ? ext->ka : 0.0f; a.kd = ext ? ext->kd : 0.0f; a.ks = ext ? ext->ks : 0.0f;
    a.gradEps = ext ? ext->gradEps : 0.0f;
    a.specPow2 = ext ? ext->specPow2 : 0u;
    a.ert = (ext && ext->ertOverride) ? ext->ertThreshold : 0.01f;   // brats_rt.slang:117
    a.half = fmt == MRIRT_OUT_RGBA16F ? 1u : 0u;
    a.labels = labCells ? nullptr : static_cast<const uint32_t*>(labels);
    a.preds = labCells ? nullptr : static_cast<const uint32_t*>(preds);
    a.labCell = labCells ? static_cast<const uint2*>(labels) : nullptr;
    a.classStream = nullptr; a.rayOffsets = nullptr;
    a.skipDist = nullptr; a.mX = a.mXY = a.mY = a.mZ = 0; a.leap = 0;
    fill_exp_consts(a.ec);
    a.expSmall = (fabsf(p->intensityAlpha * p->stepSize) <= 0.125f) ? 1u : 0u;   // val is in [0, 1]
    a.out = nullptr; a.stats = nullptr;
    a.debugFlags = (variant >> 7) & 15u;
    if (variant & 32768u) a.debugFlags |= 256u;                        // the tagged twin of the benched kernel (launch_pipe)
    if (variant & 4096u) a.debugFlags |= 32u;                          // ring kernel: three planes instead of four
    cfg.layout = layout; cfg.math = math;
    cfg.shade = ext && ext->shadeMode != 0;
    cfg.pipe = a.nch >= 1 && !(variant & 4u);
    // the LDS-staged kernel (brats_slab.hip): VGA grids, one modality, no overlays, one packet per workgroup; variant bit 6
    cfg.slab = layout == MRIRT_LAYOUT_VGA && (variant & 64u) != 0 && cfg.pipe && a.nch == 1 && p->showSeg == 0 && p->showPred == 0 &&
               a.map.blockPx == 8;
    // the plane-synchronous LDS ring kernel (brats_ring.hip): the same launches; variant bit 11
    cfg.ring = layout == MRIRT_LAYOUT_VGA && (variant & 2048u) != 0 && (variant & 64u) == 0 && cfg.pipe && a.nch == 1 && p->showSeg == 0 &&
               p->showPred == 0 && a.map.blockPx == 8 && p->dims[0] >= 16 && p->dims[1] >= 16 && p->dims[2] >= 16;
    return MRIRT_OK;
}

// ---------------------------------------------------------------------------------------
// Exact empty-space skipping: the per-launch mask.  Bit = 1 when, for every sample whose base cell lies in
// the macro cell, val <= 0 is certain (the same weighted sum / wSum division / window test as the march,
// evaluated on per-cell upper bounds of the trilinear fetch: every step is monotone, so bound in -> bound
// out) and no shown label grid holds a label there.
// ---------------------------------------------------------------------------------------
struct SkipArgs {
    uint32_t cells, nch;
    const float* ub[4];          // compacted like K1Args::chan
    float w[4];
    UDiv wsum;
    float tfLo;
    const uint32_t* seg;
    const uint32_t* pred;
    uint32_t* mask;
};

// (scratch layout: skip_bit_words / skip_map_stride, mrirt_device.h)
// where lane 0 of the wave whose first cell is `cell` stores its ballot (two words), or -1: no store
MRIRT_HD int64_t skip_ballot_word(uint32_t cell, uint32_t cells) { return cell < ((cells + 63u) & ~63u) ? (int64_t)(cell >> 5) : -1; }

template <bool STRICT>
__global__ __launch_bounds__(256) void skip_mask_kernel(const SkipArgs k) {
    using Mm = M<STRICT>;
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    bool empty = false;
    if (cell < k.cells) {
        float v = 0.0f;
        for (uint32_t c = 0; c < k.nch; ++c) v = Mm::mad(k.ub[c][cell], k.w[c], v);
        if (k.wsum.d > 0.0f) v = Mm::divu_data(v, k.wsum);
        empty = v <= k.tfLo;                                         // NaN / inf bounds: not empty
        if (k.seg != nullptr && k.seg[cell] != 0u) empty = false;
        if (k.pred != nullptr && k.pred[cell] != 0u) empty = false;
    }
    const uint64_t bits = __ballot(empty);
    const int64_t w = skip_ballot_word(cell, k.cells);               // mask holds whole ballots only
    if ((threadIdx.x & 63u) == 0u && w >= 0) {
        k.mask[w] = (uint32_t)bits;
        k.mask[w + 1] = (uint32_t)(bits >> 32);
    }
}

// The distance map from the mask, one axis at a time (box emptiness is separable).  r(c) = largest r <= cap such that
// every in-grid cell within r - 1 of c along the axes done so far has the property; cells outside the grid never hold a
// sample, so they do not constrain.  Pass x reads the bits, passes y and z read the previous pass's bytes.
// `at(cell)` = the previous pass's value of a cell (pass x: cap or 0 from the bit).
template <int AXIS, class At>
MRIRT_HD uint32_t skip_dist_cell(uint32_t c, uint32_t mx, uint32_t my, uint32_t mz, At at) {
    const uint32_t xyz[3] = { c % mx, (c / mx) % my, c / (mx * my) }, ext[3] = { mx, my, mz };
    const uint32_t stride = AXIS == 0 ? 1u : AXIS == 1 ? mx : mx * my;
    // m = smallest value within distance r of c; radius r + 1 is good when m >= r + 1
    uint32_t m = at(c), r = 0;
    while (r < m && r < kSkipDistCap) {
        ++r;
        if (xyz[AXIS] >= r) { const uint32_t v = at(c - r * stride); m = v < m ? v : m; }
        if (xyz[AXIS] + r < ext[AXIS]) { const uint32_t v = at(c + r * stride); m = v < m ? v : m; }
    }
    return r;
}
MRIRT_HD uint32_t skip_bit_value(const uint32_t* mask, uint32_t cell) { return ((mask[cell >> 5] >> (cell & 31u)) & 1u) != 0 ? kSkipDistCap : 0u; }

template <int AXIS>
__global__ __launch_bounds__(256) void skip_dist_kernel(const uint32_t* __restrict__ mask, const uint8_t* __restrict__ prev,
                                                        uint8_t* __restrict__ next, uint32_t mx, uint32_t my, uint32_t mz) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= mx * my * mz) return;
    next[c] = (uint8_t)skip_dist_cell<AXIS>(c, mx, my, mz, [&](uint32_t cell) -> uint32_t {
        if constexpr (AXIS == 0) return skip_bit_value(mask, cell);
        else return prev[cell];
    });
}

}  // namespace mrirt

using namespace mrirt;

// kernelVariant toggles (experiments; 0 = library default):
//   bit 0: row-major instead of Morton lane order      bit 1: flip 64- / 256-thread workgroups (256 is the default on VGA)
//   bit 2: no software pipelining                      bits 3-5: XCD band height (prepare(); bits 4-5 = 1: 16 px)
//   bit 6: the LDS-staged kernel of brats_slab.hip     bit 7: ... counts its LDS-served samples in stats[1]; the skipping
//                                                             kernels: the samples they did NOT fetch (flagged or leapt)
//   bit 8: skipping one step at a time (no leaps); in the slab kernel: count ring misses
//   bit 9: every band of an XCD starts at x = 0 (no per-band shift of the workgroup order: PixelMap::bandShift)
//   bit 15: the tagged twin of the benched kernel (same code, another symbol: bench.py's side measurements)
extern "C" int mrirt_render_brats_ex(const MrirtBratsParams* p, const MrirtRenderExt* ext,
                                     const void* const vol[4], const void* labels, const void* preds,
                                     void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream) {
    if (!out_rgba) return MRIRT_ERR_NULL;
    K1Args a;
    Prepared cfg;
    int rc = prepare(p, ext, vol, labels, preds, true, pitch_px, a, cfg);
    if (rc != MRIRT_OK) return rc;
    if (p->showPred != 0 && !preds && a.labCell == nullptr) return MRIRT_ERR_NULL;
    a.out = out_rgba;
    a.stats = stats_dev;
    if (a.map.numBlocks == 0) return MRIRT_OK;   // a rank that owns no tile
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_family_probe != nullptr && (cfg.slab || cfg.ring)) { *g_family_probe = cfg.slab ? MRIRT_KERNEL_SLAB : MRIRT_KERNEL_RING; return MRIRT_OK; }
    if (cfg.slab) return launch_slab_march(a, cfg.math == MRIRT_MATH_STRICT, cfg.shade, s);
    if (cfg.ring) return launch_ring_march(a, cfg.math == MRIRT_MATH_STRICT, cfg.shade, s);
    return cfg.math == MRIRT_MATH_STRICT ? launch_layout<true>(a, cfg.layout, cfg.shade, cfg.pipe, s)
                                         : launch_layout<false>(a, cfg.layout, cfg.shade, cfg.pipe, s);
}

// Does this launch march with an empty-radius map?  Skipping is sound only where "upper bound <= window floor" implies
// "contributes nothing": positive window width and gamma (pow(0, g) = 0), non-negative weights (monotone sum), a bound for
// every enabled modality and a label summary for every shown overlay — and it pays only where the launch has a SKIP kernel:
// on VG / VGA grids the pipelined kernel (one modality) or the rolling kernel (2-4 modalities, no overlays), on QUAD grids
// the pipelined kernel (launch()); STRICT with gamma == 1, or FAST (launch_pipe() / launch_roll()).  Anything else would pay
// the four pre-pass launches for a kernel that ignores the map.  Otherwise: the ordinary launch, no map built.
static bool skip_applicable(const MrirtBratsParams* p, const K1Args& a, const Prepared& cfg, const MrirtSkip* skip) {
    bool ok = skip != nullptr && p->ww > 0.0f && p->gamma > 0.0f && a.nch >= 1;      // (the scratch itself: the caller's check)
    const bool wide = a.grid.wide != 0;
    const bool overlays = p->showSeg != 0 || p->showPred != 0;
    const bool layoutOk = ((cfg.layout == MRIRT_LAYOUT_VG && !wide) || cfg.layout == MRIRT_LAYOUT_VGA) ? (a.nch == 1 || !overlays)   // pipelined / rolling
                        : ((cfg.layout == MRIRT_LAYOUT_QUAD || cfg.layout == MRIRT_LAYOUT_MOD4) && !wide && !cfg.shade);
    const bool mathOk = cfg.math == MRIRT_MATH_FAST || p->gamma == 1.0f;
    ok = ok && cfg.pipe && !cfg.slab && !cfg.ring && layoutOk && mathOk;
    for (int k = 0; k < 3; ++k) ok = ok && (p->dims[k] + 7) / 8 <= 256;   // macro coordinates travel through 8-bit wave reductions
    for (uint32_t c = 0; c < a.nch && ok; ++c)
        ok = skip->macroUb[a.chan[c]] != nullptr && a.weight[a.chan[c]] >= 0.0f;
    if (p->showSeg != 0 && !skip->macroSeg) ok = false;
    if (p->showPred != 0 && !skip->macroPred) ok = false;
    // the label overlays of a launch that takes the generic kernel (label grids >= 2^30 elements: launch()) are not skipped
    return ok;
}

extern "C" int mrirt_render_brats_skip(const MrirtBratsParams* p, const MrirtRenderExt* ext,
                                       const void* const vol[4], const void* labels, const void* preds,
                                       const MrirtSkip* skip, void* out_rgba, int64_t pitch_px,
                                       uint64_t* stats_dev, void* stream) {
    if (!out_rgba) return MRIRT_ERR_NULL;
    if (!skip) return mrirt_render_brats_ex(p, ext, vol, labels, preds, out_rgba, pitch_px, stats_dev, stream);
    K1Args a;
    Prepared cfg;
    int rc = prepare(p, ext, vol, labels, preds, true, pitch_px, a, cfg);
    if (rc != MRIRT_OK) return rc;
    if (p->showPred != 0 && !preds && a.labCell == nullptr) return MRIRT_ERR_NULL;
    a.out = out_rgba;
    a.stats = stats_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool ok = skip_applicable(p, a, cfg, skip) && skip->mask != nullptr;
    if (ok && (int64_t)skip->maskWords < mrirt_skip_mask_words(p->dims)) return MRIRT_ERR_ARG;
    // (a rank that owns no tile still builds the map: "this call returned MRIRT_OK" must mean "skip->mask holds the map",
    // which is what a caller's mapReady on the next frame rests on)
    if (ok && skip->mapReady != 0) {
        // the scratch already holds this configuration's map (the caller vouches for it: MrirtSkip::mapReady)
        const uint32_t mx = (p->dims[0] + 7) / 8, my = (p->dims[1] + 7) / 8, mz = (p->dims[2] + 7) / 8, cells = mx * my * mz;
        a.skipDist = reinterpret_cast<uint8_t*>(skip->mask + skip_bit_words(cells));
        a.mX = mx; a.mXY = mx * my; a.mY = my; a.mZ = mz;
        a.leap = (a.debugFlags & 2u) == 0u ? 1u : 0u;
    } else if (ok && g_family_probe != nullptr) {
        a.skipDist = reinterpret_cast<uint8_t*>(skip->mask);          // (probe: the launchers only test it against NULL)
    } else if (ok) {
        SkipArgs k;
        const uint32_t mx = (p->dims[0] + 7) / 8, my = (p->dims[1] + 7) / 8, mz = (p->dims[2] + 7) / 8;
        k.cells = mx * my * mz; k.nch = a.nch;
        for (uint32_t c = 0; c < 4; ++c) { k.ub[c] = c < a.nch ? skip->macroUb[a.chan[c]] : nullptr; k.w[c] = c < a.nch ? a.weight[a.chan[c]] : 0.0f; }
        k.wsum = a.wsum; k.tfLo = a.tfLo;
        k.seg = p->showSeg != 0 ? skip->macroSeg : nullptr;
        k.pred = p->showPred != 0 ? skip->macroPred : nullptr;
        k.mask = skip->mask;
        const dim3 grid((k.cells + 255) / 256), block(256);
        if (cfg.math == MRIRT_MATH_STRICT) hipLaunchKernelGGL(skip_mask_kernel<true>, grid, block, 0, s, k);
        else                               hipLaunchKernelGGL(skip_mask_kernel<false>, grid, block, 0, s, k);
        MRIRT_HIP(hipGetLastError());
        // bits -> distance bytes; the two byte maps follow the bit words in the same scratch (mrirt_skip_mask_words)
        uint8_t* mapA = reinterpret_cast<uint8_t*>(skip->mask + skip_bit_words(k.cells));
        uint8_t* mapB = mapA + skip_map_stride(k.cells);
        hipLaunchKernelGGL(skip_dist_kernel<0>, grid, block, 0, s, skip->mask, (const uint8_t*)nullptr, mapA, mx, my, mz);
        hipLaunchKernelGGL(skip_dist_kernel<1>, grid, block, 0, s, skip->mask, (const uint8_t*)mapA, mapB, mx, my, mz);
        hipLaunchKernelGGL(skip_dist_kernel<2>, grid, block, 0, s, skip->mask, (const uint8_t*)mapB, mapA, mx, my, mz);
        MRIRT_HIP(hipGetLastError());
        a.skipDist = mapA; a.mX = mx; a.mXY = mx * my; a.mY = my; a.mZ = mz;
        a.leap = (a.debugFlags & 2u) == 0u ? 1u : 0u;                 // kernelVariant bit 8: first level only (A/B timing)
    }
    if (a.map.numBlocks == 0) return MRIRT_OK;
    return cfg.math == MRIRT_MATH_STRICT ? launch_layout<true>(a, cfg.layout, cfg.shade, cfg.pipe, s)
                                         : launch_layout<false>(a, cfg.layout, cfg.shade, cfg.pipe, s);
}

// 1: mrirt_render_brats_skip with these arguments builds (or reuses) an empty-radius map and marches with it; 0: it is the
// plain launch and never touches skip->mask (callers then need no scratch); < 0: the status the render call would return.
extern "C" int mrirt_brats_skip_applicable(const MrirtBratsParams* p, const MrirtRenderExt* ext, const void* const vol[4],
                                           const void* labels, const void* preds, const MrirtSkip* skip) {
    if (!skip) return 0;
    K1Args a;
    Prepared cfg;
    const int rc = prepare(p, ext, vol, labels, preds, true, p ? (int64_t)p->imageSize[0] : 0, a, cfg);
    if (rc != MRIRT_OK) return rc;
    return skip_applicable(p, a, cfg, skip) ? 1 : 0;
}

// Which kernel family the render call with these arguments launches (host-only, nothing is launched): MrirtKernelFamily,
// possibly with MRIRT_KERNEL_SKIPPING / MRIRT_KERNEL_LABEL_CELLS or'ed in; < 0: the status the render call would return.
extern "C" int mrirt_brats_kernel_family(const MrirtBratsParams* p, const MrirtRenderExt* ext, const void* const vol[4],
                                         const void* labels, const void* preds, const MrirtSkip* skip) {
    int family = MRIRT_KERNEL_NONE;
    g_family_probe = &family;
    const int64_t pitch = p ? (int64_t)p->imageSize[0] : 0;
    const int rc = mrirt_render_brats_skip(p, ext, vol, labels, preds, skip, reinterpret_cast<void*>(uintptr_t(16)), pitch, nullptr, nullptr);
    g_family_probe = nullptr;
    return rc != MRIRT_OK ? rc : family;
}

extern "C" int mrirt_render_brats(const MrirtBratsParams* params, const float* const vol[4],
                                  const uint32_t* labels, const uint32_t* preds,
                                  float* out_rgba, int64_t pitch_px, void* stream) {
    if (!vol) return MRIRT_ERR_NULL;
    const void* v[4] = { vol[0], vol[1], vol[2], vol[3] };
    return mrirt_render_brats_ex(params, nullptr, v, labels, preds, out_rgba, pitch_px, nullptr, stream);
}

extern "C" int mrirt_render_brats_stream(const MrirtBratsParams* p, const MrirtRenderExt* ext,
                                         const void* const vol[4], const void* labels,
                                         const int16_t* classes, const int64_t* offsets,
                                         void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream) {
    if (!out_rgba || !classes || !offsets) return MRIRT_ERR_NULL;
    if (ext && ext->tileSize != 0) return MRIRT_ERR_ARG;         // whole-frame only
    K1Args a;
    Prepared cfg;
    int rc = prepare(p, ext, vol, labels, nullptr, true, pitch_px, a, cfg);
    if (rc != MRIRT_OK) return rc;
    if (a.labCell != nullptr) return MRIRT_ERR_LAYOUT;           // (the class stream replaces gPreds; label cells carry both grids)
    if (p->showPred == 0) return MRIRT_ERR_ARG;
    a.classStream = classes; a.rayOffsets = offsets;
    a.out = out_rgba; a.stats = stats_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the generic kernel: the pipelined ones do not read a class stream (Stage::issue)
    return cfg.math == MRIRT_MATH_STRICT ? launch_layout<true>(a, cfg.layout, cfg.shade, false, s)
                                         : launch_layout<false>(a, cfg.layout, cfg.shade, false, s);
}

extern "C" int mrirt_brats_sample_counts(const MrirtBratsParams* p, const MrirtRenderExt* ext, uint32_t* counts, void* stream) {
    if (!counts) return MRIRT_ERR_NULL;
    if (ext && ext->tileSize != 0) return MRIRT_ERR_ARG;
    K1Args a;
    Prepared cfg;
    int rc = prepare(p, ext, nullptr, nullptr, nullptr, false, p ? p->imageSize[0] : 0, a, cfg);
    if (rc != MRIRT_OK) return rc;
    hipLaunchKernelGGL(sample_count_kernel, dim3(a.map.chunk * kXcds), dim3(a.map.blockPx == 8 ? 64 : 256), 0,
                       static_cast<hipStream_t>(stream), a, counts);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <bool STRICT>
static int launch_emit(const K1Args& a, const EmitArgs& e, uint32_t layout, hipStream_t s) {
    if (layout > MRIRT_LAYOUT_QUAD && layout != MRIRT_LAYOUT_MOD4) return MRIRT_ERR_LAYOUT;      // the C5 passes read LINEAR / BRICK / VG / QUAD / MOD4 grids
    const dim3 grid(a.map.chunk * kXcds), block(a.map.blockPx == 8 ? 64 : 256);
    switch (layout) {
        case MRIRT_LAYOUT_LINEAR: hipLaunchKernelGGL((emit_samples_kernel<STRICT, 0>), grid, block, 0, s, a, e); break;
        case MRIRT_LAYOUT_BRICK:  hipLaunchKernelGGL((emit_samples_kernel<STRICT, 1>), grid, block, 0, s, a, e); break;
        case MRIRT_LAYOUT_VG:     hipLaunchKernelGGL((emit_samples_kernel<STRICT, 2>), grid, block, 0, s, a, e); break;
        case MRIRT_LAYOUT_MOD4:   hipLaunchKernelGGL((emit_samples_kernel<STRICT, MRIRT_LAYOUT_MOD4>), grid, block, 0, s, a, e); break;
        default:                  hipLaunchKernelGGL((emit_samples_kernel<STRICT, 3>), grid, block, 0, s, a, e); break;
    }
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

extern "C" int mrirt_brats_emit_samples(const MrirtBratsParams* p, const MrirtRenderExt* ext, const void* const vol[4],
                                        const float zmu[4], const float zsigma[4], const int64_t* offsets,
                                        float* coords, float* feats, void* stream) {
    if (!vol || !zmu || !zsigma || !offsets || !coords || !feats) return MRIRT_ERR_NULL;
    for (int m = 0; m < 4; ++m) if (!vol[m]) return MRIRT_ERR_NULL;          // the MLP reads all four modalities
    if (ext && ext->tileSize != 0) return MRIRT_ERR_ARG;
    K1Args a;
    Prepared cfg;
    int rc = prepare(p, ext, vol, nullptr, nullptr, false, p ? p->imageSize[0] : 0, a, cfg);
    if (rc != MRIRT_OK) return rc;
    EmitArgs e;
    for (int m = 0; m < 4; ++m) { e.zmu[m] = zmu[m]; e.zsigma[m] = make_udiv(zsigma[m]); }
    for (int k = 0; k < 3; ++k) { e.dimM1[k] = (double)(p->dims[k] - 1); e.rdimM1[k] = 1.0 / e.dimM1[k]; }
    e.offsets = offsets; e.coords = coords; e.feats = reinterpret_cast<float4*>(feats); e.mix = nullptr; e.seg = nullptr; e.geom = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return cfg.math == MRIRT_MATH_STRICT ? launch_emit<true>(a, e, cfg.layout, s) : launch_emit<false>(a, e, cfg.layout, s);
}

// ---------------------------------------------------------------------------------------
// C5 as one call: the chunked, ERT-aware per-sample INR render (see c5_plan_kernel).  Everything is enqueued on
// `stream`; batch sizes travel through device memory, so there is no host synchronisation inside the frame.
// ---------------------------------------------------------------------------------------
namespace mrirt {

constexpr uint32_t kC5MaxPasses = 1024;

__global__ void c5_sum_kernel(const uint32_t* __restrict__ counters, uint32_t n, uint64_t* __restrict__ queries) {
    uint64_t s = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 64) s += counters[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(queries), (unsigned long long)s);
}

struct C5Scratch { uint32_t* counters; C5Ray* rays; float4* geom; uint2* rowOwner; float* coords; float* feats; float* mix; uint32_t* seg; int16_t* classes; int64_t cap, bytes; };

static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static int c5_carve(const MrirtBratsParams* p, uint32_t chunk, void* base, C5Scratch& sc) {
    if (!p || chunk == 0 || chunk > 4096) return MRIRT_ERR_ARG;
    const int64_t px = (int64_t)p->imageSize[0] * p->imageSize[1];
    if (px <= 0) return MRIRT_ERR_DIMS;
    sc.cap = px * chunk;
    if (sc.cap >= ((int64_t)1 << 32) - ((int64_t)1 << 21)) return MRIRT_ERR_ARG;   // 32-bit row numbers per pass (+ one grid stride of headroom)
    char* b = static_cast<char*>(base);
    int64_t o = 0;
    sc.counters = reinterpret_cast<uint32_t*>(b + o); o += align256((int64_t)kC5MaxPasses * 8);     // batch sizes, then the refinement's segment tickets
    sc.rays = reinterpret_cast<C5Ray*>(b + o);        o += align256(px * (int64_t)sizeof(C5Ray));
    sc.geom = reinterpret_cast<float4*>(b + o);       o += align256(px * 32);
    sc.rowOwner = reinterpret_cast<uint2*>(b + o);    o += align256(sc.cap * 8);
    sc.coords = reinterpret_cast<float*>(b + o);      o += align256(sc.cap * 12);
    sc.feats = reinterpret_cast<float*>(b + o);       o += align256(sc.cap * 16);
    sc.mix = reinterpret_cast<float*>(b + o);         o += align256(sc.cap * 16);
    sc.seg = reinterpret_cast<uint32_t*>(b + o);      o += align256(sc.cap * 4);
    sc.classes = reinterpret_cast<int16_t*>(b + o);   o += align256(sc.cap * 2);
    sc.bytes = o;
    return MRIRT_OK;
}

template <bool STRICT>
static int c5_launch_plan(const K1Args& a, const EmitArgs& e, uint32_t layout, bool shade, const C5Scratch& sc, uint32_t pass,
                          uint32_t chunk, hipStream_t s) {
    // the C5 passes read LINEAR / BRICK / VG / QUAD / MOD4 grids; QUAD and MOD4 carry no gradients
    const bool flat = layout == MRIRT_LAYOUT_QUAD || layout == MRIRT_LAYOUT_MOD4;
    if ((layout > MRIRT_LAYOUT_QUAD && layout != MRIRT_LAYOUT_MOD4) || (flat && shade)) return MRIRT_ERR_LAYOUT;
    uint32_t* counter = sc.counters + pass;
    if (pass == 0) {                                                  // later passes are planned by the composite kernel
        hipLaunchKernelGGL(c5_plan_kernel, dim3(a.map.chunk * kXcds), dim3(a.map.blockPx == 8 ? 64 : 256), 0, s,
                           a, sc.rays, sc.geom, sc.rowOwner, counter, chunk);
        MRIRT_HIP(hipGetLastError());
    }
    const int64_t blocksWanted = (sc.cap + 255) / 256;
    const dim3 grid((uint32_t)(blocksWanted < 4096 ? blocksWanted : 4096)), block(256);
#define MRIRT_C5E(L, SH) hipLaunchKernelGGL((c5_emit_kernel<STRICT, L, SH>), grid, block, 0, s, a, e, sc.rowOwner, counter)
    switch (layout) {
        case MRIRT_LAYOUT_LINEAR: if (shade) MRIRT_C5E(0, true); else MRIRT_C5E(0, false); break;
        case MRIRT_LAYOUT_BRICK:  if (shade) MRIRT_C5E(1, true); else MRIRT_C5E(1, false); break;
        case MRIRT_LAYOUT_VG:     if (shade) MRIRT_C5E(2, true); else MRIRT_C5E(2, false); break;
        case MRIRT_LAYOUT_MOD4:   MRIRT_C5E(MRIRT_LAYOUT_MOD4, false); break;
        default:                  MRIRT_C5E(3, false); break;
    }
#undef MRIRT_C5E
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <bool STRICT>
static int c5_launch_composite(const K1Args& a, bool shade, const C5Scratch& sc, uint32_t* nextCounter, uint32_t chunk, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(a.map.blockPx == 8 ? 64 : 256);     // the plan kernel's pixel -> lane map
    if (shade) hipLaunchKernelGGL((c5_composite_kernel<STRICT, true>), grid, block, 0, s, a, sc.rays, sc.geom, sc.classes, sc.mix, sc.seg, sc.rowOwner, nextCounter, chunk);
    else       hipLaunchKernelGGL((c5_composite_kernel<STRICT, false>), grid, block, 0, s, a, sc.rays, sc.geom, sc.classes, sc.mix, sc.seg, sc.rowOwner, nextCounter, chunk);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

}  // namespace mrirt

extern "C" int64_t mrirt_brats_inr_scratch_bytes(const MrirtBratsParams* p, uint32_t chunk_steps) {
    C5Scratch sc;
    return c5_carve(p, chunk_steps, nullptr, sc) == MRIRT_OK ? sc.bytes : 0;
}

extern "C" int mrirt_render_brats_inr(const MrirtBratsParams* p, const MrirtRenderExt* ext, const void* const vol[4],
                                      const void* labels, const MrirtInrDesc* net, const float zmu[4], const float zsigma[4],
                                      uint32_t chunk_steps, void* scratch, int64_t scratch_bytes,
                                      void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream) {
    if (!out_rgba || !scratch || !net || !vol || !zmu || !zsigma) return MRIRT_ERR_NULL;
    const bool mod4 = ext && ext->layout == MRIRT_LAYOUT_MOD4;
    for (int m = 0; m < (mod4 ? 1 : 4); ++m) if (!vol[m]) return MRIRT_ERR_NULL;     // the MLP reads all four modalities
    if (ext && ext->tileSize != 0) return MRIRT_ERR_ARG;                      // whole-frame only
    if ((net->kind != MRIRT_INR_FOURIER_RELU && net->kind != MRIRT_INR_SIREN) || net->numMods != 4) return MRIRT_ERR_ARG;
    K1Args a;
    Prepared cfg;
    int rc = prepare(p, ext, vol, labels, nullptr, true, pitch_px, a, cfg);
    if (rc != MRIRT_OK) return rc;
    if (a.labCell != nullptr) return MRIRT_ERR_LAYOUT;           // the C5 passes read the ground-truth grid themselves (LINEAR / BRICK)
    if (p->showPred == 0) return MRIRT_ERR_ARG;
    C5Scratch sc;
    rc = c5_carve(p, chunk_steps, scratch, sc);
    if (rc != MRIRT_OK) return rc;
    if (scratch_bytes < sc.bytes) return MRIRT_ERR_ARG;
    // passes: a ray's chord is at most the box diagonal; the march counts it with the running fp32 sum t += stepSize, whose
    // increments are each off by up to half an ulp of t — a relative drift of up to ulp(tFar) / (2 stepSize) over the chord
    // when stepSize is only a few tens of ulps of t.  The bound carries twice that drift (tFar = the farthest any
    // sample can be, as in prepare()), so that no ray is still marching when the last pass has been composited.
    double diag2 = 0.0, dist2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double e = (double)p->voxelSize[k] * (double)p->dims[k];
        const double c = (double)p->volMin[k] + 0.5 * e - (double)p->eye[k];
        diag2 += e * e; dist2 += c * c;
    }
    double tFar = sqrt(dist2) + sqrt(diag2);
    if (ext && ext->cameraMode == 1u) tFar += fabs((double)ext->orthoHalfHeight) * (1.0 + (double)p->imageSize[0] / fmax(1.0, (double)p->imageSize[1]));
    const double ulpFar = (double)(nextafterf((float)tFar, INFINITY) - (float)tFar);
    const double drift = 1.0 + ulpFar / (double)p->stepSize;
    const uint64_t maxSteps = (uint64_t)(sqrt(diag2) / (double)p->stepSize * drift) + 3;
    const uint64_t passes = (maxSteps + chunk_steps - 1) / chunk_steps;
    if (passes > kC5MaxPasses) return MRIRT_ERR_ARG;
    EmitArgs e;
    for (int m = 0; m < 4; ++m) { e.zmu[m] = zmu[m]; e.zsigma[m] = make_udiv(zsigma[m]); }
    for (int k = 0; k < 3; ++k) { e.dimM1[k] = (double)(p->dims[k] - 1); e.rdimM1[k] = 1.0 / e.dimM1[k]; }
    e.offsets = nullptr; e.coords = sc.coords; e.feats = reinterpret_cast<float4*>(sc.feats); e.mix = sc.mix; e.seg = sc.seg; e.geom = sc.geom;
    a.out = out_rgba;
    a.stats = stats_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MRIRT_HIP(hipMemsetAsync(sc.counters, 0, (size_t)kC5MaxPasses * 8, s));
    const bool strict = cfg.math == MRIRT_MATH_STRICT;
    for (uint32_t c = 0; c < (uint32_t)passes; ++c) {
        rc = strict ? c5_launch_plan<true>(a, e, cfg.layout, cfg.shade, sc, c, chunk_steps, s)
                    : c5_launch_plan<false>(a, e, cfg.layout, cfg.shade, sc, c, chunk_steps, s);
        if (rc != MRIRT_OK) return rc;
        rc = inr_forward_dev_n(net, sc.coords, sc.feats, sc.cap, sc.counters + c, sc.classes, sc.counters + kC5MaxPasses + c, s);
        if (rc != MRIRT_OK) return rc;
        uint32_t* next = c + 1 < (uint32_t)passes ? sc.counters + c + 1 : nullptr;
        rc = strict ? c5_launch_composite<true>(a, cfg.shade, sc, next, chunk_steps, s)
                    : c5_launch_composite<false>(a, cfg.shade, sc, next, chunk_steps, s);
        if (rc != MRIRT_OK) return rc;
    }
    if (stats_dev != nullptr) {                                               // stats_dev[2] += MLP queries of the frame
        hipLaunchKernelGGL(c5_sum_kernel, dim3(1), dim3(64), 0, s, sc.counters, (uint32_t)passes, stats_dev + 2);
        MRIRT_HIP(hipGetLastError());
    }
    return MRIRT_OK;
}
