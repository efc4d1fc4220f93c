// Device code shared by the K1 march kernels (brats_march.hip: the register-gather kernels; brats_slab.hip: the
// LDS-staged kernel): the kernel argument block, sampleLinear's cell arithmetic, the gathers of one cell per layout
// (Taps), the transfer function + compositing step, ray set-up.  Reference: inr/viewer/brats_rt.slang:36-168.
#pragma once
#include "mrirt_host.h"

namespace mrirt {

constexpr uint32_t kMaxStepsPerRay = 1u << 20;   // prepare(): stepSize may not cut the box diagonal finer

struct K1Args {
    Camera cam;
    PixelMap map;
    GridDims grid;
    VgaDims vga;             // LAYOUT 4: the three axis-flat copies' addressing
    LabelAddr lab;
    float bmin[3], bmax[3];
    UDiv vox[3];             // voxelSize per axis (pIdx = (p - bmin) / voxelSize)
    UDiv wsum;               // sum of the enabled volWeight, accumulated in slot order
    UDiv wwDiv;              // ww
    float halfInvVoxel[3];   // 0.5f / voxelSize      (gradient to world units)
    float hiLin[3];          // float(dims) - 1.001f   (sampleLinear clamp)
    float hiLab[3];          // float(dims) - 1.0f     (sampleLabel clamp)
    float stepSize, nearT, farT;
    float bg[3];
    uint32_t enabled[4];
    float weight[4];
    float tfLo;              // wl - ww*0.5
    float intensityAlpha, gamma;
    uint32_t showSeg, showPred;
    float lut[8][4];
    float segAlpha[8], predAlpha[8];   // the overlays' per-label opacities (exp of a launch constant: computed on the host)
    float ka, kd, ks, gradEps, ert;
    uint32_t specPow2;
    uint32_t half;           // 1: rgba16_float output
    uint32_t nch, chan[4];   // the enabled modalities, compacted in ascending order (pipelined kernel)
    const void* vol[4];
    const uint32_t* labels;
    const uint32_t* preds;
    const uint2* labCell;         // MRIRT_LAYOUT_LABCELL: both overlays' corner labels per cell (QUAD grids); labels / preds unused then
    const int16_t* classStream;   // C5: prediction label of sample k of ray p at classStream[rayOffsets[p] + k]
    const int64_t* rayOffsets;
    const uint8_t* skipDist;      // exact empty-space skipping: byte per 8^3 macro cell; 0 = may contribute, r >= 1 = this cell
                                  // and every macro cell within Chebyshev distance r - 1 of it contribute nothing
    uint32_t mX, mXY, mY, mZ;     // macro cells per row / per slice; rows; slices
    uint32_t leap;                // 1: take several steps at once where skipDist allows (second level)
    void* out;
    uint64_t* stats;
    ExpConsts ec;                 // fp64 constants of the strict exp, SGPR-resident
    uint32_t expSmall;            // intensityAlpha * stepSize <= 1/8: the intensity exp needs no range reduction
    uint32_t debugFlags;          // experiments only (kernelVariant bits 7..): bit 0 = the slab kernel counts LDS-served samples in stats[1]
};

template <bool STRICT>
__device__ __forceinline__ float trilerp(float c000, float c100, float c010, float c110,
                                         float c001, float c101, float c011, float c111,
                                         float fx, float fy, float fz) {
    using Mm = M<STRICT>;   // nesting order of sampleLinear, brats_rt.slang:74-75
    return Mm::lerp(Mm::lerp(Mm::lerp(c000, c100, fx), Mm::lerp(c010, c110, fx), fy),
                    Mm::lerp(Mm::lerp(c001, c101, fx), Mm::lerp(c011, c111, fx), fy), fz);
}

// ---------------------------------------------------------------------------------------
// One sample's position in index space: sampleLinear's clamp / floor / fract (shared by all
// modalities) plus the unclamped pIdx the label fetch rounds.
// ---------------------------------------------------------------------------------------
struct Cell {
    float q[3];              // pIdx
    float fx, fy, fz;
    uint32_t ix, iy, iz;
};

template <bool STRICT>
__device__ __forceinline__ void locate(const K1Args& a, const float ro[3], const float rd[3], float t, Cell& c) {
    using Mm = M<STRICT>;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p = Mm::mad(t, rd[k], ro[k]);                    // o + t*d (the sum commutes)
        c.q[k] = Mm::divu(p - a.bmin[k], a.vox[k]);                  // brats_rt.slang:119-120
    }
    const float cx = clampf(c.q[0], 0.0f, a.hiLin[0]);               // :62-64
    const float cy = clampf(c.q[1], 0.0f, a.hiLin[1]);
    const float cz = clampf(c.q[2], 0.0f, a.hiLin[2]);
    const float flx = floorf(cx), fly = floorf(cy), flz = floorf(cz);
    c.ix = (uint32_t)flx; c.iy = (uint32_t)fly; c.iz = (uint32_t)flz;
    c.fx = cx - flx; c.fy = cy - fly; c.fz = cz - flz;
}

// ---------------------------------------------------------------------------------------
// Taps: the gathers of one modality at one cell (issue), and their blend (eval).  Splitting the
// two lets the pipelined kernel keep a whole step of gathers in flight.
// ---------------------------------------------------------------------------------------
template <int LAYOUT, bool SHADE> struct Taps;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A gather the COMPILER DOES NOT KNOW IS A LOAD (pipelined march kernels, brats_march.hip).  hipcc's own wait insertion
// de-pipelines the two-stage march: it guards the address arithmetic in front of stage B's gathers — temporaries that
// share registers with B's load destinations — with s_waitcnt vmcnt(7) .. vmcnt(1), i.e. it waits for stage A's gathers
// before it ISSUES stage B's, whichever way the loop is written (it rotates it back); seen in the ISA of every
// pipelined kernel, profiles/r03_pipeline_waits.txt.  Issued as asm, a gather is just "an instruction that defines its
// destination": no wait is inserted anywhere, and the march places the ONE wait it needs itself (Taps::arrive<N>: the
// stage's registers are in/out operands of the s_waitcnt, so no use can be scheduled above it).  What keeps this sound:
// (1) the destination is live from the asm to its uses, so the allocator cannot hand the register to anything else while
// the load is in flight; (2) nothing may READ it before arrive() — tools/check_async_loads.py scans the built library's
// disassembly for a read (or a copy / spill) of a gather destination between the gather and its wait and fails the build
// check if there is one; (3) every other vector-memory operation of these kernels (frame store, counters) comes after the loop;
// (4) THE BASE POINTER IS RE-MATERIALISED BY A SCALAR MOVE INSIDE THE ASM STATEMENT.  gfx9 has a hazard the hardware does not
// interlock: a vector-memory instruction that reads an SGPR needs five wait states after a VALU instruction wrote that
// SGPR.  The compiler inserts the s_nops for loads it knows — not for inline asm (LLVM's hazard recogniser: "doesn't attempt
// to address all possible inline asm hazards").  Under SGPR pressure it keeps a modality's base pointer spilled in VGPR
// lanes and restores it with v_readlane_b32 — a VALU write of an SGPR — right in front of the gather: two wait states
// instead of five in brats_march_pipe_kernel<strict, QUAD, 3 modalities, labels>, whose gathers then could read the
// register's PREVIOUS content as the high half of their address — the one unexplained memory-access fault of rounds 3 and 4
// (tests/test_gpu_skip.py::test_skip_is_bit_identical[strict-quad-False-3]).  An s_mov_b64 is a scalar
// instruction: its read of a VALU-written SGPR is interlocked, and the gather's read of an SALU-written SGPR has no
// hazard, so the copy is the cheapest sound form (one SALU instruction per gather, no stall).  tools/check_async_loads.py
// now also scans EVERY kernel of the library for the hazard itself (a VALU write of an SGPR fewer than five wait states in
// front of a vector-memory instruction that reads it) and fails the build on a hit.
typedef uint64_t sbase_t;
__device__ __forceinline__ void async_load_vec4(float4& dst, const void* __restrict__ base, uint32_t elem) {
    f32x4 t;
    sbase_t b;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=v"(t), "=&s"(b) : "v"(elem << 4), "s"(base));
    dst = __builtin_bit_cast(float4, t);
}
__device__ __forceinline__ void async_load_u32(uint32_t& dst, const void* __restrict__ base, uint32_t byteOff) {
    uint32_t t;
    sbase_t b;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dword %0, %2, %1" : "=v"(t), "=&s"(b) : "v"(byteOff), "s"(base));
    dst = t;
}

template <bool SHADE> struct Taps<2, SHADE> {        // VG: 8 x (v, dx, dy, dz)
    float4 c[8];
    template <bool WIDE>
    __device__ __forceinline__ void issue(const void* __restrict__ vbuf, const GridDims& gd, const Cell& s) {
        const CellOffsets k = vec4_cell(gd, s.ix, s.iy, s.iz);
        constexpr bool w = WIDE;
        const uint32_t o10 = k.o + k.dx, o01 = k.o + k.dy, o11 = o10 + k.dy;
        c[0] = load_vec4<w>(vbuf, k.o);        c[1] = load_vec4<w>(vbuf, o10);
        c[2] = load_vec4<w>(vbuf, o01);        c[3] = load_vec4<w>(vbuf, o11);
        c[4] = load_vec4<w>(vbuf, k.o + k.dz); c[5] = load_vec4<w>(vbuf, o10 + k.dz);
        c[6] = load_vec4<w>(vbuf, o01 + k.dz); c[7] = load_vec4<w>(vbuf, o11 + k.dz);
    }
    // the same eight gathers as inline asm (issue_async) and the wait that retires them (arrive): see async_load_vec4
    __device__ __forceinline__ void issue_async(const void* __restrict__ vbuf, const CellOffsets& k) {
        const uint32_t o10 = k.o + k.dx, o01 = k.o + k.dy, o11 = o10 + k.dy;
        // ONE statement: the base copied once by a scalar move (async_load_vec4, point 4), then the eight gathers.  The
        // destinations are early-clobber: a statement's outputs may otherwise share registers with its inputs, and these
        // are written (asynchronously) while later gathers of the statement still have their address to read.
        f32x4 t0, t1, t2, t3, t4, t5, t6, t7;
        sbase_t b;
        asm volatile("s_mov_b64 %8, %17\n\t"
                     "global_load_dwordx4 %0, %9, %8\n\tglobal_load_dwordx4 %1, %10, %8\n\t"
                     "global_load_dwordx4 %2, %11, %8\n\tglobal_load_dwordx4 %3, %12, %8\n\t"
                     "global_load_dwordx4 %4, %13, %8\n\tglobal_load_dwordx4 %5, %14, %8\n\t"
                     "global_load_dwordx4 %6, %15, %8\n\tglobal_load_dwordx4 %7, %16, %8"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&s"(b)
                     : "v"(k.o << 4), "v"(o10 << 4), "v"(o01 << 4), "v"(o11 << 4),
                       "v"((k.o + k.dz) << 4), "v"((o10 + k.dz) << 4), "v"((o01 + k.dz) << 4), "v"((o11 + k.dz) << 4), "s"(vbuf));
        c[0] = __builtin_bit_cast(float4, t0); c[1] = __builtin_bit_cast(float4, t1); c[2] = __builtin_bit_cast(float4, t2);
        c[3] = __builtin_bit_cast(float4, t3); c[4] = __builtin_bit_cast(float4, t4); c[5] = __builtin_bit_cast(float4, t5);
        c[6] = __builtin_bit_cast(float4, t6); c[7] = __builtin_bit_cast(float4, t7);
    }
    template <int YOUNGER>
    __device__ __forceinline__ void arrive() {
        f32x4 t[8];                                          // plain vector types (HIP's float4 class is a memory operand to asm); renames
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_bit_cast(f32x4, c[i]);
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]) : "n"(YOUNGER));
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_bit_cast(float4, t[i]);
    }
    template <bool STRICT>
    __device__ __forceinline__ void eval(const Cell& s, float& v, float g[3]) const {
        if constexpr (SHADE) {
            // (v, dx) and (dy, dz) blend as register pairs: 42 packed instructions instead of 84
#define MRIRT_LO(i) f32x2{ c[i].x, c[i].y }
#define MRIRT_HI(i) f32x2{ c[i].z, c[i].w }
            const f32x2 lo = trilerp2<STRICT>(MRIRT_LO(0), MRIRT_LO(1), MRIRT_LO(2), MRIRT_LO(3),
                                              MRIRT_LO(4), MRIRT_LO(5), MRIRT_LO(6), MRIRT_LO(7), s.fx, s.fy, s.fz);
            const f32x2 hi = trilerp2<STRICT>(MRIRT_HI(0), MRIRT_HI(1), MRIRT_HI(2), MRIRT_HI(3),
                                              MRIRT_HI(4), MRIRT_HI(5), MRIRT_HI(6), MRIRT_HI(7), s.fx, s.fy, s.fz);
#undef MRIRT_LO
#undef MRIRT_HI
            v = lo.x; g[0] = lo.y; g[1] = hi.x; g[2] = hi.y;
        } else {
            v = trilerp<STRICT>(c[0].x, c[1].x, c[2].x, c[3].x, c[4].x, c[5].x, c[6].x, c[7].x, s.fx, s.fy, s.fz);
        }
    }
};

template <bool SHADE> struct Taps<4, SHADE> : Taps<2, SHADE> {   // VGA: VG voxels, axis-flat bricks, copy chosen per wave
    using Taps<2, SHADE>::issue_async;
    template <bool WIDE>
    __device__ __forceinline__ void issue(const void* __restrict__ vbuf, const FlatAxis& f, const Cell& s) {
        const CellOffsets k = flat_cell(f, s.ix, s.iy, s.iz);
        const uint32_t o10 = k.o + k.dx, o01 = k.o + k.dy, o11 = o10 + k.dy;
        this->c[0] = load_vec4<false>(vbuf, k.o);        this->c[1] = load_vec4<false>(vbuf, o10);
        this->c[2] = load_vec4<false>(vbuf, o01);        this->c[3] = load_vec4<false>(vbuf, o11);
        this->c[4] = load_vec4<false>(vbuf, k.o + k.dz); this->c[5] = load_vec4<false>(vbuf, o10 + k.dz);
        this->c[6] = load_vec4<false>(vbuf, o01 + k.dz); this->c[7] = load_vec4<false>(vbuf, o11 + k.dz);
    }
};

// What a kernel hands Taps<LAYOUT>::issue as the grid description: the launch's GridDims, or (VGA) the copy this
// wave reads — with the buffer pointer moved to that copy.
template <int LAYOUT> struct WaveGrid {
    const GridDims* g;
    __device__ __forceinline__ const void* base(const void* vol) const { return vol; }
    __device__ __forceinline__ const GridDims& dims() const { return *g; }
};
template <> struct WaveGrid<4> {
    FlatAxis f;
    __device__ __forceinline__ const void* base(const void* vol) const { return static_cast<const char*>(vol) + f.baseBytes; }
    __device__ __forceinline__ const FlatAxis& dims() const { return f; }
};

// VGA: which copy this wave reads.  A ray that enters the box through a face of axis k keeps its samples of step
// n on the plane  p_k = face + n dt d_k  (t0 lies ON the face), so the wave's step-n samples form a sheet normal
// to k as long as its rays agree on the face: vote by lane.  Rays that start inside the box vote for their
// dominant direction.  Uniform result (scalar registers); any choice gives the same bits.
__device__ __forceinline__ int vga_pick_axis(const K1Args& a, const float ro[3], const float rd[3], bool marches) {
    float best = -INFINITY, bestDir = -1.0f;
    int axis = 0, axisDir = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = fabsf(rd[k]) < 1e-6f ? 1e-6f : rd[k];
        const float rcp = __builtin_amdgcn_rcpf(d);
        const float tn = fminf((a.bmin[k] - ro[k]) * rcp, (a.bmax[k] - ro[k]) * rcp);     // this slab's near plane
        if (tn > best) { best = tn; axis = k; }
        if (fabsf(rd[k]) > bestDir) { bestDir = fabsf(rd[k]); axisDir = k; }
    }
    if (!(best > 0.0f)) axis = axisDir;                                 // the eye is inside the box (or on a face)
    const uint32_t n0 = __builtin_popcountll(__ballot(marches && axis == 0));
    const uint32_t n1 = __builtin_popcountll(__ballot(marches && axis == 1));
    const uint32_t n2 = __builtin_popcountll(__ballot(marches && axis == 2));
    return n0 >= n1 && n0 >= n2 ? 0 : (n1 >= n2 ? 1 : 2);
}

template <> struct Taps<3, false> {                  // QUAD: the z and z+1 xy-quads (x0y0, x0y1, x1y0, x1y1)
    float4 q0, q1;
    template <bool WIDE>
    __device__ __forceinline__ void issue(const void* __restrict__ vbuf, const GridDims& gd, const Cell& s) {
        const CellOffsets k = vec4_cell(gd, s.ix, s.iy, s.iz);
        q0 = load_vec4<WIDE>(vbuf, k.o);
        q1 = load_vec4<WIDE>(vbuf, k.o + k.dz);
    }
    __device__ __forceinline__ void issue_async(const void* __restrict__ vbuf, const CellOffsets& k) {
        f32x4 t0, t1;                                   // one statement, the base copied once: see Taps<2>::issue_async
        sbase_t b;
        asm volatile("s_mov_b64 %2, %5\n\tglobal_load_dwordx4 %0, %3, %2\n\tglobal_load_dwordx4 %1, %4, %2"
                     : "=&v"(t0), "=&v"(t1), "=&s"(b) : "v"(k.o << 4), "v"((k.o + k.dz) << 4), "s"(vbuf));
        q0 = __builtin_bit_cast(float4, t0);
        q1 = __builtin_bit_cast(float4, t1);
    }
    template <bool STRICT>
    __device__ __forceinline__ void eval(const Cell& s, float& v, float*) const {
        // q = (c00, c01, c10, c11) [x then y]: the x blend of the y0 and y1 rows is one packed lerp per plane
        using Mm = M<STRICT>;
        const f32x2 r0 = lerp2<STRICT>(f32x2{ q0.x, q0.y }, f32x2{ q0.z, q0.w }, s.fx);
        const f32x2 r1 = lerp2<STRICT>(f32x2{ q1.x, q1.y }, f32x2{ q1.z, q1.w }, s.fx);
        v = Mm::lerp(Mm::lerp(r0.x, r0.y, s.fy), Mm::lerp(r1.x, r1.y, s.fy), s.fz);
    }
};

struct __attribute__((packed, aligned(4))) LinearPair { float a, b; };   // two x-neighbours of a LINEAR grid: one 8-byte gather

template <int LAYOUT, bool SHADE> struct TapsScalar {        // LINEAR / BRICK fp32 grids
    float c[8];
    float n[SHADE ? 24 : 1];                                 // +-1 neighbours of the 8 corners per axis
    template <bool WIDE>
    __device__ __forceinline__ void issue(const void* __restrict__ vbuf, const GridDims& gd, const Cell& s) {
        using A = Addr<LAYOUT>;
        const float* __restrict__ buf = static_cast<const float*>(vbuf);
        const uint32_t x0 = A::ox(gd, s.ix), x1 = A::ox(gd, s.ix + 1);
        const uint32_t y0 = A::oy(gd, s.iy), y1 = A::oy(gd, s.iy + 1);
        const uint32_t z0 = A::oz(gd, s.iz), z1 = A::oz(gd, s.iz + 1);
        // core 2x2x2 (sampleLinear, brats_rt.slang:69-72)
        if constexpr (LAYOUT == 0) {
            // the reference's own buffers (x fastest): the clamp dims - 1.001 keeps ix <= X - 2, so the x + 1 neighbour is the
            // next word — four 8-byte gathers (4-byte aligned: global memory takes them) instead of eight 4-byte ones
            const LinearPair p00 = *reinterpret_cast<const LinearPair*>(buf + (x0 + y0 + z0)), p10 = *reinterpret_cast<const LinearPair*>(buf + (x0 + y1 + z0));
            const LinearPair p01 = *reinterpret_cast<const LinearPair*>(buf + (x0 + y0 + z1)), p11 = *reinterpret_cast<const LinearPair*>(buf + (x0 + y1 + z1));
            c[0] = p00.a; c[1] = p00.b; c[2] = p10.a; c[3] = p10.b; c[4] = p01.a; c[5] = p01.b; c[6] = p11.a; c[7] = p11.b;
        } else {
            c[0] = buf[x0 + y0 + z0]; c[1] = buf[x1 + y0 + z0]; c[2] = buf[x0 + y1 + z0]; c[3] = buf[x1 + y1 + z0];
            c[4] = buf[x0 + y0 + z1]; c[5] = buf[x1 + y0 + z1]; c[6] = buf[x0 + y1 + z1]; c[7] = buf[x1 + y1 + z1];
        }
        if constexpr (SHADE) {
            const uint32_t xm = A::ox(gd, s.ix > 0 ? s.ix - 1 : 0), xp = A::ox(gd, min(s.ix + 2, gd.X - 1));
            const uint32_t ym = A::oy(gd, s.iy > 0 ? s.iy - 1 : 0), yp = A::oy(gd, min(s.iy + 2, gd.Y - 1));
            const uint32_t zm = A::oz(gd, s.iz > 0 ? s.iz - 1 : 0), zp = A::oz(gd, min(s.iz + 2, gd.Z - 1));
            n[0] = buf[xm + y0 + z0]; n[1] = buf[xp + y0 + z0]; n[2] = buf[xm + y1 + z0]; n[3] = buf[xp + y1 + z0];
            n[4] = buf[xm + y0 + z1]; n[5] = buf[xp + y0 + z1]; n[6] = buf[xm + y1 + z1]; n[7] = buf[xp + y1 + z1];
            n[8] = buf[x0 + ym + z0]; n[9] = buf[x1 + ym + z0]; n[10] = buf[x0 + yp + z0]; n[11] = buf[x1 + yp + z0];
            n[12] = buf[x0 + ym + z1]; n[13] = buf[x1 + ym + z1]; n[14] = buf[x0 + yp + z1]; n[15] = buf[x1 + yp + z1];
            n[16] = buf[x0 + y0 + zm]; n[17] = buf[x1 + y0 + zm]; n[18] = buf[x0 + y1 + zm]; n[19] = buf[x1 + y1 + zm];
            n[20] = buf[x0 + y0 + zp]; n[21] = buf[x1 + y0 + zp]; n[22] = buf[x0 + y1 + zp]; n[23] = buf[x1 + y1 + zp];
        }
    }
    template <bool STRICT>
    __device__ __forceinline__ void eval(const Cell& s, float& v, float g[3]) const {
        v = trilerp<STRICT>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], s.fx, s.fy, s.fz);
        if constexpr (SHADE) {
            // corner (0,dy,dz): v[i+1]-v[i-1]; corner (1,dy,dz): v[i+2]-v[i]; likewise in y and z
            g[0] = trilerp<STRICT>(c[1] - n[0], n[1] - c[0], c[3] - n[2], n[3] - c[2],
                                   c[5] - n[4], n[5] - c[4], c[7] - n[6], n[7] - c[6], s.fx, s.fy, s.fz);
            g[1] = trilerp<STRICT>(c[2] - n[8], c[3] - n[9], n[10] - c[0], n[11] - c[1],
                                   c[6] - n[12], c[7] - n[13], n[14] - c[4], n[15] - c[5], s.fx, s.fy, s.fz);
            g[2] = trilerp<STRICT>(c[4] - n[16], c[5] - n[17], c[6] - n[18], c[7] - n[19],
                                   n[20] - c[0], n[21] - c[1], n[22] - c[2], n[23] - c[3], s.fx, s.fy, s.fz);
        }
    }
};
template <> struct Taps<0, true> : TapsScalar<0, true> {};
// LINEAR, unshaded — the plain ABI's march on the reference's own buffers (mrirt_render_brats): the cell's four x-pairs
// (y, z) in {0, 1}^2 as 8-byte gathers, kept as register pairs so that the pipelined kernel can issue them as asm
// (async_load_pair) and retire a whole stage with one wait, like the float4 layouts.
__device__ __forceinline__ void async_load_pair(f32x2& dst, const void* __restrict__ base, uint32_t byteOff) {
    f32x2 t;
    sbase_t b;                                        // (the base through a scalar move: see async_load_vec4, point 4)
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx2 %0, %2, %1" : "=v"(t), "=&s"(b) : "v"(byteOff), "s"(base));
    dst = t;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void async_load_words2(u32x2& dst, const void* __restrict__ base, uint32_t byteOff) {     // (label cells)
    u32x2 t;
    sbase_t b;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx2 %0, %2, %1" : "=v"(t), "=&s"(b) : "v"(byteOff), "s"(base));
    dst = t;
}
template <> struct Taps<0, false> {
    f32x2 p[4];                                      // (x0, x1) at (y0, z0), (y1, z0), (y0, z1), (y1, z1)
    template <bool WIDE>
    __device__ __forceinline__ void issue(const void* __restrict__ vbuf, const GridDims& gd, const Cell& s) {
        const float* __restrict__ buf = static_cast<const float*>(vbuf);
        const uint32_t o = s.ix + s.iy * gd.sY + s.iz * gd.sZ;
        const uint32_t off[4] = { o, o + gd.sY, o + gd.sZ, o + gd.sY + gd.sZ };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const LinearPair q = *reinterpret_cast<const LinearPair*>(buf + off[i]);
            p[i] = f32x2{ q.a, q.b };
        }
    }
    __device__ __forceinline__ void issue_async(const void* __restrict__ vbuf, const CellOffsets& k) {   // k in ELEMENTS (floats); grid < 4 GiB
        f32x2 t0, t1, t2, t3;                           // one statement, the base copied once: see Taps<2>::issue_async
        sbase_t b;
        asm volatile("s_mov_b64 %4, %9\n\tglobal_load_dwordx2 %0, %5, %4\n\tglobal_load_dwordx2 %1, %6, %4\n\t"
                     "global_load_dwordx2 %2, %7, %4\n\tglobal_load_dwordx2 %3, %8, %4"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(b)
                     : "v"(k.o << 2), "v"((k.o + k.dy) << 2), "v"((k.o + k.dz) << 2), "v"((k.o + k.dy + k.dz) << 2), "s"(vbuf));
        p[0] = t0; p[1] = t1; p[2] = t2; p[3] = t3;
    }
    template <bool STRICT>
    __device__ __forceinline__ void eval(const Cell& s, float& v, float*) const {
        v = trilerp<STRICT>(p[0].x, p[0].y, p[1].x, p[1].y, p[2].x, p[2].y, p[3].x, p[3].y, s.fx, s.fy, s.fz);
    }
};
template <bool SHADE> struct Taps<1, SHADE> : TapsScalar<1, SHADE> {};

__device__ __forceinline__ uint32_t sample_label(const uint32_t* __restrict__ buf, const LabelAddr& la,
                                                 const float q[3], const float hi[3]) {
    // sampleLabel, brats_rt.slang:78-83; roundf = half away from zero (Metal round)
    const uint32_t ix = (uint32_t)roundf(clampf(q[0], 0.0f, hi[0]));
    const uint32_t iy = (uint32_t)roundf(clampf(q[1], 0.0f, hi[1]));
    const uint32_t iz = (uint32_t)roundf(clampf(q[2], 0.0f, hi[2]));
    return buf[la.off(ix, iy, iz)];
}

// ---------------------------------------------------------------------------------------
// Everything after the blended intensity v (and weighted gradient g) is known:
// transfer function, emission-absorption step, label overlays.  brats_rt.slang:130-162.
// ---------------------------------------------------------------------------------------
struct RayState { float C0, C1, C2, T; uint32_t nLive, nShaded; };

// the two nearest-label gathers of a sample (issued with the intensity gathers, consumed in composite)
struct Labels { uint32_t seg, pred; };
// LABCELL: which nibble of the cell's label words is sampleLabel's voxel — round(clamp(q, 0, d-1)) is the cell's base index
// or one more, per axis (include/mrirt.h) — as a shift count
__device__ __forceinline__ uint32_t label_corner_shift(const K1Args& a, const Cell& s) {
    const uint32_t rx = (uint32_t)roundf(clampf(s.q[0], 0.0f, a.hiLab[0])) - s.ix;
    const uint32_t ry = (uint32_t)roundf(clampf(s.q[1], 0.0f, a.hiLab[1])) - s.iy;
    const uint32_t rz = (uint32_t)roundf(clampf(s.q[2], 0.0f, a.hiLab[2])) - s.iz;
    return (rx + 2u * ry + 4u * rz) << 2;
}
__device__ __forceinline__ void labels_from_cell(const K1Args& a, uint32_t segWord, uint32_t predWord, uint32_t shift, Labels& l) {
    l.seg = a.showSeg != 0 ? (segWord >> shift) & 15u : 0u;
    l.pred = a.showPred != 0 ? (predWord >> shift) & 15u : 0u;
}
__device__ __forceinline__ void fetch_labels(const K1Args& a, const Cell& s, Labels& l) {
    if (a.labCell != nullptr) {                                                      // uniform
        const uint2 w = a.labCell[AddrVec4::ox(a.grid, s.ix) + AddrVec4::oy(a.grid, s.iy) + AddrVec4::oz(a.grid, s.iz)];
        labels_from_cell(a, w.x, w.y, label_corner_shift(a, s), l);
        return;
    }
    l.seg = a.showSeg != 0 ? sample_label(a.labels, a.lab, s.q, a.hiLab) : 0u;      // :144
    l.pred = a.showPred != 0 ? sample_label(a.preds, a.lab, s.q, a.hiLab) : 0u;    // :155
}
// ... and with the prediction label taken from C5's class stream (one class per sample of the ray; generic kernel only)
__device__ __forceinline__ void fetch_labels_stream(const K1Args& a, const Cell& s, Labels& l, int64_t streamRow) {
    if (a.labCell != nullptr && a.classStream == nullptr) { fetch_labels(a, s, l); return; }
    l.seg = a.showSeg != 0 ? sample_label(a.labels, a.lab, s.q, a.hiLab) : 0u;
    if (a.showPred == 0) l.pred = 0u;
    else if (a.classStream != nullptr) l.pred = (uint32_t)(uint16_t)a.classStream[streamRow];
    else l.pred = sample_label(a.preds, a.lab, s.q, a.hiLab);
}

// GAMMA1: gamma == 1 (the viewer's constant, brats_viewer.py:422), where pow(val, 1) == val exactly;
// compiling the fp64 pow out of the hot kernels frees the registers its temporaries would claim.
// The overlays' per-label constants (lutColorAlpha[l].rgb and the two opacities) are indexed by a per-lane label.  Straight
// from the kernel arguments that is a VECTOR-memory load from the kernarg segment inside a divergent branch — and a load
// the compiler cannot count past: every s_waitcnt vmcnt of the march then drops to 0 and the software pipeline (the next
// step's gathers in flight under this step's compositing) is gone (seen in the ISA of the config-2 kernel; VERDICT r2 #5 had
// it "latency-bound").  So every kernel that draws overlays copies the sixteen rows into LDS once per workgroup
// (stage_lut) and reads them with ds_read_b128: lgkmcnt, not vmcnt.  Rows 0-7: (rgb, segAlpha), rows 8-15: (rgb, predAlpha).
__device__ __forceinline__ const float4* stage_lut(const K1Args& a, float4* lutS) {
    if (threadIdx.x < 16u) {
        const uint32_t l = threadIdx.x & 7u;
        lutS[threadIdx.x] = make_float4(a.lut[l][0], a.lut[l][1], a.lut[l][2], threadIdx.x < 8u ? a.segAlpha[l] : a.predAlpha[l]);
    }
    __syncthreads();
    return lutS;
}

template <bool STRICT, bool SHADE, bool GAMMA1 = false, bool LABELS = true>
__device__ __forceinline__ void composite(const K1Args& a, const float rd[3], const Labels& lb, float v, const float g[3],
                                          RayState& r, const float4* lutS = nullptr) {
    using Mm = M<STRICT>;
    // wSum (brats_rt.slang:123-130) is the same for every sample: summed on the host
    if (a.wsum.d > 0.0f && a.wsum.d != 1.0f) v = Mm::divu_data(v, a.wsum);      // x / 1 == x: skip the three instructions
    float val = satf(Mm::divu_data(v - a.tfLo, a.wwDiv));                 // :132
    if constexpr (!GAMMA1) val = Mm::pow(val, a.gamma);              // :133
    ++r.nLive;
    if (val > 0.0f) {
        const float ex = -(val * a.intensityAlpha) * a.stepSize;
        float e;
        // constants: SGPR operands in the lean kernels; literals where the shading / overlay state already fills
        // the SGPR file (the 32 extra SGPRs spill there: measured +6 % on the 4-modality + overlay frame)
        constexpr bool LIT = SHADE || LABELS;
        if (a.expSmall) e = LIT ? Mm::exp_small_lit(ex) : Mm::exp_small(ex, a.ec);        // uniform: |ex| <= 1/8 for every sample
        else            e = LIT ? Mm::exp_lit(ex) : Mm::exp(ex, a.ec);
        const float alpha = 1.0f - e;
        float emis = val;
        if constexpr (SHADE) {
            // headlight Blinn-Phong on the lattice gradient (build-defined extension):
            // world gradient = index-space difference * 0.5/voxelSize; n.l = |g.d| / |g|
            const float gx = g[0] * a.halfInvVoxel[0], gy = g[1] * a.halfInvVoxel[1], gz = g[2] * a.halfInvVoxel[2];
            const float len2 = dot3(gx, gy, gz, gx, gy, gz);
            const float glen = STRICT ? sqrtf(len2) : __builtin_amdgcn_sqrtf(len2);
            float shade = a.ka + a.kd;
            if (glen > a.gradEps) {
                const float gd = fabsf(dot3(gx, gy, gz, rd[0], rd[1], rd[2]));
                const float ndl = fminf(STRICT ? gd / glen : gd * __builtin_amdgcn_rcpf(glen), 1.0f);
                float spec = ndl;
                for (uint32_t k = 0; k < a.specPow2; ++k) spec = spec * spec;
                shade = (a.ka + a.kd * ndl) + a.ks * spec;
            }
            emis = val * shade;
            ++r.nShaded;
        }
        const float c = (alpha * r.T) * emis;
        r.C0 += c; r.C1 += c; r.C2 += c;
        r.T *= (1.0f - alpha);
    }
    if (LABELS && a.showSeg != 0) {                                  // :143-151
        const uint32_t l = lb.seg;
        if (l > 0 && l < 8) {
            const float4 e = lutS[l];                                    // (rgb, 1 - exp(-lut[l].w * dt)): host-made, LDS-resident
            const float alpha = e.w;
            const float at = alpha * r.T;
            r.C0 += at * e.x; r.C1 += at * e.y; r.C2 += at * e.z;
            r.T *= (1.0f - alpha);
        }
    }
    if (LABELS && a.showPred != 0) {                                 // :154-162
        const uint32_t l = lb.pred;
        if (l > 0 && l < 8) {
            const float4 e = lutS[8u + l];                               // (rgb, 1 - exp(-lut[l].w * dt * 1.5))
            const float alpha = e.w;
            const float at = alpha * r.T;
            r.C0 += at * e.x; r.C1 += at * e.y; r.C2 += at * e.z;
            r.T *= (1.0f - alpha);
        }
    }
}

// ray generation + slab clip (brats_rt.slang:91-109); returns whether the ray marches
__device__ __forceinline__ bool setup_ray(const K1Args& a, uint32_t px, uint32_t py, float ro[3], float rd[3],
                                          float& t0, float& t1) {
    primary_ray(a.cam, px, py, ro, rd);
    float tmin = -INFINITY, tmax = INFINITY;   // rcp uses the nudged direction, marching the true one
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = fabsf(rd[k]) < 1e-6f ? 1e-6f : rd[k];
        const float rcp = 1.0f / d;
        const float ta = (a.bmin[k] - ro[k]) * rcp, tb = (a.bmax[k] - ro[k]) * rcp;
        tmin = fmaxf(tmin, fminf(ta, tb));
        tmax = fminf(tmax, fmaxf(ta, tb));
    }
    const bool hit = tmax >= fmaxf(tmin, 0.0f);
    t0 = fmaxf(tmin, fmaxf(0.0f, a.nearT));
    t1 = fminf(tmax, a.farT > 0.0f ? a.farT : tmax);
    return hit && !(t1 <= t0);
}

__device__ __forceinline__ void finish(const K1Args& a, int kind, int64_t oidx, const RayState& r) {
    if (kind != 0) {
        if (a.half) store_rgba<true>(a.out, oidx, r.C0, r.C1, r.C2, 1.0f);
        else        store_rgba<false>(a.out, oidx, r.C0, r.C1, r.C2, 1.0f);
    }
    if (a.stats != nullptr) {
        wave_count_add(a.stats + 0, r.nLive);
        wave_count_add(a.stats + 1, r.nShaded);
    }
}

// brats_slab.hip: the LDS-staged march (VGA layout, one modality, no overlays), selected by brats_march.hip
int launch_slab_march(const K1Args& a, bool strict, bool shade, hipStream_t s);
// brats_ring.hip: the plane-synchronous LDS ring march (same launches)
int launch_ring_march(const K1Args& a, bool strict, bool shade, hipStream_t s);

}  // namespace mrirt
