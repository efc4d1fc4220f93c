// Device-side helpers shared by the gfx950 kernels: fp32 math in two flavours, ray generation,
// brick addressing, block->pixel mapping.  Compiled with -ffp-contract=off so that the STRICT
// flavour is unfused IEEE fp32 in the order the reference shaders write it; the FAST flavour
// asks for FMA explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrirt {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kTilePx = 8;         // one wave = 8x8 pixels (the reference's numthreads(8,8,1) group)
constexpr int kBlockPx = 16;       // one 256-thread workgroup = 2x2 waves = 16x16 pixels
constexpr int kXcds = 8;

// Pure index arithmetic (which element / byte / pixel an id maps to) is compiled for the host as well: the kernels use it as
// always, and tests/native/index_harness.hip walks the same functions on the CPU under AddressSanitizer + UBSan against
// buffers of exactly the sizes the host wrappers allocate (GPU sanitizers are not available; VERDICT r3 #1b).
#define MRIRT_HD __host__ __device__ __forceinline__
MRIRT_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);                       // both factors < 2^24 (callers' contract): v_mul_u32_u24
#else
    return a * b;
#endif
}

// ---------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------
template <bool STRICT> struct M;

// A per-launch constant divisor d with its correctly rounded reciprocal r = RN(1/d) (host).
// exact == 0 marks the one case Markstein's theorem excludes (significand of d all ones).
struct UDiv { float d, r; uint32_t exact; };

// Correctly rounded fp32 exp via fp64: 2^k * P13(x - k ln2).  Same "round a <1-ulp fp64 value
// once" contract as (float)exp((double)x) on the host, at a third of OCML's instruction count
// (the argument here is -(val*alpha)*dt: no need for the full double domain).
// The sixteen fp64 constants come in through the kernel arguments (ExpConsts, filled by the host), i.e. they
// sit in SGPR pairs and feed v_fma_f64 directly: as literals each one costs a v_mov_b64 into a VGPR pair
// per use (a 64-bit literal cannot be an operand), ten extra VALU instructions per exp in the march loop.
struct ExpConsts { double log2e, ln2hi, ln2lo, c[13]; };

__device__ __forceinline__ float exp_f64_to_f32(float xf, const ExpConsts& e) {
    double x = fmin(fmax((double)xf, -200.0), 100.0);        // fp32 result is 0 / inf outside anyway
    const double k = __builtin_rint(x * e.log2e);
    double r = __builtin_fma(-k, e.ln2hi, x);                // ln2 hi / lo split
    r = __builtin_fma(-k, e.ln2lo, r);
    double p = e.c[0];                                       // 1/13!
#pragma unroll
    for (int i = 1; i < 13; ++i) p = __builtin_fma(p, r, e.c[i]);      // ... 1/12!, ..., 1/2!, 1
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, (int)k);
}

// |x| <= 1/8 (the host checks intensityAlpha * stepSize, and val <= 1): no range reduction (k = 0) and the
// series to x^10 — the first dropped term, x^11/11! <= 3e-18, is 30x below half an ulp of the fp64 result,
// the same accuracy class as the full form — in 12 instructions instead of 25.
__device__ __forceinline__ float exp_small_f64_to_f32(float xf, const ExpConsts& e) {
    const double x = (double)xf;
    double p = e.c[3];                                       // 1/10!
#pragma unroll
    for (int i = 4; i < 13; ++i) p = __builtin_fma(p, x, e.c[i]);
    p = __builtin_fma(p, x, 1.0);
    return (float)p;
}
__device__ __forceinline__ float exp_small_f64_to_f32(float xf) {
    const double x = (double)xf;
    double p = 2.755731922398589e-07;
    p = __builtin_fma(p, x, 2.7557319223985893e-06);
    p = __builtin_fma(p, x, 2.48015873015873e-05);
    p = __builtin_fma(p, x, 1.984126984126984e-04);
    p = __builtin_fma(p, x, 1.3888888888888889e-03);
    p = __builtin_fma(p, x, 8.333333333333333e-03);
    p = __builtin_fma(p, x, 4.1666666666666664e-02);
    p = __builtin_fma(p, x, 1.6666666666666666e-01);
    p = __builtin_fma(p, x, 0.5);
    p = __builtin_fma(p, x, 1.0);
    p = __builtin_fma(p, x, 1.0);
    return (float)p;
}

// The same polynomial with literal constants (identical values, identical result).  Kept for the
// gradient-shaded kernel: measured 2-3 % faster there with literals (its time is set by the L1 tag
// pipeline, and the SGPR operands change the schedule for the worse), 2-3 % slower everywhere else.
__device__ __forceinline__ float exp_f64_to_f32(float xf) {
    double x = fmin(fmax((double)xf, -200.0), 100.0);
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.3888888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, (int)k);
}

template <> struct M<true> {
    // bit-faithful to oracle_c.c / oracle_np.py: unfused fp32 in the written order; divisions
    // and exp are correctly rounded by construction
    static __device__ __forceinline__ float lerp(float a, float b, float t) { return a + t * (b - a); }
    static __device__ __forceinline__ float exp(float x, const ExpConsts& e) { return exp_f64_to_f32(x, e); }
    static __device__ __forceinline__ float exp_lit(float x) { return exp_f64_to_f32(x); }
    static __device__ __forceinline__ float exp_small(float x, const ExpConsts& e) { return exp_small_f64_to_f32(x, e); }
    static __device__ __forceinline__ float exp_small_lit(float x) { return exp_small_f64_to_f32(x); }
    // x / u.d, IEEE-exact in 3 instructions (Markstein: q = RN(x r); e = x - q d exactly by FMA;
    // RN(q + e r) is the correctly rounded quotient when r = RN(1/d))
    static __device__ __forceinline__ float divu(float x, const UDiv& u) {
        if (!u.exact) return x / u.d;
        const float q = x * u.r;
        const float e = __builtin_fmaf(-q, u.d, x);
        return __builtin_fmaf(e, u.r, q);
    }
    // The same for numerators that come from VOXEL DATA (the weighted intensity, the window test, the z-score): a volume may
    // hold +-inf or NaN, and Markstein's residual is then inf - inf = NaN where the IEEE quotient is +-inf — saturate() would
    // turn that into 0 instead of 1.  q = x r already IS the IEEE quotient whenever it is not finite (inf
    // for an infinite x with the quotient's sign, NaN for a NaN), so it is returned as is: one compare and one select.
    // (Domain note: a FINITE x whose product with r overflows where x / d itself would round to the largest float is not
    // distinguished — |x / d| >= 2^127, forty orders of magnitude beyond anything a window test can tell apart.)
    static __device__ __forceinline__ float divu_data(float x, const UDiv& u) {
        if (!u.exact) return x / u.d;
        const float q = x * u.r;
        const float e = __builtin_fmaf(-q, u.d, x);
        const float res = __builtin_fmaf(e, u.r, q);
        return __builtin_fabsf(q) < INFINITY ? res : q;
    }
    static __device__ __forceinline__ float pow(float x, float y) {
        return y == 1.0f ? x : (float)::pow((double)x, (double)y);   // pow(x,1) == x exactly
    }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }  // IEEE (hipcc default)
    static __device__ __forceinline__ float mad(float a, float b, float c) { return a * b + c; }
};

template <> struct M<false> {
    static __device__ __forceinline__ float divu(float x, const UDiv& u) { return x * u.r; }
    static __device__ __forceinline__ float divu_data(float x, const UDiv& u) { return x * u.r; }
    static __device__ __forceinline__ float lerp(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
    static __device__ __forceinline__ float exp(float x, const ExpConsts&) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
    static __device__ __forceinline__ float exp_lit(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
    static __device__ __forceinline__ float exp_small(float x, const ExpConsts&) { return exp_lit(x); }
    static __device__ __forceinline__ float exp_small_lit(float x) { return exp_lit(x); }
    static __device__ __forceinline__ float pow(float x, float y) {
        return y == 1.0f ? x : (x > 0.0f ? __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)) : 0.0f);
    }
    static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static __device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};

// v_med3_f32: one instruction; with a NaN input it returns min3 of the others, i.e. lo — the same
// result as fminf(fmaxf(NaN, lo), hi) and as HLSL saturate(NaN) = 0 (requires lo <= hi).
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float satf(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
    float n = sqrtf(dot3(x, y, z, x, y, z));
    x = x / n; y = y / n; z = z / n;
}

// ---------------------------------------------------------------------------------------
// camera block shared by the three kernels (host fills it; invTanHalf/tanHalf are computed
// on the host as (float)tan((double)(0.5f*fovY)) so both flavours agree with the oracle)
// ---------------------------------------------------------------------------------------
struct Camera {
    float eye[3], U[3], V[3], W[3];
    float invTanHalf;      // 1 / tan(0.5 fovY)            (K1, K3)
    float tanHalf;         // tan(0.5 fovY)                (K2)
    float aspect;          // W / max(1,H)   (K3: W / H)
    float orthoHalfHeight;
    uint32_t mode;         // 0 perspective, 1 orthographic
    uint32_t width, height;
};

// Perspective ray of brats_rt.slang:36-46 / raymarch.slang:45-58 (or the orthographic extension).
__device__ __forceinline__ void primary_ray(const Camera& c, uint32_t px, uint32_t py,
                                            float ro[3], float rd[3]) {
    float dimx = (float)c.width, dimy = (float)c.height;
    float uvx = (((float)px + 0.5f) / dimx) * 2.0f - 1.0f;
    float uvy = (((float)py + 0.5f) / dimy) * 2.0f - 1.0f;
    if (c.mode == 0) {
        float f = c.invTanHalf;
        float cx = uvx * c.aspect / f, cy = -uvy / f, cz = 1.0f;
        normalize3(cx, cy, cz);
#pragma unroll
        for (int k = 0; k < 3; ++k) { ro[k] = c.eye[k]; rd[k] = (cx * c.U[k] + cy * c.V[k]) + cz * c.W[k]; }
        normalize3(rd[0], rd[1], rd[2]);
    } else {
        float sx = uvx * c.orthoHalfHeight * c.aspect, sy = -uvy * c.orthoHalfHeight;
#pragma unroll
        for (int k = 0; k < 3; ++k) { ro[k] = (c.eye[k] + c.U[k] * sx) + c.V[k] * sy; rd[k] = c.W[k]; }
    }
}

// ---------------------------------------------------------------------------------------
// grid addressing.  Both layouts are separable: off(x,y,z) = ox(x) + oy(y) + oz(z).
// BRICK: 4x4x2-voxel bricks (32 elements = one 128-B line of fp32), bricks in x-fastest order.
// ---------------------------------------------------------------------------------------
struct GridDims {
    uint32_t X, Y, Z;
    uint32_t sY, sZ;       // LINEAR: X, X*Y.          BRICK: NBX*32, NBX*NBY*32
    uint32_t wide;         // VG / QUAD: grid is >= 4 GiB, byte offsets need 64 bits
};

// VG / QUAD gathers: byte offsets of the 2x2x2 cell's corners.  The cell origin costs one
// ox+oy+oz; its +1 neighbours are a per-axis delta (inside the brick, or across to the next one),
// so the other seven corners are seven adds — and while the grid is < 4 GiB the offsets stay 32-bit
// and go out as  global_load_dwordx4 v, voff, s[base]  with no 64-bit address arithmetic at all.
struct CellOffsets {
    uint32_t o, dx, dy, dz;    // float4 units
};
MRIRT_HD CellOffsets vec4_cell(const GridDims& g, uint32_t ix, uint32_t iy, uint32_t iz) {
    const uint32_t bx = ix & 1u, by = iy & 1u, bz = iz & 1u;
    CellOffsets c;
    c.o = (((ix >> 1) << 3) + bx) + ((iy >> 1) * g.sY + (by << 1)) + ((iz >> 1) * g.sZ + (bz << 2));
    c.dx = bx ? 7u : 1u;                 // next brick (+8) and back to its x = 0 lane (-1)
    c.dy = by ? g.sY - 2u : 2u;
    c.dz = bz ? g.sZ - 4u : 4u;
    return c;
}
// VGA ("VG, axis-flat"): the same float4 (v, dx, dy, dz) voxels stored THREE times, in 128-B bricks that are one
// voxel thick along x, y or z (1x4x2, 4x1x2, 4x2x1).  The samples a wave takes at one march step lie on a sheet
// parallel to the face its rays entered through (t = t0 + k dt with t0 ON that face), so the copy whose bricks are
// flat in that direction turns every line a gather touches into eight useful voxels instead of four: fewer distinct
// lines per wave-level gather, which is what the vector L1's tag pipeline (one look-up per line per clock) charges
// for.  The wave picks its copy once per packet (entry face of most of its rays); all copies hold the same bits.
// Separable like the other layouts: element = base + sum over axes of (i >> sh) * mul + (i & mask) * inner.
struct FlatAxis {
    uint32_t sh[3], mask[3], inner[3], mul[3];
    uint32_t wrap[3];          // step to the +1 neighbour from the last slot of a brick: mul - mask * inner
    uint32_t pad;
    uint64_t baseBytes;        // byte offset of this copy inside the grid buffer (each copy < 4 GiB)
};
struct VgaDims { FlatAxis ax[3]; };

MRIRT_HD CellOffsets flat_cell(const FlatAxis& f, uint32_t ix, uint32_t iy, uint32_t iz) {
    const uint32_t i[3] = { ix, iy, iz };
    uint32_t o = 0, d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t lo = i[k] & f.mask[k];
        o += mul24(i[k] >> f.sh[k], f.mul[k]) + mul24(lo, f.inner[k]);      // < 2^24 each factor (dims <= 2^13)
        d[k] = lo == f.mask[k] ? f.wrap[k] : f.inner[k];
    }
    CellOffsets c;
    c.o = o; c.dx = d[0]; c.dy = d[1]; c.dz = d[2];
    return c;
}

template <bool WIDE> __device__ __forceinline__ float4 load_vec4(const void* __restrict__ base, uint32_t elem) {
    const char* b = static_cast<const char*>(base);
    if constexpr (WIDE) return *reinterpret_cast<const float4*>(b + ((uint64_t)elem << 4));
    else return *reinterpret_cast<const float4*>(b + (uint32_t)(elem << 4));
}

// two fp32 lanes per instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): the x-y and z-w
// halves of a float4 tap are already adjacent registers, so blending them as pairs needs no moves
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool STRICT> __device__ __forceinline__ f32x2 lerp2(f32x2 a, f32x2 b, float t) {
    const f32x2 tt = { t, t };
    if constexpr (STRICT) return a + tt * (b - a);               // unfused, element-wise as M<true>::lerp
    else return __builtin_elementwise_fma(tt, b - a, a);
}
template <bool STRICT>
__device__ __forceinline__ f32x2 trilerp2(f32x2 c000, f32x2 c100, f32x2 c010, f32x2 c110,
                                          f32x2 c001, f32x2 c101, f32x2 c011, f32x2 c111,
                                          float fx, float fy, float fz) {
    return lerp2<STRICT>(lerp2<STRICT>(lerp2<STRICT>(c000, c100, fx), lerp2<STRICT>(c010, c110, fx), fy),
                         lerp2<STRICT>(lerp2<STRICT>(c001, c101, fx), lerp2<STRICT>(c011, c111, fx), fy), fz);
}

template <int LAYOUT> struct Addr;
template <> struct Addr<0> {
    static MRIRT_HD uint32_t ox(const GridDims&, uint32_t x) { return x; }
    static MRIRT_HD uint32_t oy(const GridDims& g, uint32_t y) { return y * g.sY; }
    static MRIRT_HD uint32_t oz(const GridDims& g, uint32_t z) { return z * g.sZ; }
};
template <> struct Addr<1> {
    static MRIRT_HD uint32_t ox(const GridDims&, uint32_t x) { return ((x >> 2) << 5) + (x & 3u); }
    static MRIRT_HD uint32_t oy(const GridDims& g, uint32_t y) { return (y >> 2) * g.sY + ((y & 3u) << 2); }
    static MRIRT_HD uint32_t oz(const GridDims& g, uint32_t z) { return (z >> 1) * g.sZ + ((z & 1u) << 4); }
};
// VG / QUAD: one float4 per voxel, 2x2x2-voxel bricks (8 x 16 B = one 128-B line); offsets in
// float4 units; sY = NBX*8, sZ = NBX*NBY*8 with NB* = ceil(dim/2).
struct AddrVec4 {
    static MRIRT_HD uint32_t ox(const GridDims&, uint32_t x) { return ((x >> 1) << 3) + (x & 1u); }
    static MRIRT_HD uint32_t oy(const GridDims& g, uint32_t y) { return (y >> 1) * g.sY + ((y & 1u) << 1); }
    static MRIRT_HD uint32_t oz(const GridDims& g, uint32_t z) { return (z >> 1) * g.sZ + ((z & 1u) << 2); }
};
template <> struct Addr<2> : AddrVec4 {};
template <> struct Addr<3> : AddrVec4 {};

// Label grids (1-2 nearest fetches per sample) take their layout at run time: a separable
// shift/mask form that covers LINEAR (shift 0, mask 0) and BRICK.
struct LabelAddr {
    uint32_t sh[3], mask[3], inner[3], mul[3];
    MRIRT_HD uint32_t off(uint32_t x, uint32_t y, uint32_t z) const {
        return ((x >> sh[0]) * mul[0] + (x & mask[0]) * inner[0]) +
               ((y >> sh[1]) * mul[1] + (y & mask[1]) * inner[1]) +
               ((z >> sh[2]) * mul[2] + (z & mask[2]) * inner[2]);
    }
};

// ---------------------------------------------------------------------------------------
// workgroup -> pixel mapping, XCD-aware, with optional tile sharding
// ---------------------------------------------------------------------------------------
struct PixelMap {
    uint32_t width, height;
    uint32_t blocksX, numBlocks;       // 16x16-pixel workgroups over the part of the image this call renders
    uint32_t chunk;                    // ceil(numBlocks / 8): logical ids one XCD owns
    uint32_t tileSize, tileRank, tileWorld, tilesX;   // tileSize == 0: whole frame
    uint32_t tileSkew;                 // dealt tile t -> column (t % tilesX + tileSkew * row) % tilesX (MrirtRenderExt::tileSkew)
    uint32_t laneOrder;                // 0 row-major 8x8 packet, 1 Morton (2x2 pixel quads per 4 lanes)
    uint32_t blockPx;                  // 16: 256-thread workgroups (2x2 packets); 8: one packet per workgroup
    uint32_t bandBlocks;               // whole-frame XCD interleave: workgroups per band (0 = contiguous runs)
    uint32_t bandShift;                // whole frame: an XCD's j-th band starts bandShift * j workgroups further along x (mod blocksX)
    int64_t  pitch;                    // pixels per output row (whole-frame mode)
};

// Workgroups are dealt round-robin over the 8 XCDs (b % 8 = XCD group), so logical id
// (b % 8) * chunk + b / 8 hands each XCD's L2 a contiguous run of neighbouring ray packets.
// Grid is chunk*8 workgroups.  Returns 0: this lane has no pixel (whole workgroups past
// numBlocks, or whole-frame lanes beyond the image edge); 1: march pixel (px,py) and store at
// outIndex; 2: (tile mode) a compact-buffer slot outside the image: store background only.
// tile sharding: where the dealt tile t sits in the image, and which dealt index a tile position has (inverse).
// `skew` is the host's MrirtRenderExt::tileSkew reduced modulo tilesX (fill_pixel_map, mrirt_detile) and tile rows number < 2^16
// (images < 2^20 px, tiles >= 16 px), so skew * ty stays in 32 bits.
MRIRT_HD void tile_position(uint32_t t, uint32_t tilesX, uint32_t skew, uint32_t& tx, uint32_t& ty) {
    ty = t / tilesX;
    tx = (t % tilesX + (skew * ty) % tilesX) % tilesX;
}
MRIRT_HD uint32_t tile_dealt_index(uint32_t tx, uint32_t ty, uint32_t tilesX, uint32_t skew) {
    return ty * tilesX + (tx + tilesX - (skew * ty) % tilesX) % tilesX;
}

MRIRT_HD int map_pixel_at(const PixelMap& m, uint32_t b, uint32_t tid, uint32_t& px, uint32_t& py, int64_t& outIndex) {
    uint32_t logical;
    if (m.bandBlocks != 0) {
        // whole frame: horizontal bands of bandBlocks workgroups (a few packet rows) dealt round-robin
        // to the XCDs — XCD x marches bands x, x+8, x+16, ... so the mostly-empty top and bottom of
        // the image are shared out, while each band is still a contiguous slab for one L2
        const uint32_t xcd = b % kXcds, k = b / kXcds;
        const uint32_t j = k / m.bandBlocks, band = j * kXcds + xcd;
        uint32_t p = k % m.bandBlocks;
        // Inside an XCD the dispatcher deals consecutive workgroups over CUs and SIMDs, so workgroups a whole band apart land
        // on the same SIMD: without a shift a SIMD marches the SAME image column in every band it gets — all long centre
        // rays on one SIMD, all short edge rays on its neighbour (config 2: the frame took as long as the centre-column
        // SIMDs).  Starting each band 3/8 of a row further along spreads every SIMD's packets over the width of the image.
        if (m.bandShift != 0) { const uint32_t row = p / m.blocksX, x = (p % m.blocksX + j * m.bandShift) % m.blocksX; p = row * m.blocksX + x; }
        logical = band * m.bandBlocks + p;
    } else {
        logical = (b % kXcds) * m.chunk + b / kXcds;     // one contiguous run per XCD
    }
    if (logical >= m.numBlocks) return 0;
    uint32_t wave = tid >> 6, lane = tid & 63u;
    uint32_t lx, ly;
    if (m.laneOrder == 0) {            // row-major 8x8: lane = x + 8y (the reference's thread group)
        lx = lane & 7u; ly = lane >> 3;
    } else {                           // Morton: every aligned group of 4 lanes is a 2x2 pixel quad
        lx = (lane & 1u) | ((lane >> 1) & 2u) | ((lane >> 2) & 4u);
        ly = ((lane >> 1) & 1u) | ((lane >> 2) & 2u) | ((lane >> 3) & 4u);
    }
    lx += (wave & 1u) << 3; ly += (wave >> 1) << 3;
    if (m.tileSize == 0) {
        uint32_t bx = logical % m.blocksX, by = logical / m.blocksX;
        px = bx * m.blockPx + lx; py = by * m.blockPx + ly;
        outIndex = (int64_t)py * m.pitch + px;
    } else {
        uint32_t bpr = m.tileSize / m.blockPx, bpt = bpr * bpr;
        uint32_t lt = logical / bpt, sb = logical % bpt;
        uint32_t t = m.tileRank + lt * m.tileWorld;
        uint32_t tx, ty;
        tile_position(t, m.tilesX, m.tileSkew, tx, ty);
        uint32_t ix = (sb % bpr) * m.blockPx + lx, iy = (sb / bpr) * m.blockPx + ly;
        px = tx * m.tileSize + ix; py = ty * m.tileSize + iy;
        outIndex = ((int64_t)lt * m.tileSize + iy) * m.tileSize + ix;
        // slots of an edge tile that fall outside the image are written as background so the
        // compact buffer is fully defined; de-tiling drops them.
        return (px < m.width && py < m.height) ? 1 : 2;
    }
    return (px < m.width && py < m.height) ? 1 : 0;
}
__device__ __forceinline__ int map_pixel(const PixelMap& m, uint32_t& px, uint32_t& py, int64_t& outIndex) {
    return map_pixel_at(m, blockIdx.x, threadIdx.x, px, py, outIndex);
}

// RGBA store, fp32 or the reference's rgba16_float
template <bool HALF>
__device__ __forceinline__ void store_rgba(void* out, int64_t idx, float r, float g, float b, float a) {
    if constexpr (HALF) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 v = { (_Float16)r, (_Float16)g, (_Float16)b, (_Float16)a };
        reinterpret_cast<h4*>(out)[idx] = v;
    } else {
        reinterpret_cast<float4*>(out)[idx] = make_float4(r, g, b, a);
    }
}

// Exact empty-space skipping: the scratch MrirtSkip::mask points at (mrirt_skip_mask_words 32-bit words) holds the macro
// cells' bits as whole 64-lane ballots, then two byte maps of skip_map_stride(cells) bytes each (the empty-radius map, and
// the scratch of its separable passes).
MRIRT_HD uint32_t skip_bit_words(uint32_t cells) { return ((cells + 63u) / 64u) * 2u; }
MRIRT_HD uint32_t skip_map_stride(uint32_t cells) { return (cells + 3u) & ~3u; }

__device__ __forceinline__ void wave_count_add(uint64_t* counter, uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(reinterpret_cast<unsigned long long*>(counter), (unsigned long long)v);
}

}  // namespace mrirt
