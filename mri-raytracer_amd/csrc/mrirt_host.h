// Host-side helpers of libmrirt.so: status/error plumbing, camera + pixel-map setup.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mrirt.h"
#include "mrirt_device.h"

namespace mrirt {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return MRIRT_ERR_LAUNCH;
}

#define MRIRT_HIP(expr)                                         \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) return ::mrirt::hip_fail(e_);     \
    } while (0)

// Process-environment switches exist only in development builds (-DMRIRT_DEBUG_ENV): the shipped library's behaviour is a
// function of its arguments (getenv at every launch is invisible at the ABI and races with setenv).
inline bool debug_env(const char* name) {
#ifdef MRIRT_DEBUG_ENV
    return getenv(name) != nullptr;
#else
    (void)name;
    return false;
#endif
}

// divisor + correctly rounded reciprocal for M<STRICT>::divu (Markstein); see mrirt_device.h
inline UDiv make_udiv(float d) {
    UDiv u;
    u.d = d;
    u.r = 1.0f / d;
    uint32_t bits;
    memcpy(&bits, &d, sizeof bits);
    u.exact = (isnormal(d) && isnormal(u.r) && (bits & 0x7FFFFFu) != 0x7FFFFFu) ? 1u : 0u;
    return u;
}

// the constants of exp_f64_to_f32 (mrirt_device.h), in the order its Horner loop consumes them
inline void fill_exp_consts(ExpConsts& e) {
    e.log2e = 1.4426950408889634;
    e.ln2hi = 6.93147180369123816490e-01;
    e.ln2lo = 1.90821492927058770002e-10;
    const double c[13] = { 1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
                           2.7557319223985893e-06, 2.48015873015873e-05, 1.984126984126984e-04, 1.3888888888888889e-03,
                           8.333333333333333e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, 0.5, 1.0 };
    for (int i = 0; i < 13; ++i) e.c[i] = c[i];
}

// correctly rounded fp32 tan: same expression as the oracle's tanf_cr(0.5f * fovY)
inline float tan_half_fov(float fovY) { return (float)tan((double)(0.5f * fovY)); }

inline void fill_camera(Camera& c, const float eye[3], const float U[3], const float V[3], const float W[3],
                        float fovY, uint32_t width, uint32_t height, const MrirtRenderExt* ext, bool k3Aspect) {
    for (int k = 0; k < 3; ++k) { c.eye[k] = eye[k]; c.U[k] = U[k]; c.V[k] = V[k]; c.W[k] = W[k]; }
    float th = tan_half_fov(fovY);
    c.tanHalf = th;
    c.invTanHalf = 1.0f / th;
    float dimx = (float)width, dimy = (float)height;
    c.aspect = k3Aspect ? dimx / dimy : dimx / fmaxf(1.0f, dimy);
    c.mode = ext ? ext->cameraMode : 0u;
    c.orthoHalfHeight = ext ? ext->orthoHalfHeight : 0.0f;
    c.width = width;
    c.height = height;
}

// returns MRIRT_OK or an error; grid size = map.chunk * 8 workgroups of 256 threads
// blockPx: 16 (256-thread workgroups) or 8 (64-thread workgroups); laneOrder: see PixelMap
inline int fill_pixel_map(PixelMap& m, uint32_t width, uint32_t height, int64_t pitch, const MrirtRenderExt* ext,
                          uint32_t blockPx = kBlockPx, uint32_t laneOrder = 0, uint32_t bandPx = 0, bool shiftBands = false) {
    if (width == 0 || height == 0) return MRIRT_ERR_DIMS;
    m.width = width; m.height = height; m.pitch = pitch;
    m.tileSize = ext ? ext->tileSize : 0u;
    m.tileRank = ext ? ext->tileRank : 0u;
    m.tileWorld = ext ? ext->tileWorld : 0u;
    m.tileSkew = 0u;
    m.laneOrder = laneOrder;
    m.blockPx = blockPx;
    if (m.tileSize == 0) {
        if (pitch < (int64_t)width) return MRIRT_ERR_ARG;
        m.tilesX = 0;
        m.blocksX = (width + blockPx - 1) / blockPx;
        m.numBlocks = m.blocksX * ((height + blockPx - 1) / blockPx);
    } else {
        if (m.tileSize % kBlockPx != 0 || m.tileWorld == 0 || m.tileRank >= m.tileWorld) return MRIRT_ERR_ARG;
        if (width > (1u << 20) || height > (1u << 20)) return MRIRT_ERR_DIMS;       // (tile rows < 2^16: tile_position's 32-bit products)
        m.tilesX = (width + m.tileSize - 1) / m.tileSize;
        m.tileSkew = (ext ? ext->tileSkew : 0u) % m.tilesX;
        int64_t local = mrirt_tiles_for_rank(width, height, m.tileSize, m.tileRank, m.tileWorld);
        uint32_t bpr = m.tileSize / blockPx;
        m.blocksX = 0;
        m.numBlocks = (uint32_t)local * bpr * bpr;
    }
    m.chunk = (m.numBlocks + kXcds - 1) / kXcds;
    m.bandBlocks = 0;
    m.bandShift = 0;
    if (m.tileSize == 0 && bandPx >= blockPx) {
        // XCD-interleaved bands (see map_pixel): bandPx / blockPx workgroup rows per band
        const uint32_t bandRows = bandPx / blockPx, blocksY = (height + blockPx - 1) / blockPx;
        const uint32_t bands = (blocksY + bandRows - 1) / bandRows;
        m.bandBlocks = m.blocksX * bandRows;
        m.chunk = ((bands + kXcds - 1) / kXcds) * m.bandBlocks;
        if (shiftBands && m.blocksX >= 8) m.bandShift = (3 * m.blocksX) / 8;
    } else if (m.tileSize != 0 && bandPx >= blockPx && m.tileSize % bandPx == 0 && m.numBlocks != 0) {
        // tile mode: a tile's workgroups are row-major inside the tile, so bandPx-high slabs of each tile are
        // runs of consecutive workgroups; deal THOSE round-robin to the XCDs (a rank's tiles are row-major over
        // the image: contiguous eighths would again hand the long centre rays to two or three XCDs)
        m.bandBlocks = (m.tileSize / blockPx) * (bandPx / blockPx);
        const uint32_t bands = (m.numBlocks + m.bandBlocks - 1) / m.bandBlocks;
        m.chunk = ((bands + kXcds - 1) / kXcds) * m.bandBlocks;
    }
    return MRIRT_OK;
}

inline void fill_grid_dims(GridDims& g, const uint32_t dims[3], uint32_t layout) {
    g.X = dims[0]; g.Y = dims[1]; g.Z = dims[2];
    g.wide = 0;
    if (layout == MRIRT_LAYOUT_VG || layout == MRIRT_LAYOUT_QUAD)
        g.wide = ((uint64_t)mrirt_vec4_elems(dims) << 4) >= (1ull << 32) ? 1u : 0u;
    if (layout == MRIRT_LAYOUT_LINEAR) {
        g.sY = dims[0];
        g.sZ = dims[0] * dims[1];
    } else if (layout == MRIRT_LAYOUT_VG || layout == MRIRT_LAYOUT_QUAD) {
        uint32_t nbx = (dims[0] + 1) / 2, nby = (dims[1] + 1) / 2;
        g.sY = nbx * 8;
        g.sZ = nbx * nby * 8;
    } else {
        uint32_t nbx = (dims[0] + 3) / 4, nby = (dims[1] + 3) / 4;
        g.sY = nbx * 32;
        g.sZ = nbx * nby * 32;
    }
}

// VGA: copy `a` has bricks one voxel thick along axis a and 4 x 2 along the other two (in ascending axis order).
// Bricks (128-B lines) are stored x-fastest in rows of rowLines lines and slices of sliceLines lines.  The row and the
// slice pitch are PADDED off the powers of two that 2^n volumes would give them: with a pitch that is a multiple of the
// vector L1's set count, every line a wave-level gather touches in the two transverse brick directions falls into the
// same few cache sets (profiles/r03_c3_sets: TCP_READ_TAGCONFLICT_STALL 20 % and TCP_PENDING_STALL 27 % of the cycles on
// the 512^3 grid) — and into the same L2 channels.  kVgaRowPhase / kVgaSlicePhase are the pitches' residues modulo 64
// lines: a 4 x 2 x 2-brick neighbourhood then spreads over 16 different sets.
constexpr uint32_t kVgaSetLines = 64, kVgaRowPhase = 8, kVgaSlicePhase = 36;
inline uint32_t vga_pad_to_phase(uint64_t lines, uint32_t phase) {
    static const bool off = debug_env("MRIRT_NO_STRIDE_PAD");             // A/B measurements, development builds only
    if (off || lines < kVgaSetLines) return (uint32_t)lines;
    const uint32_t r = (uint32_t)(lines % kVgaSetLines);
    return (uint32_t)(lines + (phase + kVgaSetLines - r) % kVgaSetLines);
}
struct VgaGeometry { uint32_t nb[3], rowLines; uint64_t sliceLines; };
inline VgaGeometry vga_geometry(const uint32_t dims[3], int a) {
    VgaGeometry g;
    int k4 = 1;                                         // the first non-flat axis gets the 4, the second the 2
    for (int k = 0; k < 3; ++k) {
        if (k == a) g.nb[k] = dims[k];
        else { g.nb[k] = k4 ? (dims[k] + 3) / 4 : (dims[k] + 1) / 2; k4 = 0; }
    }
    g.rowLines = vga_pad_to_phase(g.nb[0], kVgaRowPhase);
    g.sliceLines = vga_pad_to_phase((uint64_t)g.rowLines * g.nb[1], kVgaSlicePhase);
    return g;
}
// Returns the float4 elements of copy a (bricks padded to whole lines, rows and slices to their padded pitches).
inline uint64_t vga_copy_elems(const uint32_t dims[3], int a) {
    const VgaGeometry g = vga_geometry(dims, a);
    return g.sliceLines * g.nb[2] * 8;
}
inline void fill_vga_dims(VgaDims& v, const uint32_t dims[3]) {
    uint64_t base = 0;
    for (int a = 0; a < 3; ++a) {
        FlatAxis& f = v.ax[a];
        const VgaGeometry g = vga_geometry(dims, a);
        uint32_t innerStride = 1;
        int k4 = 1;
        for (int k = 0; k < 3; ++k) {
            if (k == a) { f.sh[k] = 0; f.mask[k] = 0; f.inner[k] = 0; }
            else if (k4) { f.sh[k] = 2; f.mask[k] = 3; f.inner[k] = innerStride; innerStride *= 4; k4 = 0; }
            else { f.sh[k] = 1; f.mask[k] = 1; f.inner[k] = innerStride; innerStride *= 2; }
        }
        f.mul[0] = 8; f.mul[1] = g.rowLines * 8; f.mul[2] = (uint32_t)(g.sliceLines * 8);       // bricks x-fastest, 8 float4 each
        for (int k = 0; k < 3; ++k) f.wrap[k] = f.mul[k] - f.mask[k] * f.inner[k];
        f.pad = 0;
        f.baseBytes = base << 4;
        base += vga_copy_elems(dims, a);
    }
}

inline void fill_label_addr(LabelAddr& a, const uint32_t dims[3], uint32_t layout) {
    if (layout == MRIRT_LAYOUT_LINEAR) {
        const uint32_t mul[3] = { 1u, dims[0], dims[0] * dims[1] };
        for (int k = 0; k < 3; ++k) { a.sh[k] = 0; a.mask[k] = 0; a.inner[k] = 0; a.mul[k] = mul[k]; }
    } else {
        const uint32_t nbx = (dims[0] + 3) / 4, nby = (dims[1] + 3) / 4;
        const uint32_t sh[3] = { 2, 2, 1 }, mask[3] = { 3, 3, 1 }, inner[3] = { 1, 4, 16 }, mul[3] = { 32, nbx * 32, nbx * nby * 32 };
        for (int k = 0; k < 3; ++k) { a.sh[k] = sh[k]; a.mask[k] = mask[k]; a.inner[k] = inner[k]; a.mul[k] = mul[k]; }
    }
}

// inr_mlp.hip: the MLP forward with the point count in device memory (argmax only); segTicket: nullptr, or a zeroed device
// word the near-tie refinement deals its segments with
int inr_forward_dev_n(const MrirtInrDesc* desc, const float* coords, const float* feats, int64_t nMax,
                      const uint32_t* nDev, int16_t* argmax, uint32_t* segTicket, hipStream_t s);

}  // namespace mrirt
