// Load-time grid layout conversion (linear <-> 4x4x2 bricks), tile bookkeeping and the
// de-tiling copy of the multi-GPU framebuffer gather; plus the ABI's small utility entries.
//
// Reference counterparts: the upload `create_buffer + copy_from_numpy`
// (inr/viewer/brats_viewer.py:219-230) is where a caller converts a linear grid once per case;
// tiles/de-tiling have no reference counterpart (single device) — SURVEY.md section 8e.
#include "mrirt_host.h"

namespace mrirt {

thread_local int g_last_hip_error = 0;

// One thread per 16-byte (fp32) / 4-byte (u8) x-run of a brick: reads are 4 consecutive linear
// voxels, writes are 4 consecutive bricked voxels; a wave writes 8 whole 128-B bricks.
template <typename T, bool TO_BRICK>
__global__ __launch_bounds__(256) void brick_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                    GridDims lin, GridDims brk, uint32_t nbx, uint32_t nby, uint32_t nbz) {
    const uint64_t run = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (brick, yy, zz) run of 4 x-voxels
    const uint64_t nruns = (uint64_t)nbx * nby * nbz * 8;
    if (run >= nruns) return;
    const uint32_t r = (uint32_t)(run & 7u);            // yy + 4*zz
    const uint64_t b = run >> 3;
    const uint32_t bx = (uint32_t)(b % nbx), by = (uint32_t)((b / nbx) % nby), bz = (uint32_t)(b / ((uint64_t)nbx * nby));
    const uint32_t y = by * 4 + (r & 3u), z = bz * 2 + (r >> 2);
    // edge bricks replicate the last voxel (never sampled: taps are clamped inside dims)
    const uint32_t yc = min(y, lin.Y - 1), zc = min(z, lin.Z - 1);
    const uint64_t boff = (uint64_t)bx * 32 + (uint64_t)by * brk.sY + (uint64_t)bz * brk.sZ + ((r & 3u) << 2) + ((r >> 2) << 4);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t x = bx * 4 + i, xc = min(x, lin.X - 1);
        const uint64_t loff = (uint64_t)xc + (uint64_t)yc * lin.sY + (uint64_t)zc * lin.sZ;
        if constexpr (TO_BRICK) dst[boff + i] = src[loff];
        else if (x < lin.X && y < lin.Y && z < lin.Z) dst[loff] = src[boff + i];
    }
}

// linear fp32 -> float4-per-voxel grid in 2x2x2 bricks.  One thread per (padded) voxel, in
// destination order, so a wave writes 8 whole 128-B bricks.
//   VG   = (v, v[x+1]-v[x-1], v[y+1]-v[y-1], v[z+1]-v[z-1]) with neighbour indices clamped to
//          the grid — exactly the differences the gradient shader forms per corner;
//   QUAD = (v[x,y], v[x,y+1], v[x+1,y], v[x+1,y+1]) with indices clamped: the x0 pair then the x1 pair, so the
//          x blend of both rows is one packed operation on (xy) and (zw).
template <bool VG>
__global__ __launch_bounds__(256) void vec4_build_kernel(const float* __restrict__ src, float4* __restrict__ dst,
                                                         uint32_t X, uint32_t Y, uint32_t Z, uint32_t nbx, uint32_t nby, uint64_t total) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index in dst
    if (e >= total) return;
    const uint32_t in = (uint32_t)(e & 7u);
    const uint64_t b = e >> 3;
    const uint32_t bx = (uint32_t)(b % nbx), by = (uint32_t)((b / nbx) % nby), bz = (uint32_t)(b / ((uint64_t)nbx * nby));
    const uint32_t x = bx * 2 + (in & 1u), y = by * 2 + ((in >> 1) & 1u), z = bz * 2 + (in >> 2);
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (x < X && y < Y && z < Z) {
        const uint64_t sY = X, sZ = (uint64_t)X * Y;
        auto at = [&](uint32_t xx, uint32_t yy, uint32_t zz) { return src[xx + yy * sY + zz * sZ]; };
        const uint32_t xp = min(x + 1, X - 1), yp = min(y + 1, Y - 1), zp = min(z + 1, Z - 1);
        if constexpr (VG) {
            const uint32_t xm = x > 0 ? x - 1 : 0, ym = y > 0 ? y - 1 : 0, zm = z > 0 ? z - 1 : 0;
            o.x = at(x, y, z);
            o.y = at(xp, y, z) - at(xm, y, z);
            o.z = at(x, yp, z) - at(x, ym, z);
            o.w = at(x, y, zp) - at(x, y, zm);
        } else {
            o = make_float4(at(x, y, z), at(x, yp, z), at(xp, y, z), at(xp, yp, z));
        }
    }
    dst[e] = o;
}

// linear fp32 -> VGA: the VG voxel (v and its three lattice differences) written to its slot in each of the three
// axis-flat copies.  One thread per (padded) voxel of copy `a`, in destination order: a wave writes 8 whole lines.
__global__ __launch_bounds__(256) void vga_build_kernel(const float* __restrict__ src, float4* __restrict__ dst, FlatAxis f,
                                                        uint32_t X, uint32_t Y, uint32_t Z, uint32_t nb0, uint32_t nb1, uint64_t total) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index inside this copy
    if (e >= total) return;
    const uint32_t in = (uint32_t)(e & 7u);
    const uint64_t b = e >> 3;                                               // line: rows of nb0 lines, slices of nb1 lines (padded pitches)
    const uint32_t inSlice = (uint32_t)(b % nb1);
    const uint32_t bc[3] = { inSlice % nb0, inSlice / nb0, (uint32_t)(b / nb1) };   // pad lines decode to voxels outside the grid: zeros
    uint32_t p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)                                               // invert o_k(i) = (i >> sh) mul + (i & mask) inner
        p[k] = (bc[k] << f.sh[k]) + (f.mask[k] ? (in / f.inner[k]) & f.mask[k] : 0u);
    const uint32_t x = p[0], y = p[1], z = p[2];
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (x < X && y < Y && z < Z) {
        const uint64_t sY = X, sZ = (uint64_t)X * Y;
        auto at = [&](uint32_t xx, uint32_t yy, uint32_t zz) { return src[xx + yy * sY + zz * sZ]; };
        const uint32_t xp = min(x + 1, X - 1), yp = min(y + 1, Y - 1), zp = min(z + 1, Z - 1);
        const uint32_t xm = x > 0 ? x - 1 : 0, ym = y > 0 ? y - 1 : 0, zm = z > 0 ? z - 1 : 0;
        o.x = at(x, y, z);
        o.y = at(xp, y, z) - at(xm, y, z);
        o.z = at(x, yp, z) - at(x, ym, z);
        o.w = at(x, y, zp) - at(x, y, zm);
    }
    dst[e] = o;
}

template <bool HALF>
__global__ __launch_bounds__(256) void detile_kernel(const void* __restrict__ gathered, void* __restrict__ frame,
                                                     uint32_t width, uint32_t height, int64_t pitch,
                                                     uint32_t ts, uint32_t world, uint32_t tilesX, uint32_t maxLocal, uint32_t skew) {
    const uint32_t px = blockIdx.x * 16 + (threadIdx.x & 15u), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= width || py >= height) return;
    const uint32_t t = tile_dealt_index(px / ts, py / ts, tilesX, skew);
    const uint32_t rank = t % world, lt = t / world;
    const int64_t src = (((int64_t)rank * maxLocal + lt) * ts + (py % ts)) * ts + (px % ts);
    const int64_t dst = (int64_t)py * pitch + px;
    if constexpr (HALF) reinterpret_cast<uint2*>(frame)[dst] = reinterpret_cast<const uint2*>(gathered)[src];
    else reinterpret_cast<float4*>(frame)[dst] = reinterpret_cast<const float4*>(gathered)[src];
}

// BC4 / RGTC1-unorm slices -> u8 voxels [D][H][W] (scripts/volumeRendering/app.py:200-250).  One thread per
// 4x4 block: 8 bytes in (two endpoints, 16 three-bit codes), up to 16 bytes out.  r0 > r1: six
// interpolants ((7-i) r0 + i r1 + 3) / 7; else four ((5-i) r0 + i r1 + 2) / 5, then 0 and 255.
__global__ __launch_bounds__(256) void bc4_decode_kernel(const uint2* __restrict__ blocks, uint8_t* __restrict__ out,
                                                         uint32_t width, uint32_t height, uint32_t depth,
                                                         uint32_t bw, uint32_t bh) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (uint64_t)bw * bh * depth) return;
    const uint32_t bx = (uint32_t)(b % bw), by = (uint32_t)((b / bw) % bh), z = (uint32_t)(b / ((uint64_t)bw * bh));
    const uint2 q = blocks[b];
    const int r0 = (int)(q.x & 0xffu), r1 = (int)((q.x >> 8) & 0xffu);
    const uint64_t bits = (uint64_t)(q.x >> 16) | ((uint64_t)q.y << 16);          // bytes 2..7, little endian
    int pal[8];
    pal[0] = r0; pal[1] = r1;
    const bool six = r0 > r1;
#pragma unroll
    for (int i = 1; i <= 6; ++i) {
        const int a6 = ((7 - i) * r0 + i * r1 + 3) / 7;
        const int a4 = i <= 4 ? ((5 - i) * r0 + i * r1 + 2) / 5 : (i == 5 ? 0 : 255);
        pal[i + 1] = six ? a6 : a4;
    }
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
        const uint32_t y = by * 4 + ty;
        if (y >= height) break;
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) {
            const uint32_t x = bx * 4 + tx;
            if (x >= width) break;
            const uint32_t code = (uint32_t)(bits >> (3 * (ty * 4 + tx))) & 7u;
            int v = pal[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) v = code == (uint32_t)k ? pal[k] : v;      // no dynamic register indexing
            out[((uint64_t)z * height + y) * width + x] = (uint8_t)v;
        }
    }
}

// Macro-cell summaries for exact empty-space skipping (brats_march.hip): macro cell m covers the voxels
// [8m, 8m+8] per axis (inclusive, so every trilinear cell whose base index lies in [8m, 8m+7] is inside).
// ub = max + 2e-6 max|v|: three nested unfused lerps exceed the largest corner by at most ~12 ulp of the
// largest magnitude; 2e-6 is twice that.  A NaN voxel makes the bound +inf (never skipped).
__global__ __launch_bounds__(256) void macro_max_kernel(const float* __restrict__ lin, float* __restrict__ ub,
                                                        uint32_t X, uint32_t Y, uint32_t Z, uint32_t mx, uint32_t my, uint32_t cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cells) return;
    const uint32_t cx = c % mx, cy = (c / mx) % my, cz = c / (mx * my);
    const uint32_t x1 = min(8 * cx + 8, X - 1), y1 = min(8 * cy + 8, Y - 1), z1 = min(8 * cz + 8, Z - 1);
    float vmax = -INFINITY, amax = 0.0f;
    bool nan = false;
    for (uint32_t z = 8 * cz; z <= z1; ++z)
        for (uint32_t y = 8 * cy; y <= y1; ++y)
            for (uint32_t x = 8 * cx; x <= x1; ++x) {
                const float v = lin[x + (size_t)X * (y + (size_t)Y * z)];
                nan |= v != v;
                vmax = fmaxf(vmax, v);
                amax = fmaxf(amax, fabsf(v));
            }
    ub[c] = nan ? INFINITY : vmax + 2e-6f * amax;
}

__global__ __launch_bounds__(256) void macro_label_kernel(const uint32_t* __restrict__ lin, uint32_t* __restrict__ any,
                                                          uint32_t X, uint32_t Y, uint32_t Z, uint32_t mx, uint32_t my, uint32_t cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cells) return;
    const uint32_t cx = c % mx, cy = (c / mx) % my, cz = c / (mx * my);
    const uint32_t x1 = min(8 * cx + 8, X - 1), y1 = min(8 * cy + 8, Y - 1), z1 = min(8 * cz + 8, Z - 1);
    uint32_t acc = 0;
    for (uint32_t z = 8 * cz; z <= z1; ++z)
        for (uint32_t y = 8 * cy; y <= y1; ++y)
            for (uint32_t x = 8 * cx; x <= x1; ++x) acc |= lin[x + (size_t)X * (y + (size_t)Y * z)];
    any[c] = acc;
}

static int brick_common(const void* src, void* dst, const uint32_t dims[3], uint32_t elem_bytes, bool toBrick, void* stream) {
    if (!src || !dst || !dims) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (dims[k] < 1) return MRIRT_ERR_DIMS;
    if (elem_bytes != 1 && elem_bytes != 4) return MRIRT_ERR_LAYOUT;
    GridDims lin, brk;
    fill_grid_dims(lin, dims, MRIRT_LAYOUT_LINEAR);
    fill_grid_dims(brk, dims, MRIRT_LAYOUT_BRICK);
    const uint32_t nbx = (dims[0] + 3) / 4, nby = (dims[1] + 3) / 4, nbz = (dims[2] + 1) / 2;
    const uint64_t nruns = (uint64_t)nbx * nby * nbz * 8;
    const dim3 grid((uint32_t)((nruns + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (elem_bytes == 4) {
        if (toBrick) hipLaunchKernelGGL((brick_kernel<uint32_t, true>), grid, block, 0, s, (const uint32_t*)src, (uint32_t*)dst, lin, brk, nbx, nby, nbz);
        else         hipLaunchKernelGGL((brick_kernel<uint32_t, false>), grid, block, 0, s, (const uint32_t*)src, (uint32_t*)dst, lin, brk, nbx, nby, nbz);
    } else {
        if (toBrick) hipLaunchKernelGGL((brick_kernel<uint8_t, true>), grid, block, 0, s, (const uint8_t*)src, (uint8_t*)dst, lin, brk, nbx, nby, nbz);
        else         hipLaunchKernelGGL((brick_kernel<uint8_t, false>), grid, block, 0, s, (const uint8_t*)src, (uint8_t*)dst, lin, brk, nbx, nby, nbz);
    }
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

}  // namespace mrirt

using namespace mrirt;

extern "C" int64_t mrirt_brick_elems(const uint32_t dims[3]) {
    if (!dims) return 0;
    return (int64_t)((dims[0] + 3) / 4) * ((dims[1] + 3) / 4) * ((dims[2] + 1) / 2) * 32;
}

extern "C" int mrirt_brick_grid(const void* linear, void* bricked, const uint32_t dims[3], uint32_t elem_bytes, void* stream) {
    return brick_common(linear, bricked, dims, elem_bytes, true, stream);
}

extern "C" int mrirt_unbrick_grid(const void* bricked, void* linear, const uint32_t dims[3], uint32_t elem_bytes, void* stream) {
    return brick_common(bricked, linear, dims, elem_bytes, false, stream);
}

extern "C" int64_t mrirt_vec4_elems(const uint32_t dims[3]) {
    if (!dims) return 0;
    return (int64_t)((dims[0] + 1) / 2) * ((dims[1] + 1) / 2) * ((dims[2] + 1) / 2) * 8;
}

extern "C" int64_t mrirt_vga_elems(const uint32_t dims[3]) {
    if (!dims) return 0;
    return (int64_t)(vga_copy_elems(dims, 0) + vga_copy_elems(dims, 1) + vga_copy_elems(dims, 2));
}

extern "C" int mrirt_build_vec4_grid(const float* linear, void* vec4_grid, const uint32_t dims[3], uint32_t layout, void* stream) {
    if (!linear || !vec4_grid || !dims) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (dims[k] < 1) return MRIRT_ERR_DIMS;
    if (layout == MRIRT_LAYOUT_VGA) {
        VgaDims v;
        fill_vga_dims(v, dims);
        hipStream_t s = static_cast<hipStream_t>(stream);
        for (int a = 0; a < 3; ++a) {
            const uint64_t total = vga_copy_elems(dims, a);
            if (total >= (1ull << 28)) return MRIRT_ERR_DIMS;                      // 32-bit byte offsets inside a copy
            const FlatAxis& f = v.ax[a];
            const uint32_t nb0 = f.mul[1] / 8, nb1 = f.mul[2] / 8;                  // row pitch, slice pitch (lines)
            hipLaunchKernelGGL(vga_build_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, linear,
                               reinterpret_cast<float4*>(static_cast<char*>(vec4_grid) + f.baseBytes), f,
                               dims[0], dims[1], dims[2], nb0, nb1, total);
            MRIRT_HIP(hipGetLastError());
        }
        return MRIRT_OK;
    }
    if (layout != MRIRT_LAYOUT_VG && layout != MRIRT_LAYOUT_QUAD) return MRIRT_ERR_LAYOUT;
    const uint32_t nbx = (dims[0] + 1) / 2, nby = (dims[1] + 1) / 2;
    const uint64_t total = (uint64_t)mrirt_vec4_elems(dims);
    if (total >= (1ull << 32)) return MRIRT_ERR_DIMS;
    const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (layout == MRIRT_LAYOUT_VG)
        hipLaunchKernelGGL((vec4_build_kernel<true>), grid, block, 0, s, linear, (float4*)vec4_grid, dims[0], dims[1], dims[2], nbx, nby, total);
    else
        hipLaunchKernelGGL((vec4_build_kernel<false>), grid, block, 0, s, linear, (float4*)vec4_grid, dims[0], dims[1], dims[2], nbx, nby, total);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

// MOD4 (include/mrirt.h): the four modalities' values of a voxel as one float4, elements in the VG grid's order
__global__ __launch_bounds__(256) void mod4_build_kernel(const float* __restrict__ m0, const float* __restrict__ m1,
                                                         const float* __restrict__ m2, const float* __restrict__ m3, float4* __restrict__ dst,
                                                         uint32_t X, uint32_t Y, uint32_t Z, uint32_t nbx, uint32_t nby, uint64_t total) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index in dst
    if (e >= total) return;
    const uint32_t in = (uint32_t)(e & 7u);
    const uint64_t b = e >> 3;
    const uint32_t bx = (uint32_t)(b % nbx), by = (uint32_t)((b / nbx) % nby), bz = (uint32_t)(b / ((uint64_t)nbx * nby));
    const uint32_t x = bx * 2 + (in & 1u), y = by * 2 + ((in >> 1) & 1u), z = bz * 2 + (in >> 2);
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (x < X && y < Y && z < Z) {
        const uint64_t i = x + (uint64_t)y * X + (uint64_t)z * X * Y;
        o = make_float4(m0[i], m1[i], m2[i], m3[i]);
    }
    dst[e] = o;
}

extern "C" int mrirt_build_mod4_grid(const float* const linear[4], void* mod4_grid, const uint32_t dims[3], void* stream) {
    if (!linear || !mod4_grid || !dims) return MRIRT_ERR_NULL;
    for (int m = 0; m < 4; ++m) if (!linear[m]) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (dims[k] < 2) return MRIRT_ERR_DIMS;
    const uint32_t nbx = (dims[0] + 1) / 2, nby = (dims[1] + 1) / 2;
    const uint64_t total = (uint64_t)mrirt_vec4_elems(dims);
    if (total >= (1ull << 32)) return MRIRT_ERR_DIMS;
    hipLaunchKernelGGL(mod4_build_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       linear[0], linear[1], linear[2], linear[3], static_cast<float4*>(mod4_grid), dims[0], dims[1], dims[2], nbx, nby, total);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

// LABCELL (include/mrirt.h): per voxel = cell base, the nearest-label candidates of BOTH label grids: the labels of the cell's
// eight corners as nibbles (corner (dx,dy,dz) at bits 4 (dx + 2 dy + 4 dz), neighbours clamped, labels >= 8 -> 8: the shader
// draws 1..7 only), .x ground truth, .y prediction; elements in the QUAD grid's order.
__global__ __launch_bounds__(256) void label_cells_kernel(const uint32_t* __restrict__ seg, const uint32_t* __restrict__ pred,
                                                          uint2* __restrict__ dst, uint32_t X, uint32_t Y, uint32_t Z,
                                                          uint32_t nbx, uint32_t nby, uint64_t total) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // element index in dst (the vec4 grids' order)
    if (e >= total) return;
    const uint32_t in = (uint32_t)(e & 7u);
    const uint64_t b = e >> 3;
    const uint32_t bx = (uint32_t)(b % nbx), by = (uint32_t)((b / nbx) % nby), bz = (uint32_t)(b / ((uint64_t)nbx * nby));
    const uint32_t x = bx * 2 + (in & 1u), y = by * 2 + ((in >> 1) & 1u), z = bz * 2 + (in >> 2);
    uint2 o = make_uint2(0u, 0u);
    if (x < X && y < Y && z < Z) {
        const uint64_t sY = X, sZ = (uint64_t)X * Y;
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const uint32_t xx = min(x + (c & 1u), X - 1), yy = min(y + ((c >> 1) & 1u), Y - 1), zz = min(z + (c >> 2), Z - 1);
            const uint64_t i = xx + yy * sY + zz * sZ;
            if (seg != nullptr)  o.x |= min(seg[i], 8u) << (4u * c);
            if (pred != nullptr) o.y |= min(pred[i], 8u) << (4u * c);
        }
    }
    dst[e] = o;
}

extern "C" int mrirt_build_label_cells(const uint32_t* seg_linear, const uint32_t* pred_linear, const uint32_t dims[3],
                                       void* cells, void* stream) {
    if (!cells || !dims) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (dims[k] < 2) return MRIRT_ERR_DIMS;
    const uint32_t nbx = (dims[0] + 1) / 2, nby = (dims[1] + 1) / 2;
    const uint64_t total = (uint64_t)mrirt_vec4_elems(dims);
    if (total >= (1ull << 29)) return MRIRT_ERR_DIMS;                        // 32-bit byte offsets of 8-byte elements
    hipLaunchKernelGGL(label_cells_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       seg_linear, pred_linear, static_cast<uint2*>(cells), dims[0], dims[1], dims[2], nbx, nby, total);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

extern "C" int64_t mrirt_macro_cells(const uint32_t dims[3]) {
    if (!dims) return 0;
    return (int64_t)((dims[0] + 7) / 8) * ((dims[1] + 7) / 8) * ((dims[2] + 7) / 8);
}

extern "C" int64_t mrirt_skip_mask_words(const uint32_t dims[3]) {      // whole wave ballots: 2 words per 64 cells
    if (!dims) return 0;
    // the 8^3 macro-cell bits, then two byte maps (one byte per macro cell, padded to words): the distance map and the
    // scratch of its separable passes
    const int64_t cells = mrirt_macro_cells(dims);
    if (cells >= (1ll << 31)) return 0;
    return (int64_t)skip_bit_words((uint32_t)cells) + 2 * (int64_t)(skip_map_stride((uint32_t)cells) / 4u);
}

static int macro_common(const void* lin, void* out, const uint32_t dims[3], bool labels, void* stream) {
    if (!lin || !out || !dims) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (dims[k] < 1) return MRIRT_ERR_DIMS;
    const int64_t cells = mrirt_macro_cells(dims);
    if (cells >= (1ll << 31)) return MRIRT_ERR_DIMS;
    const uint32_t mx = (dims[0] + 7) / 8, my = (dims[1] + 7) / 8;
    const dim3 grid((uint32_t)((cells + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (labels) hipLaunchKernelGGL(macro_label_kernel, grid, block, 0, s, (const uint32_t*)lin, (uint32_t*)out, dims[0], dims[1], dims[2], mx, my, (uint32_t)cells);
    else        hipLaunchKernelGGL(macro_max_kernel, grid, block, 0, s, (const float*)lin, (float*)out, dims[0], dims[1], dims[2], mx, my, (uint32_t)cells);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

extern "C" int mrirt_build_macro_max(const float* linear, const uint32_t dims[3], float* macro_ub, void* stream) {
    return macro_common(linear, macro_ub, dims, false, stream);
}

extern "C" int mrirt_build_macro_labels(const uint32_t* labels_linear, const uint32_t dims[3], uint32_t* macro_any, void* stream) {
    return macro_common(labels_linear, macro_any, dims, true, stream);
}

extern "C" int mrirt_bc4_decode(const void* blocks, uint32_t width, uint32_t height, uint32_t depth,
                                uint8_t* out_u8, void* stream) {
    if (!blocks || !out_u8) return MRIRT_ERR_NULL;
    if (width == 0 || height == 0 || depth == 0) return MRIRT_ERR_DIMS;
    const uint32_t bw = (width + 3) / 4, bh = (height + 3) / 4;
    const uint64_t nblk = (uint64_t)bw * bh * depth;
    if (nblk >= (1ull << 31) * 256) return MRIRT_ERR_DIMS;
    if ((reinterpret_cast<uintptr_t>(blocks) & 7u) != 0) return MRIRT_ERR_ARG;      // 8-byte blocks, read as uint2
    hipLaunchKernelGGL(bc4_decode_kernel, dim3((uint32_t)((nblk + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint2*>(blocks), out_u8, width, height, depth, bw, bh);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

extern "C" int64_t mrirt_tiles_for_rank(uint32_t width, uint32_t height, uint32_t tileSize, uint32_t rank, uint32_t world) {
    if (tileSize == 0 || world == 0 || rank >= world) return 0;
    const int64_t tiles = (int64_t)((width + tileSize - 1) / tileSize) * ((height + tileSize - 1) / tileSize);
    return tiles > rank ? (tiles - rank + world - 1) / world : 0;
}

extern "C" int mrirt_detile(const void* gathered, void* frame, uint32_t width, uint32_t height, int64_t pitch_px,
                            uint32_t tileSize, uint32_t world, uint32_t tileSkew, uint32_t outFormat, void* stream) {
    if (!gathered || !frame) return MRIRT_ERR_NULL;
    if (width == 0 || height == 0) return MRIRT_ERR_DIMS;
    if (tileSize == 0 || world == 0 || pitch_px < (int64_t)width) return MRIRT_ERR_ARG;
    if (outFormat > MRIRT_OUT_RGBA16F) return MRIRT_ERR_LAYOUT;
    if (width > (1u << 20) || height > (1u << 20) || tileSize < 16) return MRIRT_ERR_DIMS;
    const uint32_t tilesX = (width + tileSize - 1) / tileSize;
    tileSkew %= tilesX;
    const uint32_t maxLocal = (uint32_t)mrirt_tiles_for_rank(width, height, tileSize, 0, world);
    const dim3 grid((width + 15) / 16, (height + 15) / 16), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (outFormat == MRIRT_OUT_RGBA16F)
        hipLaunchKernelGGL((detile_kernel<true>), grid, block, 0, s, gathered, frame, width, height, pitch_px, tileSize, world, tilesX, maxLocal, tileSkew);
    else
        hipLaunchKernelGGL((detile_kernel<false>), grid, block, 0, s, gathered, frame, width, height, pitch_px, tileSize, world, tilesX, maxLocal, tileSkew);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

extern "C" int mrirt_abi_version(void) { return MRIRT_ABI_VERSION; }

extern "C" const char* mrirt_status_string(int status) {
    switch (status) {
        case MRIRT_OK: return "ok";
        case MRIRT_ERR_NULL: return "required pointer is NULL";
        case MRIRT_ERR_DIMS: return "bad volume or image dimensions";
        case MRIRT_ERR_LAYOUT: return "unknown layout / dtype / mode";
        case MRIRT_ERR_LAUNCH: return "HIP runtime error (see mrirt_last_hip_error)";
        case MRIRT_ERR_ARG: return "inconsistent argument";
        case MRIRT_ERR_NO_DEVICE: return "no gfx950 device";
        default: return "unknown status";
    }
}

extern "C" int mrirt_last_hip_error(void) { return g_last_hip_error; }

extern "C" uint32_t mrirt_sizeof(uint32_t which) {
    switch (which) {
        case 0: return (uint32_t)sizeof(MrirtBratsParams);
        case 1: return (uint32_t)sizeof(MrirtRenderExt);
        case 2: return (uint32_t)sizeof(MrirtVolumeParams);
        case 3: return (uint32_t)sizeof(MrirtSdfParams);
        case 4: return (uint32_t)sizeof(MrirtInrDesc);
        case 5: return (uint32_t)sizeof(MrirtSkip);
        default: return 0;
    }
}
