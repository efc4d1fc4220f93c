// K2: fixed-step near/far-plane volume marcher — gfx950 HIP replacement for the Slang compute
// shader `volume_cs` (reference: scripts/volumeRendering/volume_render.slang:104-148; sampling
// helpers :28-65).  K3: analytic-SDF sphere tracer `raymarch_cs`
// (scripts/raymarch/raymarch.slang:60-99).
//
// K2 voxel modes: the reference's one-u32-per-u8-voxel uint4 buffer (app.py:150-153), real
// bytes, or fp32 (build-defined generalisation used by BASELINE config 1).
#include "mrirt_host.h"

namespace mrirt {

struct K2Args {
    Camera cam;
    PixelMap map;
    uint32_t dim[3];
    float dimM1[3];        // float(d) - 1
    float nearP, farP;     // already max(0,near), max(near,far)
    float steps;           // max(1, stepCount)
    const void* vol;
    void* out;
    uint64_t* stats;
};

// byte / 255.0f, IEEE-exact in three instructions (Markstein, as M<true>::divu: 255's significand is not all
// ones, so RN(q + (x - q*255) * RN(1/255)) is the correctly rounded quotient; tests check all 256 bytes).  The
// compiler's own x / 255.0f is a ten-instruction division — eight of them per sample.
__device__ __forceinline__ float unorm8(uint32_t b) {
    const float x = (float)b, r = 1.0f / 255.0f;                     // the constant folds to RN(1/255)
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

template <int MODE>
__device__ __forceinline__ float fetch_k2(const void* __restrict__ vol, uint32_t idx) {
    if constexpr (MODE == 0) return unorm8(static_cast<const uint32_t*>(vol)[idx] & 0xffu);           // :33-38
    else if constexpr (MODE == 1) return unorm8(static_cast<const uint8_t*>(vol)[idx]);
    else return static_cast<const float*>(vol)[idx];
}

template <bool STRICT, int MODE, bool HALF>
__global__ __launch_bounds__(256) void volume_march_kernel(const K2Args a) {
    using Mm = M<STRICT>;
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    if (a.map.numBlocks == 0) return;
    float accum = 0.0f;
    uint32_t nLive = 0;
    if (kind == 1) {
        const Camera& c = a.cam;
        const float invx = 1.0f / (float)c.width, invy = 1.0f / (float)c.height;
        const float uvx = ((float)px + 0.5f) * invx, uvy = ((float)py + 0.5f) * invy;
        const float ndcx = uvx * 2.0f - 1.0f, ndcy = 1.0f - uvy * 2.0f;
        const float n = a.nearP, f = a.farP;
        float pos[3], sv[3];
        if (c.mode == 0) {
            const float vx = ndcx * c.aspect * c.tanHalf, vy = ndcy * c.tanHalf, vz = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float wn = ((c.eye[k] + c.U[k] * (vx * n)) + c.V[k] * (vy * n)) + c.W[k] * (vz * n);
                const float wf = ((c.eye[k] + c.U[k] * (vx * f)) + c.V[k] * (vy * f)) + c.W[k] * (vz * f);
                pos[k] = wn; sv[k] = (wf - wn) / a.steps;
            }
        } else {
            const float sx = ndcx * c.aspect * c.orthoHalfHeight, sy = ndcy * c.orthoHalfHeight;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float wn = ((c.eye[k] + c.U[k] * sx) + c.V[k] * sy) + c.W[k] * n;
                const float wf = ((c.eye[k] + c.U[k] * sx) + c.V[k] * sy) + c.W[k] * f;
                pos[k] = wn; sv[k] = (wf - wn) / a.steps;
            }
        }
        const float scale = 4.0f / a.steps;
        const uint32_t nsteps = (uint32_t)a.steps;
        const uint32_t sY = a.dim[0], sZ = a.dim[0] * a.dim[1];
        // The shader walks all `steps` positions and tests each against the cube (:136); most of them lie outside
        // it (96 % at the bench geometry).  The positions are a running fp32 sum, so they cannot be jumped to —
        // but the steps before the ray can possibly have entered need only the three adds, and the steps after
        // it has certainly left change nothing (the cube is convex).  [iEnter, iExit) is the slab interval of
        // the exact line, widened by the worst-case drift of the running sum (in steps) plus two.
        uint32_t iEnter = 0, iExit = nsteps;
        {
            float lo = 0.0f, hi = (float)nsteps;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float end = pos[k] + (float)nsteps * sv[k];
                const float drift = (float)nsteps * 1.2e-7f * fmaxf(fabsf(pos[k]), fabsf(end)) + 1e-6f;
                if (fabsf(sv[k]) > 1e-30f) {
                    const float r = 1.0f / sv[k];
                    const float i0 = (-1.0f - pos[k]) * r, i1 = (1.0f - pos[k]) * r;
                    const float m = fabsf(drift * r) + 2.0f;
                    lo = fmaxf(lo, fminf(i0, i1) - m);
                    hi = fminf(hi, fmaxf(i0, i1) + m);
                } else if (!(fabsf(pos[k]) < 1.0f + drift)) {
                    hi = -1.0f;                                        // parallel to this slab and outside it
                }
            }
            if (hi < lo) { iEnter = iExit = 0; }
            else { iEnter = (uint32_t)floorf(lo); iExit = min(nsteps, (uint32_t)ceilf(hi) + 1u); iEnter = min(iEnter, iExit); }
        }
        for (uint32_t i = 0; i < iEnter; ++i) { pos[0] += sv[0]; pos[1] += sv[1]; pos[2] += sv[2]; }
        for (uint32_t i = iEnter; i < iExit; ++i) {
            const bool inside = pos[0] < 1.0f && pos[1] < 1.0f && pos[2] < 1.0f &&
                                pos[0] > -1.0f && pos[1] > -1.0f && pos[2] > -1.0f;
            if (inside && accum < 1.0f) {
                uint32_t p0[3], p1[3];
                float t[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float x = satf(0.5f * (pos[k] + 1.0f)) * a.dimM1[k];
                    const float fl = floorf(x);
                    p0[k] = (uint32_t)fl;
                    p1[k] = min(p0[k] + 1, a.dim[k] - 1);
                    t[k] = x - fl;
                }
                const uint32_t y0 = p0[1] * sY, y1 = p1[1] * sY, z0 = p0[2] * sZ, z1 = p1[2] * sZ;
                float c000, c100, c010, c110, c001, c101, c011, c111;
                if constexpr (MODE == 3) {
                    // CELL8: voxel (x,y,z) holds the eight bytes of ITS cell, neighbours clamped as :52-54 —
                    // one 8-byte gather per sample instead of eight (mrirt_build_cell8)
                    const uint2 c = static_cast<const uint2*>(a.vol)[p0[0] + y0 + z0];
                    c000 = unorm8(c.x & 0xffu); c100 = unorm8((c.x >> 8) & 0xffu); c010 = unorm8((c.x >> 16) & 0xffu); c110 = unorm8(c.x >> 24);
                    c001 = unorm8(c.y & 0xffu); c101 = unorm8((c.y >> 8) & 0xffu); c011 = unorm8((c.y >> 16) & 0xffu); c111 = unorm8(c.y >> 24);
                    (void)y1; (void)z1;
                } else {
                    c000 = fetch_k2<MODE>(a.vol, p0[0] + y0 + z0); c100 = fetch_k2<MODE>(a.vol, p1[0] + y0 + z0);
                    c010 = fetch_k2<MODE>(a.vol, p0[0] + y1 + z0); c110 = fetch_k2<MODE>(a.vol, p1[0] + y1 + z0);
                    c001 = fetch_k2<MODE>(a.vol, p0[0] + y0 + z1); c101 = fetch_k2<MODE>(a.vol, p1[0] + y0 + z1);
                    c011 = fetch_k2<MODE>(a.vol, p0[0] + y1 + z1); c111 = fetch_k2<MODE>(a.vol, p1[0] + y1 + z1);
                }
                const float c00 = Mm::lerp(c000, c100, t[0]), c01 = Mm::lerp(c001, c101, t[0]);
                const float c10 = Mm::lerp(c010, c110, t[0]), c11 = Mm::lerp(c011, c111, t[0]);
                const float c0 = Mm::lerp(c00, c10, t[1]), c1 = Mm::lerp(c01, c11, t[1]);
                const float s = Mm::lerp(c0, c1, t[2]) * scale;
                accum = Mm::mad(1.0f - accum, s, accum);
                ++nLive;
            }
            pos[0] += sv[0]; pos[1] += sv[1]; pos[2] += sv[2];      // incremental, as the shader (:143)
            if (accum > 0.995f) break;
        }
    }
    if (kind != 0) store_rgba<HALF>(a.out, oidx, accum, accum, accum, 1.0f);
    if (a.stats != nullptr) wave_count_add(a.stats, nLive);
}

template <bool STRICT, int MODE>
static int launch_k2(const K2Args& a, bool half, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(256);
    if (half) hipLaunchKernelGGL((volume_march_kernel<STRICT, MODE, true>), grid, block, 0, s, a);
    else      hipLaunchKernelGGL((volume_march_kernel<STRICT, MODE, false>), grid, block, 0, s, a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <bool STRICT>
static int launch_k2_mode(const K2Args& a, uint32_t mode, bool half, hipStream_t s) {
    switch (mode) {
        case MRIRT_VOX_U32X4: return launch_k2<STRICT, 0>(a, half, s);
        case MRIRT_VOX_U8: return launch_k2<STRICT, 1>(a, half, s);
        case MRIRT_VOX_CELL8: return launch_k2<STRICT, 3>(a, half, s);
        default: return launch_k2<STRICT, 2>(a, half, s);
    }
}

// ---------------------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------------------
struct K3Args {
    Camera cam;
    PixelMap map;
    uint32_t maxSteps;
    float maxDistance, hitThreshold;
    float* out;
};

__global__ __launch_bounds__(256) void sdf_march_kernel(const K3Args a) {
    uint32_t px, py;
    int64_t oidx;
    if (map_pixel(a.map, px, py, oidx) != 1) return;
    float ro[3], rd[3];
    primary_ray(a.cam, px, py, ro, rd);
    float t = 0.0f, p[3] = { ro[0], ro[1], ro[2] };
    bool hit = false;
    for (uint32_t i = 0; i < a.maxSteps; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = ro[k] + t * rd[k];
        const float d = sqrtf(dot3(p[0], p[1], p[2], p[0], p[1], p[2])) - 0.6f;   // sdSphere(p, 0.6), :24-31
        if (d < a.hitThreshold) { hit = true; break; }
        t += clampf(d, 0.01f, 0.25f);
        if (t > a.maxDistance) break;
    }
    float r, g, b;
    if (hit) {
        float nx = p[0], ny = p[1], nz = p[2];
        normalize3(nx, ny, nz);
        const float u = (float)atan2((double)nz, (double)nx) / (2.0f * 3.14159265f) + 0.5f;
        r = u; g = ny * 0.5f + 0.5f; b = 1.0f - u;
    } else {
        float dx = rd[0], dy = rd[1], dz = rd[2];
        normalize3(dx, dy, dz);
        const float tbg = 0.5f * (dy + 1.0f);
        r = M<true>::lerp(0.05f, 0.2f, tbg); g = M<true>::lerp(0.06f, 0.25f, tbg); b = M<true>::lerp(0.08f, 0.3f, tbg);
    }
    reinterpret_cast<float4*>(a.out)[oidx] = make_float4(r, g, b, 1.0f);
}

// u8 voxels -> CELL8 (one thread per voxel): the eight bytes sampleTrilinear reads for the cell based at this
// voxel, p1 = min(p0 + 1, d - 1) as volume_render.slang:52-54.  Accepts the reference's u32-per-voxel upload too.
template <bool WIDE_SRC>
__global__ __launch_bounds__(256) void cell8_build_kernel(const void* __restrict__ src, uint2* __restrict__ dst,
                                                          uint32_t X, uint32_t Y, uint32_t Z, uint64_t total) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t x = (uint32_t)(i % X), y = (uint32_t)((i / X) % Y), z = (uint32_t)(i / ((uint64_t)X * Y));
    const uint32_t xp = min(x + 1, X - 1), yp = min(y + 1, Y - 1), zp = min(z + 1, Z - 1);
    auto at = [&](uint32_t xx, uint32_t yy, uint32_t zz) -> uint32_t {
        const uint64_t k = xx + (uint64_t)X * (yy + (uint64_t)Y * zz);
        if constexpr (WIDE_SRC) return static_cast<const uint32_t*>(src)[k] & 0xffu;
        else return static_cast<const uint8_t*>(src)[k];
    };
    uint2 c;
    c.x = at(x, y, z) | (at(xp, y, z) << 8) | (at(x, yp, z) << 16) | (at(xp, yp, z) << 24);
    c.y = at(x, y, zp) | (at(xp, y, zp) << 8) | (at(x, yp, zp) << 16) | (at(xp, yp, zp) << 24);
    dst[i] = c;
}

}  // namespace mrirt

using namespace mrirt;

extern "C" int mrirt_build_cell8(const void* voxels, uint32_t src_mode, const uint32_t dims[3], void* cell8, void* stream) {
    if (!voxels || !dims || !cell8) return MRIRT_ERR_NULL;
    if (src_mode != MRIRT_VOX_U32X4 && src_mode != MRIRT_VOX_U8) return MRIRT_ERR_LAYOUT;
    for (int k = 0; k < 3; ++k) if (dims[k] < 1) return MRIRT_ERR_DIMS;
    const uint64_t total = (uint64_t)dims[0] * dims[1] * dims[2];
    if (total >= (1ull << 32)) return MRIRT_ERR_DIMS;
    const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (src_mode == MRIRT_VOX_U32X4) hipLaunchKernelGGL(cell8_build_kernel<true>, grid, block, 0, s, voxels, static_cast<uint2*>(cell8), dims[0], dims[1], dims[2], total);
    else                             hipLaunchKernelGGL(cell8_build_kernel<false>, grid, block, 0, s, voxels, static_cast<uint2*>(cell8), dims[0], dims[1], dims[2], total);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

extern "C" int mrirt_render_volume(const MrirtVolumeParams* p, const MrirtRenderExt* ext, const void* volume,
                                   uint32_t mode, void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream) {
    if (!p || !volume || !out_rgba) return MRIRT_ERR_NULL;
    if (mode > MRIRT_VOX_CELL8) return MRIRT_ERR_LAYOUT;
    for (int k = 0; k < 3; ++k) if (p->volDim[k] < 1) return MRIRT_ERR_DIMS;
    if ((uint64_t)p->volDim[0] * p->volDim[1] * p->volDim[2] >= (1ull << 32)) return MRIRT_ERR_DIMS;
    const uint32_t math = ext ? ext->math : (uint32_t)MRIRT_MATH_STRICT;
    const uint32_t fmt = ext ? ext->outFormat : (uint32_t)MRIRT_OUT_RGBA32F;
    if (math > MRIRT_MATH_FAST || fmt > MRIRT_OUT_RGBA16F) return MRIRT_ERR_LAYOUT;
    // the loop runs uint(max(1, stepCount)) times per pixel (volume_render.slang:131): bound it, and refuse NaN / inf
    if (!isfinite(p->stepCount) || p->stepCount > (float)(1u << 20)) return MRIRT_ERR_ARG;
    if (!isfinite(p->nearPlane) || !isfinite(p->farPlane)) return MRIRT_ERR_ARG;
    K2Args a;
    fill_camera(a.cam, p->eye, p->U, p->V, p->W, p->fovY, p->imageSize[0], p->imageSize[1], ext, false);
    // 16x16-pixel workgroups, one workgroup row per band, bands interleaved over the XCDs (load balance)
    int rc = fill_pixel_map(a.map, p->imageSize[0], p->imageSize[1], pitch_px, ext, kBlockPx, 0, kBlockPx);
    if (rc != MRIRT_OK) return rc;
    for (int k = 0; k < 3; ++k) { a.dim[k] = p->volDim[k]; a.dimM1[k] = (float)p->volDim[k] - 1.0f; }
    a.nearP = fmaxf(0.0f, p->nearPlane);
    a.farP = fmaxf(a.nearP, p->farPlane);
    a.steps = fmaxf(1.0f, p->stepCount);
    a.vol = volume; a.out = out_rgba; a.stats = stats_dev;
    if (a.map.numBlocks == 0) return MRIRT_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool half = fmt == MRIRT_OUT_RGBA16F;
    return math == MRIRT_MATH_STRICT ? launch_k2_mode<true>(a, mode, half, s) : launch_k2_mode<false>(a, mode, half, s);
}

extern "C" int mrirt_render_sdf(const MrirtSdfParams* p, uint32_t width, uint32_t height,
                                float* out_rgba, int64_t pitch_px, void* stream) {
    if (!p || !out_rgba) return MRIRT_ERR_NULL;
    if (p->maxSteps > (1u << 20)) return MRIRT_ERR_ARG;             // raymarch.slang:70: `i < maxSteps` iterations per pixel
    K3Args a;
    fill_camera(a.cam, p->gEye, p->gU, p->gV, p->gW, p->fovY, width, height, nullptr, true);
    int rc = fill_pixel_map(a.map, width, height, pitch_px, nullptr);
    if (rc != MRIRT_OK) return rc;
    a.maxSteps = p->maxSteps; a.maxDistance = p->maxDistance; a.hitThreshold = p->hitThreshold;
    a.out = out_rgba;
    hipLaunchKernelGGL(sdf_march_kernel, dim3(a.map.chunk * kXcds), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}
