// K1: the BraTS volume ray-marcher — hand-written gfx950 HIP replacement for the Slang compute
// shader `brats_main` (reference: inr/viewer/brats_rt.slang:85-168; helpers :36-83).
//
// One lane = one ray; one wave64 = an 8x8 pixel packet (the reference's numthreads(8,8,1)
// group), one 256-thread workgroup = 2x2 packets.  The march loop is a plain divergent loop:
// the wave's EXEC mask IS the ballot of live rays, and the backend leaves the loop with
// s_cbranch_execz when the last lane of the packet has terminated (t >= t1 or T <= 0.01), so
// every lane stops accumulating exactly where the scalar shader does.
//
// Template axes: STRICT (bit-faithful unfused fp32 / FAST FMA + hardware exp2), LAYOUT (the
// reference's linear grid / 4x4x2 bricks), SHADE (lattice-gradient Blinn-Phong extension),
// HALF (rgba16_float output like the reference's texture).
#include "mrirt_host.h"

namespace mrirt {

struct K1Args {
    Camera cam;
    PixelMap map;
    GridDims grid;
    float bmin[3], bmax[3], voxelSize[3], invVoxel[3];
    float hiLin[3];          // float(dims) - 1.001f   (sampleLinear clamp)
    float hiLab[3];          // float(dims) - 1.0f     (sampleLabel clamp)
    float stepSize, nearT, farT;
    float bg[3];
    uint32_t enabled[4];
    float weight[4];
    float tfLo;              // wl - ww*0.5
    float ww, intensityAlpha, gamma;
    uint32_t showSeg, showPred;
    float lut[8][4];
    float ka, kd, ks, gradEps, ert;
    uint32_t specPow2;
    const float* vol[4];
    const uint32_t* labels;
    const uint32_t* preds;
    void* out;
    uint64_t* stats;
};

template <bool STRICT, int LAYOUT, bool SHADE>
__device__ __forceinline__ void sample_channel(const float* __restrict__ buf, const GridDims& gd,
                                               uint32_t ix, uint32_t iy, uint32_t iz,
                                               float fx, float fy, float fz, float& v, float g[3]) {
    using A = Addr<LAYOUT>;
    using Mm = M<STRICT>;
    const uint32_t x0 = A::ox(gd, ix), x1 = A::ox(gd, ix + 1);
    const uint32_t y0 = A::oy(gd, iy), y1 = A::oy(gd, iy + 1);
    const uint32_t z0 = A::oz(gd, iz), z1 = A::oz(gd, iz + 1);
    // core 2x2x2 (sampleLinear, brats_rt.slang:69-72)
    const float c000 = buf[x0 + y0 + z0], c100 = buf[x1 + y0 + z0];
    const float c010 = buf[x0 + y1 + z0], c110 = buf[x1 + y1 + z0];
    const float c001 = buf[x0 + y0 + z1], c101 = buf[x1 + y0 + z1];
    const float c011 = buf[x0 + y1 + z1], c111 = buf[x1 + y1 + z1];
    v = Mm::lerp(Mm::lerp(Mm::lerp(c000, c100, fx), Mm::lerp(c010, c110, fx), fy),
                 Mm::lerp(Mm::lerp(c001, c101, fx), Mm::lerp(c011, c111, fx), fy), fz);
    if constexpr (SHADE) {
        // 24 more voxels: the +-1 neighbours of the 8 corners along each axis (indices clamped)
        const uint32_t xm = A::ox(gd, ix > 0 ? ix - 1 : 0), xp = A::ox(gd, min(ix + 2, gd.X - 1));
        const uint32_t ym = A::oy(gd, iy > 0 ? iy - 1 : 0), yp = A::oy(gd, min(iy + 2, gd.Y - 1));
        const uint32_t zm = A::oz(gd, iz > 0 ? iz - 1 : 0), zp = A::oz(gd, min(iz + 2, gd.Z - 1));
        {   // d/dx: corner (0,dy,dz): v[i+1]-v[i-1]; corner (1,dy,dz): v[i+2]-v[i]
            const float d000 = c100 - buf[xm + y0 + z0], d100 = buf[xp + y0 + z0] - c000;
            const float d010 = c110 - buf[xm + y1 + z0], d110 = buf[xp + y1 + z0] - c010;
            const float d001 = c101 - buf[xm + y0 + z1], d101 = buf[xp + y0 + z1] - c001;
            const float d011 = c111 - buf[xm + y1 + z1], d111 = buf[xp + y1 + z1] - c011;
            g[0] = Mm::lerp(Mm::lerp(Mm::lerp(d000, d100, fx), Mm::lerp(d010, d110, fx), fy),
                            Mm::lerp(Mm::lerp(d001, d101, fx), Mm::lerp(d011, d111, fx), fy), fz);
        }
        {   // d/dy
            const float d000 = c010 - buf[x0 + ym + z0], d100 = c110 - buf[x1 + ym + z0];
            const float d010 = buf[x0 + yp + z0] - c000, d110 = buf[x1 + yp + z0] - c100;
            const float d001 = c011 - buf[x0 + ym + z1], d101 = c111 - buf[x1 + ym + z1];
            const float d011 = buf[x0 + yp + z1] - c001, d111 = buf[x1 + yp + z1] - c101;
            g[1] = Mm::lerp(Mm::lerp(Mm::lerp(d000, d100, fx), Mm::lerp(d010, d110, fx), fy),
                            Mm::lerp(Mm::lerp(d001, d101, fx), Mm::lerp(d011, d111, fx), fy), fz);
        }
        {   // d/dz
            const float d000 = c001 - buf[x0 + y0 + zm], d100 = c101 - buf[x1 + y0 + zm];
            const float d010 = c011 - buf[x0 + y1 + zm], d110 = c111 - buf[x1 + y1 + zm];
            const float d001 = buf[x0 + y0 + zp] - c000, d101 = buf[x1 + y0 + zp] - c100;
            const float d011 = buf[x0 + y1 + zp] - c010, d111 = buf[x1 + y1 + zp] - c110;
            g[2] = Mm::lerp(Mm::lerp(Mm::lerp(d000, d100, fx), Mm::lerp(d010, d110, fx), fy),
                            Mm::lerp(Mm::lerp(d001, d101, fx), Mm::lerp(d011, d111, fx), fy), fz);
        }
    }
}

template <int LAYOUT>
__device__ __forceinline__ uint32_t sample_label(const uint32_t* __restrict__ buf, const GridDims& gd,
                                                 const float q[3], const float hi[3]) {
    using A = Addr<LAYOUT>;
    // sampleLabel, brats_rt.slang:78-83; roundf = half away from zero (Metal round)
    const uint32_t ix = (uint32_t)roundf(clampf(q[0], 0.0f, hi[0]));
    const uint32_t iy = (uint32_t)roundf(clampf(q[1], 0.0f, hi[1]));
    const uint32_t iz = (uint32_t)roundf(clampf(q[2], 0.0f, hi[2]));
    return buf[A::ox(gd, ix) + A::oy(gd, iy) + A::oz(gd, iz)];
}

template <bool STRICT, int LAYOUT, bool SHADE, bool HALF>
__global__ __launch_bounds__(256) void brats_march_kernel(const K1Args a) {
    using Mm = M<STRICT>;
    uint32_t px, py;
    int64_t oidx;
    const int kind = map_pixel(a.map, px, py, oidx);
    if (a.map.numBlocks == 0) return;

    float C0 = a.bg[0], C1 = a.bg[1], C2 = a.bg[2];
    uint32_t nLive = 0, nShaded = 0;

    if (kind == 1) {
        float ro[3], rd[3];
        primary_ray(a.cam, px, py, ro, rd);
        // slab test, brats_rt.slang:95-102 (rcp uses the nudged direction, marching the true one)
        float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = fabsf(rd[k]) < 1e-6f ? 1e-6f : rd[k];
            const float rcp = 1.0f / d;
            const float t0 = (a.bmin[k] - ro[k]) * rcp, t1 = (a.bmax[k] - ro[k]) * rcp;
            tmin = fmaxf(tmin, fminf(t0, t1));
            tmax = fminf(tmax, fmaxf(t0, t1));
        }
        const bool hit = tmax >= fmaxf(tmin, 0.0f);
        const float t0 = fmaxf(tmin, fmaxf(0.0f, a.nearT));
        const float t1 = fminf(tmax, a.farT > 0.0f ? a.farT : tmax);
        if (hit && !(t1 <= t0)) {
            float T = 1.0f, t = t0;
            while (t < t1 && T > a.ert) {
                float q[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float p = Mm::mad(t, rd[k], ro[k]);            // o + t*d (sum is commutative)
                    q[k] = STRICT ? (p - a.bmin[k]) / a.voxelSize[k] : (p - a.bmin[k]) * a.invVoxel[k];
                }
                // sampleLinear's clamp/floor/fract is identical for all four modalities
                const float cx = clampf(q[0], 0.0f, a.hiLin[0]);
                const float cy = clampf(q[1], 0.0f, a.hiLin[1]);
                const float cz = clampf(q[2], 0.0f, a.hiLin[2]);
                const float flx = floorf(cx), fly = floorf(cy), flz = floorf(cz);
                const uint32_t ix = (uint32_t)flx, iy = (uint32_t)fly, iz = (uint32_t)flz;
                const float fx = cx - flx, fy = cy - fly, fz = cz - flz;

                float v = 0.0f, wSum = 0.0f, g[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (a.enabled[m] != 0) {
                        float s, gm[3];
                        sample_channel<STRICT, LAYOUT, SHADE>(a.vol[m], a.grid, ix, iy, iz, fx, fy, fz, s, gm);
                        v = Mm::mad(s, a.weight[m], v);
                        wSum += a.weight[m];
                        if constexpr (SHADE) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) g[k] = Mm::mad(gm[k], a.weight[m], g[k]);
                        }
                    }
                }
                if (wSum > 0.0f) {
                    v = v / wSum;
                    if constexpr (SHADE) { g[0] = g[0] / wSum; g[1] = g[1] / wSum; g[2] = g[2] / wSum; }
                }
                // transfer function, brats_rt.slang:132-133
                float val = satf(Mm::div(v - a.tfLo, a.ww));
                val = Mm::pow(val, a.gamma);
                ++nLive;
                if (val > 0.0f) {
                    const float alpha = 1.0f - Mm::exp(-(val * a.intensityAlpha) * a.stepSize);
                    float emis = val;
                    if constexpr (SHADE) {
                        const float gx = STRICT ? (g[0] * 0.5f) / a.voxelSize[0] : (g[0] * 0.5f) * a.invVoxel[0];
                        const float gy = STRICT ? (g[1] * 0.5f) / a.voxelSize[1] : (g[1] * 0.5f) * a.invVoxel[1];
                        const float gz = STRICT ? (g[2] * 0.5f) / a.voxelSize[2] : (g[2] * 0.5f) * a.invVoxel[2];
                        const float glen = sqrtf(dot3(gx, gy, gz, gx, gy, gz));
                        float shade = a.ka + a.kd;
                        if (glen > a.gradEps) {
                            const float ndl = fminf(fabsf(dot3(gx / glen, gy / glen, gz / glen, rd[0], rd[1], rd[2])), 1.0f);
                            float spec = ndl;
                            for (uint32_t s = 0; s < a.specPow2; ++s) spec = spec * spec;
                            shade = (a.ka + a.kd * ndl) + a.ks * spec;
                        }
                        emis = val * shade;
                        ++nShaded;
                    }
                    const float c = (alpha * T) * emis;
                    C0 += c; C1 += c; C2 += c;
                    T *= (1.0f - alpha);
                }
                if (a.showSeg != 0) {                                  // brats_rt.slang:143-151
                    const uint32_t l = sample_label<LAYOUT>(a.labels, a.grid, q, a.hiLab);
                    if (l > 0 && l < 8) {
                        const float alpha = 1.0f - Mm::exp(-a.lut[l][3] * a.stepSize);
                        const float at = alpha * T;
                        C0 += at * a.lut[l][0]; C1 += at * a.lut[l][1]; C2 += at * a.lut[l][2];
                        T *= (1.0f - alpha);
                    }
                }
                if (a.showPred != 0) {                                 // brats_rt.slang:154-162
                    const uint32_t l = sample_label<LAYOUT>(a.preds, a.grid, q, a.hiLab);
                    if (l > 0 && l < 8) {
                        const float alpha = 1.0f - Mm::exp(-a.lut[l][3] * a.stepSize * 1.5f);
                        const float at = alpha * T;
                        C0 += at * a.lut[l][0]; C1 += at * a.lut[l][1]; C2 += at * a.lut[l][2];
                        T *= (1.0f - alpha);
                    }
                }
                t += a.stepSize;
            }
        }
    }
    if (kind != 0) store_rgba<HALF>(a.out, oidx, C0, C1, C2, 1.0f);
    if (a.stats != nullptr) {
        wave_count_add(a.stats + 0, nLive);
        wave_count_add(a.stats + 1, nShaded);
    }
}

template <bool STRICT, int LAYOUT, bool SHADE>
static int launch_half(const K1Args& a, bool half, hipStream_t s) {
    const dim3 grid(a.map.chunk * kXcds), block(256);
    if (half) hipLaunchKernelGGL((brats_march_kernel<STRICT, LAYOUT, SHADE, true>), grid, block, 0, s, a);
    else      hipLaunchKernelGGL((brats_march_kernel<STRICT, LAYOUT, SHADE, false>), grid, block, 0, s, a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <bool STRICT, int LAYOUT>
static int launch_shade(const K1Args& a, bool shade, bool half, hipStream_t s) {
    return shade ? launch_half<STRICT, LAYOUT, true>(a, half, s) : launch_half<STRICT, LAYOUT, false>(a, half, s);
}

}  // namespace mrirt

using namespace mrirt;

extern "C" int mrirt_render_brats_ex(const MrirtBratsParams* p, const MrirtRenderExt* ext,
                                     const void* const vol[4], const void* labels, const void* preds,
                                     void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream) {
    if (!p || !out_rgba || !vol) return MRIRT_ERR_NULL;
    for (int k = 0; k < 3; ++k) if (p->dims[k] < 2) return MRIRT_ERR_DIMS;
    const uint32_t layout = ext ? ext->layout : (uint32_t)MRIRT_LAYOUT_LINEAR;
    const uint32_t math = ext ? ext->math : (uint32_t)MRIRT_MATH_STRICT;
    const uint32_t fmt = ext ? ext->outFormat : (uint32_t)MRIRT_OUT_RGBA32F;
    if (layout > MRIRT_LAYOUT_BRICK || math > MRIRT_MATH_FAST || fmt > MRIRT_OUT_RGBA16F) return MRIRT_ERR_LAYOUT;
    if (mrirt_brick_elems(p->dims) >= (int64_t)1 << 32) return MRIRT_ERR_DIMS;   // 32-bit element offsets
    for (int m = 0; m < 4; ++m) if (p->volEnabled[m] != 0 && !vol[m]) return MRIRT_ERR_NULL;
    if ((p->showSeg != 0 && !labels) || (p->showPred != 0 && !preds)) return MRIRT_ERR_NULL;

    K1Args a;
    fill_camera(a.cam, p->eye, p->U, p->V, p->W, p->fovY, p->imageSize[0], p->imageSize[1], ext, false);
    int rc = fill_pixel_map(a.map, p->imageSize[0], p->imageSize[1], pitch_px, ext);
    if (rc != MRIRT_OK) return rc;
    fill_grid_dims(a.grid, p->dims, layout);
    for (int k = 0; k < 3; ++k) {
        a.bmin[k] = p->volMin[k];
        a.bmax[k] = p->volMin[k] + p->voxelSize[k] * (float)p->dims[k];
        a.voxelSize[k] = p->voxelSize[k];
        a.invVoxel[k] = 1.0f / p->voxelSize[k];
        a.hiLin[k] = (float)p->dims[k] - 1.001f;
        a.hiLab[k] = (float)p->dims[k] - 1.0f;
        a.bg[k] = p->bgColor[k];
    }
    a.stepSize = p->stepSize; a.nearT = p->nearT; a.farT = p->farT;
    for (int m = 0; m < 4; ++m) {
        a.enabled[m] = p->volEnabled[m];
        a.weight[m] = p->volWeight[m];
        a.vol[m] = static_cast<const float*>(vol[m]);
    }
    a.tfLo = p->wl - p->ww * 0.5f;
    a.ww = p->ww; a.intensityAlpha = p->intensityAlpha; a.gamma = p->gamma;
    a.showSeg = p->showSeg; a.showPred = p->showPred;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) a.lut[i][j] = p->lutColorAlpha[i][j];
    const bool shade = ext && ext->shadeMode != 0;
    a.ka = ext ? ext->ka : 0.0f; a.kd = ext ? ext->kd : 0.0f; a.ks = ext ? ext->ks : 0.0f;
    a.gradEps = ext ? ext->gradEps : 0.0f;
    a.specPow2 = ext ? ext->specPow2 : 0u;
    a.ert = (ext && ext->ertOverride) ? ext->ertThreshold : 0.01f;   // brats_rt.slang:117
    a.labels = static_cast<const uint32_t*>(labels);
    a.preds = static_cast<const uint32_t*>(preds);
    a.out = out_rgba;
    a.stats = stats_dev;
    if (a.map.numBlocks == 0) return MRIRT_OK;   // a rank that owns no tile

    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool half = fmt == MRIRT_OUT_RGBA16F;
    if (math == MRIRT_MATH_STRICT)
        return layout == MRIRT_LAYOUT_LINEAR ? launch_shade<true, 0>(a, shade, half, s) : launch_shade<true, 1>(a, shade, half, s);
    return layout == MRIRT_LAYOUT_LINEAR ? launch_shade<false, 0>(a, shade, half, s) : launch_shade<false, 1>(a, shade, half, s);
}

extern "C" int mrirt_render_brats(const MrirtBratsParams* params, const float* const vol[4],
                                  const uint32_t* labels, const uint32_t* preds,
                                  float* out_rgba, int64_t pitch_px, void* stream) {
    if (!vol) return MRIRT_ERR_NULL;
    const void* v[4] = { vol[0], vol[1], vol[2], vol[3] };
    return mrirt_render_brats_ex(params, nullptr, v, labels, preds, out_rgba, pitch_px, nullptr, stream);
}
