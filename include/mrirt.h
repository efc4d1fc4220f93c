/* mrirt.h — C ABI of the MI355X-native MRI volume ray-marcher (libmrirt.so).
 *
 * This is the drop-in boundary for the reference's render call.  The reference
 * (klukaszek/MRI-RayTracer) has no FFI of its own: its render call is the third-party slangpy
 *     kernel.dispatch(thread_count=[W,H,1], vars={...}, command_encoder=ce)
 * at inr/viewer/brats_viewer.py:431-442, scripts/volumeRendering/app.py:350-358 and
 * scripts/raymarch/app.py:212-223, binding buffers BY NAME to the Slang globals.  Each entry
 * point below replaces one of those dispatches: same inputs (the Slang cbuffer as a POD, the
 * StructuredBuffers as plain device pointers), same output (one RGBA pixel per thread).
 *
 * Rules of the boundary:
 *   - extern "C", plain pointers and sizes, no torch / C++ types;
 *   - every data pointer is a DEVICE pointer owned by the caller; nothing is retained;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and NOT synchronised;
 *   - returns MRIRT_OK (0) or a negative MrirtStatus; never throws across the boundary;
 *   - parameter structs are read on the host during the call (they may live on the stack).
 */
#ifndef MRIRT_H
#define MRIRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: MRIRT_LAYOUT_LABCELL + mrirt_build_label_cells, MrirtInrDesc.flags / tieSigmas (they replace the process-environment
 *    switches of version 2), mrirt_brats_skip_applicable, mrirt_install_abort_trace. */
/* 4: MRIRT_LAYOUT_MOD4 + mrirt_build_mod4_grid (the four modalities of a BraTS case as ONE float4 grid: unshaded K1 and
 *    mrirt_render_brats_inr); the packed INR image ends in 32 KiB of slack more (mrirt_inr_pack_bytes says how much to allocate). */
#define MRIRT_ABI_VERSION 4

typedef enum MrirtStatus {
    MRIRT_OK = 0,
    MRIRT_ERR_NULL = -1,       /* a required pointer is NULL                         */
    MRIRT_ERR_DIMS = -2,       /* volume dims < 2 on an axis, or image size 0        */
    MRIRT_ERR_LAYOUT = -3,     /* unknown layout / dtype / mode                      */
    MRIRT_ERR_LAUNCH = -4,     /* hipLaunch / HIP runtime error (see mrirt_last_hip_error) */
    MRIRT_ERR_ARG = -5,        /* inconsistent argument (pitch < width, bad tile spec, ...) */
    MRIRT_ERR_NO_DEVICE = -6   /* no gfx950 device visible                           */
} MrirtStatus;

/* ------------------------------------------------------------------------------------ */
/* K1  brats_main                                                                        */
/* ------------------------------------------------------------------------------------ */

/* Field-for-field mirror of `struct Params`, inr/viewer/brats_rt.slang:12-31, including
 * its hand padding (so the 16-byte cbuffer layout is preserved).                        */
typedef struct MrirtBratsParams {
    uint32_t imageSize[2]; float fovY; float pad0;
    float eye[3]; float pad1;
    float U[3]; float pad2; float V[3]; float pad3; float W[3]; float pad4;
    float volMin[3]; float pad5; float voxelSize[3]; float pad6; uint32_t dims[3]; uint32_t pad7;
    float stepSize; float nearT; float farT; float pad8;
    float bgColor[3]; float pad9;
    uint32_t volEnabled[4];
    float volWeight[4];
    float ww; float wl; float intensityAlpha; float padInt;
    float gamma; float gradBoost; float gradScale; float padTone;   /* gradBoost/gradScale: unread, as in the shader */
    uint32_t showSeg;
    uint32_t showPred;
    uint32_t padFlags[2];
    float lutColorAlpha[8][4];
} MrirtBratsParams;

/* Grid storage layouts.  LINEAR is the reference's (x fastest: x + y*X + z*X*Y,
 * inr/viewer/brats_viewer.py:64).  BRICK is this library's HBM layout (DESIGN.md "Data
 * layout"): 4x4x2-voxel bricks, one 128-byte line per fp32 brick, produced by mrirt_brick_*. */
typedef enum MrirtLayout {
    MRIRT_LAYOUT_LINEAR = 0,
    MRIRT_LAYOUT_BRICK = 1,
    /* fp32 intensity grids only: one float4 per voxel, 2x2x2-voxel bricks (8 x 16 B = one 128-B
     * line).  VG   = (v, dv/dx, dv/dy, dv/dz) with the lattice central differences
     *                v[clamp(c+e)] - v[clamp(c-e)] precomputed at load time (mrirt_build_vec4_grid):
     *                a gradient-shaded sample is 8 dwordx4 gathers instead of 32 dword gathers,
     *                bit-identical arithmetic.
     *        QUAD = (v[x,y], v[x,y+1], v[x+1,y], v[x+1,y+1]) (indices clamped): an unshaded
     *                trilinear sample is 2 dwordx4 gathers instead of 8 dword gathers.          */
    MRIRT_LAYOUT_VG = 2,
    MRIRT_LAYOUT_QUAD = 3,
    /* VGA = the VG voxels stored three times, in 128-B bricks one voxel thick along x, y or z (1x4x2 / 4x1x2 /
     *       4x2x1 voxels).  A ray packet's samples of one march step lie on a sheet parallel to the face the rays
     *       entered through; the kernel reads the copy whose bricks are flat in that direction (chosen per packet),
     *       so a gather touches about half the cache lines of the 2x2x2 bricks.  Same bits as VG; 3x the memory
     *       (6 GiB for a 512^3 channel: sized for 288 GB of HBM).  mrirt_vga_elems / mrirt_build_vec4_grid.        */
    MRIRT_LAYOUT_VGA = 4,
    /* Label grids only (labelLayout), with QUAD intensity grids: BOTH overlays' nearest labels per CELL.  sampleLabel's
     * rounded voxel (brats_rt.slang:78-83) is always one of the eight corners of the cell sampleLinear blends
     * (round(clamp(q, 0, d-1)) - floor(clamp(q, 0, d-1.001)) is 0 or 1 per axis), and the shader draws labels 1..7 only
     * (:145,156), so one 8-byte element per voxel holds what a sample needs of both grids: .x = the ground-truth labels of
     * the cell's corners as eight nibbles (corner (dx,dy,dz) in bits 4 (dx + 2 dy + 4 dz) .., neighbours clamped to the grid,
     * labels >= 8 stored as 8), .y = the prediction's likewise; elements in the QUAD grid's order (mrirt_vec4_elems).  A
     * sample then takes ONE 8-byte gather at the offset its intensity taps already have instead of two nearest-voxel
     * gathers.  mrirt_build_label_cells makes it; `labels` points at it, `preds` is ignored.                              */
    MRIRT_LAYOUT_LABCELL = 5,
    /* MOD4 = the FOUR modalities of one case interleaved: float4 (gIntensity0..3)[voxel] in the VG grid's element order
     *        (2x2x2-voxel bricks of 8 x 16 B, mrirt_vec4_elems).  An unshaded sample of a multi-modality frame — the reference
     *        viewer's own frame, and the per-sample network of mrirt_render_brats_inr — takes eight 16-byte gathers for the eight
     *        corners of ALL four modalities (the same 128 B per sample as four QUAD grids) from a grid a quarter of the size:
     *        268 MB instead of 1.07 GB for 256^3 x 4, one set of cache lines per sample instead of four.  Trilinear arithmetic
     *        unchanged (sampleLinear's order): same bits.  mrirt_build_mod4_grid makes it; vol[0] points at it (vol[1..3] are
     *        ignored); which modalities are drawn stays volEnabled; label grids as before or as MRIRT_LAYOUT_LABCELL.
     *        mrirt_render_brats_ex / _skip (the pipelined march) and mrirt_render_brats_inr.  MRIRT_ERR_LAYOUT with gradient
     *        shading (no gradients in it), with a class stream (mrirt_render_brats_stream), and for grids of 4 GiB or label
     *        grids of 2^30 elements upwards.                                                                                 */
    MRIRT_LAYOUT_MOD4 = 6
} MrirtLayout;

typedef enum MrirtMath {
    MRIRT_MATH_STRICT = 0,  /* unfused fp32 in the oracle's order, fp64-backed exp/pow: bit-faithful */
    MRIRT_MATH_FAST = 1     /* FMA contraction, hardware exp2/rcp; within the 1e-4 image tolerance   */
} MrirtMath;

typedef enum MrirtOutFormat { MRIRT_OUT_RGBA32F = 0, MRIRT_OUT_RGBA16F = 1 /* reference's rgba16_float */ } MrirtOutFormat;

/* Build-defined extensions (no reference counterpart).  A NULL pointer
 * or an all-zero struct selects exactly the reference's behaviour.                         */
typedef struct MrirtRenderExt {
    uint32_t cameraMode;        /* 0 perspective (brats_rt.slang:36-46), 1 orthographic        */
    float    orthoHalfHeight;
    uint32_t shadeMode;         /* 0 off, 1 lattice central-difference gradient + Blinn-Phong  */
    float    ka, kd, ks;
    uint32_t specPow2;          /* specular exponent 2^specPow2 by repeated squaring           */
    float    gradEps;
    uint32_t ertOverride;       /* 0: the reference's T > 0.01; 1: use ertThreshold            */
    float    ertThreshold;
    uint32_t math;              /* MrirtMath                                                   */
    uint32_t outFormat;         /* MrirtOutFormat                                              */
    uint32_t layout;            /* MrirtLayout of ALL bound fp32 intensity grids               */
    uint32_t labelLayout;       /* MrirtLayout (LINEAR, BRICK or LABCELL) of the labels / preds grids */
    /* Image-tile sharding (one process per GPU).  tileSize == 0: whole image into
     * out[y*pitch + x].  Otherwise this call renders the tiles t with t % tileWorld ==
     * tileRank (t = ty*tilesX + tx, tiles of tileSize^2 pixels) into a COMPACT buffer
     * out[local_tile][tileSize][tileSize][4]; pitch is ignored.                            */
    uint32_t tileSize, tileRank, tileWorld;
    uint32_t kernelVariant;     /* 0 = library default; others select experimental kernels (bench/tests) */
    /* Tile sharding, load balance: the dealt tile t sits at row ty = t / tilesX, column tx = (t % tilesX + tileSkew * ty) % tilesX
     * (0: tx = t % tilesX, the plain row-major deal).  When tilesX is a multiple of tileWorld the plain deal hands every rank
     * whole tile COLUMNS (rank = tx % world); a skew of (tilesX - 1) % tileWorld turns that into diagonals, rank = (tx + ty) % world,
     * so every rank draws from every column and row of the image.  mrirt_detile takes the same value.                         */
    uint32_t tileSkew;
    uint32_t reserved;
} MrirtRenderExt;

/* Argument checks shared by every K1 entry point (MRIRT_ERR_ARG, nothing is launched): stepSize must be a
 * finite positive number that still advances t in fp32 at the far end of the box (t + stepSize > t) and may
 * not cut the box diagonal into more than 2^20 steps; voxelSize finite and > 0; volMin / eye finite.  The
 * reference's UI clamps its slider to >= 0.001 (inr/viewer/brats_viewer.py:168); the shader itself would
 * spin.  K2: stepCount finite and <= 2^20, near / far finite.  K3: maxSteps <= 2^20.                      */

/* Drop-in for kernel.dispatch(...) of brats_main, inr/viewer/brats_viewer.py:431-442.
 *   vol[m]  = gIntensity<m>  (fp32, X*Y*Z, LINEAR; may be NULL when volEnabled[m] == 0)
 *   labels  = gLabels, preds = gPreds (uint32 per voxel; may be NULL when showSeg/showPred == 0)
 *   out_rgba= gOutput as fp32 RGBA, pitch_px pixels per row (>= imageSize[0])             */
int mrirt_render_brats(const MrirtBratsParams* params, const float* const vol[4],
                       const uint32_t* labels, const uint32_t* preds,
                       float* out_rgba, int64_t pitch_px, void* stream);

/* As above with extensions; grids are in ext->layout, out in ext->outFormat.
 * stats_dev (optional): 2 device uint64 counters, atomically incremented by
 * {live samples, gradient-shaded samples} — used for sample accounting, not timed.       */
int mrirt_render_brats_ex(const MrirtBratsParams* params, const MrirtRenderExt* ext,
                          const void* const vol[4], const void* labels, const void* preds,
                          void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream);

/* Host CPU twin? No: the product has no CPU fallback.  The CPU restatement lives in oracle/. */

/* Per-sample INR query along the rays (build-defined, BASELINE config 5): the label
 * of the prediction overlay (brats_rt.slang:154-162) comes from an MLP evaluated AT each sample
 * instead of sampleLabel(gPreds).  Three device passes around mrirt_inr_forward:
 *   1. mrirt_brats_sample_counts: counts[y*W + x] = steps ray (x,y) takes in [t0,t1) (no ERT: the
 *      classes are needed to know T).  Caller: exclusive prefix sum -> offsets (int64), total.
 *   2. mrirt_brats_emit_samples: MLP inputs of sample k of ray p at row offsets[p] + k:
 *      coords[row][3] = 2*clamp(pIdx,0,dim-1)/(dim-1) - 1 (== predict_volume's coordinate at lattice
 *      points, inr/inr/model.py:124-128), feats[row][4] = the four trilinear samples z-scored as
 *      (v - zmu[m]) / zsigma[m] (inr/viewer/brats_viewer.py:281-287).  All four grids must be bound.
 *      Caller: mrirt_inr_forward(desc, coords, feats, total, NULL, classes).
 *   3. mrirt_render_brats_stream: K1 as mrirt_render_brats_ex, the prediction label of sample k of
 *      ray p read from classes[offsets[p] + k] (requires params->showPred != 0; whole-frame only). */
int mrirt_brats_sample_counts(const MrirtBratsParams* params, const MrirtRenderExt* ext, uint32_t* counts, void* stream);
int mrirt_brats_emit_samples(const MrirtBratsParams* params, const MrirtRenderExt* ext, const void* const vol[4],
                             const float zmu[4], const float zsigma[4], const int64_t* offsets,
                             float* coords, float* feats, void* stream);
int mrirt_render_brats_stream(const MrirtBratsParams* params, const MrirtRenderExt* ext,
                              const void* const vol[4], const void* labels,
                              const int16_t* classes, const int64_t* offsets,
                              void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream);

/* The one-call, chunked and ERT-aware form of this render is mrirt_render_brats_inr (declared after the INR section). */

/* ------------------------------------------------------------------------------------ */
/* Brick layout conversion (load-time; replaces create_buffer + copy_from_numpy,         */
/* inr/viewer/brats_viewer.py:219-230)                                                   */
/* ------------------------------------------------------------------------------------ */

/* Number of ELEMENTS (not bytes) a bricked grid of `dims` occupies. */
int64_t mrirt_brick_elems(const uint32_t dims[3]);
/* linear (x fastest) -> bricked; elem_bytes in {1,4}; dst holds mrirt_brick_elems(dims) elements. */
int mrirt_brick_grid(const void* linear, void* bricked, const uint32_t dims[3], uint32_t elem_bytes, void* stream);
/* inverse, for round-trip tests */
int mrirt_unbrick_grid(const void* bricked, void* linear, const uint32_t dims[3], uint32_t elem_bytes, void* stream);
/* Number of float4 ELEMENTS a VG / QUAD grid of `dims` occupies. */
int64_t mrirt_vec4_elems(const uint32_t dims[3]);
/* Number of float4 ELEMENTS the three copies of a VGA grid of `dims` occupy together. */
int64_t mrirt_vga_elems(const uint32_t dims[3]);
/* linear fp32 (x fastest) -> VG, QUAD or VGA float4 grid (layout = MRIRT_LAYOUT_VG / _QUAD / _VGA). */
int mrirt_build_vec4_grid(const float* linear, void* vec4_grid, const uint32_t dims[3], uint32_t layout, void* stream);
/* four linear fp32 grids (gIntensity0..3, brats_viewer.py:64-69; all required) -> the MRIRT_LAYOUT_MOD4 grid:
 * mrirt_vec4_elems(dims) float4 elements, device. */
int mrirt_build_mod4_grid(const float* const linear[4], void* mod4_grid, const uint32_t dims[3], void* stream);
/* linear uint32 label grids (the reference's gLabels / gPreds uploads, brats_viewer.py:71-73; either may be NULL = no labels)
 * -> the MRIRT_LAYOUT_LABCELL grid: mrirt_vec4_elems(dims) elements of 8 bytes, device. */
int mrirt_build_label_cells(const uint32_t* seg_linear, const uint32_t* pred_linear, const uint32_t dims[3], void* cells, void* stream);
/* BC4 / RGTC1-unorm slices (8-byte blocks, [depth][ceil(h/4)][ceil(w/4)], device, 8-byte aligned) -> u8 voxels
 * [depth][height][width]: the decode scripts/volumeRendering/app.py:200-250 does on the host. */
int mrirt_bc4_decode(const void* blocks, uint32_t width, uint32_t height, uint32_t depth, uint8_t* out_u8, void* stream);

/* ------------------------------------------------------------------------------------ */
/* K2  volume_cs                                                                         */
/* ------------------------------------------------------------------------------------ */

/* Mirror of `struct Params`, scripts/volumeRendering/volume_render.slang:9-21. */
typedef struct MrirtVolumeParams {
    uint32_t imageSize[2]; float fovY; float stepCount;
    float nearPlane; float farPlane;
    float eye[3]; float padEye; float U[3]; float padU; float V[3]; float padV; float W[3]; float padW;
    uint32_t volDim[3]; uint32_t padDim;
} MrirtVolumeParams;

typedef enum MrirtVoxelMode {
    MRIRT_VOX_U32X4 = 0,   /* reference: StructuredBuffer<uint4>, one u32 per u8 voxel (app.py:150-153) */
    MRIRT_VOX_U8 = 1,      /* real bytes (1 B/voxel)                                                     */
    MRIRT_VOX_F32 = 2,     /* fp32 grid (build-defined generalisation)                    */
    MRIRT_VOX_CELL8 = 3    /* 8 B/voxel: the eight bytes of the trilinear cell based at the voxel, built */
                           /* once by mrirt_build_cell8 — one gather per sample, same bytes, same frame  */
} MrirtVoxelMode;

/* Drop-in for kernel.dispatch(...) of volume_cs, scripts/volumeRendering/app.py:350-358.
 * volume = gVolumeU8 in `mode`; ext may be NULL (uses cameraMode/orthoHalfHeight/math/outFormat/tiles). */
int mrirt_render_volume(const MrirtVolumeParams* params, const MrirtRenderExt* ext, const void* volume,
                        uint32_t mode, void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream);
/* load time: u8 voxels (src_mode MRIRT_VOX_U8, or the reference's one-u32-per-voxel MRIRT_VOX_U32X4) ->
 * MRIRT_VOX_CELL8 grid of dims[0]*dims[1]*dims[2] 8-byte elements */
int mrirt_build_cell8(const void* voxels, uint32_t src_mode, const uint32_t dims[3], void* cell8, void* stream);

/* ------------------------------------------------------------------------------------ */
/* K3  raymarch_cs                                                                       */
/* ------------------------------------------------------------------------------------ */

/* Mirror of `struct Params` + the four float3 globals, scripts/raymarch/raymarch.slang:7-21. */
typedef struct MrirtSdfParams {
    uint32_t imageSize[2]; float fovY; uint32_t maxSteps;
    float maxDistance; float hitThreshold; float normalEps; float pad0;
    float gEye[3]; float pad1; float gU[3]; float pad2; float gV[3]; float pad3; float gW[3]; float pad4;
} MrirtSdfParams;

/* Drop-in for kernel.dispatch(...) of raymarch_cs, scripts/raymarch/app.py:212-223.
 * width/height are the render_texture dimensions (the shader queries the texture, :64). */
int mrirt_render_sdf(const MrirtSdfParams* params, uint32_t width, uint32_t height,
                     float* out_rgba, int64_t pitch_px, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Tile sharding helpers (multi-GPU)                               */
/* ------------------------------------------------------------------------------------ */
/* Exact empty-space skipping (build-defined acceleration).  The frame and the
 * sample counters are bit-identical to mrirt_render_brats_ex: a sample in a skipped macro cell still
 * counts as a march step, it just fetches and composites nothing, and a macro cell (8^3 voxels) is skipped
 * only when an upper bound of every enabled modality there cannot lift the transfer function above 0 for
 * THIS launch's window / weights and no shown label grid has a label there.  Applies to the VG / QUAD
 * pipelined kernels; any other configuration renders without skipping.                                   */
typedef struct MrirtSkip {
    const float* macroUb[4];      /* per modality: mrirt_build_macro_max output, or NULL (modality unused)   */
    const uint32_t* macroSeg;     /* mrirt_build_macro_labels of gLabels (needed when showSeg)               */
    const uint32_t* macroPred;    /* ... of gPreds (needed when showPred)                                    */
    uint32_t* mask;               /* scratch of maskWords uint32 (8^3 macro-cell bits, then the byte maps of the
                                     empty-radius transform); written by a launch unless mapReady             */
    uint32_t maskWords;           /* size of `mask` in words: must be >= mrirt_skip_mask_words(dims) (checked: a caller
                                     built against an older formula gets MRIRT_ERR_ARG, not device writes past its buffer) */
    uint32_t mapReady;            /* != 0: `mask` still holds the map a previous launch built from the SAME dims, window
                                     (ww, wl), gamma, volEnabled, volWeight, showSeg / showPred, math mode and summaries —
                                     the four pre-pass launches are skipped (a viewer frame changes the camera, not these).
                                     The caller orders that launch before this one (same stream, or an event).            */
} MrirtSkip;
int64_t mrirt_macro_cells(const uint32_t dims[3]);
int64_t mrirt_skip_mask_words(const uint32_t dims[3]);
/* from the LINEAR (x fastest) fp32 / uint32 grids, before any layout conversion */
int mrirt_build_macro_max(const float* linear, const uint32_t dims[3], float* macro_ub, void* stream);
int mrirt_build_macro_labels(const uint32_t* labels_linear, const uint32_t dims[3], uint32_t* macro_any, void* stream);
int mrirt_render_brats_skip(const MrirtBratsParams* params, const MrirtRenderExt* ext,
                            const void* const vol[4], const void* labels, const void* preds,
                            const MrirtSkip* skip, void* out_rgba, int64_t pitch_px,
                            uint64_t* stats_dev, void* stream);
/* Host-only query, nothing is launched: 1 = mrirt_render_brats_skip with these arguments marches with an empty-radius map
 * (it builds one into skip->mask, or trusts skip->mapReady); 0 = it is the plain launch and never reads or writes
 * skip->mask (window width or gamma <= 0, a negative weight, a layout / modality count / math mode without a skipping
 * kernel, more than 256 macro cells on an axis, a missing summary): a caller that caches maps must not mark such a scratch
 * as holding one.  < 0: the MrirtStatus the render call would return for these arguments.                               */
int mrirt_brats_skip_applicable(const MrirtBratsParams* params, const MrirtRenderExt* ext, const void* const vol[4],
                                const void* labels, const void* preds, const MrirtSkip* skip);
/* Host-only query, nothing is launched: which march kernel family mrirt_render_brats_skip (skip != NULL) / mrirt_render_brats_ex
 * (skip == NULL) takes for these arguments — the generic kernel (any layout, 64-bit offsets: grids >= 4 GiB, label grids
 * >= 2^30 elements, BRICK), the software-pipelined one, the rolling one (2-4 shaded modalities), or the opt-in LDS kernels —
 * with MRIRT_KERNEL_SKIPPING / MRIRT_KERNEL_LABEL_CELLS or'ed in.  MRIRT_KERNEL_NONE: a rank that owns no tile.
 * < 0: the MrirtStatus the render call would return.  (Tests and the bench line name the kernel they measured with it.)  */
typedef enum MrirtKernelFamily {
    MRIRT_KERNEL_NONE = 0, MRIRT_KERNEL_GENERIC = 1, MRIRT_KERNEL_PIPELINED = 2, MRIRT_KERNEL_ROLLING = 3,
    MRIRT_KERNEL_SLAB = 4, MRIRT_KERNEL_RING = 5,
    MRIRT_KERNEL_SKIPPING = 16, MRIRT_KERNEL_LABEL_CELLS = 32
} MrirtKernelFamily;
int mrirt_brats_kernel_family(const MrirtBratsParams* params, const MrirtRenderExt* ext, const void* const vol[4],
                              const void* labels, const void* preds, const MrirtSkip* skip);

/* ------------------------------------------------------------------------------------ */
/* number of tiles rank `rank` of `world` renders for a W x H image */
int64_t mrirt_tiles_for_rank(uint32_t width, uint32_t height, uint32_t tileSize, uint32_t rank, uint32_t world);
/* scatter gathered compact tile buffers [world][max_local][ts][ts][4] back into a pitch-linear frame */
int mrirt_detile(const void* gathered, void* frame, uint32_t width, uint32_t height, int64_t pitch_px,
                 uint32_t tileSize, uint32_t world, uint32_t tileSkew, uint32_t outFormat, void* stream);

/* ------------------------------------------------------------------------------------ */
/* INR forward (inr/inr/model.py:11-50,119-141; notebooks/neumors_inr.ipynb:1165-1178)   */
/* ------------------------------------------------------------------------------------ */
typedef enum MrirtInrKind {
    MRIRT_INR_FOURIER_RELU = 0,   /* inputs built on the device: coords, Fourier features, modalities; ReLU */
    MRIRT_INR_SIREN = 1,          /* inputs (coords, modalities); sin activations, first layer scaled by w0 */
    MRIRT_INR_RAW_RELU = 2,       /* feats IS the [n][inDim] input matrix (apply_mlp); ReLU                 */
    MRIRT_INR_RAW_SIREN = 3       /* the same with the SIREN activations                                    */
} MrirtInrKind;

typedef struct MrirtInrDesc {
    uint32_t kind;           /* MrirtInrKind                                                    */
    uint32_t numLayers;      /* linear layers incl. head (<= 8)                                 */
    uint32_t inDim;          /* raw input width (3 + 6K + M for Fourier; 7 for the SIREN)       */
    uint32_t outDim;         /* classes (<= 16)                                                 */
    uint32_t hidden;         /* hidden width (multiple of 32, <= 256)                           */
    uint32_t fourierFreqs;   /* K (Fourier kind)                                                */
    uint32_t numMods;        /* M (<= 8 for kinds 0 and 1: the kernel stages 3 + 8 raw inputs per point) */
    float    w0;             /* SIREN first-layer frequency (30)                                */
    const void* weights;     /* device: packed bf16 weights, see mrirt_inr_pack_bytes           */
    const float* biases;     /* device: fp32 biases, layers concatenated, each padded to its padded out width */
    uint32_t flags;          /* MrirtInrFlags; 0 = the library's defaults                       */
    float    tieSigmas;      /* near-tie mark: top-2 gap < tieSigmas * sqrt(2) * calibrated rms error; 0 = default (3) */
} MrirtInrDesc;

/* A/B switches of the INR forward (measurement and tests; every combination computes the same classes up to the near-tie
 * refinement they switch off).  Version 2 read these from the process environment at every launch.                      */
typedef enum MrirtInrFlags {
    MRIRT_INR_NO_WEIGHT_STATIONARY = 1,  /* 4 x 256 SIRENs take the streaming kernel instead of the weight-stationary one */
    MRIRT_INR_NO_REFINE = 2,             /* no near-tie marking and no second pass: classes are the bf16 pass's           */
    MRIRT_INR_MARK_ONLY = 4              /* mark near-ties (bit 14 of the stored class) but run no second pass            */
} MrirtInrFlags;

/* bytes to allocate for the packed weight buffer of a network shape (0: unsupported shape): the bf16 MFMA image,
 * 64 KiB of slack (the kernel's last weight-chunk prefetch reads and ignores it), the split-bf16 (hi + lo) image the
 * near-tie refinement reads, and a 1 KiB calibration record. */
int64_t mrirt_inr_pack_bytes(const MrirtInrDesc* desc);
/* pack fp32 row-major [in,out] weights (device, layers concatenated unpadded) into `packed`.  The images depend on
 * desc->kind and desc->w0 (the SIREN's w0 / 2 pi and 1 / 2 pi are folded in): pack and forward with the same
 * descriptor.  When desc->weights == packed and desc->biases is set (the usual call), the network is also
 * CALIBRATED here (mrirt_inr_calibrate); this synchronises `stream` once (load time). */
int mrirt_inr_pack_weights(const MrirtInrDesc* desc, const float* w_f32, void* packed, void* stream);
/* Near-tie calibration of a packed network: 8192 pseudo-random inputs through the bf16 pass and through the
 * split-bf16 pass; the rms difference of the logits is stored in the calibration record.  The class outputs
 * (argmax / predict_volume / mrirt_render_brats_inr) then re-evaluate every point whose two largest bf16 logits are
 * closer than 3 sqrt(2) times that error with split-bf16 operands in every layer, so the stored class agrees with an
 * fp32 evaluation except on ties ~1e-5 of the logit range apart.  Without a calibration nothing is re-evaluated. */
int mrirt_inr_calibrate(const MrirtInrDesc* desc, void* stream);
/* logits[n][outDim] (fp32) and/or argmax[n] (int16) for n points.
 * coords[n][3] in [-1,1]; feats[n][numMods].  Either output may be NULL.  logits are the bf16 pass's (those of
 * re-evaluated points, when argmax is requested too, the split-bf16 pass's). */
int mrirt_inr_forward(const MrirtInrDesc* desc, const float* coords, const float* feats, int64_t n,
                      float* logits, int16_t* argmax, void* stream);
/* the same with EVERY point evaluated by the split-bf16 pass (~16-bit operands in every layer, three MFMAs per
 * product): the accuracy reference of the bf16 path, three times its matrix work */
int mrirt_inr_forward_refined(const MrirtInrDesc* desc, const float* coords, const float* feats, int64_t n,
                              float* logits, int16_t* argmax, void* stream);
/* predict_volume (inr/inr/model.py:119-141): mods[M][H][W][D] fp32 -> pred[H][W][D] int16 */
int mrirt_inr_predict_volume(const MrirtInrDesc* desc, const float* mods, const uint32_t hwd[3],
                             int16_t* pred, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Per-sample INR render, one call (BASELINE config 5)                                   */
/* ------------------------------------------------------------------------------------ */
/* The same render as ONE call, chunked and ERT-aware (north star: "all LIVE sample points"): the march advances
 * chunk_steps per pass; each pass emits the MLP inputs of the next <= chunk_steps samples of every ray that is
 * still alive (t < t1 and T > 0.01 after the previous pass), classifies that batch with the MFMA forward and
 * composites it, so a ray that has terminated is not classified any further (a ray that terminates inside a pass
 * wastes at most chunk_steps - 1 queries).  Batch sizes stay in device memory: nothing synchronises with the host.
 *   net      : Fourier/ReLU (MRIRT_INR_FOURIER_RELU) or SIREN (MRIRT_INR_SIREN, the 7-input network of
 *              notebooks/neumors_inr.ipynb:853-899,1165-1178: x = (coords, 4 z-scored modalities)), numMods == 4
 *   scratch  : device memory of mrirt_brats_inr_scratch_bytes(params, chunk_steps) bytes, owned by the caller
 *              (58 B per pixel per step of a pass + 64 B per pixel + 12 KiB: 1.5 GB for 512 x 512 x 96)
 *   stats_dev: optional, THREE device uint64 counters, atomically incremented by
 *              {composited (live) samples, gradient-shaded samples, MLP queries}
 * The frame is bit-identical to mrirt_brats_sample_counts / _emit_samples / mrirt_render_brats_stream run over whole rays (the MLP is batch-position invariant). */
int64_t mrirt_brats_inr_scratch_bytes(const MrirtBratsParams* params, uint32_t chunk_steps);
int mrirt_render_brats_inr(const MrirtBratsParams* params, const MrirtRenderExt* ext, const void* const vol[4],
                           const void* labels, const MrirtInrDesc* net, const float zmu[4], const float zsigma[4],
                           uint32_t chunk_steps, void* scratch, int64_t scratch_bytes,
                           void* out_rgba, int64_t pitch_px, uint64_t* stats_dev, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Misc                                                                                  */
/* ------------------------------------------------------------------------------------ */
int mrirt_abi_version(void);
const char* mrirt_status_string(int status);
int mrirt_last_hip_error(void);            /* hipError_t of the last failed HIP call on this thread */
uint32_t mrirt_sizeof(uint32_t which);     /* 0 BratsParams, 1 RenderExt, 2 VolumeParams, 3 SdfParams, 4 InrDesc, 5 Skip */
/* Opt-in diagnostics (host only; the library installs nothing by itself): a SIGABRT handler that writes the native
 * backtrace of the aborting thread and, when file descriptor 2 has been redirected into a regular file (a test runner's
 * capture), the tail of that file to `fd` — a descriptor the caller duplicated before the redirection — and then chains
 * to the handler that was installed before it.  The ROCm runtime reports GPU faults as a line on stderr + abort() from
 * its event thread; under a capturing runner that line is otherwise lost with the process.                              */
int mrirt_install_abort_trace(int fd);

#ifdef __cplusplus
}
#endif
#endif /* MRIRT_H */
